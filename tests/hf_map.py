"""Upstream segment-anything key -> transformers.models.sam key. Test helper."""


def to_hf(k: str) -> str:
    k = k.replace("image_encoder.", "vision_encoder.")
    k = k.replace("patch_embed.proj", "patch_embed.projection")
    k = k.replace(".blocks.", ".layers.")
    for i in (1, 2, 3, 4):
        k = k.replace(f".norm{i}.", f".layer_norm{i}.")
    k = k.replace("neck.0.", "neck.conv1.").replace("neck.1.", "neck.layer_norm1.")
    k = k.replace("neck.2.", "neck.conv2.").replace("neck.3.", "neck.layer_norm2.")
    k = k.replace("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix",
                  "shared_image_embedding.positional_embedding")
    k = k.replace("point_embeddings.", "point_embed.")
    k = k.replace("mask_downscaling.0.", "mask_embed.conv1.").replace("mask_downscaling.1.", "mask_embed.layer_norm1.")
    k = k.replace("mask_downscaling.3.", "mask_embed.conv2.").replace("mask_downscaling.4.", "mask_embed.layer_norm2.")
    k = k.replace("mask_downscaling.6.", "mask_embed.conv3.")
    k = k.replace("transformer.norm_final_attn", "transformer.layer_norm_final_attn")
    k = k.replace("output_upscaling.0.", "upscale_conv1.").replace("output_upscaling.1.", "upscale_layer_norm.")
    k = k.replace("output_upscaling.3.", "upscale_conv2.")
    if "output_hypernetworks_mlps." in k or "iou_prediction_head." in k:
        k = k.replace("layers.0.", "proj_in.").replace("layers.1.", "layers.0.").replace("layers.2.", "proj_out.")
    return k


def load_into_hf(sd):
    from transformers import SamConfig, SamModel
    hf = SamModel(SamConfig()).eval()
    hsd = hf.state_dict()
    new = {}
    for k, v in sd.items():
        hk = to_hf(k)
        assert hk in hsd and hsd[hk].shape == v.shape, (k, hk)
        new[hk] = v
    new["prompt_encoder.shared_embedding.positional_embedding"] = \
        sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    assert not (set(hsd) - set(new))
    hf.load_state_dict(new)
    # known divergence: HF uses eps 1e-6 in the two-way block LayerNorms, upstream 1e-5
    for blk in hf.mask_decoder.transformer.layers:
        for ln in (blk.layer_norm1, blk.layer_norm2, blk.layer_norm3, blk.layer_norm4):
            ln.eps = 1e-5
    return hf
