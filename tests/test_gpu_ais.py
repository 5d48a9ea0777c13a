"""AIS end to end on the GPU: the UNETR decoder module (models/unetr.py; torch operators, widths from the checkpoint) on the
HIP encoder's embedding -> InstanceSegmentationWithDecoder (seeded watershed) and AutomaticPromptGenerator (derived prompts through the HIP
mask decoder) - shapes, determinism, state caching; the decoder's arithmetic itself is checked on the CPU (tests/test_unetr_host.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ais_and_apg_with_the_unetr_decoder(vit_b_sd, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import instance_segmentation as IS
    from micro_sam_amd import precompute_state as PS
    from micro_sam_amd import util
    from micro_sam_amd.models import unetr as U
    from micro_sam_amd.synthetic import synthetic_tile
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    torch.manual_seed(0)
    proto = U.UNETR(predictor.model.image_encoder, U._default_widths(256, 3, True))
    state = {k: v.detach().clone() for k, v in proto.state_dict().items() if not k.startswith("encoder")}
    decoder = IS.get_decoder(predictor.model.image_encoder, state, device="cuda")
    assert isinstance(decoder, IS.DecoderAdapter) and decoder.out_conv.weight.is_cuda
    tile = synthetic_tile(7, (512, 512))
    emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
    seg = IS.get_instance_segmentation_generator(predictor, is_tiled=False, decoder=decoder)
    assert type(seg) is IS.InstanceSegmentationWithDecoder
    seg.initialize(tile, emb)
    st = seg.get_state()
    assert st["foreground"].shape == (512, 512) and np.isfinite(st["center_distances"]).all() and 0 <= st["foreground"].min() <= st["foreground"].max() <= 1
    # the adapter on the predictor's embedding == the module's own decoder path
    proto.eval()                                              # (BatchNorm on its running statistics, as get_unetr leaves the loaded module)
    with torch.no_grad():
        direct = proto.to("cuda").postprocess_masks(proto.decode(predictor.features.float()), predictor.input_size, predictor.original_size)
    assert np.allclose(direct[0, 0].cpu().numpy(), st["foreground"], atol=1e-4)
    labels = seg.generate(min_size=0)
    assert labels.shape == (512, 512) and np.array_equal(labels, seg.generate(min_size=0))
    cached = PS.cache_is_state(predictor, decoder, tile, emb, str(tmp_path), verbose=False)
    again = PS.cache_is_state(predictor, decoder, tile, emb, str(tmp_path), verbose=False)
    assert np.array_equal(cached.get_state()["boundary_distances"], again.get_state()["boundary_distances"])
    apg = IS.get_instance_segmentation_generator(predictor, is_tiled=False, decoder=decoder, segmentation_mode="apg")
    apg.initialize(tile, emb)
    out = apg.generate(foreground_threshold=float(np.quantile(st["foreground"], 0.7)), center_distance_threshold=float(np.quantile(st["center_distances"], 0.3)),
                       boundary_distance_threshold=float(np.quantile(st["boundary_distances"], 0.3)))
    assert out.shape == (512, 512) and out.dtype == np.uint32
