"""AIS end to end on the GPU: the UNETR decoder (parameters in models/unetr.py, inference on the library's fp32 kernels:
models/unetr_hip.py) on the HIP encoder's embedding -> InstanceSegmentationWithDecoder (seeded watershed) and AutomaticPromptGenerator
(derived prompts through the HIP mask decoder) - shapes, determinism, state caching; and (round 5) the decoder's three maps and the AIS
label image against the ORACLE (oracle/unetr_ref.py: an independent restatement of the reference's DecoderAdapter graph)."""
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


@pytest.mark.parametrize("transpose", [True, False])
def test_unetr_decoder_on_hip_kernels_vs_the_oracle(vit_b_sd, transpose):
    """DecoderAdapter.forward on the device (implicit-GEMM convolutions, InstanceNorm, BatchNorm / ReLU / Sigmoid epilogues, fused
    postprocess_masks) == oracle/unetr_ref.decoder_forward (fp32 torch on the CPU) for both up-sampler flavours, a non-square original size
    (the padding crop + resize of postprocess_masks) - and the AIS label image from the HIP maps == the one from the oracle's maps."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    import time
    from micro_sam_amd import instance_segmentation as IS
    from micro_sam_amd import util
    from micro_sam_amd.models import unetr as U
    from micro_sam_amd.synthetic import synthetic_tile
    from oracle import unetr_ref as R
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    torch.manual_seed(5 + int(transpose))
    proto = U.UNETR(predictor.model.image_encoder, U._default_widths(256, 3, transpose))
    with torch.no_grad():
        for mod in proto.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.3); mod.running_var.uniform_(0.6, 1.6); mod.weight.uniform_(0.7, 1.3); mod.bias.normal_(0, 0.2)
    state = {k: v.detach().clone() for k, v in proto.state_dict().items() if not k.startswith("encoder")}
    decoder = IS.get_decoder(predictor.model.image_encoder, state, device="cuda")
    tile = synthetic_tile(9, (600, 800))
    emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
    feats = torch.as_tensor(emb["features"]).float()
    isz, osz = emb["input_size"], emb["original_size"]
    decoder(feats.cuda(), isz, osz)                                                    # first call: weight copies, code objects
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = decoder(feats.cuda(), isz, osz)
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.perf_counter()
    ref = R.decoder_forward({k: v.cpu() for k, v in state.items()}, feats.cpu(), isz, osz)
    t_ref = time.perf_counter() - t0
    assert got.shape == ref.shape == (1, 3, 600, 800)
    d = (got.cpu() - ref).abs()
    rec = {"flavour": "transposed convolutions" if transpose else "bilinear + 1x1", "max_abs_err": float(d.max()), "mean_abs_err": float(d.mean()),
           "hip_seconds": round(t_hip, 4), "oracle_cpu_seconds": round(t_ref, 2)}
    print("\nUNETR decoder, HIP vs oracle:", json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "unetr_parity.json")
        allr = json.load(open(path)) if os.path.exists(path) else {}
        allr[rec["flavour"]] = rec
        json.dump(allr, open(path, "w"), indent=1)
    except OSError:
        pass
    assert d.max().item() <= 2e-4 and d.mean().item() <= 5e-6, rec                       # sigmoid outputs in (0, 1): fp32 rounding through ~20 layers
    # the label image: the same generator fed with the HIP maps and with the oracle's maps
    seg_h = IS.get_instance_segmentation_generator(predictor, is_tiled=False, decoder=decoder)
    seg_h.initialize(tile, emb)
    lab_h = seg_h.generate(min_size=0)
    seg_o = IS.get_instance_segmentation_generator(predictor, is_tiled=False, decoder=lambda e, i, o: ref.to(e.device))
    seg_o.initialize(tile, emb)
    lab_o = seg_o.generate(min_size=0)
    assert lab_h.shape == (600, 800) and (lab_h == lab_o).mean() >= 0.999, float((lab_h == lab_o).mean())
