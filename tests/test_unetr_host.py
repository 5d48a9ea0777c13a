"""models/unetr.py: the UNETR decoder of micro_sam's AIS (reference instance_segmentation.py:688-870) restated from torch_em's published
module tree - PARITY UNPINNED against torch_em (absent).  What CAN be checked without it: the module tree is built to the shapes of a
``decoder_state`` (both up-sampler flavours, default and other widths), loads it strictly with the reference's error behaviour, the adapter
computes exactly the unetr's decoder path and its own functional restatement in plain torch, and the whole AIS route
(checkpoint file -> get_predictor_and_decoder -> InstanceSegmentationWithDecoder -> cache_is_state) runs - here on a host stand-in for the
predictor's encoder (the decoder is torch operators; the HIP encoder only supplies the embedding)."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from micro_sam_amd.models import unetr as U


class _Enc(torch.nn.Module):
    img_size = 1024

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return F.adaptive_avg_pool2d(x, 64).mean(1, keepdim=True).expand(-1, 256, -1, -1) * 0.01


def _functional(state, z12, transpose):
    """The adapter's forward graph written out over the raw tensors of a decoder_state."""
    def inorm(x):
        return F.instance_norm(x)

    def convblock(x, p):
        x = F.relu(F.conv2d(inorm(x), state[p + ".block.1.weight"], state[p + ".block.1.bias"], padding=1))
        return F.relu(F.conv2d(inorm(x), state[p + ".block.4.weight"], state[p + ".block.4.bias"], padding=1))

    def up(x, p):
        if transpose:
            return F.conv_transpose2d(x, state[p + ".block.weight"], state[p + ".block.bias"], stride=2)
        return F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), state[p + ".conv.weight"], state[p + ".conv.bias"])

    def deconv(x, p):
        x = up(x, p + ".block.0")
        x = F.conv2d(x, state[p + ".block.1.block.weight"], state[p + ".block.1.block.bias"], padding=1)
        x = F.batch_norm(x, state[p + ".block.2.running_mean"], state[p + ".block.2.running_var"], state[p + ".block.2.weight"],
                         state[p + ".block.2.bias"], False, 0.0, 1e-5)
        return F.relu(x)
    z9 = deconv(z12, "deconv1"); z6 = deconv(z9, "deconv2"); z3 = deconv(z6, "deconv3"); z0 = deconv(z3, "deconv4")
    x = convblock(z12, "base")
    for i, skip in enumerate((z9, z6, z3)):
        x = convblock(torch.cat([up(x, f"decoder.samplers.{i}"), skip], 1), f"decoder.blocks.{i}")
    x = convblock(torch.cat([up(x, "deconv_out"), z0], 1), "decoder_head")
    return torch.sigmoid(F.conv2d(x, state["out_conv.weight"], state["out_conv.bias"]))


def _random_state(transpose, widths=None, seed=0):
    torch.manual_seed(seed)
    w = widths or U._default_widths(256, 3, transpose)
    w = dict(w, use_conv_transpose=transpose)
    m = U.UNETR(_Enc(), w)
    for mod in m.modules():                                   # non-trivial BatchNorm statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
    return {k: v.clone() for k, v in m.state_dict().items() if not k.startswith("encoder")}


@pytest.mark.parametrize("transpose", [False, True])
def test_module_tree_follows_the_checkpoint_and_matches_the_functional_graph(transpose):
    state = _random_state(transpose)
    assert any(k.startswith("decoder.samplers.0.block.") for k in state) == transpose           # the reference's flavour test (:772)
    assert {"base.block.1.weight", "decoder.blocks.2.block.4.bias", "deconv3.block.1.block.weight", "deconv4.block.2.running_var",
            "decoder_head.block.1.weight", "out_conv.bias"} <= set(state)
    enc = _Enc()
    dec = U.get_decoder(enc, state, device="cpu")
    assert isinstance(dec, U.DecoderAdapter) and dec.deconv_out.__class__ is (U.SingleDeconv2DBlock if transpose else U.Upsampler2d)
    z12 = torch.randn(1, 256, 16, 16)                         # (a 16 x 16 grid keeps the CPU test small: the graph is size-agnostic)
    with torch.no_grad():
        out = dec._forward_impl(z12)
        ref = _functional(state, z12, transpose)
    assert out.shape == (1, 3, 256, 256) and torch.allclose(out, ref, atol=1e-5), float((out - ref).abs().max())
    full = dec(z12, (192, 256), (96, 128))                    # postprocess_masks: to 1024, crop the padding, to the original size
    assert full.shape == (1, 3, 96, 128)


def test_widths_come_from_the_checkpoint():
    """Another self-consistent width set (wider skips) builds and loads: nothing about the widths is hard-wired."""
    w = {"use_conv_transpose": True, "base": (256, 96), "blocks": [(96 + 40, 48), (48 + 24, 32), (32 + 16, 16)],
         "samplers": [(96, 96), (48, 48), (32, 32)], "deconv": [(256, 40), (40, 24), (24, 16), (16, 8)], "deconv_out": (16, 16),
         "head": (16 + 8, 12), "out": (12, 3)}
    state = _random_state(True, w, seed=3)
    dec = U.get_decoder(_Enc(), state, device="cpu")
    z12 = torch.randn(2, 256, 8, 8)
    with torch.no_grad():
        assert torch.allclose(dec._forward_impl(z12), _functional(state, z12, True), atol=1e-5)


def test_strict_and_flexible_loading_like_the_reference():
    state = _random_state(False)
    broken = dict(state)
    del broken["decoder_head.block.4.weight"]
    with pytest.raises(RuntimeError, match="could not be found"):
        U.get_unetr(_Enc(), broken, device="cpu")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m = U.get_unetr(_Enc(), broken, device="cpu", flexible_load_checkpoint=True)
    assert any("Could not find 'decoder_head.block.4.weight'" in str(w.message) for w in rec)
    assert torch.equal(m.base.block[1].weight, state["base.block.1.weight"])
    # no checkpoint: the default (interpolation) decoder for training, 3 output channels, Sigmoid
    fresh = U.get_unetr(_Enc(), None, device="cpu")
    assert isinstance(fresh.deconv_out, U.Upsampler2d) and isinstance(fresh.final_activation, torch.nn.Sigmoid) and fresh.out_channels == 3
    with torch.no_grad():
        y = fresh(torch.rand(1, 3, 96, 128) * 255)
    assert y.shape == (1, 3, 96, 128) and float(y.min()) >= 0 and float(y.max()) <= 1


def test_ais_route_through_a_checkpoint_file(tmp_path, monkeypatch):
    """checkpoint {"model_state", "decoder_state"} -> get_predictor_and_decoder -> factory default "ais" -> initialize / generate,
    cache_is_state round trip (precompute_state.py:90-155)."""
    from micro_sam_amd import instance_segmentation as IS
    from micro_sam_amd import precompute_state as PS
    from micro_sam_amd import util
    state = _random_state(True, seed=5)

    class P:                                                  # host stand-in for the predictor (its embedding comes from the HIP encoder)
        device = "cpu"
        features = original_size = input_size = None
        is_image_set = False

        class model:
            image_encoder = _Enc()
    monkeypatch.setattr(util, "get_sam_model", lambda **kw: (P(), {"model_state": {}, "decoder_state": state}))
    predictor, decoder = IS.get_predictor_and_decoder("vit_b", checkpoint_path=str(tmp_path / "ckpt.pt"), device="cpu")
    assert isinstance(decoder, IS.DecoderAdapter)
    monkeypatch.setattr(util, "get_sam_model", lambda **kw: (P(), {"model_state": {}}))
    with pytest.raises(ValueError, match="does not contain a decoder state"):
        IS.get_predictor_and_decoder("vit_b", checkpoint_path="x", device="cpu")
    emb = {"features": np.random.default_rng(0).standard_normal((1, 256, 64, 64)).astype(np.float32) * 0.3, "input_size": (1024, 1024),
           "original_size": (128, 128)}
    seg = PS.cache_is_state(predictor, decoder, np.zeros((128, 128), np.uint8), emb, str(tmp_path), verbose=False)
    assert type(seg) is IS.InstanceSegmentationWithDecoder
    st = seg.get_state()
    assert st["foreground"].shape == (128, 128) and 0.0 <= float(st["foreground"].min()) and float(st["foreground"].max()) <= 1.0
    labels = seg.generate(min_size=0)
    assert labels.shape == (128, 128) and labels.dtype == np.uint32
    again = PS.cache_is_state(predictor, decoder, np.zeros((128, 128), np.uint8), emb, str(tmp_path), verbose=False)      # loads the state
    assert np.array_equal(again.generate(min_size=0), labels)
