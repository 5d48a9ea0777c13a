"""vit_t (MobileSAM / TinyViT-5M; BASELINE configs[0], reference micro_sam/util.py:35-43,435-439): the module tree with MobileSAM's
parameter names against the functional CPU restatement of the published architecture (oracle/tinyvit_ref.py), pinned the only way
available offline - parameter names / shapes of the published checkpoint layout, 5.78 M encoder parameters without the classification
head (MobileSAM paper), [1, 256, 64, 64] output.  The encoder runs torch operators (models/tiny_vit.py: scope note), so this runs on
the CPU; the full get_sam_model("vit_t") -> AMG plumbing of config 1 needs the GPU decoder (tests/test_gpu_model.py)."""
import torch

from micro_sam_amd import modeling
from micro_sam_amd.synthetic import synthetic_state_dict


def test_tinyvit_module_matches_the_restated_architecture():
    from oracle import tinyvit_ref as T
    sd = synthetic_state_dict("vit_t", 0)
    sam = modeling.build_sam("vit_t")
    sam.load_state_dict(sd)                                           # strict: every published key, no extra key
    enc = sam.image_encoder
    names = dict(enc.named_parameters())
    n_enc = sum(p.numel() for k, p in names.items() if not k.startswith(("head.", "norm_head.")))
    assert 5.70e6 < n_enc < 5.85e6, n_enc                              # 5.78 M (MobileSAM paper: image encoder)
    assert names["layers.1.blocks.0.attn.attention_biases"].shape == (4, 49)           # 7 x 7 window: 49 |dy|, |dx| offsets
    assert names["layers.2.blocks.0.attn.attention_biases"].shape == (5, 196)          # 14 x 14 window
    assert names["layers.2.blocks.0.attn.qkv.weight"].shape == (480, 160) and names["layers.3.blocks.1.mlp.fc1.weight"].shape == (1280, 320)
    assert names["layers.0.blocks.0.conv2.c.weight"].shape == (256, 1, 3, 3) and names["layers.1.downsample.conv1.c.weight"].shape == (160, 128, 1, 1)
    assert names["patch_embed.seq.0.c.weight"].shape == (32, 3, 3, 3) and names["neck.2.weight"].shape == (256, 256, 3, 3)
    assert "layers.3.downsample.conv1.c.weight" not in names and enc.layers[2].downsample.conv2.c.stride == (1, 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    with torch.no_grad():
        out = enc(x)
        ref = T.image_encoder(sd, x)
    assert out.shape == (1, 256, 64, 64) and torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()
    from micro_sam_amd import util
    assert util._hash_state_dict(sd).startswith("xxh128:") and util._validate_model_type(sd) == "vit_t"      # 0-dim BatchNorm counters hash too
    # uint8 path = Sam.preprocess + forward
    img = (torch.rand(1, 700, 1024, 3, generator=g) * 255).to(torch.uint8)
    xf = img.permute(0, 3, 1, 2).float()
    xf = (xf - sam.pixel_mean) / sam.pixel_std
    xf = torch.nn.functional.pad(xf, (0, 0, 0, 324))
    with torch.no_grad():
        assert (enc.forward_u8(img) - enc(xf)).abs().max().item() < 1e-5


def test_tinyvit_under_a_trainer_keeps_its_running_statistics_and_has_a_tape():
    """a trainer calls model.train(); the no-tape forward of a FROZEN TinyViT must keep normalising with the
    checkpoint's running statistics (and must not overwrite them), and a trainable TinyViT gets its gradients through forward_taped
    (reference: mobile_sam's TinyViT is fine-tuned end to end by training/sam_trainer.py)."""
    sd = synthetic_state_dict("vit_t", 0)
    sam = modeling.build_sam("vit_t")
    sam.load_state_dict(sd)
    enc = sam.image_encoder
    bn = next(m for m in enc.modules() if isinstance(m, torch.nn.BatchNorm2d))
    x = torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(2))
    enc.eval()
    ref = enc(x)
    mean0, count0 = bn.running_mean.clone(), int(bn.num_batches_tracked)
    enc.train()
    out = enc(x)
    assert torch.equal(out, ref)                                        # running statistics, not batch statistics
    assert torch.equal(bn.running_mean, mean0) and int(bn.num_batches_tracked) == count0 and bn.training
    # the taped forward: gradients reach the first convolution; train() mode updates the running statistics as torch does
    y = enc.forward_taped(x)
    assert y.requires_grad
    y.square().mean().backward()
    g = enc.patch_embed.seq[0].c.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert int(bn.num_batches_tracked) == count0 + 1
