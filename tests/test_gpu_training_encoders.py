"""Un-frozen fine-tuning on the GPU (reference micro_sam/training/trainable_sam.py:71-81,96-99 under autograd):
the rel-pos attention kernels and the wide LayerNorm backward against torch autograd, the differentiable image encoder and
prompt encoder (values + parameter gradients) against the fp32 oracle's autograd, and a few optimiser steps of the whole model.

Tolerances as in tests/test_gpu_training.py: fp32 kernels 1e-4 relative (2e-4 for attention gradients); bf16-operand GEMM
compositions: outputs 2 % of the output range, per-parameter gradient cosine similarity >= 0.99."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _dense(q, k, v, bh, bw, scale):
    BH, N, _ = q.shape
    Gh, Gw = bh.shape[2], bw.shape[2]
    s = (q * scale) @ k.transpose(1, 2)
    s = (s.view(BH, N, Gh, Gw) + bh[:, :, :, None] + bw[:, :, None, :]).view(BH, N, N)
    return s.softmax(dim=-1) @ v


@pytest.mark.parametrize("BH,Gh,Gw,D", [(6, 14, 14, 64), (2, 32, 32, 64), (3, 14, 14, 80), (1, 64, 64, 64), (2, 5, 9, 64)])
def test_relpos_attention_forward_backward(dev, BH, Gh, Gw, D):
    from micro_sam_amd.training import functional as HF
    g = torch.Generator().manual_seed(BH * Gh + D)
    N = Gh * Gw
    q, k, v = (torch.randn(BH, N, D, generator=g).to(dev).requires_grad_() for _ in range(3))
    bh = torch.randn(BH, N, Gh, generator=g).to(dev).requires_grad_()
    bw = torch.randn(BH, N, Gw, generator=g).to(dev).requires_grad_()
    do = torch.randn(BH, N, D, generator=g).to(dev)
    scale = D ** -0.5
    out = HF.relpos_attention(q, k, v, bh, bw, scale)
    out.backward(do)
    ref_in = [t.detach().double().clone().requires_grad_() for t in (q, k, v, bh, bw)]
    ref = _dense(*ref_in, scale)
    ref.backward(do.double())
    assert _rel(out.double(), ref) < 1e-4
    for name, a, b in zip(("dq", "dk", "dv", "dbias_h", "dbias_w"), (q, k, v, bh, bw), ref_in):
        assert _rel(a.grad.double(), b.grad) < 2e-4, name


@pytest.mark.parametrize("BH,Gh,Gw,D", [(6, 14, 14, 64), (2, 64, 64, 64), (3, 14, 14, 80)])
def test_relpos_attention_gemm_route_against_the_fp32_kernels(dev, BH, Gh, Gw, D):
    """The default training route (torch.bmm on bf16 operands, fp32 softmax; functional.relpos_attention_gemm) against the hand-written
    fp32 kernels on the same inputs: values and all five gradients within bf16 operand rounding."""
    from micro_sam_amd.training import functional as HF
    g = torch.Generator().manual_seed(BH + Gh + D)
    N = Gh * Gw
    ins = [torch.randn(BH, N, D, generator=g) for _ in range(3)] + [torch.randn(BH, N, Gh, generator=g), torch.randn(BH, N, Gw, generator=g)]
    do = torch.randn(BH, N, D, generator=g).to(dev)
    a = [t.to(dev).requires_grad_() for t in ins]
    b = [t.to(dev).requires_grad_() for t in ins]
    out = HF.relpos_attention_gemm(*a, D ** -0.5)
    ref = HF.relpos_attention(*b, D ** -0.5)
    assert out.dtype == torch.float32 and _rel(out, ref) < 2e-2
    out.backward(do)
    ref.backward(do)
    for name, x, y in zip(("dq", "dk", "dv", "dbias_h", "dbias_w"), a, b):
        assert _rel(x.grad, y.grad) < 3e-2, name


@pytest.mark.parametrize("dim", [768, 1024, 1280])
def test_layer_norm_backward_encoder_widths(dev, dim):
    from micro_sam_amd.training import functional as HF
    g = torch.Generator().manual_seed(dim)
    x = (torch.randn(2, 700, dim, generator=g) * 2 + 0.5).to(dev).requires_grad_()
    w = (torch.randn(dim, generator=g) * 0.2 + 1).to(dev).requires_grad_()
    b = torch.randn(dim, generator=g).to(dev).requires_grad_()
    dy = torch.randn(2, 700, dim, generator=g).to(dev)
    y = HF.layer_norm(x, w, b, 1e-6)
    y.backward(dy)
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    yr = F.layer_norm(xr, (dim,), wr, br, 1e-6)
    yr.backward(dy)
    assert _rel(y, yr) < 1e-5
    assert _rel(x.grad, xr.grad) < 1e-4 and _rel(w.grad, wr.grad) < 1e-4 and _rel(b.grad, br.grad) < 1e-4


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def test_image_encoder_forward_and_gradients_vs_oracle(dev, monkeypatch):
    """A two-block encoder (one windowed, one global; width 256, 4 heads of 64) on the real 64 x 64 token grid against the
    oracle's fp32 autograd on the CPU."""
    from micro_sam_amd import modeling
    from micro_sam_amd.training import encoders as E
    from oracle import sam_ref as S
    torch.manual_seed(0)
    enc = modeling.ImageEncoderViT(embed_dim=256, depth=2, num_heads=4, global_attn_indexes=(1,))
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if ("norm" in n and n.endswith("weight")) or n in ("neck.1.weight", "neck.3.weight"):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * (0.5 if "rel_pos" in n or "pos_embed" in n or n.endswith("bias") else p[0].numel() ** -0.5))
    sd = {"image_encoder." + k: v.detach().clone().requires_grad_() for k, v in enc.state_dict().items()}
    monkeypatch.setitem(S.VIT_CONFIGS, "vit_s", {"embed_dim": 256, "depth": 2, "num_heads": 4, "global_attn_indexes": (1,)})
    enc.to(dev)
    x = torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(1, 256, 64, 64, generator=torch.Generator().manual_seed(2))
    out = E.image_encoder_forward(enc, x.to(dev))
    (out * gout.to(dev)).sum().backward()
    ref = S.image_encoder(sd, x, model_type="vit_s", precision="fp32")
    (ref * gout).sum().backward()
    assert out.shape == (1, 256, 64, 64)
    assert (out.cpu() - ref).abs().max().item() <= 0.02 * (ref.max() - ref.min()).item()
    cos = {n: _cos(p.grad.cpu(), sd["image_encoder." + n].grad) for n, p in enc.named_parameters()}
    assert min(cos.values()) >= 0.99, sorted(cos.items(), key=lambda kv: kv[1])[:5]


def test_full_vit_b_training_forward_matches_the_inference_encoder(dev, vit_b_sd):
    """The differentiable forward and the inference kernels are two implementations of the same network (both with 16-bit
    GEMM operands): their embeddings of the same image agree to the operand rounding."""
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_tile
    from micro_sam_amd.training import TrainableSAM
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=vit_b_sd)
    model = TrainableSAM(predictor.model)
    img = torch.as_tensor(np.repeat(synthetic_tile(3)[None], 3, axis=0).astype(np.float32))
    batch = [{"image": img, "original_size": (1024, 1024)}]
    for p in model.sam.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        frozen, _ = model.image_embeddings_oft([dict(b) for b in batch])
    for p in model.sam.image_encoder.parameters():
        p.requires_grad_(True)
    taped, _ = model.image_embeddings_oft([dict(b) for b in batch])
    assert taped.requires_grad and not frozen.requires_grad and taped.shape == frozen.shape == (1, 256, 64, 64)
    d = (taped.detach() - frozen).abs()
    assert d.mean().item() <= 0.02 and d.max().item() <= 0.25, (d.mean().item(), d.max().item())
    taped.square().mean().backward()
    grads = [p.grad for p in model.sam.image_encoder.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0


def test_prompt_encoder_forward_and_gradients_vs_oracle(dev, vit_b_sd):
    from micro_sam_amd import util
    from micro_sam_amd.training import encoders as E
    from oracle import sam_ref as S
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=vit_b_sd)
    pe = predictor.model.prompt_encoder
    sd = {k: v.detach().clone().float().requires_grad_("gaussian" not in k) for k, v in vit_b_sd.items() if k.startswith("prompt_encoder.")}
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(5, 3, 2, generator=g) * 1000, torch.randint(0, 2, (5, 3), generator=g))
    x0 = torch.rand(5, 2, generator=g) * 600
    boxes = torch.cat([x0, x0 + 50 + torch.rand(5, 2, generator=g) * 300], dim=1)
    masks = torch.randn(5, 1, 256, 256, generator=g) * 3
    for args in ((pts, None, None), (None, boxes, None), (pts, boxes, masks)):
        for p in pe.parameters():
            p.grad = None
        for v in sd.values():
            v.grad = None
        dargs = tuple(None if a is None else (tuple(t.to(dev) for t in a) if isinstance(a, tuple) else a.to(dev)) for a in args)
        sparse, dense = E.prompt_encoder_forward(pe, *dargs)
        rs, rd = S.prompt_encoder(sd, *args)
        assert (sparse.cpu() - rs).abs().max().item() < 2e-4
        # (mask prompts: the three small convolutions take bf16 operands - 2 % of the embedding's range)
        assert (dense.cpu() - rd).abs().max().item() <= (0.02 * (rd.max() - rd.min()).item() if args[2] is not None else 1e-6)
        gs, gd = torch.randn(rs.shape, generator=g), torch.randn(rd.shape, generator=g)
        ((sparse * gs.to(dev)).sum() + (dense * gd.to(dev)).sum()).backward()
        ((rs * gs).sum() + (rd * gd).sum()).backward()
        for n, p in pe.named_parameters():
            r = sd["prompt_encoder." + n].grad
            if r is not None and r.abs().max() > 0:
                assert _cos(p.grad.cpu(), r) >= 0.99, n


def test_unfrozen_training_steps(dev, vit_b_sd):
    """Three SGD steps of the WHOLE model on one batch: every part receives finite gradients and the loss goes down; the
    inference kernels pick the changed encoder weights up afterwards."""
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_tile
    from micro_sam_amd.training import TrainableSAM
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=vit_b_sd)
    model = TrainableSAM(predictor.model)
    img = torch.as_tensor(np.repeat(synthetic_tile(5, (512, 512))[None], 3, axis=0).astype(np.float32))
    yy, xx = np.mgrid[0:512, 0:512]
    target = torch.as_tensor(((yy - 250) ** 2 + (xx - 260) ** 2 < 90 ** 2).astype(np.float32))[None, None].to(dev)
    rec = {"image": img, "original_size": (512, 512), "point_coords": torch.tensor([[[520.0, 500.0]]]), "point_labels": torch.tensor([[1]])}
    # (lr: with the synthetic weights the mask logits are of order 100; 1e-3 overshoots, 1e-5 / 1e-6 descend - CPU dry run)
    opt = torch.optim.SGD(model.sam.parameters(), lr=3e-6)
    with torch.no_grad():
        before, _ = model.image_embeddings_oft([dict(rec)])
    losses = []
    for _ in range(3):
        opt.zero_grad()
        emb, inputs = model.image_embeddings_oft([dict(rec)])
        out = model(inputs, emb, multimask_output=False)[0]
        loss = F.binary_cross_entropy_with_logits(out["masks"], target)
        loss.backward()
        for part in (model.sam.image_encoder, model.sam.prompt_encoder, model.sam.mask_decoder):
            gs = [p.grad for p in part.parameters() if p.grad is not None]
            assert gs and all(torch.isfinite(g).all() for g in gs) and sum(float(g.abs().sum()) for g in gs) > 0
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    with torch.no_grad():
        after, _ = model.image_embeddings_oft([dict(rec)])
    assert (after - before).abs().max().item() > 0                           # the inference copies were rebuilt


def test_get_trainable_sam_model_freeze_and_lora(dev, vit_b_sd):
    """training.get_trainable_sam_model (reference micro_sam/training/util.py:77-151): `freeze`, LoRA surgery, and one taped
    step of a LoRA model - only the low-rank matrices receive gradients (B starts at zero, so dA = 0 and dB != 0)."""
    from micro_sam_amd.synthetic import synthetic_tile
    from micro_sam_amd.training import get_trainable_sam_model
    m = get_trainable_sam_model("vit_b", device=dev, state_dict=vit_b_sd, freeze=["image_encoder", "prompt_encoder"])
    assert {n.split(".")[0] for n, p in m.sam.named_parameters() if p.requires_grad} == {"mask_decoder"}
    m = get_trainable_sam_model("vit_b", device=dev, state_dict=vit_b_sd)
    assert all(p.requires_grad for p in m.sam.parameters())
    with pytest.raises(ValueError):
        get_trainable_sam_model("vit_b", device=dev, state_dict=vit_b_sd, peft_kwargs={"rank": 4}, freeze="image_encoder")
    m = get_trainable_sam_model("vit_b", device=dev, state_dict=vit_b_sd, peft_kwargs={"rank": 4}, freeze=["prompt_encoder", "mask_decoder"])
    trainable = [n for n, p in m.sam.named_parameters() if p.requires_grad]
    assert len(trainable) == 12 * 4 and all(".w_a_linear_" in n or ".w_b_linear_" in n for n in trainable)
    img = torch.as_tensor(np.repeat(synthetic_tile(3)[None], 3, axis=0).astype(np.float32))
    emb, _ = m.image_embeddings_oft([{"image": img, "original_size": (1024, 1024)}])
    assert emb.requires_grad
    emb.square().mean().backward()
    for n, p in m.sam.named_parameters():
        if ".w_b_linear_" in n:
            assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, n
        elif ".w_a_linear_" in n:
            assert p.grad is not None and float(p.grad.abs().sum()) == 0, n
        else:
            assert p.grad is None, n
