"""Un-frozen fine-tuning, host half: the differentiable image encoder / prompt encoder of micro_sam_amd.training.encoders
(reference micro_sam/training/trainable_sam.py:71-81,96-99 under autograd) are compositions of three HIP primitives
(functional.linear / layer_norm / relpos_attention).  Here the primitives are replaced by torch stand-ins and the COMPOSITION -
patch gather, window partition / padding, rel-pos tables and bias einsums, head split, residuals, the neck as GEMMs, the prompt
encoder's embeddings and patch convolutions - is checked against the oracle's fp32 functions, values and gradients, on the
CPU.  The primitives themselves are checked on the GPU (tests/test_gpu_training_encoders.py); the formulas their backward
kernels implement are checked below against autograd through a numpy transcription of the kernels' loops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from micro_sam_amd import modeling
from micro_sam_amd.training import encoders as E
from micro_sam_amd.training import functional as HF
from oracle import sam_ref as S


def _dense_relpos_attention(q, k, v, bias_h, bias_w, scale):
    BH, N, D = q.shape
    Gh, Gw = bias_h.shape[2], bias_w.shape[2]
    s = (q * scale) @ k.transpose(1, 2)
    s = (s.view(BH, N, Gh, Gw) + bias_h[:, :, :, None] + bias_w[:, :, None, :]).view(BH, N, N)
    return s.softmax(dim=-1) @ v


@pytest.fixture()
def torch_primitives(monkeypatch):
    monkeypatch.setattr(HF, "linear", lambda x, w, b=None: F.linear(x, w, b))
    monkeypatch.setattr(HF, "layer_norm", lambda x, w, b, eps: F.layer_norm(x, (x.shape[-1],), w, b, eps))
    monkeypatch.setattr(HF, "relpos_attention", _dense_relpos_attention)
    monkeypatch.setattr(HF, "RELPOS_ATTENTION_IMPL", "kernel")      # the composition is checked exactly; the bf16 GEMM route below


def _small_encoder(seed=0):
    """Two blocks (one windowed, one global) of width 128 / 2 heads on the real 64 x 64 token grid."""
    torch.manual_seed(seed)
    enc = modeling.ImageEncoderViT(embed_dim=128, depth=2, num_heads=2, global_attn_indexes=(1,))
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "norm" in n and n.endswith("weight") or n in ("neck.1.weight", "neck.3.weight"):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * (0.5 if "rel_pos" in n or "pos_embed" in n or n.endswith("bias") else p[0].numel() ** -0.5))
    sd = {"image_encoder." + k: v.detach().clone().requires_grad_() for k, v in enc.state_dict().items()}
    return enc, sd


def test_image_encoder_composition_matches_oracle_values_and_gradients(torch_primitives, monkeypatch):
    enc, sd = _small_encoder()
    monkeypatch.setitem(S.VIT_CONFIGS, "vit_s", {"embed_dim": 128, "depth": 2, "num_heads": 2, "global_attn_indexes": (1,)})
    x = torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(1))
    out = E.image_encoder_forward(enc, x)
    ref = S.image_encoder(sd, x, model_type="vit_s", precision="fp32")
    assert out.shape == ref.shape == (1, 256, 64, 64)
    assert (out - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    (out * g).sum().backward()
    (ref * g).sum().backward()
    worst = {}
    for n, p in enc.named_parameters():
        r = sd["image_encoder." + n].grad
        assert p.grad is not None and r is not None, n
        worst[n] = (p.grad - r).abs().max().item() / (r.abs().max().item() + 1e-12)
    assert max(worst.values()) <= 2e-3, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    assert {"pos_embed", "blocks.0.attn.rel_pos_h", "blocks.1.attn.rel_pos_w", "patch_embed.proj.weight", "neck.2.weight"} <= set(worst)


def test_rel_pos_table_interpolation_and_window_round_trip():
    t = torch.randn(27, 64)
    assert torch.equal(E._rel_pos_table(t, 14), S._get_rel_pos(14, 14, t))
    t2 = torch.randn(127, 64)                                   # a checkpoint table of another length is interpolated
    assert torch.allclose(E._rel_pos_table(t2, 14), S._get_rel_pos(14, 14, t2))
    x = torch.randn(2, 64, 64, 8)
    w, pad_hw = E._window_partition(x, 14)
    assert w.shape == (2 * 25, 14, 14, 8) and pad_hw == (70, 70)
    assert torch.equal(E._window_unpartition(w, 14, pad_hw, (64, 64)), x)


@pytest.mark.parametrize("kind", ["points", "boxes", "points+boxes", "masks+boxes", "points+masks"])
def test_prompt_encoder_composition_matches_oracle(torch_primitives, kind):
    torch.manual_seed(3)
    pe = modeling.PromptEncoder()
    with torch.no_grad():
        for p in pe.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    sd = {"prompt_encoder." + k: v.detach().clone().requires_grad_(v.is_floating_point() and "gaussian" not in k)
          for k, v in pe.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(5, 3, 2, generator=g) * 1000, torch.randint(0, 2, (5, 3), generator=g))
    x0 = torch.rand(5, 2, generator=g) * 600
    boxes = torch.cat([x0, x0 + 50 + torch.rand(5, 2, generator=g) * 300], dim=1)
    masks = torch.randn(5, 1, 256, 256, generator=g) * 3
    args = (pts if "points" in kind else None, boxes if "boxes" in kind else None, masks if "masks" in kind else None)
    sparse, dense = E.prompt_encoder_forward(pe, *args)
    rs, rd = S.prompt_encoder(sd, *args)
    assert sparse.shape == rs.shape and dense.shape == rd.shape == (5, 256, 64, 64)
    assert torch.allclose(sparse, rs, atol=1e-5) and torch.allclose(dense, rd, atol=1e-4)
    gs, gd = torch.randn(sparse.shape, generator=g), torch.randn(dense.shape, generator=g)
    ((sparse * gs).sum() + (dense * gd).sum()).backward()
    ((rs * gs).sum() + (rd * gd).sum()).backward()
    for n, p in pe.named_parameters():
        r = sd["prompt_encoder." + n].grad
        if r is None:
            assert p.grad is None or not p.grad.abs().any(), n
        else:
            assert torch.allclose(p.grad, r, rtol=1e-3, atol=1e-4 * r.abs().max().item() + 1e-6), n


def _kernel_loops(q, k, v, bh, bw, dout, scale):
    """numpy transcription of relpos_fwd_kernel / relpos_bwd_q_kernel / relpos_bwd_kv_kernel (csrc/train.hip), one (bh) slice."""
    N, D = q.shape
    Gh, Gw = bh.shape[1], bw.shape[1]
    out, lse = np.zeros((N, D)), np.zeros(N)
    for i in range(N):                                           # forward: online softmax over (kh, kw)
        qv, m, l, acc = q[i] * scale, -3.0e38, 0.0, np.zeros(D)
        for kh in range(Gh):
            for kw in range(Gw):
                j = kh * Gw + kw
                s = bh[i, kh] + bw[i, kw] + qv @ k[j]
                mn = max(m, s); a = np.exp(m - mn); p = np.exp(s - mn)
                l = l * a + p; acc = acc * a + p * v[j]; m = mn
        out[i] = acc / l; lse[i] = m + np.log(l)
    dq, dbh, dbw, delta = np.zeros((N, D)), np.zeros((N, Gh)), np.zeros((N, Gw)), np.zeros(N)
    for i in range(N):                                           # backward, query side
        qv, dl, acc = q[i] * scale, dout[i] @ out[i], np.zeros(D)
        for kh in range(Gh):
            d_h = 0.0
            for kw in range(Gw):
                j = kh * Gw + kw
                s = bh[i, kh] + bw[i, kw] + qv @ k[j]
                ds = np.exp(s - lse[i]) * (dout[i] @ v[j] - dl)
                acc += ds * k[j]; d_h += ds; dbw[i, kw] += ds
            dbh[i, kh] = d_h
        dq[i] = acc * scale; delta[i] = dl
    dk, dv = np.zeros((N, D)), np.zeros((N, D))
    for j in range(N):                                           # backward, key side
        kh, kw = divmod(j, Gw)
        ak, av = np.zeros(D), np.zeros(D)
        for i in range(N):
            p = np.exp((q[i] @ k[j]) * scale + bh[i, kh] + bw[i, kw] - lse[i])
            ds = p * (dout[i] @ v[j] - delta[i])
            ak += ds * q[i]; av += p * dout[i]
        dk[j] = ak * scale; dv[j] = av
    return out, dq, dk, dv, dbh, dbw


def test_relpos_attention_kernel_formulas_match_autograd():
    g = torch.Generator().manual_seed(5)
    Gh, Gw, D = 3, 4, 8
    N = Gh * Gw
    q, k, v = (torch.randn(1, N, D, generator=g, dtype=torch.float64).requires_grad_() for _ in range(3))
    bh = torch.randn(1, N, Gh, generator=g, dtype=torch.float64).requires_grad_()
    bw = torch.randn(1, N, Gw, generator=g, dtype=torch.float64).requires_grad_()
    dout = torch.randn(1, N, D, generator=g, dtype=torch.float64)
    out = _dense_relpos_attention(q, k, v, bh, bw, 0.3)
    out.backward(dout)
    got = _kernel_loops(*(t[0].detach().numpy() for t in (q, k, v, bh, bw)), dout[0].numpy(), 0.3)
    want = (out[0].detach(), q.grad[0], k.grad[0], v.grad[0], bh.grad[0], bw.grad[0])
    for a, b in zip(got, want):
        assert np.allclose(a, b.numpy(), rtol=1e-9, atol=1e-11)


def test_relpos_attention_gemm_route_is_the_dense_attention_at_bf16_operand_precision():
    """The default training route (library batched GEMMs on bf16 operands, fp32 scores and softmax) against the fp64 dense attention."""
    g = torch.Generator().manual_seed(6)
    Gh, Gw, D = 14, 14, 64
    N = Gh * Gw
    q, k, v = (torch.randn(3, N, D, generator=g).requires_grad_() for _ in range(3))
    bh = torch.randn(3, N, Gh, generator=g).requires_grad_()
    bw = torch.randn(3, N, Gw, generator=g).requires_grad_()
    dout = torch.randn(3, N, D, generator=g)
    out = HF.relpos_attention_gemm(q, k, v, bh, bw, D ** -0.5)
    assert out.dtype == torch.float32
    out.backward(dout)
    ref_in = [t.detach().double().requires_grad_() for t in (q, k, v, bh, bw)]
    ref = _dense_relpos_attention(*ref_in, D ** -0.5)
    ref.backward(dout.double())
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    assert rel(out, ref) < 2e-2
    for a, b in zip((q, k, v, bh, bw), ref_in):
        assert rel(a.grad, b.grad) < 3e-2


def test_lora_branches_train_and_equal_the_merged_weights(torch_primitives, monkeypatch):
    """LoRA surgery (reference models/peft_sam.py:35-131): the taped forward keeps the low-rank branches as separate products
    (so that A and B get gradients); its output equals the oracle on the MERGED weights W + B A, and dA / dB equal the oracle's
    autograd through that merge."""
    from micro_sam_amd.models.peft_sam import LoRASurgery
    enc, sd = _small_encoder(seed=7)
    for p in enc.parameters():
        p.requires_grad_(False)
    surgeries = [LoRASurgery(rank=4, block=blk, update_matrices=["q", "v", "mlp"]) for blk in enc.blocks]
    torch.manual_seed(8)
    lora = {}
    for i, blk in enumerate(enc.blocks):
        for name, mod in (("attn.qkv", blk.attn.qkv), ("mlp", blk.mlp)):
            for pn, p in mod.named_parameters():
                if pn.startswith("w_"):
                    with torch.no_grad():
                        p.copy_(torch.randn_like(p) * 0.1)
                    lora[f"blocks.{i}.{name}.{pn}"] = p
    assert len(lora) == 2 * (4 + 4) and all(p.requires_grad for p in lora.values())
    # the oracle on merged weights, with the LoRA matrices as leaves
    leaves = {k: v.detach().clone().requires_grad_() for k, v in lora.items()}
    msd = {k: v.detach() for k, v in sd.items()}
    for i in range(2):
        pre, D = f"image_encoder.blocks.{i}.", 128
        w = msd[pre + "attn.qkv.weight"].clone()
        upd_q = leaves[f"blocks.{i}.attn.qkv.w_b_linear_q.weight"] @ leaves[f"blocks.{i}.attn.qkv.w_a_linear_q.weight"]
        upd_v = leaves[f"blocks.{i}.attn.qkv.w_b_linear_v.weight"] @ leaves[f"blocks.{i}.attn.qkv.w_a_linear_v.weight"]
        msd[pre + "attn.qkv.weight"] = torch.cat([w[:D] + upd_q, w[D:2 * D], w[2 * D:] + upd_v])
        for j in (1, 2):
            msd[pre + f"mlp.lin{j}.weight"] = msd[pre + f"mlp.lin{j}.weight"] + \
                leaves[f"blocks.{i}.mlp.w_b_linear_{j}.weight"] @ leaves[f"blocks.{i}.mlp.w_a_linear_{j}.weight"]
    monkeypatch.setitem(S.VIT_CONFIGS, "vit_s", {"embed_dim": 128, "depth": 2, "num_heads": 2, "global_attn_indexes": (1,)})
    x = torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(1))
    out = E.image_encoder_forward(enc, x)
    ref = S.image_encoder(msd, x, model_type="vit_s", precision="fp32")
    assert (out - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    # the inference path's merged weight is the same matrix
    assert torch.allclose(enc.blocks[0].attn.qkv.weight, msd["image_encoder.blocks.0.attn.qkv.weight"].detach(), atol=1e-6)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    (out * g).sum().backward()
    (ref * g).sum().backward()
    for k, p in lora.items():
        r = leaves[k].grad
        assert p.grad is not None and (p.grad - r).abs().max().item() <= 2e-3 * r.abs().max().item() + 1e-9, k
    assert all(p.grad is None for n, p in enc.named_parameters() if ".w_" not in n)         # the frozen weights get no gradient
    assert len(surgeries) == 2
