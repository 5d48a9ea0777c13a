"""GPU diagnostic sweep: runs every kernel / stage check against torch or the CPU oracle, never stops at a failure,
prints error statistics and quick timings.  Usage on the GPU box:  python tools/gpu_diag.py [section ...]
Sections: gemm norm misc attn encoder decoder post amg perf   (default: all)
"""
import math
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from micro_sam_amd import _lib, modeling, ops  # noqa: E402
import debug_helpers as _debug  # noqa: E402  (tools/debug_helpers.py)
from micro_sam_amd import util as mutil  # noqa: E402
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402
from oracle import sam_ref as S  # noqa: E402

dev = torch.device("cuda")
RESULTS = []


def report(name, ok, detail=""):
    RESULTS.append((name, ok))
    print(f"[{'PASS' if ok else 'FAIL'}] {name} {detail}", flush=True)


def stats(got, ref):
    got, ref = got.float(), ref.float()
    d = (got - ref).abs()
    return f"max|d|={d.max().item():.4g} mean|d|={d.mean().item():.4g} ref_absmax={ref.abs().max().item():.4g} " \
           f"nan={int(torch.isnan(got).sum())}"


def close(got, ref, atol, rtol):
    got, ref = got.float(), ref.float()
    return bool(torch.isfinite(got).all()) and bool(((got - ref).abs() <= atol + rtol * ref.abs()).all())


def bf(x):
    return x.to(torch.bfloat16)


def section(fn):
    def run():
        print(f"\n===== {fn.__name__} =====", flush=True)
        try:
            fn()
        except Exception:
            traceback.print_exc()
            report(fn.__name__ + " (exception)", False)
    run.__name__ = fn.__name__
    return run


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


# --------------------------------------------------------------------------------------------------------------
@section
def gemm():
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K) in [(128, 128, 64), (300, 128, 128), (4096, 768, 768), (4100, 256, 2304), (448, 2048, 256)]:
        a = bf(torch.randn(M, K, generator=g)).to(dev)
        w = bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.float() @ w.float().t() + bias
        for glds in (0, 1):
            out = ops.gemm(a, w, bias, use_glds=glds)
            report(f"gemm f32 M{M} N{N} K{K} glds{glds}", close(out, ref, 1e-3, 1e-4), stats(out, ref))
    # epilogue variants
    M, N, K = 4096 * 2, 256, 128
    a = bf(torch.randn(M, K, generator=g)).to(dev)
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    table = torch.randn(4096, 128, generator=g).to(dev)
    resid_f = torch.randn(M, N, generator=g).to(dev)
    resid_b = bf(torch.randn(4096, N, generator=g)).to(dev)
    base = a.float() @ w.float().t() + bias
    ref = base.clone(); ref[:, :128] += table.repeat(2, 1)
    for glds in (0, 1):
        out = ops.gemm(a, w, bias, table=table, table_cols=128, use_glds=glds)
        report(f"gemm table glds{glds}", close(out, ref, 1e-3, 1e-4), stats(out, ref))
        out = ops.gemm(a, w, bias, resid=resid_f, use_glds=glds)
        report(f"gemm resid f32 glds{glds}", close(out, base + resid_f, 1e-3, 1e-4), stats(out, base + resid_f))
        out = ops.gemm(a, w, bias, resid=resid_b, resid_rows=4096, use_glds=glds)
        r2 = base + resid_b.float().repeat(2, 1)
        report(f"gemm resid bf16 mod glds{glds}", close(out, r2, 1e-3, 1e-4), stats(out, r2))
        out = ops.gemm(a, w, bias, act=ops.ACT_GELU, out_dtype=torch.bfloat16, use_glds=glds)
        r3 = F.gelu(base)
        report(f"gemm gelu bf16 glds{glds}", close(out, r3, 2e-2, 1e-2), stats(out, r3))
        out = ops.gemm(a, w, bias, act=ops.ACT_RELU, out_dtype=torch.bfloat16, use_glds=glds)
        report(f"gemm relu bf16 glds{glds}", close(out, F.relu(base), 2e-2, 1e-2), stats(out, F.relu(base)))
        # in-place residual (out aliases resid)
        x = resid_f.clone()
        ops.gemm(a, w, bias, resid=x, out=x, use_glds=glds)
        report(f"gemm in-place resid glds{glds}", close(x, base + resid_f, 1e-3, 1e-4), stats(x, base + resid_f))
    # qkv split
    B, heads, D = 2, 12, 768
    a = bf(torch.randn(B * 4096, D, generator=g)).to(dev)
    w = bf(torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(dev)
    bias = torch.randn(3 * D, generator=g).to(dev)
    ref = (a.float() @ w.float().t() + bias).reshape(B, 4096, 3, heads, 64).permute(2, 0, 3, 1, 4)
    for glds in (0, 1):
        q, k, v = ops.gemm_qkv(a, w, bias, B, heads, use_glds=glds)
        for nm, t, r in (("q", q, ref[0]), ("k", k, ref[1]), ("v", v, ref[2])):
            report(f"gemm qkv-split {nm} glds{glds}", close(t, r, 3e-2, 1e-2), stats(t, r))
    # kv split (transposed v)
    a = bf(torch.randn(2 * 4096, 256, generator=g)).to(dev)
    w = bf(torch.randn(256, 256, generator=g) / 16).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    table = torch.randn(4096, 128, generator=g).to(dev)
    ref = a.float() @ w.float().t() + bias
    ref[:, :128] += table.repeat(2, 1)
    for glds in (0, 1):
        k, vT = ops.gemm_kv(a, w, bias, table, 4096, use_glds=glds)
        report(f"gemm kv-split k glds{glds}", close(k, ref[:, :128], 3e-2, 1e-2), stats(k, ref[:, :128]))
        rv = ref[:, 128:].reshape(2, 4096, 128).permute(0, 2, 1)
        report(f"gemm kv-split vT glds{glds}", close(vT, rv, 3e-2, 1e-2), stats(vT, rv))


@section
def norm():
    g = torch.Generator().manual_seed(1)
    for dim in (64, 256, 768, 1024, 1280, 96):
        rows = 1000 if dim != 64 else 4096
        x = (torch.randn(rows, dim, generator=g) * 3 + 1).to(dev)
        w = torch.randn(dim, generator=g).to(dev); b = torch.randn(dim, generator=g).to(dev)
        ref = F.layer_norm(x, (dim,), w, b, eps=1e-6)
        out = ops.layernorm(x, w, b, 1e-6)
        report(f"layernorm f32 dim{dim}", close(out, ref, 2e-5, 1e-5), stats(out, ref))
        out = ops.layernorm(x, w, b, 1e-6, out_dtype=torch.bfloat16, gelu=True)
        report(f"layernorm bf16+gelu dim{dim}", close(out, F.gelu(ref), 2e-2, 1e-2), stats(out, F.gelu(ref)))
    x = torch.randn(2 * 4096, 256, generator=g).to(dev)
    w = torch.randn(256, generator=g).to(dev); b = torch.randn(256, generator=g).to(dev)
    out = ops.layernorm(x, w, b, 1e-6, nchw_hw=4096)
    ref = F.layer_norm(x, (256,), w, b, eps=1e-6).reshape(2, 4096, 256).permute(0, 2, 1)
    report("layernorm nchw", close(out, ref, 2e-5, 1e-5), stats(out, ref))


@section
def misc():
    g = torch.Generator().manual_seed(2)
    img = torch.randn(2, 3, 1024, 1024, generator=g).to(dev)
    out = ops.patchify(img)
    ref = F.unfold(img, kernel_size=16, stride=16).permute(0, 2, 1).reshape(2 * 4096, 768)
    report("patchify", bool((out.float() == bf(ref).float()).all()), stats(out, ref))
    u8 = torch.randint(0, 256, (2, 700, 1024, 3), generator=g, dtype=torch.uint8).to(dev)
    out = ops.patchify_u8(u8)
    pre = S.preprocess(u8.permute(0, 3, 1, 2))
    ref = F.unfold(pre, kernel_size=16, stride=16).permute(0, 2, 1).reshape(2 * 4096, 768)
    report("patchify_u8", bool((out.float() == bf(ref).float()).all()), stats(out, ref))
    x = bf(torch.randn(2, 64, 64, 256, generator=g)).to(dev)
    out = ops.im2col3x3(x)
    ref = F.unfold(x.float().permute(0, 3, 1, 2), kernel_size=3, padding=1)          # [B, C*9, 4096] (c, ky, kx)
    ref = ref.reshape(2, 256, 9, 4096).permute(0, 3, 2, 1).reshape(2 * 4096, 9 * 256)
    report("im2col3x3", bool((out.float() == ref).all()), stats(out, ref))


def _attn_ref(x_bf, sd_blk, heads, window):
    """Oracle attention (bf16 rounding points) on LN output x [B,64,64,D] with identity out-projection."""
    D = x_bf.shape[-1]
    sd = {"a.qkv.weight": sd_blk["qkv_w"], "a.qkv.bias": sd_blk["qkv_b"], "a.rel_pos_h": sd_blk["rel_h"],
          "a.rel_pos_w": sd_blk["rel_w"], "a.proj.weight": torch.eye(D, device=x_bf.device),
          "a.proj.bias": torch.zeros(D, device=x_bf.device)}
    p = S.Prec("bf16")
    y = x_bf.float()
    if window:
        yw, pad_hw = S._window_partition(y, 14)
        o = S._attention_relpos(sd, "a.", yw, heads, p)
        o = S._window_unpartition(o, 14, pad_hw, (64, 64))
    else:
        o = S._attention_relpos(sd, "a.", y, heads, p)
    return o


@section
def attn():
    g = torch.Generator().manual_seed(3)
    B, heads, D = 1, 12, 768
    x = bf(torch.randn(B, 64, 64, D, generator=g)).to(dev)
    for window in (True, False):
        Sz = 14 if window else 64
        blk = {"qkv_w": (torch.randn(3 * D, D, generator=g) * 1.3 / math.sqrt(D)).to(dev),
               "qkv_b": (torch.randn(3 * D, generator=g) * 0.3).to(dev),
               "rel_h": (torch.randn(2 * Sz - 1, 64, generator=g) * 0.08).to(dev),
               "rel_w": (torch.randn(2 * Sz - 1, 64, generator=g) * 0.08).to(dev)}
        q, k, v = ops.gemm_qkv(x.reshape(-1, D), bf(blk["qkv_w"]), blk["qkv_b"], B, heads)
        if window:
            out = ops.window_attention(q, k, v, bf(blk["rel_h"]), bf(blk["rel_w"]), blk["qkv_b"])
        else:
            out = ops.global_attention(q, k, v, bf(blk["rel_h"]), bf(blk["rel_w"]))
        ref = _attn_ref(x, blk, heads, window).reshape(-1, D)
        # p.linear with identity proj rounds the attention output to bf16 like the kernel's store
        report(f"{'window' if window else 'global'} attention", close(out, ref, 2e-2, 2e-2), stats(out, ref))
        # per-head breakdown if it failed
        if not close(out, ref, 2e-2, 2e-2):
            d = (out.float() - ref).abs().reshape(4096, heads, 64)
            print("   per-head max err:", [round(float(x), 3) for x in d.amax(dim=(0, 2))])
            print("   per-row-block max err (64 rows):", [round(float(x), 3) for x in d.reshape(64, 64, heads, 64).amax(dim=(1, 2, 3))][:16])


def _model(sd, model_type="vit_b", glds=0):
    sam = modeling.build_sam(model_type)
    sam.load_state_dict(sd)
    sam.to(dev)
    sam.use_glds = glds
    sam.image_encoder.use_glds = glds
    return sam


@section
def encoder():
    sd = synthetic_state_dict("vit_b", 0)
    img = A.to_image(synthetic_tile(0))
    x = S.preprocess(torch.as_tensor(img).permute(2, 0, 1)[None])
    t0 = time.time()
    with torch.no_grad():
        ref_b, taps_b = S.image_encoder(sd, x, precision="bf16", return_blocks=True)
        ref_f = S.image_encoder(sd, x, precision="fp32")
    print(f"oracle encoder x2: {time.time() - t0:.1f}s", flush=True)
    for glds in (0, 1):
        sam = _model(sd, glds=glds)
        for tb in (0, 2, 11):
            out, tap = sam.image_encoder(x.to(dev), tap_block=tb)
            r = taps_b[tb].reshape(-1, 768)
            report(f"encoder residual stream after block {tb} (vs bf16 oracle) glds{glds}",
                   close(tap.cpu(), r, 0.05 * r.abs().max().item() + 0.05, 0.0), stats(tap.cpu(), r))
        out = sam.image_encoder(x.to(dev)).cpu()
        report(f"encoder out vs bf16-mode oracle glds{glds}", close(out, ref_b, 0.08, 0.0), stats(out, ref_b))
        print("   vs fp32 oracle:", stats(out, ref_f), " | bf16-oracle vs fp32-oracle:", stats(ref_b, ref_f))
        u8 = torch.as_tensor(img)[None].to(dev)
        out8 = sam.image_encoder.forward_u8(u8).cpu()
        report(f"encoder u8 path == f32 path glds{glds}", close(out8, out, 1e-5, 0), stats(out8, out))
    torch.save({"emb_hip": out, "emb_ref_bf16": ref_b, "emb_ref_f32": ref_f}, os.path.join(ROOT, "gpurun_out", "enc.pt"))


@section
def decoder():
    sd = synthetic_state_dict("vit_b", 0)
    encp = os.path.join(ROOT, "gpurun_out", "enc.pt")
    if os.path.exists(encp):
        feats = torch.load(encp)["emb_ref_bf16"]
    else:
        feats = torch.randn(1, 256, 64, 64, generator=torch.Generator().manual_seed(5))
    g = torch.Generator().manual_seed(4)
    P = 8
    pts = (torch.rand(P, 1, 2, generator=g) * 1024)
    lbl = torch.ones(P, 1, dtype=torch.int)
    dbg_b = {}
    with torch.no_grad():
        _, iou_b, low_b = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True,
                                          precision="bf16", debug=dbg_b)
        _, iou_f, low_f = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True)
    sam = _model(sd)
    # dense PE
    pe = sam.prompt_encoder.get_dense_pe().cpu()
    report("dense_pe", close(pe, S.get_dense_pe(sd), 2e-4, 0), stats(pe, S.get_dense_pe(sd)))
    # staged debug runs
    for nl in (0, 1, 2):
        os.environ["MSAM_DEBUG_DEC_LAYERS"] = str(nl)
        try:
            sam.decode(feats.to(dev), pts.to(dev), lbl.to(dev))
            torch.cuda.synchronize()
            v = _debug.decoder_workspace_views(sam._dec_ws, P, 7)
            if nl == 0:
                report("decoder tokens (prompt encoder)", close(v["qpe"].cpu(), dbg_b["tokens"], 2e-4, 1e-4),
                       stats(v["qpe"].cpu(), dbg_b["tokens"]))
            else:
                q_ref, k_ref = dbg_b[f"queries{nl - 1}"], dbg_b[f"keys{nl - 1}"]
                report(f"decoder queries after layer {nl - 1}", close(v["queries"].cpu(), q_ref, 0.05, 0.02),
                       stats(v["queries"].cpu(), q_ref))
                report(f"decoder keys after layer {nl - 1}", close(v["keys"].cpu(), k_ref, 0.06, 0.02),
                       stats(v["keys"].cpu(), k_ref))
        finally:
            del os.environ["MSAM_DEBUG_DEC_LAYERS"]
    for glds in (0, 1):
        sam.use_glds = glds
        sam.invalidate()
        low, iou = sam.decode(feats.to(dev), pts.to(dev), lbl.to(dev))
        torch.cuda.synchronize()
        v = _debug.decoder_workspace_views(sam._dec_ws, P, 7)
        if glds == 0:
            report("decoder final queries", close(v["queries"].cpu(), dbg_b["queries_final"], 0.05, 0.02),
                   stats(v["queries"].cpu(), dbg_b["queries_final"]))
            report("decoder hyper", close(v["hyper"][:, :, :32].cpu(), dbg_b["hyper"], 0.05 * dbg_b["hyper"].abs().max().item(), 0.02),
                   stats(v["hyper"][:, :, :32].cpu(), dbg_b["hyper"]))
            up1_ref = dbg_b["up1"].reshape(P, 64, 64, 2, 64, 2).permute(0, 2, 4, 3, 5, 1).reshape(P, 4096, 4, 64)
            report("decoder up1 (ConvT1+LN+GELU)", close(v["up1"].cpu(), up1_ref, 0.05, 0.02), stats(v["up1"].cpu(), up1_ref))
        scale = low_b.abs().max().item()
        report(f"decoder low_res vs bf16-mode oracle glds{glds}", close(low.cpu(), low_b, 0.01 * scale, 0.0), stats(low.cpu(), low_b))
        report(f"decoder iou vs bf16-mode oracle glds{glds}", close(iou.cpu(), iou_b, 2e-3, 0), stats(iou.cpu(), iou_b))
        agree = ((low.cpu() > 0) == (low_b > 0)).float().mean().item()
        agree_f = ((low.cpu() > 0) == (low_f > 0)).float().mean().item()
        print(f"   low-res sign agreement: vs bf16 oracle {agree:.6f}, vs fp32 oracle {agree_f:.6f}; "
              f"bf16-oracle vs fp32-oracle {((low_b > 0) == (low_f > 0)).float().mean().item():.6f}")
    # single-mask + box prompt
    bx = torch.tensor([[100., 100., 400., 300.], [600., 200., 900., 700.]])
    with torch.no_grad():
        _, iou_r, low_r = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), None, None, boxes=bx,
                                          multimask_output=False, return_logits=True, precision="bf16")
    sam.use_glds = 0
    sam.invalidate()
    low, iou = sam.decode(feats.to(dev), None, None, boxes=bx.to(dev), multimask_output=False)
    report("decoder box prompt, single mask", close(low.cpu(), low_r, 0.01 * low_r.abs().max().item(), 0), stats(low.cpu(), low_r))


@section
def post():
    g = torch.Generator().manual_seed(6)
    low = (torch.randn(6, 256, 256, generator=g) * 3)
    low = F.avg_pool2d(low[None], 5, stride=1, padding=2)[0] * 4      # smoother fields
    low[4] = -5.0                                                     # empty mask
    low[5] = 5.0                                                      # full mask
    for (in_hw, out_hw) in (((1024, 1024), (1024, 1024)), ((1024, 768), (1024, 768)), ((1024, 1024), (512, 512)),
                            ((683, 1024), (400, 600))):
        ref_logits = S.postprocess_masks(low[None], in_hw, out_hw)[0]
        res = ops.postprocess_masks(low.to(dev), in_hw, out_hw, 0.0, 1.0, want_logits=True)
        lg = res["logits"].cpu()
        exact = bool((lg == ref_logits).all())
        report(f"postprocess logits bit-exact {in_hw}->{out_hw}", exact, stats(lg, ref_logits))
        m = ref_logits > 0.0
        counts_ref = torch.stack([(ref_logits > 1.0).sum((1, 2)), (ref_logits > -1.0).sum((1, 2)), m.sum((1, 2))], 1).int()
        report(f"postprocess counts {out_hw}", bool((res["counts"].cpu() == counts_ref).all()),
               f"{res['counts'].cpu().tolist()} vs {counts_ref.tolist()}")
        boxes_ref = A.batched_mask_to_box(m)
        report(f"postprocess boxes {out_hw}", bool((res["boxes"].cpu() == boxes_ref).all()),
               f"{res['boxes'].cpu().tolist()} vs {boxes_ref.tolist()}")
        um = ops.unpack_bits(res["bits"], out_hw[0]).cpu()
        report(f"postprocess bits {out_hw}", bool((um == m).all()), f"mismatch={(um != m).sum().item()}")
        counts, offsets = ops.rle_encode(res["bits"], out_hw[0], out_hw[1])
        rles = ops.rles_to_list(counts, offsets, out_hw[0], out_hw[1])
        rles_ref = A.mask_to_rle(m)
        report(f"rle {out_hw}", rles == rles_ref, f"runs={[len(r['counts']) for r in rles]}")
    # noisy masks (many runs) + vendored API
    from micro_sam_amd import _vendored
    m = torch.rand(3, 300, 200, generator=g) > 0.5
    m[0, 0, 0] = True
    r = _vendored.mask_to_rle_pytorch(m.to(dev))
    report("rle noisy masks (vendored API)", r == A.mask_to_rle(m), f"runs={[len(x['counts']) for x in r]}")
    b = _vendored.batched_mask_to_box(m.to(dev)).cpu()
    report("batched_mask_to_box", bool((b == A.batched_mask_to_box(m)).all()))


@section
def amg():
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    sd = synthetic_state_dict("vit_b", 0)
    tile = synthetic_tile(0)
    predictor = mutil.get_sam_model("vit_b", device="cuda", state_dict=sd)
    t0 = time.time()
    emb = mutil.precompute_image_embeddings(predictor, tile, verbose=False)
    amg_ = AutomaticMaskGenerator(predictor)
    amg_.initialize(tile, emb)
    torch.cuda.synchronize()
    t1 = time.time()
    seg = amg_.generate()
    t2 = time.time()
    print(f"HIP embed+initialize {t1 - t0:.2f}s, generate {t2 - t1:.2f}s (first call, includes weight prep)")
    d = amg_.crop_list[0]
    st = d["stability_score"].cpu()
    print("   masks:", len(d["rles"]), "iou>0.88:", int((d["iou_preds"].cpu() > 0.88).sum()), "stab>=0.95:", int((st >= 0.95).sum()),
          "instances:", int(seg.max()), "median runs:", int(np.median([len(r["counts"]) for r in d["rles"]])))
    # oracle on the HIP embedding (isolates decoder+post) with 128 of the 1024 prompts
    feats = torch.as_tensor(emb["features"])
    state = PR.amg_initialize(sd, A.to_image(tile), feats, emb["input_size"], emb["original_size"], precision="bf16",
                              max_batches=2)
    dr = state["crop_list"][0]
    n = len(dr["rles"])
    iou_err = (d["iou_preds"].cpu()[:n] - dr["iou_preds"]).abs().max().item()
    ious = []
    for i in range(n):
        a_, b_ = A.rle_to_mask(d["rles"][i]), A.rle_to_mask(dr["rles"][i])
        u = (a_ | b_).sum()
        ious.append(1.0 if u == 0 else (a_ & b_).sum() / u)
    ious = np.array(ious)
    report("AMG initialize vs bf16-mode oracle: mask IoU >= 0.999 (first 384 masks)", bool((ious >= 0.999).mean() > 0.99),
           f"min={ious.min():.5f} frac>=0.999={np.mean(ious >= 0.999):.4f} iou_pred_err={iou_err:.2e}")
    sb = dr["stability_score"]
    print("   stability max|d|:", (st[:n] - sb).abs().max().item(), " boxes equal frac:",
          (d["boxes"].cpu()[:n] == dr["boxes"]).all(1).float().mean().item())
    # generate() parity: run the oracle generate on the HIP state (integer post-processing must be identical)
    hip_state = amg_.get_state()
    cl = A.MaskData(**{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in hip_state["crop_list"][0].items()})
    seg_ref = PR.amg_generate({"crop_list": [cl], "crop_boxes": hip_state["crop_boxes"], "original_size": hip_state["original_size"]})
    report("AMG generate (host integer post-processing) identical ids", bool(np.array_equal(seg, seg_ref)),
           f"instances {int(seg.max())} vs {int(seg_ref.max())}")


@section
def perf():
    sd = synthetic_state_dict("vit_b", 0)
    sam = _model(sd)
    g = torch.Generator().manual_seed(7)
    for glds in (0, 1):
        sam.image_encoder.use_glds = glds
        sam.image_encoder.invalidate()
        for B in (1, 4):
            u8 = torch.randint(0, 256, (B, 1024, 1024, 3), generator=g, dtype=torch.uint8).to(dev)
            ms = timeit(lambda: sam.image_encoder.forward_u8(u8), n=3, warm=1)
            print(f"encoder B={B} glds{glds}: {ms:.2f} ms  ({0.938 * B / ms:.1f} TFLOP/s algorithmic)", flush=True)
    feats = torch.randn(1, 256, 64, 64, generator=g).to(dev)
    for glds in (0, 1):
        sam.use_glds = glds
        sam.invalidate()
        for P in (64, 256, 1024):
            pts = (torch.rand(P, 1, 2, generator=g) * 1024).to(dev); lbl = torch.ones(P, 1, dtype=torch.int, device=dev)
            ms = timeit(lambda: sam.decode(feats, pts, lbl), n=3, warm=1)
            print(f"decoder P={P} glds{glds}: {ms:.2f} ms  ({3.61e-3 * P / ms:.1f} TFLOP/s algorithmic)", flush=True)
    low = torch.randn(3072, 256, 256, generator=g).to(dev)
    ms = timeit(lambda: ops.postprocess_masks(low, (1024, 1024), (1024, 1024)), n=3, warm=1)
    print(f"postprocess 3072 masks: {ms:.2f} ms", flush=True)
    res = ops.postprocess_masks(F.avg_pool2d(low[None], 9, 1, 4)[0] * 5, (1024, 1024), (1024, 1024))
    ms = timeit(lambda: ops.rle_encode(res["bits"], 1024, 1024), n=3, warm=1)
    print(f"rle 3072 masks: {ms:.2f} ms", flush=True)
    for (M, N, K) in [(4096 * 4, 2304, 768), (4096 * 4, 3072, 768), (4096 * 4, 768, 3072), (4096 * 64, 256, 256), (4096 * 64, 128, 256)]:
        a = bf(torch.randn(M, K, generator=g)).to(dev); w = bf(torch.randn(N, K, generator=g)).to(dev)
        for glds in (0, 1):
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ms = timeit(lambda: ops.gemm(a, w, None, out=out, use_glds=glds), n=5, warm=2)
            print(f"gemm {M}x{N}x{K} glds{glds}: {ms:.3f} ms {2 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)


ALL = {"gemm": gemm, "norm": norm, "misc": misc, "attn": attn, "encoder": encoder, "decoder": decoder, "post": post,
       "amg": amg, "perf": perf}

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    print("device:", torch.cuda.get_device_name(0), "| lib:", _lib.lib_path(), flush=True)
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
    print("\n===== SUMMARY =====")
    for name, ok in RESULTS:
        if not ok:
            print("FAIL:", name)
    print(f"{sum(ok for _, ok in RESULTS)}/{len(RESULTS)} checks passed")
