"""Benchmark of the hot path: 1024^2 tiles/s, embed + AMG (vit_b, bf16 MFMA operands), BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  When the script is NOT already running under a launcher (no WORLD_SIZE in the
environment) it re-executes itself under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1`` and fails loudly if fewer than N GPUs are visible; under a launcher WORLD_SIZE must equal N.

One step = TILES_PER_STEP synthetic tiles per rank through: batched image encoder -> per tile AutomaticMaskGenerator
initialize (32x32 grid prompts, fused mask post-processing to bit masks on the device) -> generate (default thresholds,
box NMS, merge to a uint32 label image); with N > 1 the label tiles of all ranks are all-gathered (RCCL) inside the timed
region.  Inputs (uint8 RGB tiles, output of util._to_image; 256 distinct tiles per rank, visited in order) are resident
in HBM before the timed region.  Prints ONE JSON line on rank 0 with, next to the contract fields:

  roofline          dominant kernel family of the timed region against the MFMA roofline SURVEY.md 8(d) defines
                    (algorithmic FLOP of the reference's formulation per launch / live HIP-event duration / 2.5 PFLOP/s);
                    the family's HBM figures and the other families are kept alongside; whole_path_frac = 4.64 TFLOP x tiles/s
  cpu_baseline      the CPU oracle (restated reference path, fp32 torch) timed on full tiles on the host cores
  mask_iou_vs_ref   per-instance mask IoU of the HIP path against that fp32 CPU reference on the same tiles (the metric says
                    "mask IoU vs ref"): distribution over the instances the reference keeps, keep-set and label agreement
  pcie_inclusive    tiles/s of a separate pass that starts from the raw host tiles (H2D upload, util._to_image on the device)
                    and ends with the label images back in host memory
  config3_side / fp8_side / train_side   (N = 1, default run) BASELINE configs[2], [4], [3] as short side runs of this script /
                    tools/train_bench.py in their own processes after the timed region: the driver's record carries every named
                    configuration; `--no-side` or `--no-config-sides` skips them
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILES_PER_STEP = 64     # 4 encoder batches of 16; the decode lanes start on a batch's tiles while the encoder works on the next batch
ENC_BATCH = 16     # M = 65536 rows: every encoder GEMM is a whole number of 256-workgroup rounds (B = 8 left 1.5-round tails)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0           # MI355X dense fp8 (MX-scaled MFMA), --encoder-dtype fp8 only
PEAK_HBM_GBS = 8000.0              # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s achievable)
TILE_TFLOP_ALGORITHMIC = 4.64      # encoder 0.938 + AMG decode 3.70
# SURVEY.md 8(d) per-prompt algorithmic work of the reference's formulation that each decoder stream kernel replaces
# (GFLOP per prompt per launch): image->token attention of one layer = Q + out projections of 4096 image tokens (0.537);
# token->image attention of layer 1 / final = K + V projections (0.537); up-scaling = ConvT1 0.537 + ConvT2 0.268 + hyper product 0.02
# chained kernels (csrc/decfold_tok.hip): i2t0_t2i replaces the image->token attention of layer 0 AND the token->image attention of
# layer 1 (2 x 0.537; it also recomputes nothing algorithmic), i2t01 the image->token attention of layer 1 (0.537; its
# recomputation of the layer-0 tile is executed, not algorithmic, work)
ALG_GFLOP_PER_PROMPT = {2: 0.537, 3: 0.537, 4: 0.825, 6: 1.074, 7: 0.537}


LINE_LIMIT = 6144          # the driver keeps an 8000-character stdout tail: the final line must fit it with room to spare


def _finite(o):
    """Strict-JSON form of a result tree: NaN / +-inf become null (json.dumps would otherwise print non-standard tokens)."""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, (np.floating,)):
        return _finite(float(o))
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def _parity_summary(rep):
    """The six numbers of one parity leg the final line carries (everything else: bench_extras)."""
    if not isinstance(rep, dict):
        return rep
    if "error" in rep:
        return {"error": str(rep["error"])[:120]}
    iou = rep.get("iou", rep)
    labels = rep.get("labels", {}) or {}
    r4 = lambda v: round(v, 4) if isinstance(v, float) else v                     # noqa: E731
    out = {"n_instances": iou.get("n_instances"), "frac_ge_0.999": r4(iou.get("frac_ge_0.999")), "min": r4(iou.get("min")),
           "keep_set": iou.get("keep_set"), "identical_id_frac_foreground": r4(labels.get("identical_id_frac_foreground"))}
    for k in ("tiles_per_s", "tiles_per_s_segment_slices"):
        if k in rep:
            out[k] = rep[k]
    return out


def _side_summary(rec, keys=("value", "unit")):
    if not isinstance(rec, dict):
        return rec
    if "error" in rec:
        return {"error": str(rec["error"])[:120]}
    return {k: rec[k] for k in keys if k in rec}


def compact_line(out, extras_path=None):
    """The ONE final stdout line: contract fields + `config` + `roofline` (dominant kernel only) + `cpu_baseline` + six numbers per parity leg
    + value/unit per side run, strict JSON (no NaN tokens), shorter than LINE_LIMIT.  The full result tree (worst-instance lists, the other
    kernel families, stage tables, side anatomy) goes to `extras_path` and to an EARLIER stdout line (the 24 KB line of round 5
    overflowed the driver's stdout tail and was recorded as unparsed)."""
    out = _finite(out)
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    clip = lambda v, n=300: (v[:n] if isinstance(v, str) else v)                  # noqa: E731
    line = {k: clip(v) for k, v in line.items()}
    cfg = out.get("config") or {}
    line["config"] = {k: clip(cfg[k]) for k in ("workload", "tiles_per_step_per_gpu", "encoder_batch", "distinct_tiles_per_gpu", "weights", "parallelism",
                                           "decode_lanes", "pipelined_labels_equal_serial", "instances_per_tile", "precision_mode") if k in cfg}
    roof = out.get("roofline")
    if isinstance(roof, dict):
        line["roofline"] = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_us",
                                                  "avg_launch_gflop_algorithmic", "hbm_gbytes_per_s", "whole_path_tflops", "whole_path_frac")
                            if k in roof}
        if isinstance(line["roofline"].get("kernel"), str):
            line["roofline"]["kernel"] = line["roofline"]["kernel"][:100]
        if roof.get("traffic") is None and roof.get("traffic_source"):
            line["roofline"]["traffic_note"] = str(roof["traffic_source"])[:160]
    else:
        line["roofline"] = roof
    cpu = out.get("cpu_baseline")
    line["cpu_baseline"] = ({k: (str(cpu[k])[:200] if k == "sample" else cpu[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
                            if isinstance(cpu, dict) else cpu)
    for k in ("mask_iou_vs_ref", "mask_iou_vs_ref_strict", "mask_iou_vs_ref_split16", "mask_iou_vs_ref_fp16_encoder", "mask_iou_vs_ref_plain_bf16_encoder"):
        if k in out:
            line[k] = _parity_summary(out[k])
    tp = out.get("mask_iou_vs_ref_trained")
    if isinstance(tp, dict):
        line["mask_iou_vs_ref_trained"] = ({"error": str(tp["error"])[:120]} if "error" in tp else
                                           {m: _parity_summary(tp[m]) for m in ("default", "split16", "strict") if m in tp})
    for k in ("pcie_inclusive", "api_inclusive", "config3_side", "split16_side", "fp8_side"):
        if k in out:
            line[k] = _side_summary(out[k])
    if isinstance(out.get("interactive_side"), dict):
        line["interactive_side"] = _side_summary(out["interactive_side"], ("predict_ms_median",))
    ts = out.get("train_side")
    if isinstance(ts, dict):
        line["train_side"] = {m: _side_summary(r, ("value", "unit", "non_hip_device_time_frac", "kernel_launches_per_step")) for m, r in ts.items()}
    if extras_path:
        line["extras"] = extras_path
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:            # never let the line outgrow the driver's tail again: shed the optional blocks, largest first
        for k in sorted((k for k in line if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                       "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")),
                        key=lambda k: -len(json.dumps(line[k]))):
            line[k] = "see extras"
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    return text


def emit(out, name="bench_extras.json"):
    """Write the full result tree to gpurun_out/<name> and to an earlier stdout line (prefixed, so that it is never taken for the bench
    line), then print the compact final line."""
    rel = os.path.join("gpurun_out", name)
    full = json.dumps(_finite(out), allow_nan=False)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as fh:
            fh.write(full + "\n")
    except OSError as exc:
        log(f"bench.py: could not write {rel}: {exc}")
        rel = None
    print("bench_extras: " + full, flush=True)
    print(compact_line(out, rel), flush=True)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _make_tile(seed):
    from micro_sam_amd.synthetic import synthetic_tile
    return synthetic_tile(seed)


def make_tiles(seeds):
    """Distinct synthetic tiles (SURVEY.md 8(d) config 2 generator); a process pool, started before CUDA is touched."""
    import multiprocessing as mp
    n_proc = max(1, min(16, (os.cpu_count() or 2) // 2, len(seeds)))
    if n_proc == 1:
        return [_make_tile(s) for s in seeds]
    with mp.get_context("fork").Pool(n_proc) as pool:
        return pool.map(_make_tile, seeds, chunksize=4)


def cpu_reference(sd, tiles_np, n_threads):
    """The CPU oracle (restated reference hot path, fp32 torch) on FULL tiles: embedding + 16 decoder batches of 64 prompts +
    mask post-processing + generate, no extrapolation.  Returns the baseline record and the per-tile states / label images
    (the reference side of mask_iou_vs_ref).  Reported only, never used as a target."""
    from oracle import amg_ref as A
    from oracle import pipeline_ref as PR
    torch.set_num_threads(n_threads)
    per_tile, states, segs, stages = [], [], [], []
    for tile in tiles_np:
        img = A.to_image(tile)
        t0 = time.perf_counter()
        feats, osz, isz = PR.compute_embeddings(sd, [img], "vit_b", "fp32")
        t1 = time.perf_counter()
        tm = {}
        state = PR.amg_initialize(sd, img, feats, isz[0], osz[0], precision="fp32", timings=tm)
        t2 = time.perf_counter()
        seg = PR.amg_generate(state)
        t3 = time.perf_counter()
        per_tile.append(t3 - t0)
        stages.append({"encoder": round(t1 - t0, 2), "decode_x16": round(tm.get("decode", 0.0), 2),
                       "mask_data_x16": round(tm.get("mask_data", 0.0), 2), "generate": round(t3 - t2, 2)})
        states.append(state); segs.append(seg)
        log(f"  cpu reference tile: {t3 - t0:.1f} s {stages[-1]}")
    med = float(np.median(per_tile))
    rec = {"value": round(1.0 / med, 5), "unit": "tiles/s", "cores": n_threads, "kind": "port",
           "sample": f"{len(tiles_np)} full tiles (encoder + 16 x 64 prompts + post-processing + generate each), median "
                     f"{med:.1f} s per tile, fp32 torch on {n_threads} host threads; no extrapolation",
           "seconds_per_tile": [round(t, 2) for t in per_tile], "stages_seconds": stages}
    return rec, states, segs


def mask_iou_vs_ref(predictor, amg, tiles_np, ref_states, ref_segs):
    """Per-instance mask IoU of the HIP path against the fp32 CPU reference (oracle/parity.py) on the same tiles."""
    from micro_sam_amd import ops, util
    from oracle import parity as PT
    reps, labs = [], []
    for tile, ref, seg_ref in zip(tiles_np, ref_states, ref_segs):
        emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
        amg.initialize(tile, emb)
        data = amg.crop_list[0]
        n = len(data)
        cand = data.shallow_copy()
        cand["cand"] = torch.arange(n, device=data["iou_preds"].device)
        kept_test = amg._postprocess_batch(cand, amg.crop_boxes[0], amg.original_size, 0.88, 0.95, 0.7)["cand"].cpu().numpy()
        kept_ref = PT.kept_candidates(ref)
        h = amg.original_size[0]
        bits = data["bits"]
        test_scores = {"iou_pred": data["iou_preds"].float().cpu().numpy(), "stability": data["stability_score"].float().cpu().numpy()}
        rep = PT.iou_report(kept_ref, kept_test, PT.oracle_mask_fn(ref),
                            lambda i: ops.unpack_bits(bits[i:i + 1], h)[0].cpu().numpy(),
                            PT.oracle_scores(ref), test_scores)
        reps.append(rep)
        labs.append(PT.label_agreement(seg_ref, amg.generate().astype(seg_ref.dtype)))
    out = PT.public(PT.merge_reports(reps))
    out["tiles"] = len(reps)
    out["labels"] = {k: (round(float(np.mean([la[k] for la in labs])), 5) if "frac" in k or "agreement" in k else
                         [la[k] for la in labs]) for k in labs[0]}
    out["reference"] = "fp32 CPU oracle (restated reference path) on the same tiles and weights"
    return out


def strict_leg(predictor, amg, tiles_np, ref_states, ref_segs, cpu_tiles_per_s, mode="strict", slice_tiles=16):
    """mask_iou_vs_ref in a reference-formulation precision mode ("strict": fp32 kernels; "split16": every product on fp16 operand pairs) + that
    mode's throughput two ways: the literal per-tile API loop (precompute_image_embeddings -> AutomaticMaskGenerator.initialize -> generate,
    host arrays in and out, one lane) and the product's own slice loop (multi_dimensional_segmentation.segment_slices: encoder batches of 8,
    decode lanes)."""
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd import util
    predictor.set_precision(mode)
    rep = mask_iou_vs_ref(predictor, amg, tiles_np, ref_states, ref_segs)        # (also the warm-up of the mode's kernels)

    def whole(tile):
        emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
        amg.initialize(tile, emb)
        return amg.generate()
    n = 6
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        whole(tiles_np[k % len(tiles_np)])
    torch.cuda.synchronize()
    tps = n / (time.perf_counter() - t0)
    rep["tiles_per_s"] = round(tps, 2)
    rep["times_cpu_baseline"] = round(tps / cpu_tiles_per_s, 1) if cpu_tiles_per_s else None
    try:
        stack = np.stack([tiles_np[k % len(tiles_np)] for k in range(slice_tiles)])
        mds.segment_slices(stack[:8], predictor, amg, batch_size=8)              # warm-up: lane streams and their workspaces
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mds.segment_slices(stack, predictor, amg, batch_size=8)
        torch.cuda.synchronize()
        rep["tiles_per_s_segment_slices"] = round(slice_tiles / (time.perf_counter() - t0), 2)
    except Exception as exc:                # a side measurement must not cost the parity report
        rep["tiles_per_s_segment_slices"] = None
        rep["segment_slices_error"] = repr(exc)[:200]
    rep["mode"] = {"strict": "predictor.set_precision('strict'): image encoder, prompt encoder and mask decoder in the reference's formulation on fp32 "
                             "kernels (f32-input MFMA products, erf GELU, expf softmax)",
                   "split16": "predictor.set_precision('split16'): the same formulation, every product on fp16 operand pairs (hi + lo of each fp32 "
                              "operand, 3 MFMAs of the 16-bit pipe, fp32 accumulation; LayerNorm / softmax / GELU in fp32) with the up-scaling's second half, "
                              "the image->token block and the k|v projections fused"}[mode] + \
                  "; tiles_per_s = the literal per-tile API loop on one lane, tiles_per_s_segment_slices = one segment_slices call over 16 tiles"
    return rep


def csrc_sha16():
    """sha256 (first 16 hex digits) of the kernel sources + the ABI header: names the code a PMC / rocprof table was measured on
    (tools/csrc_sha.py prints the same on the GPU box; tools/summarize_profiles.py stores it in profiles/<tag>_pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "micro_sam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "msam_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC table - only when that table was measured on THIS code
    (csrc_sha16 recorded with it); a table of other code is refused (a round-1 constant was shipped in round 2's line)."""
    sha = csrc_sha16()
    best = None
    try:
        names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_pmc_traffic.json"))
    except OSError:
        names = []
    for name in reversed(names):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        meta = pmc.get("_meta", {})
        for kname, rec in pmc.items():
            if kname != "_meta" and isinstance(rec, dict) and kname.split("<")[0] in kernel_name:
                if meta.get("csrc_sha16") == sha:
                    return rec.get("hbm_bytes_per_launch"), f"profiles/{name} (measured on this code, csrc_sha16 {sha})"
                best = best or (f"profiles/{name} holds a table for this kernel measured on other code "
                                f"(csrc_sha16 {meta.get('csrc_sha16')}, this code {sha}): not reported")
    return None, best or "no PMC table for this kernel under profiles/"


def api_inclusive(predictor, amg, tiles_np, n_api, enc_batch):
    """SURVEY.md 8(d) config 2 through the drop-in API itself, host arrays in and out, two ways:
    value          the product's own slice loop, ONE call: multi_dimensional_segmentation.segment_slices(stack, predictor, amg,
                   batch_size) (reference _segment_slices :385-416: embeddings + per slice initialize / generate with running id offsets)
                   - inside the product it is a device pipeline (encoder batches, decode lanes, double-buffered label download)
    literal_loop   the same calls written out by the caller: precompute_image_embeddings(stack, ndim=3, batch_size) -> per slice
                   AutomaticMaskGenerator.initialize(i=z) + generate() (numpy label image per slice: one synchronisation per slice)"""
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd import util
    stack = np.stack(tiles_np[:n_api])
    mds.segment_slices(stack[:4], predictor, amg, batch_size=enc_batch)          # warm-up: lane streams, page-locked buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seg, _ = mds.segment_slices(stack, predictor, amg, batch_size=enc_batch)
    torch.cuda.synchronize()
    t_pipe = time.perf_counter() - t0
    loop_segs = np.zeros(stack.shape, dtype=np.uint32)          # (allocated and touched before the clock, as a caller's result volume is)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = util.precompute_image_embeddings(predictor, stack, ndim=3, batch_size=enc_batch, verbose=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for z in range(n_api):
        amg.initialize(stack[z], emb, i=z)
        loop_segs[z] = amg.generate()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n_inst, offset, differing = 0, 0, []
    for z, s in enumerate(loop_segs):               # (outside the clock) the pipelined volume == the caller's loop + running offsets
        m = int(s.max())
        n_inst += m
        if not np.array_equal(np.where(s != 0, s + np.uint32(offset), 0).astype(np.uint32), seg[z]):
            differing.append(z)
        offset += m
    same = not differing
    if differing:
        log(f"api_inclusive: pipelined volume differs from the caller's loop in slices {differing}")
        if os.environ.get("MSAM_API_DEBUG"):
            third, _ = mds.segment_slices(stack, predictor, amg, batch_size=enc_batch, decode_lanes=0)
            again, _ = mds.segment_slices(stack, predictor, amg, batch_size=enc_batch)
            z = differing[0]
            off = int(sum(int(loop_segs[k].max()) for k in range(z)))
            exp = np.where(loop_segs[z] != 0, loop_segs[z] + np.uint32(off), 0).astype(np.uint32)
            log(f"  slice {z}: px differing pipeline/loop {int((exp != seg[z]).sum())}, serial-product/loop {int((exp != third[z]).sum())}, "
                f"pipeline-again/loop {int((exp != again[z]).sum())}, pipeline/pipeline-again {int((seg[z] != again[z]).sum())}; max ids loop "
                f"{int(loop_segs[z].max())} pipeline {int(seg[z].max())} serial-product {int(third[z].max())}; fg px loop {int((exp != 0).sum())} pipeline {int((seg[z] != 0).sum())}")
    return {"value": round(n_api / t_pipe, 2), "unit": "tiles/s", "tiles": n_api,
            "what": f"multi_dimensional_segmentation.segment_slices(stack[{n_api},1024,1024] uint8 host array, predictor, "
                    f"AutomaticMaskGenerator, batch_size={enc_batch}) -> uint32 [Z,Y,X] host volume with the serial loop's running id "
                    "offsets; ONE product call (the reference's _segment_slices), pipelined inside the product",
            "labels_equal_literal_loop": same, "instances_per_tile_mean": round(n_inst / n_api, 1),
            "literal_loop": {"value": round(n_api / (t2 - t0), 2), "unit": "tiles/s",
                             "embed_seconds_per_tile": round((t1 - t0) / n_api, 5), "amg_seconds_per_tile": round((t2 - t1) / n_api, 5),
                             "what": f"util.precompute_image_embeddings(stack, ndim=3, batch_size={enc_batch}) then per slice "
                                     "AutomaticMaskGenerator.initialize(stack[z], emb, i=z) + generate() -> numpy label image; the caller's "
                                     "loop synchronises wherever the API returns host data"}}


def config_sides(timeout_s: float = 420.0, fp8: bool = False):
    """BASELINE configs[2] / [4] / [3] as short runs in their own processes (clean allocator, their own model): the last JSON line of
    each, reduced to the fields that name the measurement.  A failing or slow side run costs its field, never the bench line."""
    def run(cmd, keep):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode or not line:
                return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
            d = json.loads(line[-1])
            out = {k: d[k] for k in keep if k in d}
            out["command"] = "python " + " ".join(cmd)
            out["wall_seconds"] = round(time.perf_counter() - t0, 1)
            return out
        except Exception as exc:
            return {"error": repr(exc)[:300]}
    sides = {}
    # the parity-meeting precision mode in the headline's own harness: tiles resident in HBM, 3 decode lanes, label tiles in HBM (mask_iou_vs_ref_split16
    # carries its parity; its tiles_per_s_segment_slices is the host-array-in / host-volume-out rate of the same mode)
    sides["split16_side"] = run(["bench.py", "--precision", "split16", "--no-cpu-baseline", "--no-side", "--steps", "2", "--warmup", "1", "--tiles-per-step", "32",
                                 "--distinct-tiles", "64", "--enc-batch", "8"], ("metric", "value", "unit", "ms_per_step"))
    if fp8:         # configs[4]: a precision mode, not a performance mode (rounds 3-5: 175 vs 170-174 tiles/s; DESIGN "fp8") - on request only
        sides["fp8_side"] = run(["bench.py", "--encoder-dtype", "fp8", "--no-cpu-baseline", "--no-side", "--steps", "3", "--warmup", "1"],
                                ("metric", "value", "unit", "ms_per_step", "dtype"))
    sides["config3_side"] = run(["bench.py", "--workload", "config3", "--steps", "3", "--warmup", "1", "--slices", "8"],
                                ("metric", "value", "unit", "ms_per_step", "dtype", "config"))
    # parity on weights that are not hand-designed (tools/trained_parity.py: the "cells" checkpoint after 100 AdamW steps of this package's
    # trainer), default and strict precision mode against the fp32 CPU oracle of the same run
    tp = run([os.path.join("tools", "trained_parity.py"), "--steps", "100", "--strict", "--quartile", "--stability-thresh", "0.8"],
             ("steps", "train_seconds", "checkpoint_digest", "weight_distance_from_designed", "iou", "labels", "split16", "strict", "pred_iou_thresh", "stability_score_thresh",
              "embedding_mean_abs_err", "oracle_seconds"))
    if "iou" in tp:
        tp["default"] = {"iou": tp.pop("iou"), "labels": tp.pop("labels")}
        tp["what"] = ("per-instance mask IoU vs the fp32 CPU oracle on a fine-tuned (non-designed) checkpoint, tile 1000, 16 x 16 prompts, "
                      "thresholds = lower quartile of the reference's predicted IoUs / stability 0.8 (the fine-tuned masks are soft: the "
                      "default thresholds keep nothing); 'default' = the 16-bit path, 'strict' = set_precision('strict')")
    sides["mask_iou_vs_ref_trained"] = tp
    tb = os.path.join("tools", "train_bench.py")
    keep_t = ("metric", "value", "unit", "model", "ms_per_step", "config", "non_hip_device_time_frac", "kernel_launches_per_step")
    sides["train_side"] = {"vit_b": run([tb, "--model", "vit_b", "--steps", "5", "--warmup", "2", "--device-time"], keep_t),
                           "vit_h": run([tb, "--model", "vit_h", "--steps", "4", "--warmup", "1"], keep_t)}
    return sides


def bench_config3(args, rank, world, dev):
    """BASELINE configs[2], one GPU's share: vit_l, Z slices of 2048 x 2048 (the 64-slice volume sharded per slice over 8 GPUs = 8 per
    GPU), tiled embeddings (tile 768 + halo 128: outer tiles up to 1024^2, 9 per slice) + TiledAutomaticMaskGenerator per slice with the
    reference's running id offsets (multi_dimensional_segmentation.segment_slices; the cross-slice merge of
    automatic_3d_segmentation is the host step after the gather).  One step = all Z slices."""
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd import util
    from micro_sam_amd.instance_segmentation import TiledAutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    Z = args.slices
    vol = np.stack([synthetic_tile(3000 + rank * Z + z, (2048, 2048)) for z in range(Z)])
    sd = synthetic_state_dict("vit_l", 0, variant=args.weights)
    predictor = util.get_sam_model("vit_l", device=dev, state_dict=sd)
    predictor.model.image_encoder.set_precision(args.encoder_dtype)
    segmentor = TiledAutomaticMaskGenerator(predictor)          # (one generator for all steps: its decode lanes keep their workspaces)

    def step():
        # batch_size = the tiles of TWO slices: tiles are batched by shape (a slice has 4 + 1 + 2 + 2 tiles of four shapes) and the encoder of
        # the next pair of slices runs underneath the decode lanes of the current pair (round 5: _segment_slices_tiled_overlapped)
        seg, _ = mds.segment_slices(vol, predictor, segmentor, tile_shape=(768, 768), halo=(128, 128), batch_size=18)
        return seg
    for _ in range(args.warmup):
        seg = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seg = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n = Z * args.steps * world
        print(json.dumps({
            "metric": "2048^2 slices/s tiled embed+AMG (vit_l bf16), BASELINE configs[2] per-slice path", "value": round(n / elapsed, 4),
            "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"{args.encoder_dtype} image encoder, " + ("fp16" if _lib_decoder_is_f16() else "bf16") + " mask decoder",
            "data": "synthetic",
            "config": {"workload": f"configs[2] share of one GPU: vit_l, {Z} slices of 2048x2048 uint8, tile_shape 768 + halo 128 "
                                   "(9 tiles per slice), TiledAutomaticMaskGenerator defaults (32x32 grid per tile), host arrays in / "
                                   "uint32 label volume out (multi_dimensional_segmentation.segment_slices)",
                       "tiles_per_second": round(9 * n / elapsed, 2), "instances_in_last_volume": int(seg.max()),
                       "weights": f"seeded synthetic checkpoint (synthetic.py variant '{args.weights}')"},
            "roofline": None, "cpu_baseline": None}), flush=True)


def _lib_decoder_is_f16():
    from micro_sam_amd import _lib
    return _lib.decoder_dtype() == torch.float16


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks."""
    if not args.dry_run:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to run a smaller job "
                             f"under this label")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("bench.py: launching", " ".join(cmd))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU reference leg (cpu_baseline, mask_iou_vs_ref)")
    ap.add_argument("--cpu-tiles", type=int, default=3, help="full tiles the CPU reference runs (median reported)")
    ap.add_argument("--glds", type=int, default=0)
    ap.add_argument("--tiles-per-step", type=int, default=TILES_PER_STEP, help="tiles per rank and step")
    ap.add_argument("--distinct-tiles", type=int, default=256, help="distinct synthetic tiles per rank, visited in order")
    ap.add_argument("--enc-batch", type=int, default=ENC_BATCH, help="tiles per image-encoder call")
    ap.add_argument("--encoder-dtype", choices=("bf16", "fp16", "fp8"), default="bf16",
                    help="bf16: the headline configuration (\"vit_b bf16\"); fp16: IEEE fp16 instead of bf16 operands in the image encoder "
                         "(same kernels and MFMA rate; side measurement of what the bf16 operand rounding costs in mask IoU); fp8: "
                         "BASELINE config 5 (encoder projections on fp8 e4m3 MX MFMA, bf16 decoder); NOT the headline metric")
    ap.add_argument("--weights", choices=("cells", "blobs", "field"), default="cells",
                    help="synthetic checkpoint variant (micro_sam_amd/synthetic.py)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="tiles are decoded round-robin on this many HIP streams (each with its own predictor state and "
                         "decoder workspace): the latency-bound token-side launches of one tile run underneath the "
                         "streaming kernels of another")
    ap.add_argument("--serial-generate", action="store_true",
                    help="run generate() on the main stream (default: side stream overlapping the next tile's decode)")
    ap.add_argument("--device-chunk", type=int, default=1024,
                    help="grid prompts decoded per decoder pass (results do not depend on it)")
    ap.add_argument("--workload", choices=("config2", "config3"), default="config2",
                    help="config2 (default): the headline metric; config3: BASELINE configs[2] per-slice path (vit_l, tiled 2048^2 slices)")
    ap.add_argument("--slices", type=int, default=8, help="--workload config3: slices per GPU and step")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements (pcie_inclusive, api_inclusive, rle_side, interactive_side): profiling runs, so "
                         "that per-kernel averages hold the hot path's launches only")
    ap.add_argument("--api-tiles", type=int, default=TILES_PER_STEP, help="tiles of the api_inclusive side measurement (default: one step's worth)")
    ap.add_argument("--no-config-sides", action="store_true",
                    help="skip config3_side / fp8_side / train_side (the other named configurations as short side runs, N = 1 only)")
    ap.add_argument("--precision", choices=("default", "split16", "strict"), default="default",
                    help="precision mode of the timed region (predictor.set_precision): default = the headline's 16-bit path; split16 / strict = the reference's "
                         "formulation on fp16 operand pairs / fp32 kernels (side measurements: `split16_side` of the default run is this script with --precision split16)")
    ap.add_argument("--fp8-side", action="store_true", help="also run the fp8-encoder configuration (BASELINE configs[4]) as a side run")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: gloo, no GPU work, prints the JSON line with n_gpus = world size")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if args.dry_run:
        # launcher / rendezvous check without a GPU (gloo).  --workload config3 additionally walks the sharded bench path of BASELINE configs[2]:
        # every rank's share of the 64-slice volume (slice seeds as bench_config3 draws them), the barrier, the MAX over ranks of the step time and
        # rank 0's line - so that the config-3 launch is exercised before hardware sees it
        shares = None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.barrier()
        if args.workload == "config3":
            Z = args.slices
            mine = torch.tensor([3000 + rank * Z + z for z in range(Z)], dtype=torch.int64)
            elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)          # a stand-in step time: rank r "takes" r + 1 ms
            if world > 1:
                gathered = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(gathered, mine)
                dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
                dist.barrier()
            else:
                gathered = [mine]
            shares = [g.tolist() for g in gathered]
            flat = [s_ for sh in shares for s_ in sh]
            assert len(set(flat)) == world * Z and flat == list(range(3000, 3000 + world * Z)), flat       # disjoint, contiguous, in rank order
            assert abs(float(elapsed.item()) - 0.001 * world) < 1e-12                                       # the slowest rank's time is the step's
        if world > 1:
            dist.destroy_process_group()
        if rank == 0:
            rec = {"metric": "1024^2 tiles/s embed+AMG (vit_b bf16)", "value": 0.0, "unit": "tiles/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "dry_run": True}
            if shares is not None:
                rec.update({"metric": "2048^2 slices/s tiled embed+AMG (vit_l bf16), BASELINE configs[2] per-slice path", "unit": "slices/s",
                            "config": {"workload": "config3", "slices_per_gpu": args.slices, "slice_seeds_per_rank": shares}})
            print(json.dumps(rec), flush=True)
        return

    # distinct synthetic tiles per rank (seed = global tile index), generated before CUDA is initialised (fork pool)
    n_tiles = args.tiles_per_step
    enc_batch = args.enc_batch
    n_distinct = max(n_tiles, (args.distinct_tiles // n_tiles) * n_tiles)
    t_gen = time.perf_counter()
    tiles_np = make_tiles([1000 + rank * n_distinct + i for i in range(n_distinct)]) if args.workload == "config2" else []
    log(f"rank {rank}: {n_distinct} distinct tiles generated in {time.perf_counter() - t_gen:.1f} s")

    # MSAM_FORCE_DIST=1 at N = 1: a world-size-1 "nccl" group, label tiles gathered through RCCL (smoke run of the N > 1 code path)
    force_dist = world == 1 and os.environ.get("MSAM_FORCE_DIST") == "1"
    if force_dist:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ["MSAM_FORCE_COLLECTIVES"] = "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.workload == "config3":
        bench_config3(args, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from micro_sam_amd import _lib, parallel, util
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict

    sd = synthetic_state_dict("vit_b", 0, variant=args.weights)
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=sd)
    predictor.model.use_glds = args.glds
    predictor.model.image_encoder.use_glds = args.glds
    predictor.model.image_encoder.set_precision(args.encoder_dtype)
    amg = AutomaticMaskGenerator(predictor, device_chunk=args.device_chunk)   # reference defaults: 32x32 grid, 64 points per batch

    tiles_u8 = torch.stack([torch.as_tensor(util._to_image(t)) for t in tiles_np]).to(dev)     # [n_distinct,1024,1024,3] in HBM
    torch.cuda.synchronize()
    lib = _lib.load()
    stage = {"encode": 0.0, "initialize": 0.0, "generate": 0.0, "gather": 0.0, "host_enqueue": 0.0}
    n_instances = []

    # live HIP-event measurement per kernel family (include/msam_hip.h msam_profile_collect_family)
    NF = _lib.PROFILE_FAMILIES
    FAMILY = [
        ("gemm256_kernel (256x256 tile MFMA GEMM: the image encoder's qkv / proj / MLP projections)", "mfma"),
        ("wsgemm_kernel / dec_image_layer_kernel (weights-stationary streaming kernels, > 8 tokens per prompt)", "hbm"),
        ("i2t_tok_kernel / fold_i2t_kernel (folded image->token attention + out_proj + norm4 of one decoder layer, stage-by-stage form)", "mfma"),
        ("fold_attn_kernel (folded token->image attention: final attention; layer 1 in the stage-by-stage form)", "mfma"),
        ("up_fused_kernel (fused up-scaling ConvT-LN-GELU-ConvT-GELU + hyper product)", "mfma"),
        ("gemm_kernel / gemm_ln_kernel (128x128 tile MFMA GEMM: patch embedding, neck and the latency-bound token-side "
         "launches)", "mfma"),
        ("i2t0_t2i_kernel (layer-0 image->token block recomputed per tile from the shared source, chained into the layer-1 "
         "token->image attention; no per-prompt stream read or written)", "mfma"),
        ("i2t01_kernel (layer-0 image->token block chained into the layer-1 image->token block; writes the layer-1 stream)", "mfma"),
    ]
    prof = [{"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0} for _ in range(NF)]

    def collect():
        n_, ms_, fl_, by_ = (C.c_int32 * NF)(), (C.c_double * NF)(), (C.c_double * NF)(), (C.c_double * NF)()
        lib.msam_profile_collect_family(n_, ms_, fl_, by_)
        for f in range(NF):
            prof[f]["launches"] += n_[f]; prof[f]["ms"] += ms_[f]; prof[f]["flops"] += fl_[f]; prof[f]["bytes"] += by_[f]

    # generate() of tile i (small latency-bound launches) runs on a side stream underneath the decoder kernels of tile i+1
    gen_stream = None if args.serial_generate else torch.cuda.Stream(device=dev)
    lanes = [(predictor, amg, torch.cuda.Stream(device=dev))]
    for _ in range(1, max(args.lanes, 1)):
        pk = util.get_sam_model("vit_b", device=dev, state_dict=sd)
        pk.model.use_glds = args.glds
        lanes.append((pk, AutomaticMaskGenerator(pk, device_chunk=args.device_chunk), torch.cuda.Stream(device=dev)))
    if args.precision != "default":
        for pk_, _, _ in lanes:
            pk_.set_precision(args.precision)
    pinned_labels = [None]
    # N > 1: the all_gather of step k's label tiles runs on a communication stream underneath step k + 1's encoder / decoder kernels
    # (8 ranks x 64 tiles: every rank receives 1.75 GiB per step - ~6 ms of a 370 ms step over xGMI - un-overlapped before round 4)
    comm_stream = torch.cuda.Stream(device=dev) if (world > 1 or force_dist) else None
    shape_only = np.broadcast_to(np.zeros((1, 1), dtype=np.uint8), (1024, 1024))   # initialize() reads the image SHAPE only

    flag_ring = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2)]
    flag_pending = []                  # (page-locked word, event) of steps whose convergence flags have not been read yet

    def check_flags(keep: int = 0):
        """Read the convergence flags of all but the `keep` most recent steps (their event has fired or is waited for)."""
        while len(flag_pending) > keep:
            word, ev = flag_pending.pop(0)
            ev.synchronize()
            if int(word.item()) != 0:
                raise RuntimeError("connected-component labelling did not converge in 2 passes")

    def step(timed: bool, index: int, uploads=None, serial: bool = False, wait_gather: bool = False, defer_flags: bool = False):
        """timed=True: instrumented pass with a device sync after every stage (stage breakdown only);
        timed=False: the production path, no extra synchronisation.  uploads: host tiles to convert + upload inside the step
        (the PCIe-inclusive pass), else the resident uint8 tiles of step `index` are used."""
        labels = torch.empty((n_tiles, 1024, 1024), dtype=torch.int32, device=dev)
        flags = []
        t0 = time.perf_counter()
        if uploads is not None:
            # the product's raw-tile path (util._compute_embeddings_batched_raw): pinned staging + asynchronous H2D of the
            # raw uint8 tiles (1 MiB each), util._to_image on the device
            raw = util._upload_raw_tiles(predictor, uploads)
            batch_u8 = torch.stack([util.to_image_device(raw[b]) for b in range(raw.shape[0])])
        else:
            lo = (index * n_tiles) % n_distinct
            batch_u8 = tiles_u8[lo:lo + n_tiles]
        # embeddings of the step: [n,1,256,64,64] on the device, filled one encoder batch at a time; an event after every batch lets the
        # decode lanes start on the tiles of batch j while the encoder works on batch j + 1 (--tiles-per-step > --enc-batch)
        feats = torch.empty((n_tiles, 1, 256, 64, 64), dtype=torch.float32, device=dev)
        enc_done = []
        for s in range(0, n_tiles, enc_batch):
            feats[s:s + enc_batch, 0] = predictor.model.image_encoder.forward_u8(batch_u8[s:s + enc_batch])
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            enc_done.append(ev)
        emb = {"features": feats, "input_size": (1024, 1024), "original_size": (1024, 1024)}
        if timed:
            torch.cuda.synchronize(); stage["encode"] += time.perf_counter() - t0
        if len(lanes) > 1 and not timed and not serial:
            main = torch.cuda.current_stream()
            for i in range(n_tiles):
                _, ak, st = lanes[i % len(lanes)]
                st.wait_event(enc_done[i // enc_batch])                 # this tile's embedding is ready
                with torch.cuda.stream(st):
                    ak.initialize(shape_only, emb, i=i)
                    lab, flag = ak.generate_device()
                    labels[i] = lab
                flags.append(flag)
            for _, _, st in lanes:
                main.wait_stream(st)
        for i in range(n_tiles if (len(lanes) == 1 or timed or serial) else 0):
            t1 = time.perf_counter()
            amg.initialize(shape_only, emb, i=i)
            if timed:
                torch.cuda.synchronize(); stage["initialize"] += time.perf_counter() - t1
            t2 = time.perf_counter()
            # generate() on the device, result kept in HBM (it feeds the all_gather); same labels as amg.generate()
            if gen_stream is None or timed:
                lab, flag = amg.generate_device()
                labels[i] = lab
            else:
                lab, flag = amg.generate_device(stream=gen_stream)
                with torch.cuda.stream(gen_stream):
                    labels[i] = lab
            flags.append(flag)
            if timed:
                torch.cuda.synchronize(); stage["generate"] += time.perf_counter() - t2
        if gen_stream is not None and not timed:
            torch.cuda.current_stream().wait_stream(gen_stream)          # label tiles complete before the gather
        t3 = time.perf_counter()
        if comm_stream is not None:
            comm_stream.wait_stream(torch.cuda.current_stream())         # this step's label tiles are complete
            with torch.cuda.stream(comm_stream):
                full = parallel.gather_label_tiles(labels, n_tiles * world)
            labels.record_stream(comm_stream)
            if timed or uploads is not None or wait_gather:
                torch.cuda.current_stream().wait_stream(comm_stream)     # the caller reads `full` on the compute stream
        else:
            full = labels
        if timed:
            torch.cuda.synchronize(); stage["gather"] += time.perf_counter() - t3
        if not timed:
            stage["host_enqueue"] += time.perf_counter() - t0    # host time to enqueue the whole step (no sync inside)
        host_labels = None
        if uploads is not None:                                          # PCIe-inclusive pass: label D2H into pinned memory
            if pinned_labels[0] is None or pinned_labels[0].shape != full.shape:
                pinned_labels[0] = torch.empty(full.shape, dtype=full.dtype).pin_memory()
            host_labels = pinned_labels[0].copy_(full, non_blocking=True)
        # single synchronisation point of the step: convergence flags of the connected-component labelling.  In the timed loop the
        # flags of step k are read while step k + 1 is already queued (defer_flags: a page-locked word + an event per step), so that the
        # GPU does not drain at every step boundary - the next step's first encoder batch runs under this step's last decode lanes
        word = flag_ring[index & 1]
        word.copy_(torch.stack(flags).sum().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        flag_pending.append((word, ev))
        check_flags(keep=1 if defer_flags else 0)
        if timed:
            n_instances.extend(int(v) for v in labels.flatten(1).max(dim=1).values.tolist())
        return host_labels if uploads is not None else full

    for k in range(args.warmup):
        step(False, k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Per-kernel HIP-event brackets (msam_profile_*) measure a kernel's own duration only when nothing else shares the GPU with
    # it.  With one decode lane they run inside the timed region; with several lanes kernels of different tiles overlap (that
    # is the point of the lanes) and a bracket would also count the time a launch queues behind another lane's kernels, so the
    # roofline pass is then a separate serial (one-lane) pass over the same steps right after the timed region.
    profile_in_timed_region = len(lanes) == 1
    lib.msam_profile_enable(1 if profile_in_timed_region else 0)
    stage["host_enqueue"] = 0.0
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(False, args.warmup + k, defer_flags=not profile_in_timed_region)
        if profile_in_timed_region:
            collect()        # synchronises the step's kernel events (end of step: nothing left in flight anyway)
    check_flags()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    lib.msam_profile_enable(0)
    prof_steps = args.steps
    if not profile_in_timed_region:
        prof_steps = max(1, min(args.steps, 5))
        host_enqueue_timed = stage["host_enqueue"]
        torch.cuda.synchronize()
        lib.msam_profile_enable(1)
        t_ser = time.perf_counter()
        for k in range(prof_steps):
            step(False, args.warmup + k, serial=True)
            collect()
        torch.cuda.synchronize()
        serial_elapsed = time.perf_counter() - t_ser
        lib.msam_profile_enable(0)
        stage["host_enqueue"] = host_enqueue_timed
    serial_labels = step(True, 0).clone()     # one extra instrumented pass (outside the timed region): stage breakdown, and
    pipelined_labels = step(False, 0, wait_gather=True)   # the serial result that the pipelined (lanes / side stream) step must reproduce
    torch.cuda.synchronize()
    labels_equal = bool(torch.equal(serial_labels, pipelined_labels))
    if not labels_equal:
        raise RuntimeError("bench.py: the pipelined step (lanes / side-stream generate) does not reproduce the serial labels")
    # PCIe-inclusive pass (outside the timed region): util._to_image + H2D of the uint8 tiles + label D2H inside the clock
    torch.cuda.synchronize()
    t_p = time.perf_counter()
    n_pcie = 0 if args.no_side else min(3, max(1, n_distinct // n_tiles))
    for k in range(n_pcie):
        step(False, k, uploads=tiles_np[k * n_tiles:(k + 1) * n_tiles])
    torch.cuda.synchronize()
    pcie_elapsed = time.perf_counter() - t_p
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_tiles = n_tiles * args.steps * world
        value = total_tiles / elapsed
        tiles_timed = n_tiles * prof_steps                  # tiles of the pass the kernel brackets were recorded in

        def fam(f):
            d = prof[f]
            sec = d["ms"] * 1e-3
            launches = max(d["launches"], 1)
            exec_tflops = d["flops"] / sec / 1e12 if sec > 0 else 0.0
            # algorithmic FLOP per launch: 2*M*N*K for the GEMM families; SURVEY 8(d)'s per-prompt figure x prompts per launch
            # for the decoder stream kernels (one launch = device_chunk prompts of one tile)
            if f in ALG_GFLOP_PER_PROMPT:
                prompts_per_launch = min(args.device_chunk, 1024)
                alg_flop = ALG_GFLOP_PER_PROMPT[f] * 1e9 * prompts_per_launch * d["launches"]
            else:
                alg_flop = d["flops"]
            alg_tflops = alg_flop / sec / 1e12 if sec > 0 else 0.0
            peak = PEAK_FP8_TFLOPS if (f == 0 and args.encoder_dtype == "fp8") else PEAK_BF16_TFLOPS
            g = d["bytes"] / sec / 1e9 if sec > 0 and d["bytes"] > 0 else None
            return {"kernel": FAMILY[f][0], "bound": FAMILY[f][1], "launches": d["launches"],
                    "seconds_per_tile": round(sec / tiles_timed, 5),
                    "avg_launch_us": round(d["ms"] * 1e3 / launches, 2),
                    "achieved": round(alg_tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg_tflops / peak, 4),
                    "avg_launch_gflop_algorithmic": round(alg_flop / launches / 1e9, 3),
                    "executed_tflops": round(exec_tflops, 2),
                    "hbm_gbytes_per_s": None if g is None else round(g, 1),
                    "hbm_frac": None if g is None else round(g / PEAK_HBM_GBS, 4),
                    "avg_launch_mbytes": round(d["bytes"] / launches / 1e6, 2)}

        fams = [fam(f) for f in range(NF) if prof[f]["launches"] > 0]
        fams.sort(key=lambda r: -r["seconds_per_tile"])
        # roofline of the dominant kernel (largest GPU time in the timed region); the others are kept alongside.
        # `traffic`: HBM bytes per launch from the PMC passes of profiles/ (FETCH_SIZE / WRITE_SIZE, separate runs), when a
        # table for this kernel is committed
        dom = dict(fams[0]) if fams else {"bound": "mfma", "achieved": 0.0, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": 0.0}
        traffic, traffic_source = pmc_traffic(dom.get("kernel", ""))
        whole = TILE_TFLOP_ALGORITHMIC * value / world
        roof = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                "traffic": traffic, "traffic_source": traffic_source, **{k: v for k, v in dom.items() if k not in ("bound", "achieved", "peak", "unit", "frac")},
                "whole_path_tflops": round(whole, 2), "whole_path_frac": round(whole / PEAK_BF16_TFLOPS, 4),
                "measured_in": ("the timed region (one decode lane: no two kernels of the hot path share the GPU)" if profile_in_timed_region
                                else f"a serial one-lane pass of {prof_steps} steps right after the timed region "
                                     f"({n_tiles * prof_steps / serial_elapsed:.1f} tiles/s): in the timed region {len(lanes)} lanes "
                                     "overlap the kernels of different tiles, so a HIP-event bracket there also counts queueing; "
                                     "profiles/ holds the rocprofv3 summary of the same one-lane command (--lanes 1)"),
                "other_kernels": fams[1:],
                # why frac stops where it does (measured: profiles/r06_mfma_valu_overlap.md - the matrix pipe and the vector ALU of a SIMD DO overlap,
                # round 5's "additive" reading came from a probe whose vector stream was dependent v_pk_fma_f32)
                "pipes": ("MI355X, measured (tools/mfma_valu_overlap_probe2.hip, inline-asm streams): an MFMA-only and a VALU-only wave of one SIMD take 1.09 - 1.12 x the longer "
                          "of the two in both age orders; two waves per SIMD hide up to 5 plain vector instructions per 32-cycle MFMA at 91 - 94 % matrix-pipe "
                          "occupancy, beyond that the SIMD's vector ISSUE (~one instruction per 5.5 cycles between two waves) is the limit.  The dominant kernel "
                          "carries 372 vector : 57 (16-cycle) matrix instructions per tile and wave: it is vector-issue-bound (profiles/r05_pmc_sq_counters.md: "
                          "2 x 35 % VALU-active, 31 % MFMA-busy); `frac` is what its vector stream leaves of the matrix pipe.  Clock under this load: ~2.0 GHz "
                          "(GRBM_GUI_ACTIVE / duration), not the 2.4 GHz of the nominal peak")}
        out = {
            "metric": {"bf16": "1024^2 tiles/s embed+AMG (vit_b bf16)",
                       "fp16": "1024^2 tiles/s embed+AMG (vit_b, fp16 instead of bf16 operands in the image encoder: side measurement)",
                       "fp8": "1024^2 tiles/s embed+AMG (vit_b fp8 encoder + bf16 decoder, BASELINE configs[4])"}[args.encoder_dtype] +
                      ("" if args.precision == "default" else f" [precision mode {args.precision}: side measurement]"),
            "value": round(value, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ({"bf16": "bf16", "fp16": "fp16", "fp8": "fp8 projections + bf16"}[args.encoder_dtype] + " image encoder (patch embedding "
                      "+ neck, 1.2 % of its flops, on hi+lo operand pairs of that type), " +
                      ("fp16" if _lib.decoder_dtype() == torch.float16 else "bf16") + " mask decoder (16-bit MFMA operands, fp32 accumulation)")
                     if args.precision == "default" else
                     {"split16": "fp16 operand pairs (hi + lo of every fp32 operand, 3 MFMAs of the 16-bit pipe per product, fp32 accumulation) in the reference's formulation",
                      "strict": "f32 (f32-input MFMA) in the reference's formulation"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[1]: vit_b, 1024x1024 uint8 synthetic tiles, batched embedding precompute + "
                                   "AutomaticMaskGenerator (32x32 grid, multimask, default thresholds)",
                       "tiles_per_step_per_gpu": n_tiles, "encoder_batch": enc_batch,
                       "distinct_tiles_per_gpu": n_distinct,
                       "weights": f"seeded synthetic checkpoint (synthetic.py variant '{args.weights}')",
                       "parallelism": f"dp{world} tiles, all_gather of uint32 label tiles (RCCL, on a communication stream under the next step)" +
                                      (" (MSAM_FORCE_DIST: world-size-1 nccl group, the gather runs through RCCL)" if force_dist else ""),
                       "timed_region": "uint8 RGB tiles resident in HBM -> label tiles in HBM (all-gathered when N > 1); "
                                       "util._to_image, H2D and label D2H are in pcie_inclusive, lazy RLE encoding in rle_side",
                       "decode_lanes": len(lanes), "pipelined_labels_equal_serial": labels_equal,
                       "instances_per_tile": {"median": int(np.median(n_instances)), "min": int(min(n_instances)),
                                              "max": int(max(n_instances))},
                       "stage_seconds_per_tile_synced_pass": {k: round(v / n_tiles, 5) for k, v in stage.items()
                                                              if k != "host_enqueue"},
                       "host_enqueue_seconds_per_tile": round(stage["host_enqueue"] / (n_tiles * args.steps), 5),
                       "tile_tflop_algorithmic": TILE_TFLOP_ALGORITHMIC},
            "roofline": roof,
        }
        if not args.no_side:
            out["pcie_inclusive"] = {"value": round(n_pcie * n_tiles / pcie_elapsed, 2), "unit": "tiles/s", "tiles": n_pcie * n_tiles,
                                     "includes": "pinned asynchronous H2D of the raw uint8 tiles (1 MiB each), util._to_image on the "
                                                 "device (msam_to_image), label D2H into pinned memory (4 MiB each)"}
            try:
                out["api_inclusive"] = api_inclusive(predictor, amg, tiles_np, min(args.api_tiles, n_distinct), enc_batch)
            except Exception as exc:            # a side measurement must not cost the bench line
                out["api_inclusive"] = {"error": repr(exc)}
        if not args.no_side:
            # a15 side measurement: RLE encoding of one tile's candidate masks (lazy in the product: only when rles are read)
            amg.initialize(shape_only, {"features": predictor.model.image_encoder.forward_u8(tiles_u8[:1]).unsqueeze(1),
                                        "input_size": (1024, 1024), "original_size": (1024, 1024)}, i=0)
            from micro_sam_amd import ops
            bits = amg.crop_list[0]["bits"]
            ops.rle_encode(bits.contiguous(), 1024, 1024)               # first call: loads the RLE kernels' code object
            torch.cuda.synchronize(); t_r = time.perf_counter()
            counts, offsets = ops.rle_encode(bits.contiguous(), 1024, 1024)
            torch.cuda.synchronize()
            out["rle_side"] = {"masks": int(bits.shape[0]), "ms_device": round((time.perf_counter() - t_r) * 1e3, 3),
                               "total_runs": int(offsets[-1].item()) if offsets.numel() else 0}
            # f4 side measurement: latency of one interactive point prompt (SamPredictor.predict: numpy in, 3 masks + scores + low-res
            # logits back on the host) on an embedding that is already set, as the napari annotator issues them
            util.set_precomputed(predictor, {"features": predictor.model.image_encoder.forward_u8(tiles_u8[:1]),
                                             "input_size": (1024, 1024), "original_size": (1024, 1024)})
            rng = np.random.default_rng(0)
            lat = []
            for k in range(60):
                pt = rng.uniform(32, 992, size=(1, 2))
                torch.cuda.synchronize(); t_i = time.perf_counter()
                predictor.predict(point_coords=pt, point_labels=np.ones(1), multimask_output=True)
                lat.append((time.perf_counter() - t_i) * 1e3)
            out["interactive_side"] = {"predict_ms_median": round(float(np.median(lat[10:])), 3), "predict_ms_p90": round(float(np.quantile(lat[10:], 0.9)), 3),
                                       "what": "SamPredictor.predict(one point, multimask) incl. the D2H of 3 x 1024^2 masks"}
        if not args.no_cpu_baseline and world == 1:
            log(f"timing the CPU reference on {args.cpu_tiles} full tiles ...")
            n_thr = min(os.cpu_count() or 1, 32)    # more threads than this only slow the fp32 torch ops down
            ref_tiles = tiles_np[:args.cpu_tiles]
            out["cpu_baseline"], ref_states, ref_segs = cpu_reference(sd, ref_tiles, n_thr)
            out["mask_iou_vs_ref"] = mask_iou_vs_ref(predictor, amg, ref_tiles, ref_states, ref_segs)
            # the strict precision mode (predictor.set_precision("strict"): the reference's formulation on fp32 kernels, micro_sam_amd/strict.py)
            # on the same tiles against the same reference: the point of the speed / parity curve that meets the north-star statement
            for mode in ("split16", "strict"):
                try:
                    out[f"mask_iou_vs_ref_{mode}"] = strict_leg(predictor, amg, ref_tiles, ref_states, ref_segs, out["cpu_baseline"]["value"], mode=mode)
                except Exception as exc:            # a side measurement must not cost the bench line
                    out[f"mask_iou_vs_ref_{mode}"] = {"error": repr(exc)}
                finally:
                    predictor.set_precision("default")
                    predictor.model.image_encoder.set_precision(args.encoder_dtype)
                    torch.cuda.empty_cache()    # (the modes' multi-GiB fp32 streams go back to the driver before the next leg)
            if args.encoder_dtype == "bf16":
                # what the bf16 rounding of the encoder's operands costs: the same comparison with IEEE fp16 operands in the encoder
                # (same kernels and MFMA rate; throughput of that mode: python bench.py --encoder-dtype fp16)
                try:
                    predictor.model.image_encoder.set_precision("fp16")
                    out["mask_iou_vs_ref_fp16_encoder"] = mask_iou_vs_ref(predictor, amg, ref_tiles, ref_states, ref_segs)
                except Exception as exc:        # a side measurement must not cost the bench line
                    out["mask_iou_vs_ref_fp16_encoder"] = {"error": repr(exc)}
                finally:
                    predictor.model.image_encoder.set_precision("bf16")
                # and what the hi + lo operand pairs at the patch embedding / neck buy: the same comparison with every operand plainly bf16
                # (the arithmetic of rounds 1 and 2)
                try:
                    predictor.model.image_encoder.set_split_io(False)
                    out["mask_iou_vs_ref_plain_bf16_encoder"] = mask_iou_vs_ref(predictor, amg, ref_tiles, ref_states, ref_segs)
                except Exception as exc:
                    out["mask_iou_vs_ref_plain_bf16_encoder"] = {"error": repr(exc)}
                finally:
                    predictor.model.image_encoder.set_split_io(True)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
            out["mask_iou_vs_ref"] = None
        if world == 1 and not args.no_side and not args.no_config_sides and args.encoder_dtype == "bf16":
            log("side runs of the other named configurations (configs[2], [4], [3]) ...")
            del tiles_u8
            torch.cuda.empty_cache()
            out.update(config_sides(fp8=args.fp8_side))
        emit(out)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
