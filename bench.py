"""Benchmark of the hot path: 1024^2 tiles/s, embed + AMG (vit_b, bf16 MFMA operands), BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = TILES_PER_STEP synthetic tiles per rank through: batched image encoder -> per tile AutomaticMaskGenerator
initialize (32x32 grid prompts, fused mask post-processing to bit masks on the device) -> generate
(default thresholds, box NMS, merge to a uint32 label image); with N > 1 the label tiles of all ranks are all-gathered
(RCCL) inside the timed region.  Inputs (uint8 RGB tiles, output of util._to_image) are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILES_PER_STEP = 16
ENC_BATCH = 16     # M = 65536 rows: every encoder GEMM is a whole number of 256-workgroup rounds (B = 8 left 1.5-round tails)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0           # MI355X dense fp8 (MX-scaled MFMA), --encoder-dtype fp8 only
PEAK_HBM_GBS = 8000.0              # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s achievable)
TILE_TFLOP_ALGORITHMIC = 4.64      # encoder 0.938 + AMG decode 3.70


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(sd, tile_u8):
    """The CPU oracle (restated reference hot path, fp32) on a bounded sample of the same workload:
    one tile's encoder + 2 of its 16 decoder batches (+ their mask post-processing) + generate; the per-batch cost is
    extrapolated to 16 batches.  Reported only, never used as a target."""
    from oracle import amg_ref as A
    from oracle import pipeline_ref as PR
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # more threads than this only slow the fp32 torch ops down
    img = A.to_image(tile_u8)
    t0 = time.perf_counter()
    feats, osz, isz = PR.compute_embeddings(sd, [img], "vit_b", "fp32")
    t_enc = time.perf_counter() - t0
    tm = {}
    t0 = time.perf_counter()
    state = PR.amg_initialize(sd, img, feats, isz[0], osz[0], precision="fp32", timings=tm, max_batches=2)
    t_init2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    PR.amg_generate(state)
    t_gen = time.perf_counter() - t0
    per_tile = t_enc + 8.0 * t_init2 + t_gen
    return {"value": 1.0 / per_tile, "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 tile: encoder {t_enc:.1f}s + 2/16 decoder batches {t_init2:.1f}s (x8 extrapolated) + "
                      f"generate {t_gen:.1f}s, fp32 torch on host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--glds", type=int, default=0)
    ap.add_argument("--tiles-per-step", type=int, default=TILES_PER_STEP, help="tiles per rank and step")
    ap.add_argument("--enc-batch", type=int, default=ENC_BATCH, help="tiles per image-encoder call")
    ap.add_argument("--encoder-dtype", choices=("bf16", "fp8"), default="bf16",
                    help="fp8: BASELINE config 5 (encoder projections on fp8 e4m3 MX MFMA, bf16 decoder); NOT the headline metric")
    ap.add_argument("--lanes", type=int, default=1,
                    help="tiles are decoded round-robin on this many HIP streams (each with its own predictor state and "
                         "decoder workspace): the latency-bound token-side launches of one tile run underneath the "
                         "streaming kernels of another")
    ap.add_argument("--serial-generate", action="store_true",
                    help="run generate() on the main stream (default: side stream overlapping the next tile's decode)")
    ap.add_argument("--device-chunk", type=int, default=1024,
                    help="grid prompts decoded per decoder pass (results do not depend on it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from micro_sam_amd import _lib, parallel, util
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile

    sd = synthetic_state_dict("vit_b", 0, variant="blobs")
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=sd)
    predictor.model.use_glds = args.glds
    predictor.model.image_encoder.use_glds = args.glds
    predictor.model.image_encoder.set_precision(args.encoder_dtype)
    amg = AutomaticMaskGenerator(predictor, device_chunk=args.device_chunk)   # reference defaults: 32x32 grid, 64 points per batch

    n_steps = args.warmup + args.steps
    # distinct synthetic tiles per rank and step (seed = global tile index), staged in HBM before timing
    n_tiles = args.tiles_per_step
    enc_batch = args.enc_batch
    tiles_np = [synthetic_tile(1000 + rank * n_tiles + i) for i in range(n_tiles)]
    tiles_u8 = torch.stack([torch.as_tensor(util._to_image(t)) for t in tiles_np]).to(dev)
    torch.cuda.synchronize()
    lib = _lib.load()
    stage = {"encode": 0.0, "initialize": 0.0, "generate": 0.0, "gather": 0.0, "host_enqueue": 0.0}
    n_instances = 0

    # live HIP-event measurement per kernel family (include/msam_hip.h msam_profile_collect_family)
    NF = _lib.PROFILE_FAMILIES
    FAMILY = [
        ("gemm256_kernel (256x256 tile MFMA GEMM: the image encoder's qkv / proj / MLP projections)", "mfma"),
        ("wsgemm_kernel / dec_image_layer_kernel (weights-stationary streaming kernels, > 8 tokens per prompt)", "hbm"),
        ("fold_i2t_kernel (folded image->token attention + out_proj + norm4: stream read (layer 1) + written in place)", "hbm"),
        ("fold_attn_kernel (folded token->image attention: stream read once)", "hbm"),
        ("up_fused_kernel (fused up-scaling + hyper product: stream read once, fp32 low-res logits written)", "hbm"),
        ("gemm_kernel / gemm_ln_kernel (128x128 tile MFMA GEMM: patch embedding, neck and the ~39 latency-bound token-side "
         "launches per tile)", "mfma"),
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

    def step(timed: bool):
        """timed=True: instrumented pass with a device sync after every stage (stage breakdown only);
        timed=False: the production path, no extra synchronisation."""
        nonlocal n_instances
        labels = torch.empty((n_tiles, 1024, 1024), dtype=torch.int32, device=dev)
        flags = []
        t0 = time.perf_counter()
        feats = []
        for s in range(0, n_tiles, enc_batch):
            feats.append(predictor.model.image_encoder.forward_u8(tiles_u8[s:s + enc_batch]))
        feats = torch.cat(feats).unsqueeze(1)                       # [n,1,256,64,64] on device
        emb = {"features": feats, "input_size": (1024, 1024), "original_size": (1024, 1024)}
        if timed:
            torch.cuda.synchronize(); stage["encode"] += time.perf_counter() - t0
        if len(lanes) > 1 and not timed:
            main = torch.cuda.current_stream()
            for _, _, st in lanes:
                st.wait_stream(main)                                    # embeddings ready
            for i in range(n_tiles):
                _, ak, st = lanes[i % len(lanes)]
                with torch.cuda.stream(st):
                    ak.initialize(tiles_np[i], emb, i=i)
                    lab, flag = ak.generate_device()
                    labels[i] = lab
                flags.append(flag)
            for _, _, st in lanes:
                main.wait_stream(st)
        for i in range(n_tiles if (len(lanes) == 1 or timed) else 0):
            t1 = time.perf_counter()
            amg.initialize(tiles_np[i], emb, i=i)
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
        full = parallel.gather_label_tiles(labels, n_tiles * world) if world > 1 else labels
        if timed:
            torch.cuda.synchronize(); stage["gather"] += time.perf_counter() - t3
        if not timed:
            stage["host_enqueue"] += time.perf_counter() - t0    # host time to enqueue the whole step (no sync inside)
        # single synchronisation point of the step: convergence flags of the connected-component labelling
        if int(torch.stack(flags).sum().item()) != 0:
            raise RuntimeError("connected-component labelling did not converge in 2 passes")
        n_instances = int(labels[-1].max().item())
        return full

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.msam_profile_enable(1)
    stage["host_enqueue"] = 0.0
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(False)
        collect()            # synchronises the step's GEMM events (end of step: nothing left in flight anyway)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    lib.msam_profile_enable(0)
    step(True)               # one extra instrumented pass (outside the timed region) for the stage breakdown
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_tiles = n_tiles * args.steps * world
        value = total_tiles / elapsed
        tiles_timed = n_tiles * args.steps

        def fam(f):
            d = prof[f]
            sec = d["ms"] * 1e-3
            r = {"kernel": FAMILY[f][0], "bound": FAMILY[f][1], "launches": d["launches"],
                 "seconds_per_tile": round(sec / tiles_timed, 5),
                 "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2),
                 "tflops": round(d["flops"] / sec / 1e12, 2) if sec > 0 else 0.0,
                 "gbytes_per_s": round(d["bytes"] / sec / 1e9, 1) if sec > 0 and d["bytes"] > 0 else None,
                 "avg_launch_gflop": round(d["flops"] / max(d["launches"], 1) / 1e9, 3),
                 "avg_launch_mbytes": round(d["bytes"] / max(d["launches"], 1) / 1e6, 2)}
            if FAMILY[f][1] == "mfma":
                peak = PEAK_FP8_TFLOPS if (f == 0 and args.encoder_dtype == "fp8") else PEAK_BF16_TFLOPS
                r.update(achieved=r["tflops"], peak=peak, unit="TFLOP/s", frac=round(r["tflops"] / peak, 4))
            else:
                g = r["gbytes_per_s"] or 0.0
                r.update(achieved=g, peak=PEAK_HBM_GBS, unit="GB/s", frac=round(g / PEAK_HBM_GBS, 4))
            return r

        fams = [fam(f) for f in range(NF) if prof[f]["launches"] > 0]
        fams.sort(key=lambda r: -r["seconds_per_tile"])
        # roofline of the dominant kernel (largest GPU time in the timed region); the others are kept alongside.
        # `traffic`: HBM bytes per launch from the PMC passes of profiles/ (FETCH_SIZE / WRITE_SIZE, separate runs), when a
        # table for this kernel is committed
        dom = dict(fams[0]) if fams else {"bound": "hbm", "achieved": 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": 0.0}
        traffic = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            for name, rec in pmc.items():
                if isinstance(rec, dict) and name.split("<")[0] in dom.get("kernel", ""):
                    traffic = rec.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        roof = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                "traffic": traffic, **{k: v for k, v in dom.items() if k not in ("bound", "achieved", "peak", "unit", "frac")},
                "other_kernels": fams[1:]}
        out = {
            "metric": "1024^2 tiles/s embed+AMG (vit_b bf16)" if args.encoder_dtype == "bf16" else
                      "1024^2 tiles/s embed+AMG (vit_b fp8 encoder + bf16 decoder, BASELINE configs[4])", "value": round(value, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.encoder_dtype == "bf16" else "fp8 encoder projections + bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: vit_b, 1024x1024 uint8 synthetic tiles, batched embedding precompute + "
                                   "AutomaticMaskGenerator (32x32 grid, multimask, default thresholds)",
                       "tiles_per_step_per_gpu": n_tiles, "encoder_batch": enc_batch, "weights": "seeded random init "
                       "(synthetic.py variant 'blobs')", "parallelism": f"dp{world} tiles, all_gather of uint32 label tiles",
                       "instances_last_tile": n_instances,
                       "stage_seconds_per_tile_synced_pass": {k: round(v / n_tiles, 5) for k, v in stage.items()
                                                              if k != "host_enqueue"},
                       "host_enqueue_seconds_per_tile": round(stage["host_enqueue"] / (n_tiles * args.steps), 5),
                       "tile_tflop_algorithmic": TILE_TFLOP_ALGORITHMIC,
                       "whole_path_tflops_algorithmic": round(TILE_TFLOP_ALGORITHMIC * value / world, 2)},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU oracle on a bounded sample ...")
            out["cpu_baseline"] = cpu_baseline(sd, tiles_np[0])
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
