"""Fine-tuning throughput on synthetic LIVECell-shaped batches (BASELINE configs[3]: training.sam_trainer, data-parallel):
steps/s of SamTrainer.train_iteration (iterative prompting, AdamW) for a model type and a freeze setting.  A measuring tool for
the GPU box (not part of bench.py's contract):

    python tools/train_bench.py --model vit_b --freeze image_encoder prompt_encoder --steps 10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_bench.py --model vit_h

Prints one JSON line on rank 0 (max over ranks of the timed region, barrier + synchronize on both sides)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_batch(rng, batch, shape, n_obj):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    xs, ys = [], []
    for _ in range(batch):
        y = np.zeros(shape, dtype=np.int64)
        img = rng.normal(40, 8, size=shape)
        for k in range(1, n_obj + 1):
            cy, cx, r = rng.integers(20, shape[0] - 20), rng.integers(20, shape[1] - 20), rng.integers(8, 22)
            m = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r) & (y == 0)
            if m.sum() < 30:
                continue
            y[m] = k
            img[m] += rng.uniform(60, 160)
        xs.append(np.repeat(np.clip(img, 0, 255)[None], 3, axis=0).astype(np.float32))
        ys.append(y[None])
    return torch.as_tensor(np.stack(xs)), torch.as_tensor(np.stack(ys))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_b")
    ap.add_argument("--freeze", nargs="*", default=None, help="parts to freeze: image_encoder prompt_encoder mask_decoder")
    ap.add_argument("--lora-rank", type=int, default=0)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--shape", type=int, nargs=2, default=(520, 704))
    ap.add_argument("--objects", type=int, default=25)
    ap.add_argument("--sub-iterations", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--device-time", action="store_true",
                    help="after the timed steps: one step under the torch profiler -> share of the GPU time in kernels that are NOT this "
                         "library's (at::native element-wise / reductions, rocBLAS / hipBLASLt, copies) and kernel launches per step")
    ap.add_argument("--op-profile", type=int, default=0, help="print the N heaviest torch operators (by device time, with input shapes) of one step")
    args = ap.parse_args()
    from micro_sam_amd.synthetic import synthetic_state_dict
    from micro_sam_amd.training import ConvertToSamInputs, SamTrainer, get_trainable_sam_model
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model = get_trainable_sam_model(args.model, device=dev, state_dict=synthetic_state_dict(args.model, 0), freeze=args.freeze,
                                    peft_kwargs={"rank": args.lora_rank} if args.lora_rank else None)
    params = [p for p in model.parameters() if p.requires_grad]
    trainer = SamTrainer(model, torch.optim.AdamW(params, lr=1e-5), ConvertToSamInputs(transform=model.transform),
                         n_sub_iteration=args.sub_iterations, n_objects_per_batch=args.objects, mask_prob=0.5, device=dev)
    rng = np.random.default_rng(rank)
    batches = [synthetic_batch(rng, args.batch, tuple(args.shape), args.objects + 5) for _ in range(max(args.steps, 2))]
    for i in range(args.warmup):
        trainer.train_iteration(*batches[i % len(batches)])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    if args.op_profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            trainer.train_iteration(*batches[0])
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=args.op_profile,
                                                                 max_name_column_width=40, max_shapes_column_width=70))
        print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
        return
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rec = trainer.train_iteration(*batches[i % len(batches)])
    fence()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    extra = {}
    if args.device_time and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            trainer.train_iteration(*batches[0])
            torch.cuda.synchronize()
        tot = non = 0.0
        n_launch = 0
        worst = {}
        for e in prof.key_averages():
            t = float(getattr(e, "self_device_time_total", 0.0) or getattr(e, "self_cuda_time_total", 0.0))
            if t <= 0:
                continue
            tot += t
            n_launch += int(e.count)
            name = str(e.key)
            foreign = any(k in name for k in ("at::", "Cijk_", "rocblas", "hipblas", "ck::", "miopen", "Memcpy", "Memset", "copyBuffer", "fillBuffer"))
            if foreign:
                non += t
                worst[name[:80]] = worst.get(name[:80], 0.0) + t
        extra = {"non_hip_device_time_frac": round(non / tot, 4) if tot else None, "kernel_launches_per_step": n_launch,
                 "device_ms_per_step": round(tot / 1e3, 2),
                 "largest_foreign_kernels": [[k, round(v / tot, 4)] for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]]}
    if rank == 0:
        print(json.dumps({**extra, "model": args.model, "ms_per_step": round(float(dt) / args.steps * 1e3, 1),
                          "metric": f"fine-tuning steps/s ({args.model}, batch {args.batch} x {world} GPUs, {args.objects} objects, "
                                    f"{args.sub_iterations} sub-iterations)", "value": round(args.steps / float(dt), 4), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "trained_parameters": sum(p.numel() for p in params),
                          "frozen": args.freeze or [], "lora_rank": args.lora_rank, "last_loss": rec["loss"],
                          "allreduce_bytes_per_step": rec["allreduce_bytes"], "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
