"""bench.py --gpus N must launch N ranks by itself (--gpus was never read)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def test_gpus_2_spawns_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_too_few_gpus_fails_loudly():
    # no GPU in the CPU container: asking for 2 GPUs without --dry-run must refuse instead of running 1 rank
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
