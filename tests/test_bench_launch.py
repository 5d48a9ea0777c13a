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


def test_roofline_traffic_is_reported_only_for_the_code_it_was_measured_on(monkeypatch):
    """bench.pmc_traffic: the committed PMC table counts only while the kernel sources hash to the value stored with it - a table of
    other code is named and refused (round 2 shipped a round-1 constant)."""
    import json

    import bench
    tables = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_pmc_traffic.json"))
    assert tables
    newest = json.load(open(os.path.join(ROOT, "profiles", tables[-1])))
    sha = newest["_meta"]["csrc_sha16"]
    monkeypatch.setattr(bench, "csrc_sha16", lambda: sha)
    value, source = bench.pmc_traffic("up_fused_kernel<1, 1>")
    assert value == newest["up_fused_kernel"]["hbm_bytes_per_launch"] and value > 1e9 and sha in source
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "0" * 16)
    value, source = bench.pmc_traffic("up_fused_kernel<1, 1>")
    assert value is None and "other code" in source and sha in source
    value, source = bench.pmc_traffic("no_such_kernel")
    assert value is None and "no PMC table" in source
    # the hash names the sources: it moves with any byte of csrc/ or the ABI header
    monkeypatch.undo()
    assert len(bench.csrc_sha16()) == 16 and bench.csrc_sha16() == bench.csrc_sha16()
