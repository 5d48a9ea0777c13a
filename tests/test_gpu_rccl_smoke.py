"""RCCL on one GPU: a world-size-1 "nccl" process group drives exactly the collective calls of the N > 1 data path -
``parallel.gather_label_tiles`` (all_gather_into_tensor of uint32 label tiles + the serial loop's id offsets,
reference micro_sam/multi_dimensional_segmentation.py:401-414) and ``training.sam_trainer.all_reduce_gradients`` (flat fp32 buckets) -
so the driver's 8-GPU run is not the first time that code meets RCCL.  Runs in a subprocess (its own process group)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, socket, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from micro_sam_amd import parallel
from micro_sam_amd.training.sam_trainer import GradientBuckets, all_reduce_gradients
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSAM_FORCE_COLLECTIVES="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert parallel.collectives_active()
g = torch.Generator().manual_seed(0)
labels = (torch.rand(4, 256, 256, generator=g) * 6).to(torch.int32).to(dev)          # ids 0..5 per tile
full = parallel.gather_label_tiles(labels, 4)
os.environ["MSAM_FORCE_COLLECTIVES"] = "0"
ref = parallel.gather_label_tiles(labels, 4)                                        # local path (no collective)
os.environ["MSAM_FORCE_COLLECTIVES"] = "1"
assert torch.equal(full, ref) and int(full.max()) == 20
lin = torch.nn.Linear(300, 500).to(dev)
lin.weight.grad = torch.randn(500, 300, generator=g).to(dev); lin.bias.grad = torch.randn(500, generator=g).to(dev)
w0, b0 = lin.weight.grad.clone(), lin.bias.grad.clone()
nbytes = all_reduce_gradients(lin.parameters(), bucket_bytes=256 << 10)              # several buckets
assert nbytes == (500 * 300 + 500) * 4 and torch.equal(lin.weight.grad, w0) and torch.equal(lin.bias.grad, b0)
# round 5: the overlapped form - gradients written by autograd into flat buckets, every bucket all-reduced (RCCL, communication stream)
# from the hook of its last gradient while backward runs; with one rank the averaged gradients are the plain ones
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(), torch.nn.Linear(512, 8)).to(dev)
x, t = torch.randn(256, 64, device=dev), torch.randn(256, 8, device=dev)
((net(x) - t) ** 2).mean().backward()
want = [p.grad.clone() for p in net.parameters()]
net.zero_grad()
buckets = GradientBuckets(net.parameters(), bucket_bytes=256 << 10)
for step in range(2):
    buckets.zero()
    ((net(x) - t) ** 2).mean().backward()
    nb = buckets.finish()
    torch.cuda.synchronize()
    assert nb == sum(p.numel() for p in net.parameters()) * 4 and len(buckets.buckets) >= 2
    assert all(torch.allclose(p.grad, w, atol=1e-6) for p, w in zip(net.parameters(), want))
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_world1_label_gather_and_gradient_buckets():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    run = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    tail = "\n".join((run.stdout + "\n" + run.stderr).splitlines()[-25:])
    assert run.returncode == 0 and "RCCL_WORLD1_OK" in run.stdout, tail
