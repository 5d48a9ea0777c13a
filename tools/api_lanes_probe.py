"""GPU-box probe (round 4): tiles/s of the pipelined multi_dimensional_segmentation.segment_slices over decode lanes x encoder batch sizes (64 tiles)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from micro_sam_amd import multi_dimensional_segmentation as mds
from micro_sam_amd import util
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
n = 64
sd = synthetic_state_dict("vit_b", 0, variant="cells")
p = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
amg = AutomaticMaskGenerator(p)
stack = np.stack([synthetic_tile(1000 + i) for i in range(n)])
for lanes, bs in ((3, 16), (3, 16), (4, 16), (3, 8), (4, 8), (3, 32), (2, 16)):
    mds.segment_slices(stack[:8], p, amg, batch_size=bs, decode_lanes=lanes)
    best = 0
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        seg, _ = mds.segment_slices(stack, p, amg, batch_size=bs, decode_lanes=lanes)
        torch.cuda.synchronize(); best = max(best, n / (time.perf_counter() - t0))
    print(f"lanes {lanes} batch {bs}: {best:.1f} tiles/s", flush=True)
