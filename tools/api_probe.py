"""GPU-box probe of the API path: the caller's loop vs multi_dimensional_segmentation.segment_slices (device pipeline), per-slice label
comparison and timings (before / after a pipelined call)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import multi_dimensional_segmentation as mds
from micro_sam_amd import util
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd = synthetic_state_dict("vit_b", 0, variant="cells")
p = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
amg = AutomaticMaskGenerator(p)
stack = np.stack([synthetic_tile(1000 + i) for i in range(n)])


def loop(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    emb = util.precompute_image_embeddings(p, stack, ndim=3, batch_size=16, verbose=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    segs, per = [], []
    for z in range(n):
        tz = time.perf_counter()
        amg.initialize(stack[z], emb, i=z)
        segs.append(amg.generate())
        per.append(round(1e3 * (time.perf_counter() - tz), 1))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("   per tile ms:", per, flush=True)
    print(f"{tag}: embed {1e3 * (t1 - t0) / n:.2f} ms/tile, amg {1e3 * (t2 - t1) / n:.2f} ms/tile, {n / (t2 - t0):.1f} tiles/s", flush=True)
    return segs


def pipe(tag, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg, _ = mds.segment_slices(stack, p, amg, batch_size=16, decode_lanes=lanes, **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{tag}: {n / (t1 - t0):.1f} tiles/s", flush=True)
    return seg


a = loop("loop (cold)")
a = loop("loop")
s1 = pipe("pipeline (cold)")
s2 = pipe("pipeline")
s3 = pipe("pipeline")
b = loop("loop after the pipeline")
off = 0
for z in range(n):
    exp = np.where(a[z] != 0, a[z] + np.uint32(off), 0).astype(np.uint32)
    off += int(a[z].max())
    same = [bool(np.array_equal(exp, s[z])) for s in (s1, s2, s3)]
    if not all(same) or not np.array_equal(a[z], b[z]):
        print("slice", z, "pipeline == loop:", same, "loop == loop again:", bool(np.array_equal(a[z], b[z])),
              "max ids", int(a[z].max()), [int(s[z].max()) for s in (s1, s2, s3)], "differing px", [int((exp != s[z]).sum()) for s in (s1, s2, s3)], flush=True)
print("done")
