import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from micro_sam_amd import multi_dimensional_segmentation as mds
from micro_sam_amd import util
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
n = 32
sd = synthetic_state_dict("vit_b", 0, variant="cells")
p = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
amg = AutomaticMaskGenerator(p)
stack = np.stack([synthetic_tile(1000 + i) for i in range(n)])
def loop(tag, prof=False):
    emb = util.precompute_image_embeddings(p, stack, ndim=3, batch_size=16, verbose=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    if prof: pr.enable()
    ti = tg = 0.0
    for z in range(n):
        t0 = time.perf_counter()
        amg.initialize(stack[z], emb, i=z)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        amg.generate()
        t2 = time.perf_counter()
        ti += t1 - t0; tg += t2 - t1
    if prof:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    print(tag, f"initialize {1e3*ti/n:.2f} ms, generate {1e3*tg/n:.2f} ms; reserved GB {torch.cuda.memory_reserved()/2**30:.1f}", flush=True)
loop("before"); loop("before", True)
mds.segment_slices(stack, p, amg, batch_size=16)
loop("after")
