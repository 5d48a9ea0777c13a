import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from micro_sam_amd import multi_dimensional_segmentation as mds
from micro_sam_amd import util
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
n = 8
sd = synthetic_state_dict("vit_b", 0, variant="cells")
p = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
amg = AutomaticMaskGenerator(p, device_chunk=1024)
stack = np.stack([synthetic_tile(1000 + i) for i in range(n)])
def loop():
    emb = util.precompute_image_embeddings(p, stack, ndim=3, batch_size=16, verbose=False)
    out = np.zeros(stack.shape, np.uint32); off = 0
    for z in range(n):
        amg.initialize(stack[z], emb, i=z)
        s = amg.generate()
        out[z] = np.where(s != 0, s + np.uint32(off), 0); off += int(s.max())
    return out
def cmp(tag, a, b):
    d = [int((a[z] != b[z]).sum()) for z in range(n)]
    print(tag, "differing px per slice:", d, "max ids", int(a.max()), int(b.max()), flush=True)
ref = loop()
s0, _ = mds.segment_slices(stack, p, amg, batch_size=16); cmp("pipe first   ", ref, s0)
w, _ = mds.segment_slices(stack[:4], p, amg, batch_size=16); cmp("pipe [:4]    ", ref[:4].tolist() and ref, np.concatenate([w, ref[4:]]))
s1, _ = mds.segment_slices(stack, p, amg, batch_size=16); cmp("pipe after :4", ref, s1)
s2, _ = mds.segment_slices(stack, p, amg, batch_size=16); cmp("pipe again   ", ref, s2)
cmp("loop again   ", ref, loop())
s3, _ = mds.segment_slices(stack, p, amg, batch_size=16, decode_lanes=1); cmp("pipe 1 lane  ", ref, s3)
