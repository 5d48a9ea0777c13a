"""GPU-box experiment: pure host cost of enqueueing one tile of the AMG path (queue empty at the start, so no
back-pressure), against the GPU time of the same work; with and without the per-launch HIP-event profiling of bench.py.
    python tools/host_cost.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, util  # noqa: E402
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator  # noqa: E402
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict("vit_b", 0, variant="blobs")
    predictor = util.get_sam_model("vit_b", device=dev, state_dict=sd)
    amg = AutomaticMaskGenerator(predictor, device_chunk=1024)
    tiles_np = [synthetic_tile(1000 + i) for i in range(8)]
    tiles_u8 = torch.stack([torch.as_tensor(util._to_image(t)) for t in tiles_np]).to(dev)
    lib = _lib.load()

    def encode():
        return predictor.model.image_encoder.forward_u8(tiles_u8).unsqueeze(1)

    feats = encode()
    emb = {"features": feats, "input_size": (1024, 1024), "original_size": (1024, 1024)}
    for i in range(2):
        amg.initialize(tiles_np[i], emb, i=i)
        amg.generate_device()
    torch.cuda.synchronize()
    for prof in (0, 1, 0):
        lib.msam_profile_enable(prof)
        rows = []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            amg.initialize(tiles_np[i], emb, i=i)
            t1 = time.perf_counter()
            lab, flag = amg.generate_device()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        encode()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if prof:
            import ctypes as C
            NF = _lib.PROFILE_FAMILIES
            lib.msam_profile_collect_family((C.c_int32 * NF)(), (C.c_double * NF)(), (C.c_double * NF)(), (C.c_double * NF)())
        r = rows[2:]
        n = len(r)
        print(f"profile={prof}: per tile host initialize {sum(x[0] for x in r)/n*1e3:.2f} ms, host generate "
              f"{sum(x[1] for x in r)/n*1e3:.2f} ms, enqueue->done {sum(x[2] for x in r)/n*1e3:.2f} ms; "
              f"encoder B=8 host {(t1-t0)*1e3:.2f} ms, done {(t2-t0)*1e3:.2f} ms", flush=True)
    lib.msam_profile_enable(0)
    # python-level breakdown of the host side
    import cProfile
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(8):
        amg.initialize(tiles_np[i], emb, i=i)
        amg.generate_device()
        torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
