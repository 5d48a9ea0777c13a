"""Why does TiledAutomaticMaskGenerator.initialize take 6.2 ms per 1024^2 tile in config 3 when the bench's pipelined path decodes a tile in 3.8 ms?
Wall time of initialize for 1 / 3 / 5 lanes, its host enqueue time (no synchronisation), and a host profile.  One JSON line + the profile on stderr."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import util
    from micro_sam_amd.instance_segmentation import TiledAutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    model = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
    img = synthetic_tile(3000, (2048, 2048))
    predictor = util.get_sam_model(model, device="cuda", state_dict=synthetic_state_dict(model, 0, variant="cells"))
    kw = dict(tile_shape=(768, 768), halo=(128, 128))
    emb = util.precompute_image_embeddings(predictor, img, ndim=2, batch_size=9, verbose=False, **kw)
    rec = {}
    for lanes in (1, 3, 5):
        seg = TiledAutomaticMaskGenerator(predictor, tile_lanes=lanes)
        for _ in range(2):
            seg.initialize(img, image_embeddings=emb)
        torch.cuda.synchronize()
        wall, enq = [], []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seg.initialize(img, image_embeddings=emb)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            wall.append((t2 - t0) * 1e3); enq.append((t1 - t0) * 1e3)
        rec[f"lanes_{lanes}"] = {"initialize_ms": round(float(np.median(wall)), 2), "host_enqueue_ms": round(float(np.median(enq)), 2)}
    seg = TiledAutomaticMaskGenerator(predictor, tile_lanes=3)
    seg.initialize(img, image_embeddings=emb)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    seg.initialize(img, image_embeddings=emb)
    torch.cuda.synchronize()
    pr.disable()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(40)
    print(sio.getvalue()[-6000:], file=sys.stderr)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
