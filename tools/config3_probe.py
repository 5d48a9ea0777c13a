"""Where a 2048^2 slice of BASELINE configs[2] (vit_l, tile 768 + halo 128, TiledAutomaticMaskGenerator) spends its time: synchronised
phase timings of the per-slice path (encoder of the slice's 9 tiles, initialize = decode of the tiles on the lanes, generate = filters +
NMS + cross-tile NMS + label image, host-side offset arithmetic).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd import util
    from micro_sam_amd.instance_segmentation import TiledAutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    model = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
    Z = 4
    vol = np.stack([synthetic_tile(3000 + z, (2048, 2048)) for z in range(Z)])
    predictor = util.get_sam_model(model, device="cuda", state_dict=synthetic_state_dict(model, 0, variant="cells"))
    seg = TiledAutomaticMaskGenerator(predictor)
    kw = dict(tile_shape=(768, 768), halo=(128, 128))
    mds.segment_slices(vol[:2], predictor, seg, batch_size=18, **kw)          # warm-up
    torch.cuda.synchronize()
    rec = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        rec.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3, 2))
        return out
    emb = timed("encode_all_slices_ms", lambda: util.precompute_image_embeddings(predictor, vol, ndim=3, batch_size=9 * Z, verbose=False, **kw))
    for z in range(Z):
        timed("initialize_ms", lambda: seg.initialize(vol[z], image_embeddings=emb, i=z))
        t0 = time.perf_counter()
        lab = timed("generate_ms", lambda: seg.generate())
        t1 = time.perf_counter()
        mx = int(lab.max()); lab[lab != 0] += 7
        rec.setdefault("host_offsets_ms", []).append(round((time.perf_counter() - t1) * 1e3, 2))
    for steps, fn in (("loop_slices_per_s", lambda: mds.segment_slices(vol, predictor, seg, batch_size=9 * Z, decode_lanes=0, **kw)),
                      ("overlapped_slices_per_s", lambda: mds.segment_slices(vol, predictor, seg, batch_size=18, **kw)),
                      ("overlapped_bs9_slices_per_s", lambda: mds.segment_slices(vol, predictor, seg, batch_size=9, **kw))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        rec[steps] = round(2 * Z / (time.perf_counter() - t0), 2)
    # host profile of one generate()
    import cProfile
    import pstats
    import io
    seg.initialize(vol[0], image_embeddings=emb, i=0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    seg.generate()
    pr.disable()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(25)
    print(sio.getvalue()[-3500:], file=sys.stderr)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
