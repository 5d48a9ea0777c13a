"""Latency of ONE image embedding (batch 1) per encoder, default mode: python tools/embed_latency_probe.py -> one JSON line.
(precompute_image_embeddings of one 1024^2 tile incl. the host transfers; and the encoder alone on a resident uint8 tile.)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    rec = {}
    tile = synthetic_tile(5)
    for model in ("vit_b", "vit_l", "vit_h"):
        predictor = util.get_sam_model(model, device="cuda", state_dict=synthetic_state_dict(model, 0, variant="cells"))
        for _ in range(4):
            util.precompute_image_embeddings(predictor, tile, verbose=False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            util.precompute_image_embeddings(predictor, tile, verbose=False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        u8 = torch.from_numpy(np.repeat(tile[None, :, :, None], 3, axis=3).copy()).cuda()
        enc = predictor.model.image_encoder
        for _ in range(3):
            enc.forward_u8(u8)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            enc.forward_u8(u8)
        b.record()
        torch.cuda.synchronize()
        rec[model] = {"precompute_ms": round(float(np.median(ts)) * 1e3, 2), "encoder_device_ms": round(a.elapsed_time(b) / 10, 2)}
        del predictor
        torch.cuda.empty_cache()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
