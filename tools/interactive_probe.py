"""Latency anatomy of one interactive point prompt (SamPredictor.predict on a set embedding: what the napari annotator issues per click):
host enqueue time of decode / post-processing, their device time (events), the download - and the same device work replayed from a
captured hipGraph (torch.cuda.CUDAGraph around the library calls), to see what the ~45 launches of the single-prompt decode cost.
One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import ops, util
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    p = util.get_sam_model("vit_b", device="cuda", state_dict=synthetic_state_dict("vit_b", 0, variant="cells"))
    tile = synthetic_tile(1000)
    p.set_image(util._to_image(tile))
    rng = np.random.default_rng(0)
    rec = {}
    lat = []
    for k in range(80):
        pt = rng.uniform(32, 992, size=(1, 2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p.predict(point_coords=pt, point_labels=np.ones(1), multimask_output=True)
        lat.append((time.perf_counter() - t0) * 1e3)
    rec["predict_ms_median"] = round(float(np.median(lat[10:])), 3)
    pts = torch.tensor([[[300.0, 400.0]]], device="cuda")
    lbl = torch.ones((1, 1), dtype=torch.int32, device="cuda")

    def decode():
        return p.model.decode(p.features, pts, lbl, None, None, True)

    def post(low):
        return ops.postprocess_masks(low.reshape(3, 256, 256), p.input_size, p.original_size, 0.0, 1.0, want_logits=False)
    low, iou = decode(); res = post(low)
    torch.cuda.synchronize()
    host_dec, host_post, dev_dec, dev_post = [], [], [], []
    for _ in range(50):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record(); low, iou = decode(); e1.record(); t1 = time.perf_counter()
        res = post(low); e2.record(); t2 = time.perf_counter()
        torch.cuda.synchronize()
        host_dec.append((t1 - t0) * 1e3); host_post.append((t2 - t1) * 1e3)
        dev_dec.append(e0.elapsed_time(e1)); dev_post.append(e1.elapsed_time(e2))
    rec.update(host_enqueue_decode_ms=round(float(np.median(host_dec)), 3), host_enqueue_post_ms=round(float(np.median(host_post)), 3),
               device_decode_ms=round(float(np.median(dev_dec)), 3), device_post_ms=round(float(np.median(dev_post)), 3))
    bits = res["bits"]
    t = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = ops.unpack_bits(bits, p.original_size[0]).cpu()
        t.append((time.perf_counter() - t0) * 1e3)
    rec["unpack_and_download_ms"] = round(float(np.median(t)), 3)
    # the same device work from a captured graph
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                low, iou = decode(); res = post(low)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            low_g, iou_g = decode()
            res_g = post(low_g)
        torch.cuda.synchronize()
        tt = []
        for _ in range(50):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            tt.append((time.perf_counter() - t0) * 1e3)
        rec["graph_replay_decode_post_ms"] = round(float(np.median(tt)), 3)
        low_e, iou_e = decode()
        rec["graph_equals_eager"] = bool(torch.equal(low_g, low_e) and torch.equal(iou_g, iou_e))
        tt = []
        for _ in range(50):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            low, iou = decode(); res = post(low)
            torch.cuda.synchronize()
            tt.append((time.perf_counter() - t0) * 1e3)
        rec["eager_decode_post_synced_ms"] = round(float(np.median(tt)), 3)
    except Exception as exc:                       # capture may be refused (an unsupported call inside the library's entry points)
        rec["graph_error"] = repr(exc)[:300]
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
