"""The three precision modes side by side on one GPU: per-tile time of the literal API loop (precompute_image_embeddings ->
AutomaticMaskGenerator.initialize -> generate), encoder time, and the agreement of the default / split16 masks with the strict mode's
(fp32 kernels: the on-device stand-in of the fp32 CPU oracle; tests/test_gpu_strict.py pins strict == oracle) - per-instance IoU of the kept
masks, keep sets, label images.  Also the product's rate on the path's shapes in the strict and the split16 form.  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def timed(fn, reps=3):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--modes", default="default,split16,strict")
    ap.add_argument("--no-products", action="store_true")
    ap.add_argument("--weights", default="cells")
    ap.add_argument("--slice-tiles", type=int, default=16, help="tiles of the segment_slices timing (0: skip)")
    a = ap.parse_args()
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd import ops, strict, util
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    sd = synthetic_state_dict("vit_b", 0, variant=a.weights)
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
    tiles = [synthetic_tile(1000 + i) for i in range(a.tiles)]
    amg = AutomaticMaskGenerator(predictor, device_chunk=1024)
    rec, results = {}, {}

    def whole_tile(tile):
        emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
        amg.initialize(tile, emb)
        return amg.generate()

    def snapshot(tile):
        emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
        amg.initialize(tile, emb)
        data = amg.crop_list[0]
        n = len(data)
        cand = data.shallow_copy()
        cand["cand"] = torch.arange(n, device=data["iou_preds"].device)
        kept = amg._postprocess_batch(cand, amg.crop_boxes[0], amg.original_size, 0.88, 0.95, 0.7)["cand"].cpu().numpy()
        return {"emb": torch.as_tensor(emb["features"]).float().cpu(), "bits": data["bits"].clone(), "kept": kept,
                "iou_preds": data["iou_preds"].float().cpu(), "stab": data["stability_score"].float().cpu(), "seg": amg.generate()}

    for mode in a.modes.split(","):
        predictor.set_precision(mode)
        results[mode] = [snapshot(t) for t in tiles]
        t_tile = timed(lambda: [whole_tile(t) for t in tiles]) / len(tiles)
        t_enc = timed(lambda: [util.precompute_image_embeddings(predictor, t, verbose=False) for t in tiles]) / len(tiles)
        # the product's own slice loop (encoder batches of 8, decode lanes): multi_dimensional_segmentation.segment_slices, host arrays in and out
        stack = np.stack([synthetic_tile(1000 + i) for i in range(a.slice_tiles)])
        t_pipe = None
        if a.slice_tiles:
            t_pipe = timed(lambda: mds.segment_slices(stack, predictor, amg, batch_size=8), reps=2) / a.slice_tiles
        rec[mode] = {"seconds_per_tile_api_loop": round(t_tile, 4), "tiles_per_s_api_loop": round(1.0 / t_tile, 2),
                     "tiles_per_s_segment_slices": None if t_pipe is None else round(1.0 / t_pipe, 2),
                     "encoder_seconds_per_tile": round(t_enc, 4), "decode_and_generate_seconds_per_tile": round(t_tile - t_enc, 4)}
    if "strict" in results:
        for mode in results:
            if mode == "strict":
                continue
            ious, both, only_ref, only_test, emb_err, iou_pred_err, same_px = [], 0, 0, 0, 0.0, 0.0, []
            for r, t in zip(results["strict"], results[mode]):
                kr, kt = set(r["kept"].tolist()), set(t["kept"].tolist())
                both += len(kr & kt); only_ref += len(kr - kt); only_test += len(kt - kr)
                emb_err = max(emb_err, float((r["emb"] - t["emb"]).abs().max()))
                iou_pred_err = max(iou_pred_err, float((r["iou_preds"] - t["iou_preds"]).abs().max()))
                for i in sorted(kr):
                    mr = ops.unpack_bits(r["bits"][i:i + 1], 1024)[0]
                    mt = ops.unpack_bits(t["bits"][i:i + 1], 1024)[0]
                    inter = float((mr & mt).sum()); union = float((mr | mt).sum())
                    ious.append(inter / union if union else 1.0)
                same_px.append(float((r["seg"] == t["seg"]).mean()))
            ious = np.array(ious)
            rec[mode]["vs_strict"] = {"n_instances": int(ious.size), "frac_ge_0.999": round(float((ious >= 0.999).mean()), 4),
                                      "frac_eq_1": round(float((ious == 1.0).mean()), 4), "min": round(float(ious.min()), 5),
                                      "keep_set": {"both": both, "ref_only": only_ref, "test_only": only_test},
                                      "embedding_max_abs_diff": emb_err, "iou_pred_max_abs_diff": iou_pred_err,
                                      "identical_label_px_frac": round(float(np.mean(same_px)), 6)}
    dev = torch.device("cuda")
    shapes = {"enc_qkv": (4096 * 4, 2304, 768), "enc_lin1": (4096 * 4, 3072, 768), "enc_lin2": (4096 * 4, 768, 3072),
              "dec_t2i_kv": (128 * 4096, 128, 256), "dec_up1": (128 * 4096, 256, 256), "dec_up2": (128 * 16384, 128, 64),
              "dec_i2t_out": (128 * 4096, 256, 128), "tok_mlp1": (512 * 7, 2048, 256), "tok_q": (512 * 7, 256, 256)}
    if not a.no_products:
        for split in (False, True):
            g = {}
            with strict.split_mode(split):
                for name, (M, N, K) in shapes.items():
                    A = torch.randn(M, K, device=dev)
                    W = torch.randn(N, K, device=dev) / K ** 0.5
                    out = torch.empty(M, N, device=dev)
                    t = timed(lambda: strict.gemm(A, W, out=out), reps=5)
                    ref = (A[:256].double() @ W.double().T)
                    err = float(((out[:256].double() - ref).abs().mean() / ref.abs().mean()))
                    g[name] = {"M": M, "N": N, "K": K, "us": round(t * 1e6, 1), "tflops": round(2.0 * M * N * K / t / 1e12, 1),
                               "gbytes_per_s": round((M * K + M * N) * 4 / t / 1e9, 0), "mean_rel_err_vs_fp64": err}
            rec["products_split16" if split else "products_strict"] = g
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
