"""Timing of the strict precision mode on one GPU (DESIGN.md section 4 "cost"): the fp32 image encoder per tile, the fp32 decode of the
32 x 32 prompt grid per tile, the whole tile (precompute + AutomaticMaskGenerator.initialize + generate) next to the default path, and the
f32-input MFMA product's rate on the encoder's / decoder's shapes.  Prints one JSON line; `--out` also writes it."""
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
    fn()
    fn()                                  # (the first strict pass also grows torch's caching allocator)
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
    ap.add_argument("--model", default="vit_b")
    ap.add_argument("--no-products", action="store_true", help="skip the product shapes (they are vit_b's)")
    ap.add_argument("--modes", default="default,strict")
    ap.add_argument("--sgemm-bufs", default="", help="comma list of msam_tune_set('sgemm_bufs') values to A/B (strict tile + the product shapes)")
    ap.add_argument("--srel-mfma", default="", help="comma list of msam_tune_set('srel_mfma') values to A/B (strict encoder time)")
    ap.add_argument("--ab", default="", help="key=v1,v2,...: msam_tune_set(key, v) for each v, strict tile + encoder time under each (last v stays set)")
    ap.add_argument("--ab-py", default="", help="NAME=0,1: a module flag of micro_sam_amd.strict (FUSED_I2T, ...) to A/B on the strict tile")
    a = ap.parse_args()
    from micro_sam_amd import strict, util
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    sd = synthetic_state_dict(a.model, 0, variant="cells")
    predictor = util.get_sam_model(a.model, device="cuda", state_dict=sd)
    tiles = [synthetic_tile(1000 + i) for i in range(a.tiles)]
    amg = AutomaticMaskGenerator(predictor, device_chunk=1024)
    rec = {}

    def whole_tile(tile):
        emb = util.precompute_image_embeddings(predictor, tile, verbose=False)
        amg.initialize(tile, emb)
        return amg.generate()

    for mode in a.modes.split(","):
        predictor.set_precision(mode)
        t_tile = timed(lambda: [whole_tile(t) for t in tiles]) / len(tiles)
        t_enc = timed(lambda: [util.precompute_image_embeddings(predictor, t, verbose=False) for t in tiles]) / len(tiles)
        rec[mode] = {"seconds_per_tile_api_loop": round(t_tile, 4), "tiles_per_s_api_loop": round(1.0 / t_tile, 2),
                     "encoder_seconds_per_tile": round(t_enc, 4), "decode_and_generate_seconds_per_tile": round(t_tile - t_enc, 4)}
    # the f32-input MFMA product on the path's shapes
    dev = torch.device("cuda")
    shapes = {"enc_qkv": (4096 * 4, 2304, 768), "enc_lin1": (4096 * 4, 3072, 768), "enc_lin2": (4096 * 4, 768, 3072),
              "dec_t2i_kv": (128 * 4096, 128, 256), "dec_up1": (128 * 4096, 256, 256), "dec_up2": (128 * 16384, 128, 64),
              "dec_i2t_out": (128 * 4096, 256, 128)}
    from micro_sam_amd import _lib

    def products():
        g = {}
        for name, (M, N, K) in shapes.items():
            A = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev) / K ** 0.5
            out = torch.empty(M, N, device=dev)
            t = timed(lambda: strict.gemm(A, W, out=out), reps=5)
            g[name] = {"M": M, "N": N, "K": K, "us": round(t * 1e6, 1), "tflops": round(2.0 * M * N * K / t / 1e12, 1),
                       "gbytes_per_s": round((M * K + M * N) * 4 / t / 1e9, 0)}
        return g
    if not a.no_products:
        rec["strict_gemm"] = products()
    for v in filter(None, a.sgemm_bufs.split(",")):
        _lib.check(_lib.load().msam_tune_set(b"sgemm_bufs", int(v)), "msam_tune_set")
        predictor.set_precision("strict")
        t_tile = timed(lambda: [whole_tile(t) for t in tiles]) / len(tiles)
        rec[f"sgemm_bufs_{v}"] = {"strict_seconds_per_tile_api_loop": round(t_tile, 4), "strict_gemm": products()}
    if a.sgemm_bufs:
        _lib.check(_lib.load().msam_tune_set(b"sgemm_bufs", 1), "msam_tune_set")
    for v in filter(None, a.srel_mfma.split(",")):
        _lib.check(_lib.load().msam_tune_set(b"srel_mfma", int(v)), "msam_tune_set")
        predictor.set_precision("strict")
        t_enc = timed(lambda: [util.precompute_image_embeddings(predictor, t, verbose=False) for t in tiles]) / len(tiles)
        rec[f"srel_mfma_{v}"] = {"strict_encoder_seconds_per_tile": round(t_enc, 5)}
    if a.srel_mfma:
        _lib.check(_lib.load().msam_tune_set(b"srel_mfma", 1), "msam_tune_set")
    if a.ab:
        key, vals = a.ab.split("=")
        for v in vals.split(","):
            _lib.check(_lib.load().msam_tune_set(key.encode(), int(v)), "msam_tune_set")
            predictor.set_precision("strict")
            t_tile = timed(lambda: [whole_tile(t) for t in tiles]) / len(tiles)
            t_enc = timed(lambda: [util.precompute_image_embeddings(predictor, t, verbose=False) for t in tiles]) / len(tiles)
            rec[f"{key}_{v}"] = {"strict_seconds_per_tile_api_loop": round(t_tile, 5), "strict_encoder_seconds_per_tile": round(t_enc, 5)}
    if a.ab_py:
        key, vals = a.ab_py.split("=")
        for v in vals.split(","):
            setattr(strict, key, type(getattr(strict, key))(int(v)))
            predictor.set_precision("strict")
            t_tile = timed(lambda: [whole_tile(t) for t in tiles]) / len(tiles)
            rec[f"{key}_{v}"] = {"strict_seconds_per_tile_api_loop": round(t_tile, 5)}
    rec["fp32_mfma_peak_tflops"] = 157.3
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
