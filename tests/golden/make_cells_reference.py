"""Generate tests/golden/cells_vit_b_tile{seed}.npz: the fp32 CPU reference (oracle = restated reference path) of the
BENCHMARKED configuration on one tile - synthetic "cells" checkpoint (vit_b, seed 0), synthetic tile `seed`, 32x32 prompt
grid, 64 prompts per batch, default thresholds - so that the GPU parity test compares the HIP path with reference outputs
without re-running ~80 s of CPU work on the GPU box.

    python tests/golden/make_cells_reference.py [tile_seed [weight_seed [variant [logit_scale]]]]

The default (tile 1000, weights "cells" seed 0, logit scale 1) is the benchmarked configuration; the other fixtures
(``golden_name``) are the sensitivity cases of tests/test_gpu_parity_iou.py: other weight seeds, the "field" checkpoint, and the
"cells" checkpoint with its mask logits scaled by 0.25 (softer boundaries).

Stored: scores of all 3072 candidates (predicted IoU, stability, boxes), the candidates the reference keeps
(``_postprocess_batch``), their masks as uncompressed column-major RLE (the reference's own mask format), the final
uint32 label image, and the fp32 embedding's checksum.  /root/reference cannot be imported; the oracle
is pinned against transformers' SAM and the reference's known-answer tests (tests/test_oracle_*.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import parity as PT  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402


def golden_name(tile: int, wseed: int = 0, variant: str = "cells", scale: float = 1.0) -> str:
    if wseed == 0 and variant == "cells" and scale == 1.0:
        return f"cells_vit_b_tile{tile}.npz"
    return f"{variant}_vit_b_w{wseed}_x{scale:g}_tile{tile}.npz"


def main():
    from micro_sam_amd.synthetic import scale_mask_logits
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    variant = sys.argv[3] if len(sys.argv) > 3 else "cells"
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = synthetic_state_dict("vit_b", wseed, variant=variant)
    if scale != 1.0:
        scale_mask_logits(sd, scale)
    tile = synthetic_tile(seed)
    img = A.to_image(tile)
    feats, osz, isz = PR.compute_embeddings(sd, [img], "vit_b", "fp32")
    state = PR.amg_initialize(sd, img, feats, isz[0], osz[0], precision="fp32")
    seg = PR.amg_generate(state)
    d = state["crop_list"][0]
    kept = PT.kept_candidates(state)
    counts = [np.asarray(d["rles"][i]["counts"], dtype=np.int32) for i in kept]
    offsets = np.zeros(len(kept) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(c) for c in counts])
    out = os.path.join(HERE, golden_name(seed, wseed, variant, scale))
    np.savez_compressed(
        out, iou_preds=d["iou_preds"].numpy().astype(np.float32), stability=d["stability_score"].numpy().astype(np.float32),
        boxes=d["boxes"].numpy().astype(np.int32), kept=kept.astype(np.int32),
        rle_counts=np.concatenate(counts) if counts else np.zeros(0, np.int32), rle_offsets=offsets,
        labels=seg.astype(np.uint16 if seg.max() < 65536 else np.uint32),
        embedding_sum=np.float64(feats.double().sum().item()), embedding_abs_sum=np.float64(feats.double().abs().sum().item()))
    print(out, os.path.getsize(out), "bytes; kept", len(kept), "instances", int(seg.max()))


if __name__ == "__main__":
    main()
