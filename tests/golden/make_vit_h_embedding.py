"""Generate tests/golden/vit_h_embedding_tile22.npz: the FULL 32-block vit_h image embedding of the oracle (fp32 = the
reference CPU arithmetic, and the oracle's bf16 mode = the HIP path's rounding points) for the synthetic vit_h checkpoint
(seed 2) on synthetic tile 22, so that the GPU test compares every block's effect without minutes of CPU work on the GPU box.

    python tests/golden/make_vit_h_embedding.py

Stored: every 8th channel of both embeddings as float16 ([32, 64, 64]) + whole-tensor sums.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import sam_ref as S  # noqa: E402


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = synthetic_state_dict("vit_h", 2)
    tile = synthetic_tile(22)
    x = S.preprocess(torch.as_tensor(A.to_image(tile)).permute(2, 0, 1)[None])
    out = {}
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            e = S.image_encoder(sd, x, model_type="vit_h", precision=prec)[0]
            out[f"sub_{prec}"] = e[::8].numpy().astype(np.float16)
            out[f"abs_sum_{prec}"] = np.float64(e.double().abs().sum().item())
            print(prec, float(e.std()), out[f"abs_sum_{prec}"])
    path = os.path.join(HERE, "vit_h_embedding_tile22.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
