"""Test-only helpers: mirror of the decoder workspace layout (csrc/decoder.hip ``carve_work``) so that tests can
read intermediate buffers after a ``msam_decoder_forward`` call."""
from __future__ import annotations

from typing import Dict

import torch

T, C, CI = 4096, 256, 128


def _al(x: int) -> int:
    return (x + 255) & ~255


def decoder_workspace_views(ws: torch.Tensor, P: int, Nt: int) -> Dict[str, torch.Tensor]:
    M, R = P * Nt, P * T
    off = 0
    out = {}

    def take(name, nbytes, dtype, shape):
        nonlocal off
        out[name] = ws[off:off + nbytes].view(dtype).reshape(shape)
        off += _al(nbytes)

    take("qpe", M * C * 4, torch.float32, (P, Nt, C))
    take("queries", M * C * 4, torch.float32, (P, Nt, C))
    take("tmp", M * C * 4, torch.float32, (P, Nt, C))
    for nm in ("a", "b", "qs", "ks", "vs", "attn_tok"):
        take(nm, M * C * 2, torch.bfloat16, (M, C))
    take("mlp_h", M * 2048 * 2, torch.bfloat16, (M, 2048))
    take("keys", R * C * 2, torch.bfloat16, (P, T, C))
    take("kimg", R * CI * 2, torch.bfloat16, (P, T, CI))
    take("vT", R * CI * 2, torch.bfloat16, (P, CI, T))
    take("qimg", R * CI * 2, torch.bfloat16, (P, T, CI))
    take("attn_img", R * CI * 2, torch.bfloat16, (P, T, CI))
    take("up1", R * C * 2, torch.bfloat16, (P, T, 4, 64))
    take("pre", R * C * 4, torch.float32, (P, T, C))
    take("hh0", P * C * 2, torch.bfloat16, (P, C))
    take("hh1", P * C * 2, torch.bfloat16, (P, C))
    take("hyper", P * 4 * 128 * 4, torch.float32, (P, 4, 128))
    take("iou_full", P * 128 * 4, torch.float32, (P, 128))
    return out
