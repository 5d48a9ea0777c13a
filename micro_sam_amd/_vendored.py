"""Device versions of ``micro_sam/_vendored.py``: ``batched_mask_to_box`` (:33-85) and ``mask_to_rle_pytorch``
(:114-152).  Both take bool masks [N,H,W] on the GPU; the AMG hot path never builds those (it goes from low-res
logits to bit masks directly, see ``ops.postprocess_masks``) - these entry points keep the reference's signatures
for other callers and for the parity tests."""
from __future__ import annotations

from typing import Any, Dict, List

import torch

from . import ops


def pack_bits(masks: torch.Tensor) -> torch.Tensor:
    """bool [N,H,W] -> bit masks int32 [N, ceil(H/32), W] (bit b of word yw = row yw*32 + b).  Torch ops only."""
    n, h, w = masks.shape
    wpc = (h + 31) // 32
    m = torch.zeros((n, wpc * 32, w), dtype=torch.int64, device=masks.device)
    m[:, :h] = masks.to(torch.int64)
    m = m.reshape(n, wpc, 32, w)
    weights = (1 << torch.arange(32, device=masks.device, dtype=torch.int64)).view(1, 1, 32, 1)
    words = (m * weights).sum(dim=2)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """XYXY (inclusive) boxes around bool masks [..., H, W]; [0,0,0,0] for empty masks."""
    assert masks.dtype == torch.bool, masks.dtype
    if torch.numel(masks) == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    shape = masks.shape
    h, w = shape[-2:]
    flat = masks.reshape(-1, h, w)
    rows, cols = flat.any(dim=-1), flat.any(dim=-2)
    ar_h = torch.arange(h, device=masks.device, dtype=torch.int32)
    ar_w = torch.arange(w, device=masks.device, dtype=torch.int32)
    big_h, big_w = torch.full_like(ar_h, h), torch.full_like(ar_w, w)
    bottom = torch.where(rows, ar_h, torch.zeros_like(ar_h)).amax(-1)
    top = torch.where(rows, ar_h, big_h).amin(-1)
    right = torch.where(cols, ar_w, torch.zeros_like(ar_w)).amax(-1)
    left = torch.where(cols, ar_w, big_w).amin(-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1) * (~empty).unsqueeze(-1)
    return out.reshape(*shape[:-2], 4) if len(shape) > 2 else out[0]


def mask_to_rle_pytorch(tensor: torch.Tensor, rle_implementation: str = "default") -> List[Dict[str, Any]]:
    """Column-major uncompressed RLE ({"size": [h, w], "counts": [...]}) computed by the HIP RLE kernels."""
    b, h, w = tensor.shape
    if b == 0:
        return []
    bits = pack_bits(tensor.to(torch.bool))
    counts, offsets = ops.rle_encode(bits, h, w)
    return ops.rles_to_list(counts, offsets, h, w)
