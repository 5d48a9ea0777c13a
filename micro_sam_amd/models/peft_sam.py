"""LoRA surgery of the image encoder (reference ``micro_sam/models/peft_sam.py:16-146,392-460``).

``PEFT_Sam(sam, rank, peft_module=LoRASurgery, ...)`` replaces ``block.attn.qkv`` (and ``block.mlp`` with
``update_matrices=[..., "mlp"]``) by modules with the reference's names and parameters (``qkv_proj``, ``w_a_linear_q``,
``w_b_linear_q``, ...), so LoRA checkpoints of micro_sam ``load_state_dict`` unchanged.  At inference a low-rank update is
a weight update - ``qkv(x) + B A x == (W + B A) x`` - so the HIP encoder needs no extra kernel: the modules expose the MERGED
``weight`` / ``bias`` that ``ImageEncoderViT._prepare`` turns into its 16-bit operand copies (exact; alpha = 1 as in the
reference).  Training keeps the low-rank branches as separate products so that A and B receive gradients
(``training/encoders.py`` ``_qkv_projection`` / ``_mlp``: the frozen projection plus ``alpha * B(A(x))``; first GPU run
pending, composition checked on the CPU in tests/test_training_encoders_host.py).  The other PEFT methods of the reference
(FacT, SSF, AdaptFormer, selective / classical surgery) are not provided.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn


class AttentionLoRA(nn.Module):
    def __init__(self, rank: int, block: nn.Linear, update_matrices: List[str] = ["q", "v"]):
        super().__init__()
        self.qkv_proj = block
        self.dim = self.qkv_proj.in_features
        self.alpha = 1
        self.rank = rank
        for m in ("q", "v", "k"):
            if m in update_matrices:
                setattr(self, f"w_a_linear_{m}", nn.Linear(self.dim, rank, bias=False))
                setattr(self, f"w_b_linear_{m}", nn.Linear(rank, self.dim, bias=False))
        self.reset_parameters()

    def reset_parameters(self):
        for m in ("q", "v", "k"):
            if hasattr(self, f"w_a_linear_{m}"):
                nn.init.kaiming_uniform_(getattr(self, f"w_a_linear_{m}").weight, a=math.sqrt(5))
                nn.init.zeros_(getattr(self, f"w_b_linear_{m}").weight)

    @property
    def in_features(self) -> int:
        return self.qkv_proj.in_features

    @property
    def out_features(self) -> int:
        return self.qkv_proj.out_features

    @property
    def weight(self) -> torch.Tensor:
        """[3 dim, dim] with the low-rank updates merged into the q / k / v row blocks."""
        w = self.qkv_proj.weight.detach().clone()
        for m, sl in (("q", slice(0, self.dim)), ("k", slice(self.dim, 2 * self.dim)), ("v", slice(2 * self.dim, 3 * self.dim))):
            if hasattr(self, f"w_a_linear_{m}"):
                w[sl] += self.alpha * (getattr(self, f"w_b_linear_{m}").weight.detach() @ getattr(self, f"w_a_linear_{m}").weight.detach())
        return w

    @property
    def bias(self) -> torch.Tensor:
        return self.qkv_proj.bias


class _MergedLinear:
    """``lin1`` / ``lin2`` of an ``MLPLoRA`` as the encoder reads them: merged weight, original bias."""

    def __init__(self, lin: nn.Linear, a: nn.Linear, b: nn.Linear):
        self._lin, self._a, self._b = lin, a, b

    @property
    def weight(self) -> torch.Tensor:
        return self._lin.weight.detach() + self._b.weight.detach() @ self._a.weight.detach()

    @property
    def bias(self) -> torch.Tensor:
        return self._lin.bias

    @property
    def in_features(self) -> int:
        return self._lin.in_features

    @property
    def out_features(self) -> int:
        return self._lin.out_features


class MLPLoRA(nn.Module):
    def __init__(self, rank: int, mlp_layer: nn.Module):
        super().__init__()
        self.mlp_layer = mlp_layer
        self.rank = rank
        self.w_a_linear_1 = nn.Linear(mlp_layer.lin1.in_features, rank, bias=False)
        self.w_b_linear_1 = nn.Linear(rank, mlp_layer.lin1.out_features, bias=False)
        self.w_a_linear_2 = nn.Linear(mlp_layer.lin2.in_features, rank, bias=False)
        self.w_b_linear_2 = nn.Linear(rank, mlp_layer.lin2.out_features, bias=False)
        self.activation = mlp_layer.act
        nn.init.kaiming_uniform_(self.w_a_linear_1.weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.w_a_linear_2.weight, a=math.sqrt(5))
        nn.init.zeros_(self.w_b_linear_1.weight)
        nn.init.zeros_(self.w_b_linear_2.weight)

    @property
    def lin1(self):
        return _MergedLinear(self.mlp_layer.lin1, self.w_a_linear_1, self.w_b_linear_1)

    @property
    def lin2(self):
        return _MergedLinear(self.mlp_layer.lin2, self.w_a_linear_2, self.w_b_linear_2)

    @property
    def act(self):
        return self.activation


class LoRASurgery(nn.Module):
    def __init__(self, rank: int, block: nn.Module, update_matrices: List[str] = ["q", "v"]):
        super().__init__()
        if set(update_matrices) - set(["q", "k", "v", "mlp"]):
            raise ValueError(f"Some of the expected keys for updating matrics in '{update_matrices}' are not expected.")
        self.block = block
        block.attn.qkv = AttentionLoRA(rank=rank, block=block.attn.qkv, update_matrices=update_matrices)
        if "mlp" in update_matrices:
            block.mlp = MLPLoRA(rank=rank, mlp_layer=block.mlp)

    def forward(self, x):
        return x


class PEFT_Sam(nn.Module):
    """Reference ``PEFT_Sam`` for ``peft_module=LoRASurgery``; ``.sam`` is the operated model."""

    def __init__(self, model, rank: Optional[int] = None, peft_module=LoRASurgery,
                 attention_layers_to_update: Optional[List[int]] = None, quantize: bool = False, **module_kwargs):
        super().__init__()
        if peft_module is not LoRASurgery:
            raise NotImplementedError("micro_sam_amd: only LoRASurgery is provided (FacT / SSF / AdaptFormer / selective surgery "
                                      "are not)")
        if quantize:
            raise NotImplementedError("micro_sam_amd: QLoRA (bitsandbytes 4-bit) is not provided")
        if not rank or rank <= 0:
            raise RuntimeError("The chosen PEFT method cannot run without a valid rank choice.")
        self.peft_layers = attention_layers_to_update or list(range(len(model.image_encoder.blocks)))
        self.peft_module = peft_module
        self.peft_blocks = []
        for param in model.image_encoder.parameters():
            param.requires_grad = False
        for t_layer_i, blk in enumerate(model.image_encoder.blocks):
            if t_layer_i not in self.peft_layers:
                continue
            self.peft_blocks.append(self.peft_module(rank=rank, block=blk, **module_kwargs))
        self.peft_blocks = nn.ModuleList(self.peft_blocks)
        self.sam = model
        model.image_encoder.invalidate()

    def forward(self, batched_input, multimask_output):
        raise NotImplementedError("micro_sam_amd: use the SamPredictor / TrainableSAM interfaces")
