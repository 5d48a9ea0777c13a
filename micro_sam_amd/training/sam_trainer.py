"""``SamTrainer`` (reference ``micro_sam/training/sam_trainer.py:131-425``) without the torch_em base class (absent here):
the iterative-prompting loss, the train step and a plain ``fit`` loop; data-parallel training all-reduces the gradients
over ``torch.distributed`` (RCCL on ROCm; one process per GPU) in flat fp32 buckets.

Kept from the reference, same names and arithmetic: ``_get_prompt_and_multimasking_choices``, ``_compute_iou``,
``_compute_loss`` (dice of the best of the 1 / 3 masks per object + MSE between predicted and true IoU),
``_get_best_masks``, ``_use_mask_inputs`` (rank-0 decision broadcast to all ranks), ``_compute_iterative_loss``,
``_update_prompts``, ``_preprocess_batch``, ``_interactive_train_iteration``.
"""
from __future__ import annotations

import random
from typing import Callable, Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from ..prompt_generators import IterativePromptGenerator, PromptGeneratorBase


def dice_loss_per_channel(prediction: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """torch_em ``DiceLoss(reduce_channel=None)`` (the reference's mask loss): inputs [N, C, ...]; per channel
    1 - 2 sum(p t) / (sum(p^2) + sum(t^2)) over the flattened samples."""
    C = prediction.shape[1]
    p = prediction.transpose(0, 1).reshape(C, -1)
    t = target.transpose(0, 1).reshape(C, -1)
    num = (p * t).sum(-1)
    den = (p * p).sum(-1) + (t * t).sum(-1)
    return 1.0 - 2.0 * num / den.clamp(min=eps)


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20) -> int:
    """Average the gradients over the ranks of the default process group: flat fp32 buckets of ``bucket_bytes`` (64 MiB: a few
    large RCCL all-reduces over xGMI instead of one per tensor), in place.  Returns the number of bytes reduced."""
    from ..parallel import collectives_active
    if not collectives_active():
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in parameters if p.requires_grad and p.grad is not None]
    total, bucket, size = 0, [], 0

    def flush():
        nonlocal bucket, size, total
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1).float() for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].reshape(g.shape))
            off += n
        total += flat.numel() * 4
        bucket, size = [], 0
    for g in grads:
        bucket.append(g)
        size += g.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
    return total


class GradientBuckets:
    """Data-parallel gradient exchange overlapped with the backward pass (the reference trains under torch DDP,
    ``finetuning/specialists/training/light_microscopy/livecell_multi_gpu_finetuning.py:56-82``: bucketed all-reduce from autograd hooks).

    The parameters' ``.grad`` tensors ARE slices of flat fp32 buckets (``bucket_bytes`` each; parameters in reverse registration order =
    roughly the order in which backward finishes them): no ``cat`` before the all-reduce and no copy back after it (the two extra passes
    over the gradients of the round-4 form).  A post-accumulate hook counts a bucket's finished gradients; the last one starts that
    bucket's asynchronous all-reduce - on a communication stream that waits for the producing stream at that point - while backward
    continues with the layers in front of it.  ``finish()`` starts whatever is left (parameters that got no gradient in this step),
    waits, and divides by the world size.  ``zero()`` replaces ``optimizer.zero_grad()`` (which would drop the views)."""

    def __init__(self, parameters: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20) -> None:
        self.params = [p for p in parameters if p.requires_grad]
        self.world = dist.get_world_size()
        self.buckets: List[dict] = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._make_bucket(cur)
                cur, size = [], 0
        if cur:
            self._make_bucket(cur)
        self._comm = None
        self._handles = []
        for b in self.buckets:
            for q in b["params"]:
                b["hooks"].append(q.register_post_accumulate_grad_hook(lambda _p, b=b: self._ready(b, _p)))

    def _make_bucket(self, params) -> None:
        dev = params[0].device
        flat = torch.zeros(sum(q.numel() for q in params), dtype=torch.float32, device=dev)
        views, off = [], 0
        for q in params:
            views.append(flat[off:off + q.numel()].view(q.shape))
            off += q.numel()
        self.buckets.append({"flat": flat, "params": list(params), "views": views, "ready": 0, "launched": False, "hooks": [], "fired": set()})
        self._attach(self.buckets[-1])

    @staticmethod
    def _attach(b) -> None:
        for q, v in zip(b["params"], b["views"]):
            if q.grad is not v:
                q.grad = v

    def zero(self) -> None:
        for b in self.buckets:
            b["flat"].zero_()
            b["ready"], b["launched"] = 0, False
            b["fired"].clear()
            self._attach(b)
        self._handles = []

    def _launch(self, b) -> None:
        b["launched"] = True
        for q, v in zip(b["params"], b["views"]):                     # (autograd replaced a gradient it found undefined / of another layout)
            if q.grad is not None and q.grad is not v:
                v.copy_(q.grad)
                q.grad = v
        flat = b["flat"]
        if flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat.device)
            self._comm.wait_stream(torch.cuda.current_stream(flat.device))   # the bucket's gradients are complete on the producing stream
            with torch.cuda.stream(self._comm):
                self._handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), b))
        else:
            self._handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), b))

    def _ready(self, b, p=None) -> None:
        b["ready"] += 1
        if p is not None:
            b["fired"].add(id(p))
        if b["ready"] == len(b["params"]) and not b["launched"]:
            self._launch(b)

    def finish(self) -> int:
        """Start the buckets backward did not complete, wait for all of them, average.  Returns the bytes reduced."""
        total = 0
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for work, b in self._handles:
            work.wait()
            if b["flat"].is_cuda:
                torch.cuda.current_stream(b["flat"].device).wait_stream(self._comm)
            b["flat"].div_(self.world)
            total += b["flat"].numel() * 4
        self._handles = []
        # a parameter that received no gradient in this step has `.grad = None` for the optimizer, as without the buckets (its zero-filled
        # view would make AdamW decay it and move its moments: the overlapped and the plain path must produce the same update - ADVICE r5);
        # `zero()` attaches the views again
        for b in self.buckets:
            for q in b["params"]:
                if id(q) not in b["fired"]:
                    q.grad = None
        return total

    def remove(self) -> None:
        for b in self.buckets:
            for h in b["hooks"]:
                h.remove()


class SamTrainer:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, convert_inputs: Callable, n_sub_iteration: int,
                 n_objects_per_batch: Optional[int] = None, mse_loss: Callable = torch.nn.MSELoss(),
                 prompt_generator: PromptGeneratorBase = IterativePromptGenerator(), mask_prob: float = 0.5,
                 mask_loss: Optional[Callable] = None, device=None) -> None:
        self.model, self.optimizer = model, optimizer
        self.loss = self.mask_loss = dice_loss_per_channel if mask_loss is None else mask_loss
        self.convert_inputs = convert_inputs
        self.mse_loss = mse_loss
        self.n_objects_per_batch = n_objects_per_batch
        self.n_sub_iteration = n_sub_iteration
        self.prompt_generator = prompt_generator
        self.mask_prob = mask_prob
        self.is_data_parallel = dist.is_available() and dist.is_initialized()
        self.device = device if device is not None else getattr(getattr(model, "sam", None), "device", "cpu")
        self._iteration = 0
        self.history: List[dict] = []
        # data parallel: gradients live in flat buckets that are all-reduced from autograd hooks while backward runs (GradientBuckets);
        # MSAM_DP_OVERLAP=0 keeps the round-4 form (all_reduce_gradients after backward)
        self._buckets = None

    # ---- reference :70-128
    def _get_prompt_and_multimasking_choices(self, current_iteration):
        if current_iteration % 2 == 0:      # a single point per object
            return 1, 0, False, True
        return 0, 0, True, False            # a single box per object

    def _compute_iou(self, pred, true, eps=1e-7):
        pred_mask = pred > 0.5
        overlap = pred_mask.logical_and(true).sum(dim=(1, 2, 3))
        union = pred_mask.logical_or(true).sum(dim=(1, 2, 3))
        return overlap / (union + eps)

    # ---- reference :131-172
    def _compute_loss(self, batched_outputs, y_one_hot):
        mask_loss, iou_regression_loss = 0.0, 0.0
        batch_size = len(batched_outputs)
        for batch_output, targets in zip(batched_outputs, y_one_hot):
            predicted_objects = torch.sigmoid(batch_output["masks"])
            dice_scores = torch.stack([
                self.loss(predicted_objects[:, i:i + 1].swapaxes(0, 1), targets.swapaxes(0, 1))
                for i in range(predicted_objects.shape[1])])
            dice_scores, _ = torch.min(dice_scores, dim=0)
            with torch.no_grad():
                true_iou = torch.stack([self._compute_iou(predicted_objects[:, i:i + 1], targets)
                                        for i in range(predicted_objects.shape[1])])
            iou_score = self.mse_loss(true_iou.swapaxes(0, 1), batch_output["iou_predictions"])
            mask_loss = mask_loss + torch.mean(dice_scores)
            iou_regression_loss = iou_regression_loss + iou_score
        mask_loss = mask_loss / batch_size
        iou_regression_loss = iou_regression_loss / batch_size
        return mask_loss + iou_regression_loss, mask_loss, iou_regression_loss

    # ---- reference :178-205
    def _get_best_masks(self, batched_outputs, batched_iou_predictions):
        masks = torch.stack([m["masks"] for m in batched_outputs])
        logits = torch.stack([m["low_res_masks"] for m in batched_outputs])
        best = torch.argmax(batched_iou_predictions, dim=2, keepdim=True)
        best = torch.zeros_like(batched_iou_predictions).scatter(2, best, value=1).bool()
        batch_size, n_objects = masks.shape[:2]
        h, w = masks.shape[-2:]
        masks = masks[best].view(batch_size, n_objects, 1, h, w)
        h, w = logits.shape[-2:]
        logits = logits[best].view(batch_size, n_objects, 1, h, w)
        return (masks > 0.0).float(), logits

    # ---- reference :207-241: one decision per top-level iteration, taken on rank 0 and broadcast
    def _use_mask_inputs(self, batched_inputs, y_one_hot):
        use_mask_inputs, use_zero_mask = False, False
        if self.mask_prob == 1:
            use_mask_inputs, use_zero_mask = True, self.is_data_parallel
        elif self.mask_prob > 0:
            if self.is_data_parallel:
                flag = torch.tensor(int(random.random() < self.mask_prob) if dist.get_rank() == 0 else 0, dtype=torch.uint8,
                                    device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.broadcast(flag, src=0)
                use_mask_inputs = bool(flag.item())
                use_zero_mask = use_mask_inputs
            else:
                use_mask_inputs = None
        if use_zero_mask:
            y_zeros = torch.zeros((*y_one_hot.shape[:3], 256, 256))
            for bi, curr in zip(batched_inputs, y_zeros):
                bi["mask_inputs"] = curr
        return batched_inputs, use_mask_inputs

    # ---- reference :243-289
    def _compute_iterative_loss(self, batched_inputs, y_one_hot, num_subiter, multimask_output):
        image_embeddings, batched_inputs = self.model.image_embeddings_oft(batched_inputs)
        loss, mask_loss, iou_regression_loss, mean_model_iou = 0.0, 0.0, 0.0, 0.0
        batched_inputs, use_mask_inputs = self._use_mask_inputs(batched_inputs, y_one_hot)
        for i in range(num_subiter):
            batched_outputs = self.model(batched_inputs=batched_inputs, image_embeddings=image_embeddings,
                                         multimask_output=multimask_output if i == 0 else False)
            net_loss, net_mask_loss, net_iou_loss = self._compute_loss(batched_outputs, y_one_hot)
            batched_iou_predictions = torch.stack([m["iou_predictions"] for m in batched_outputs])
            with torch.no_grad():
                net_mean_model_iou = torch.mean(batched_iou_predictions)
            loss, mask_loss = loss + net_loss, mask_loss + net_mask_loss
            iou_regression_loss, mean_model_iou = iou_regression_loss + net_iou_loss, mean_model_iou + net_mean_model_iou
            if i < num_subiter - 1:
                with torch.no_grad():
                    masks, logits = self._get_best_masks(batched_outputs, batched_iou_predictions)
                    batched_inputs = self._update_prompts(batched_inputs, y_one_hot, masks, logits, use_mask_inputs)
        n = num_subiter
        return loss / n, mask_loss / n, iou_regression_loss / n, mean_model_iou / n

    # ---- reference :291-327
    def _update_prompts(self, batched_inputs, y_one_hot, masks, logits_masks, use_mask_inputs):
        for x1, x2, _inp, logits in zip(masks, y_one_hot, batched_inputs, logits_masks):
            net_coords, net_labels, _, _ = self.prompt_generator(x2, x1)       # on the masks' device, all objects at once
            net_coords = self.model.transform.apply_coords_torch(net_coords, y_one_hot.shape[-2:])
            _inp["point_coords"] = torch.cat([_inp["point_coords"].cpu(), net_coords], dim=1) if "point_coords" in _inp else net_coords
            _inp["point_labels"] = torch.cat([_inp["point_labels"].cpu(), net_labels.float()], dim=1) \
                if "point_labels" in _inp else net_labels.float()
            if self.is_data_parallel:
                use_this_iter = use_mask_inputs
            else:
                use_mask_inputs = (random.random() < self.mask_prob) if self.mask_prob > 0 else False
                use_this_iter = use_mask_inputs
            if use_this_iter:
                _inp["mask_inputs"] = logits
            else:
                _inp.pop("mask_inputs", None)
        return batched_inputs

    # ---- reference :333-357
    def _preprocess_batch(self, batched_inputs, y, sampled_ids):
        assert len(y) == len(sampled_ids)
        n_objects = min(len(ids) for ids in sampled_ids)
        y = y.to(self.device, non_blocking=True)
        y_one_hot = torch.stack([torch.stack([target == int(seg_id) for seg_id in ids[:n_objects]])
                                 for target, ids in zip(y, sampled_ids)]).float()
        batched_inputs = [{k: (v[:n_objects] if k in ("point_coords", "point_labels", "boxes") else v) for k, v in inp.items()}
                          for inp in batched_inputs]
        return batched_inputs, y_one_hot

    def _interactive_train_iteration(self, x, y):
        n_pos, n_neg, get_boxes, multimask_output = self._get_prompt_and_multimasking_choices(self._iteration)
        batched_inputs, sampled_ids = self.convert_inputs(x, y, n_pos, n_neg, get_boxes, self.n_objects_per_batch)
        batched_inputs, y_one_hot = self._preprocess_batch(batched_inputs, y, sampled_ids)
        loss, mask_loss, iou_loss, model_iou = self._compute_iterative_loss(
            batched_inputs=batched_inputs, y_one_hot=y_one_hot, num_subiter=self.n_sub_iteration, multimask_output=multimask_output)
        return loss, mask_loss, iou_loss, model_iou, y_one_hot

    # ---- the train step of reference :384-418 (optimizer.zero_grad - forward - backward - step) + gradient all-reduce
    def train_iteration(self, x, y) -> dict:
        import os
        from ..parallel import collectives_active
        self.model.train()
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        overlap = collectives_active() and os.environ.get("MSAM_DP_OVERLAP", "1") != "0"
        if overlap:
            if self._buckets is None:
                self._buckets = GradientBuckets(params)
            self._buckets.zero()
        else:
            if self._buckets is not None:            # overlap switched off after the buckets were made: their hooks must not reduce a second time
                self._buckets.remove()
                self._buckets = None
            self.optimizer.zero_grad()
        loss, mask_loss, iou_loss, model_iou, _ = self._interactive_train_iteration(x, y)
        loss.backward()
        reduced = self._buckets.finish() if overlap else all_reduce_gradients(params)
        self.optimizer.step()
        rec = {"iteration": self._iteration, "loss": float(loss.detach()), "mask_loss": float(mask_loss.detach()),
               "iou_regression_loss": float(iou_loss.detach()), "model_iou": float(model_iou), "allreduce_bytes": reduced}
        self.history.append(rec)
        self._iteration += 1
        return rec

    def fit(self, iterations: int, train_loader: Iterable) -> List[dict]:
        it = iter(train_loader)
        for _ in range(iterations):
            try:
                x, y = next(it)
            except StopIteration:
                it = iter(train_loader)
                x, y = next(it)
            self.train_iteration(x, y)
        return self.history
