"""Host-side logic of the fine-tuning slice (CPU): losses, prompt generators, input conversion, the trainer's iterative loss
with a stub model, and the bucketed gradient all-reduce + the rank-0 mask-input decision under gloo (world size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _disks(n=4, size=96, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    y = np.zeros((size, size), dtype=np.int64)
    for k in range(n):
        cy, cx = 14 + (k // 2) * 44 + rng.integers(0, 6), 14 + (k % 2) * 44 + rng.integers(0, 6)
        y[(yy - cy) ** 2 + (xx - cx) ** 2 < 100] = k + 1
    return y


def test_dice_loss_matches_definition():
    from micro_sam_amd.training.sam_trainer import dice_loss_per_channel
    g = torch.Generator().manual_seed(0)
    p, t = torch.rand(1, 3, 8, 9, generator=g), (torch.rand(1, 3, 8, 9, generator=g) > 0.5).float()
    got = dice_loss_per_channel(p, t)
    ref = torch.stack([1 - 2 * (p[0, c] * t[0, c]).sum() / ((p[0, c] ** 2).sum() + (t[0, c] ** 2).sum()) for c in range(3)])
    assert torch.allclose(got, ref, atol=1e-6)
    assert torch.allclose(dice_loss_per_channel(t, t), torch.zeros(3), atol=1e-6)


def test_prompt_generators_and_convert_inputs():
    from micro_sam_amd.prompt_generators import IterativePromptGenerator
    from micro_sam_amd.training import ConvertToSamInputs
    np.random.seed(0)
    y = _disks()
    x = torch.zeros(1, 3, 96, 96)
    conv = ConvertToSamInputs(transform=None)
    bi, ids = conv(x, torch.as_tensor(y)[None, None], 1, 0, get_boxes=False, n_samples=3)
    assert len(ids[0]) == 3 and bi[0]["point_coords"].shape == (3, 1, 2) and bi[0]["point_labels"].shape == (3, 1)
    for k, i in enumerate(ids[0]):                                        # the positive point lies in its object, (x, y) order
        px, py = bi[0]["point_coords"][k, 0].long().tolist()
        assert y[py, px] == i
    bi2, ids2 = conv(x, torch.as_tensor(y)[None, None], 0, 0, get_boxes=True, n_samples=None)
    assert bi2[0]["boxes"].shape == (4, 4) and "point_coords" not in bi2[0]
    x0, y0, x1, y1 = bi2[0]["boxes"][0].long().tolist()
    assert (y[y0:y1, x0:x1] == ids2[0][0]).any() and not (y == ids2[0][0])[:, :x0].any()
    # iterative prompts: positive where the prediction misses the object, negative where it over-segments
    true = torch.as_tensor(np.stack([(y == i) for i in (1, 2)])[:, None]).float()
    pred = true.clone()
    pred[0, 0, :, :] = 0                                                  # object 0 completely missed
    pred[1, 0, 0:5, 0:5] = 1                                              # object 1 over-segmented in the corner
    c, l, _, _ = IterativePromptGenerator()(true, pred)
    assert c.shape == (2, 2, 2) and l.tolist() == [[1, 0], [1, 0]]
    assert true[0, 0, c[0, 0, 1], c[0, 0, 0]] == 1                        # positive point inside the missed object
    assert pred[1, 0, c[1, 1, 1], c[1, 1, 0]] == 1 and true[1, 0, c[1, 1, 1], c[1, 1, 0]] == 0


def test_iterative_prompts_regions_and_fallbacks():
    """IterativePromptGenerator (reference prompt_generators.py:252-377) for all objects at once: the point comes from the first
    non-empty region in the reference's order of preference, uniformly."""
    from micro_sam_amd.prompt_generators import IterativePromptGenerator
    torch.manual_seed(0)
    true = torch.zeros(3, 1, 40, 50)
    true[0, 0, 10:20, 10:20] = 1; true[1, 0, 5:9, 30:40] = 1; true[2, 0, 25:35, 5:15] = 1
    pred = true.clone()
    pred[0, 0, 10:20, 10:15] = 0                     # object 0: left half missed            -> positive in the missed half
    pred[1, 0, 20:24, 20:24] = 1                     # object 1: over-segmented far away     -> negative in that patch, positive on the object
    gen = IterativePromptGenerator()                 # object 2: perfect prediction          -> positive on the object, negative in the 3-px ring
    seen = set()
    for _ in range(60):
        c, l, _, _ = gen(true, pred)
        assert c.shape == (3, 2, 2) and l.tolist() == [[1, 0]] * 3
        (px, py), (nx, ny) = c[0].tolist()
        assert 10 <= py < 20 and 10 <= px < 15 and (not (10 <= ny < 20 and 10 <= nx < 20)) and 7 <= ny < 23 and 7 <= nx < 23
        (px, py), (nx, ny) = c[1].tolist()
        assert 5 <= py < 9 and 30 <= px < 40 and 20 <= ny < 24 and 20 <= nx < 24
        (px, py), (nx, ny) = c[2].tolist()
        assert 25 <= py < 35 and 5 <= px < 15 and 22 <= ny < 38 and 2 <= nx < 18 and not (25 <= ny < 35 and 5 <= nx < 15)
        seen.add((px, py))
    assert len(seen) > 30                             # uniform over the 100 object pixels, not a fixed one
    # an object that fills the image and is predicted perfectly: no ring, no background -> the degenerate (0, 0)
    full = torch.ones(1, 1, 8, 8)
    c, _, _, _ = gen(full, full)
    assert c[0, 1].tolist() == [0, 0]


def test_point_prompts_follow_the_reference_distribution():
    """Reference prompt_generators.py:105-190 as called by training/util.py:192-216 (no centre coordinates): positive points are
    random object pixels (not always the centre), negatives keep a SQUARE (Chebyshev) safety border of ``dilation_strength`` around
    the object, and missing points are filled with label-0 background pixels."""
    from micro_sam_amd.prompt_generators import PointAndBoxPromptGenerator
    np.random.seed(1)
    yy, xx = np.mgrid[:64, :64]
    obj = (yy - 32) ** 2 + (xx - 32) ** 2 <= 64                      # disk of radius 8: its grown box has free corners
    oy, ox = np.where(obj)
    seg = torch.as_tensor(obj[None, None]).float()
    gen = PointAndBoxPromptGenerator(1, 4, dilation_strength=3)
    seen_pos, min_cheb = set(), 99
    for _ in range(100):
        c, l, _, _ = gen(seg, [(24, 24, 41, 41)])
        assert l.tolist() == [[1, 0, 0, 0, 0]]
        px, py = c[0, 0].long().tolist()
        assert obj[py, px]
        seen_pos.add((px, py))
        for x, y in c[0, 1:].long().tolist():
            assert 21 <= x < 44 and 21 <= y < 44 and not obj[y, x]
            min_cheb = min(min_cheb, int(np.maximum(np.abs(oy - y), np.abs(ox - x)).min()))
    assert len(seen_pos) > 50                     # random object pixels - not one fixed centre
    assert min_cheb == 4                          # square border: no negative within Chebyshev distance 3 (a diamond would allow 2-3)
    # no negative region left (the dilated object covers the grown box): points are filled from the background with label 0
    big = np.zeros((16, 16), dtype=bool); big[2:14, 2:14] = True
    c, l, _, _ = PointAndBoxPromptGenerator(1, 2, dilation_strength=1)(torch.as_tensor(big[None, None]).float(), [(2, 2, 14, 14)])
    assert l.tolist() == [[1, 0, 0]]
    for x, y in c[0, 1:].long().tolist():
        assert not big[y, x]


class _StubModel(torch.nn.Module):
    """A differentiable stand-in with TrainableSAM's interface: masks = scale * (a Gaussian bump at the first prompt point)."""

    def __init__(self):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.tensor(1.0))
        self.bias = torch.nn.Parameter(torch.tensor(-2.0))
        from micro_sam_amd.transforms import ResizeLongestSide
        self.transform = ResizeLongestSide(96)

    def image_embeddings_oft(self, batched_inputs):
        for b in batched_inputs:
            b["input_size"] = (96, 96)
        return torch.zeros(len(batched_inputs), 1), batched_inputs

    def forward(self, batched_inputs, image_embeddings, multimask_output=False):
        outs = []
        yy, xx = torch.meshgrid(torch.arange(96.0), torch.arange(96.0), indexing="ij")
        for rec in batched_inputs:
            n = rec["point_coords"].shape[0] if "point_coords" in rec else rec["boxes"].shape[0]
            if "point_coords" in rec:
                cx, cy = rec["point_coords"][:, 0, 0], rec["point_coords"][:, 0, 1]
            else:
                cx, cy = rec["boxes"][:, [0, 2]].mean(1), rec["boxes"][:, [1, 3]].mean(1)
            bump = torch.exp(-((yy[None] - cy[:, None, None]) ** 2 + (xx[None] - cx[:, None, None]) ** 2) / 150.0)
            c = 3 if multimask_output else 1
            masks = (self.scale * 6 * bump + self.bias)[:, None].repeat(1, c, 1, 1) * torch.linspace(1.0, 0.8, c)[None, :, None, None]
            low = torch.nn.functional.interpolate(masks, (256, 256), mode="bilinear")
            outs.append({"low_res_masks": low, "masks": masks, "iou_predictions": torch.sigmoid(self.bias).expand(n, c) * 0 + 0.5 + 0 * self.scale})
        return outs


def test_trainer_iterative_loss_decreases_with_stub_model():
    from micro_sam_amd.training import ConvertToSamInputs, SamTrainer
    np.random.seed(1); torch.manual_seed(1)
    import random
    random.seed(1)
    model = _StubModel()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-2)
    tr = SamTrainer(model, opt, ConvertToSamInputs(transform=None), n_sub_iteration=3, n_objects_per_batch=3, mask_prob=0.5, device="cpu")
    y = torch.as_tensor(np.stack([_disks(seed=s) for s in (0, 1)]))[:, None]
    x = torch.zeros(2, 3, 96, 96)
    hist = tr.fit(12, [(x, y)])
    assert len(hist) == 12 and all(np.isfinite(h["loss"]) for h in hist)
    assert np.mean([h["loss"] for h in hist[-3:]]) < np.mean([h["loss"] for h in hist[:3]])
    assert [h["iteration"] for h in hist] == list(range(12))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from micro_sam_amd.training.sam_trainer import SamTrainer, all_reduce_gradients
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(300)), torch.nn.Parameter(torch.zeros(3))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[2].grad = None                                                 # a parameter without gradient is skipped
    nbytes = all_reduce_gradients(params, bucket_bytes=256)               # small buckets: several all-reduces
    import random
    random.seed(100 + rank)                                               # ranks would decide differently on their own
    tr = SamTrainer(torch.nn.Linear(1, 1), None, None, n_sub_iteration=2, mask_prob=0.5, device="cpu")
    decisions = []
    for _ in range(6):
        _, use = tr._use_mask_inputs([{}], torch.zeros(1, 2, 1, 8, 8))
        decisions.append(bool(use))
    # (numpy copies: a tensor in a queue is handed over through the SENDER's file-descriptor server, which is gone if this process ends first)
    q.put((rank, [p.grad.numpy().copy() if p.grad is not None else None for p in params], nbytes, decisions))
    dist.destroy_process_group()


def test_gradient_all_reduce_and_mask_decision_broadcast_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (g, n, d) for r, g, n, d in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        g, nbytes, _ = res[r]
        assert np.allclose(g[0], np.full((5, 7), 1.5)) and np.allclose(g[1], np.full((300,), 3.0)) and g[2] is None
        assert nbytes == (35 + 300) * 4                                   # grads all-reduce bytes = 4 * N_trainable
    assert res[0][2] == res[1][2] and any(res[0][2]) and not all(res[0][2])      # rank 0's decision everywhere


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from micro_sam_amd.training.sam_trainer import GradientBuckets
    torch.manual_seed(0)                                                  # the same model on every rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 40), torch.nn.ReLU(), torch.nn.Linear(40, 3))
    unused = torch.nn.Parameter(torch.ones(5))                           # a parameter that gets no gradient
    buckets = GradientBuckets(list(net.parameters()) + [unused], bucket_bytes=1024)      # several buckets
    out = []
    for step in range(2):                                                 # two steps: zero() re-arms the hooks, the views survive
        buckets.zero()
        g = torch.Generator().manual_seed(100 * step + rank)             # every rank its own batch
        x, t = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
        ((net(x) - t) ** 2).mean().backward()
        nbytes = buckets.finish()
        # (round 6) a parameter that got no gradient has `.grad = None` after finish() - the optimizer skips it as it does without the
        # buckets (its zero-filled view would make AdamW decay it) - and gets its view back from zero()
        assert unused.grad is None
        out.append([p.grad.numpy().copy() for p in net.parameters()] + [np.zeros(5, np.float32)])
        assert all(p.grad.data_ptr() == v.data_ptr() for b in buckets.buckets for p, v in zip(b["params"], b["views"]) if p is not unused)
    q.put((rank, out, nbytes, len(buckets.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_buckets_all_reduce_from_hooks_gloo(world):
    """GradientBuckets (round 5): gradients written by autograd straight into flat buckets, each bucket's all-reduce
    started by the hook of its last gradient - the averaged gradients equal the mean of the per-rank gradients computed here serially."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (o, n, nb) for r, o, n, nb in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 40), torch.nn.ReLU(), torch.nn.Linear(40, 3))
    n_param = sum(p.numel() for p in net.parameters()) + 5
    for step in range(2):
        want = None
        for r in range(world):
            net.zero_grad()
            g = torch.Generator().manual_seed(100 * step + r)
            x, t = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
            ((net(x) - t) ** 2).mean().backward()
            grads = [p.grad.clone() for p in net.parameters()]
            want = grads if want is None else [a + b for a, b in zip(want, grads)]
        want = [w / world for w in want]
        for r in range(world):
            got = res[r][0][step]
            assert all(np.allclose(g, w.numpy(), atol=1e-6) for g, w in zip(got[:-1], want)) and np.all(got[-1] == 0)
    assert res[0][1] == n_param * 4 and res[0][2] >= 2


def test_weight16_cache_follows_the_parameter_version():
    """The bf16 operand copies of a weight (W and W^T, zero-padded to the product kernel's tiles) are formed once per parameter
    version: same objects on the second use, new ones after an in-place update; views of a parameter are cached per view."""
    from micro_sam_amd.training import functional as HF
    p = torch.nn.Parameter(torch.randn(32, 200))
    w16, w16t = HF._weight16(p)
    assert w16.shape == (128, 256) and w16t.shape == (256, 64) and w16.dtype == torch.bfloat16
    assert torch.equal(w16[:32, :200], p.detach().to(torch.bfloat16)) and float(w16[32:].abs().sum()) == 0 and float(w16[:, 200:].abs().sum()) == 0
    assert torch.equal(w16t[:200, :32], p.detach().to(torch.bfloat16).t())
    again = HF._weight16(p)
    assert again[0] is w16 and again[1] is w16t
    with torch.no_grad():
        p.add_(1.0)
    new = HF._weight16(p)
    assert new[0] is not w16 and torch.equal(new[0][:32, :200], p.detach().to(torch.bfloat16))
    conv = torch.nn.Parameter(torch.randn(256, 4, 8, 8))
    v1, _ = HF._weight16(conv.reshape(256, -1))
    v2, _ = HF._weight16(conv.reshape(256, -1))
    assert v1 is v2 and v1.shape == (256, 256)
    plain = torch.randn(16, 64)                       # not a parameter: no caching, no padding
    a, b = HF._weight16(plain)
    assert a.shape == (16, 64) and b.shape == (64, 16)
