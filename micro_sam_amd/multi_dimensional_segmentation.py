"""Per-slice automatic segmentation of a volume / time series (reference
``micro_sam/multi_dimensional_segmentation.py:385-416`` ``_segment_slices``).

``segment_slices`` is the reference's serial loop: embeddings of all slices once (batched), then per slice
``initialize(i=z)`` + ``generate`` with a running id offset.  ``segment_slices_sharded`` runs the same loop on a
contiguous block of slices per rank (one process per GPU) and assembles the volume with the two collectives of
``parallel.gather_label_tiles``; its result equals the serial loop's on every rank.

``merge_instance_segmentation_3d`` / ``automatic_3d_segmentation`` (reference :312-382, :419-481): the overlap
of objects between consecutive slices is counted on the device (``ops.slice_overlaps``: one scatter-add pass over the label volume
into a hash table in HBM), the graph problem on those few thousand edges is solved on the host.  The reference delegates both to
un-vendored libraries (``elf.tracking.tracking_utilities.compute_edges_from_overlap`` -> ``nifty.ground_truth.overlap``;
``elf.segmentation.multicut``), absent here: the edge list is restated from their published behaviour (score = overlap / size of the
source object; edges to label 0 are not emitted), the costs follow the reference's own lines :365-373, and the multicut is solved by
greedy additive edge contraction (the warm start of elf's default Kernighan-Lin solver) - PARITY UNPINNED for the solver: on overlap
graphs of stacked slices (chains of near-1 and near-0 scores) both agree, in general a different local optimum is possible.

``segment_mask_in_volume`` (reference :105-233) is the interactive 3-d path: an object annotated in a few slices is
carried through the volume slice by slice, each step one ``prompt_based_segmentation.segment_from_mask`` call (prompts
derived from the neighbouring slice's mask, one decoder pass on the slice's precomputed embedding).
"""
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, modeling, parallel, util


_PIPELINE_KWARGS = ("pred_iou_thresh", "stability_score_thresh", "box_nms_thresh", "with_background")


def _can_pipeline(data, predictor, segmentor, embedding_path, tile_shape, kwargs) -> bool:
    """The pipelined loop covers the plain case: an untiled ``AutomaticMaskGenerator`` (one crop layer) on a GPU, embeddings computed
    here and kept in memory, ``generate`` called with threshold arguments only (label-image output)."""
    from .instance_segmentation import AutomaticMaskGenerator
    return (type(segmentor) is AutomaticMaskGenerator and segmentor._crop_n_layers == 0 and embedding_path is None and tile_shape is None
            and data.shape[0] > 1 and all(k in _PIPELINE_KWARGS for k in kwargs)
            and str(predictor.device).startswith("cuda") and torch.cuda.is_available()
            and hasattr(predictor.model.image_encoder, "forward_u8") and hasattr(predictor.model, "lane_view")
            and segmentor._predictor is predictor)


@torch.no_grad()
def _segment_slices_pipelined(data, predictor, segmentor, batch_size: int, n_lanes: int, return_device: bool, kwargs, offsets: bool = True):
    """The slice loop of ``_segment_slices`` as a device pipeline (what bench.py's step does, inside the product):

      main stream   raw slices up (page-locked ring, asynchronous), ``_to_image`` + encoder, one batch at a time, an event per batch
      decode lanes  ``n_lanes`` lane clones of the generator (own decoder scratch, own stream): slice z on lane z % n_lanes waits for ITS
                    encoder batch only, runs initialize + generate_device; kernels of different slices overlap, the encoder works on the
                    next batch underneath
      post stream   per batch: running id offsets of the serial loop (offset of slice z = sum of the max ids before it), label tiles
                    into a page-locked double buffer
      host          drains batch b - 1 into the result while batch b is queued: no per-slice synchronisation

    Same labels as the serial loop (the same initialize / generate_device per slice; the offsets are the serial loop's)."""
    Z, H, W = data.shape[0], data.shape[1], data.shape[2]
    dev = predictor.device
    main = torch.cuda.current_stream(dev)
    lanes = segmentor._decode_lanes(max(1, min(n_lanes, Z)))          # cached on the generator: clones keep their decoder workspace
    post = getattr(segmentor, "_post_stream", None)
    if post is None or post.device != torch.device(dev):
        post = segmentor._post_stream = torch.cuda.Stream(device=dev)
    batch_size = max(1, int(batch_size))
    feats = torch.empty((Z, 1, modeling.PROMPT_DIM, modeling.GRID, modeling.GRID), dtype=torch.float32, device=dev)
    labels = torch.empty((Z, H, W), dtype=torch.int32, device=dev)
    emb = {"features": feats, "input_size": None, "original_size": None}
    carry = None                                     # running id offset: lives on the post stream (allocated, updated and read there only -
    #                                                  a tensor of the main stream's pool dropped here would be recycled under the post stream's reads)
    out = None if return_device else np.empty((Z, H, W), dtype=np.uint32)
    pins = [None, None]
    pending = None                                   # (s0, s1, slot, event) of the batch whose labels are on their way to the host
    flags = []

    def drain(p):
        s0, s1, slot, ev = p
        ev.synchronize()
        np.copyto(out[s0:s1], pins[slot][: s1 - s0].numpy().view(np.uint32))

    # batch schedule: a short first encoder batch - the decode lanes start after 4 slices' worth of encoder time instead of a whole
    # batch's -, full batches after it.  The encoder's output does not depend on how slices are batched (every kernel forms a row's
    # products in the same order whatever the row count: tests/test_gpu_model.py::test_encoder_bits_do_not_depend_on_the_batch)
    bounds, s0 = [], 0
    while s0 < Z:
        s1 = min(s0 + (min(4, batch_size) if s0 == 0 and Z > batch_size else batch_size), Z)
        bounds.append((s0, s1))
        s0 = s1
    for b, (s0, s1) in enumerate(bounds):
        f, osz, isz = util._compute_embeddings_batched_raw(predictor, [np.asarray(data[z]) for z in range(s0, s1)])
        feats[s0:s1, 0] = f
        emb["input_size"], emb["original_size"] = isz[-1], osz[-1]
        enc_done = torch.cuda.Event()
        enc_done.record(main)
        used = []
        for z in range(s0, s1):
            amg, st = lanes[z % len(lanes)]
            st.wait_event(enc_done)
            with torch.cuda.stream(st):
                amg.initialize(data[z], image_embeddings=emb, i=z)
                lab, flag = amg.generate_device(**kwargs)
                labels[z] = lab
            flags.append(flag)
            if st not in used:
                used.append(st)
        for st in used:
            post.wait_stream(st)
        with torch.cuda.stream(post):
            blk = labels[s0:s1]
            if offsets:
                if carry is None:
                    carry = torch.zeros((), dtype=torch.int64, device=dev)
                mx = blk.flatten(1).amax(dim=1).to(torch.int64)
                offs = (torch.cumsum(mx, 0) - mx + carry).view(-1, 1, 1).to(blk.dtype)
                blk += torch.where(blk != 0, offs, torch.zeros_like(offs))
                carry = carry + mx.sum()
            if out is not None:
                slot = b & 1
                if pins[slot] is None:
                    pins[slot] = util.pinned_buffer(f"slices{slot}", (batch_size, H, W), torch.int32)
                pins[slot][: s1 - s0].copy_(blk, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(post)
        if out is not None:
            if b == len(bounds) - 1:
                out[s0:s1].fill(0)           # first touch of the LAST batch's result pages (fresh allocation: ~10 ms of page faults per 64 MiB)
                #                              now, while the device works: its drain is the one nothing overlaps (earlier batches fault their
                #                              pages inside their own, overlapped, drains - touching them twice only costs host time)
            if pending is not None:
                drain(pending)               # batch b - 1: complete by now or soon - batch b is already queued behind it
            pending = (s0, s1, slot, ev)
    if pending is not None:
        drain(pending)
    for _, st in lanes:
        main.wait_stream(st)
    main.wait_stream(post)
    bad = torch.stack(flags).flatten().nonzero().flatten().tolist()      # connected components that did not converge in two passes
    if bad:
        return None
    # the objects are left as the serial loop leaves them: generator initialised on the last slice, predictor holding its embedding
    last = lanes[(Z - 1) % len(lanes)][0]
    segmentor.set_state(last.get_state())
    util.set_precomputed(predictor, emb, i=Z - 1)
    return (labels if return_device else out), emb


def _can_overlap_tiled(data, predictor, segmentor, embedding_path, tile_shape, halo) -> bool:
    from .instance_segmentation import TiledAutomaticMaskGenerator
    return (type(segmentor) is TiledAutomaticMaskGenerator and embedding_path is None and tile_shape is not None and halo is not None
            and data.shape[0] > 1 and str(predictor.device).startswith("cuda") and torch.cuda.is_available()
            and hasattr(predictor.model.image_encoder, "forward_u8") and segmentor._predictor is predictor
            and isinstance(data, np.ndarray) and util._device_to_image_ok([data[0]]))


@torch.no_grad()
def _segment_slices_tiled_overlapped(data, predictor, segmentor, tile_shape, halo, batch_size: int, kwargs):
    """The tiled slice loop (BASELINE configs[2]: 2048^2 slices, tile 768 + halo 128) with the image encoder of the NEXT group of slices
    running on its own HIP stream underneath the decode lanes of the current one (the reference - and rounds 1 - 4 here -
    encode every tile of every slice first, then decode slice by slice: encoder and decoder kernels never shared the GPU).

      encoder stream   slices in groups of max(1, batch_size // tiles per slice): raw tiles up, ``_to_image`` (+ resize) + encoder, tiles
                       batched by shape; results go straight into the volume's tiled feature store ([Z, 1, 256, 64, 64] per tile), an
                       event per group
      caller's stream  waits for the group's event, then per slice ``TiledAutomaticMaskGenerator.initialize(i=z)`` (tiles on the
                       generator's decode lanes) + ``generate`` - the same calls as the loop, so the same labels

    The next group is enqueued BEFORE the current group's slices are decoded; the host synchronisations inside ``generate`` only wait for
    the decode streams.  Returns (segmentation, image_embeddings) as the loop does (the embeddings: the same in-memory tiled store as
    ``precompute_image_embeddings(ndim=3, tile_shape=...)``)."""
    from .tiling import Blocking, TileArray, TiledFeatures
    Z, shape = data.shape[0], tuple(data.shape[1:3])
    dev = predictor.device
    tiling = Blocking([0, 0], shape, tile_shape)
    n_tiles = tiling.number_of_blocks
    features = TiledFeatures(shape, tile_shape, halo)
    enc = getattr(predictor, "_encoder_stream", None)
    if enc is None or enc.device != torch.device(dev):
        enc = predictor._encoder_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    step = max(1, int(batch_size) // n_tiles)
    groups = [(z0, min(z0 + step, Z)) for z0 in range(0, Z, step)]
    outer = []
    for tile_id in range(n_tiles):
        tile = tiling.get_block_with_halo(tile_id, list(halo))
        outer.append(tuple(slice(beg, end) for beg, end in zip(tile.outer_block.begin, tile.outer_block.end)))

    # the volume's feature store, allocated on the CALLER's stream (it outlives this call in the returned embeddings; memory of the encoder
    # stream's pool handed to the caller could be recycled under the caller's later reads)
    stores = [torch.zeros((Z, 1, modeling.PROMPT_DIM, modeling.GRID, modeling.GRID), dtype=torch.float32, device=dev) for _ in range(n_tiles)]

    # page-locked staging per (group parity, tile shape): ALL uploads of a group are queued before its first encoder launch, and a buffer
    # is reused two groups later, when its copy has long completed - the shared two-slot ring of util._upload_raw_tiles would make the
    # host wait for a copy that sits on the encoder stream behind the encoder kernels of the same group
    pins = getattr(predictor, "_slice_pins", None)
    if pins is None:
        pins = predictor._slice_pins = {}
    uploaded = {}

    def encode(g, z0, z1):
        if g - 2 in uploaded:
            uploaded.pop(g - 2).synchronize()
        enc.wait_stream(main)                         # (first use: the zero fill above; later: a no-op in practice)
        with torch.cuda.stream(enc):
            by_shape = {}
            for z in range(z0, z1):
                for tile_id in range(n_tiles):
                    image = np.asarray(data[(z,) + outer[tile_id]])
                    by_shape.setdefault(image.shape, []).append((z, tile_id, image))
            batches = []
            for shp, members in by_shape.items():
                host = np.stack([im for _, _, im in members])
                if host.dtype != np.uint8:
                    host = host.astype(np.float32)
                key = (g & 1, shp, host.dtype.str)
                pin = pins.get(key)
                if pin is None or pin.numel() < host.size:
                    pin = pins[key] = torch.empty(host.size, dtype=torch.from_numpy(host[:0]).dtype).pin_memory()
                view = pin[: host.size].view(host.shape)
                view.copy_(torch.from_numpy(host))
                batches.append((members, view.to(dev, non_blocking=True)))
            up = torch.cuda.Event()
            up.record(enc)
            uploaded[g] = up
            predictor.reset_image()
            for members, dev_raw in batches:
                emb, osz, isz = util._embeddings_from_uploaded_raw(predictor, dev_raw)
                for k, (z, tile_id, _) in enumerate(members):
                    if tile_id not in features:
                        features[tile_id] = TileArray(stores[tile_id], osz[k], isz[k])
                    stores[tile_id][z, 0] = emb[k]
            ev = torch.cuda.Event()
            ev.record(enc)
        return ev
    emb = {"features": features, "input_size": None, "original_size": None}
    segmentation = np.zeros(data.shape, dtype="uint32")
    offset = 0
    pending = encode(0, *groups[0])
    for g, (z0, z1) in enumerate(groups):
        ready = pending
        if g + 1 < len(groups):
            pending = encode(g + 1, *groups[g + 1])   # queued now: runs on the encoder stream while the slices below are decoded
        main.wait_event(ready)
        for z in range(z0, z1):
            segmentor.initialize(data[z], image_embeddings=emb, verbose=False, i=z)
            seg = segmentor.generate(**kwargs)
            max_z = int(seg.max())
            if max_z == 0:
                continue
            seg[seg != 0] += offset
            offset = max_z + offset
            segmentation[z] = seg
    main.wait_stream(enc)
    return segmentation, emb


def segment_slices(data: np.ndarray, predictor, segmentor, embedding_path=None, verbose: bool = False,
                   tile_shape: Optional[Tuple[int, int]] = None, halo: Optional[Tuple[int, int]] = None,
                   batch_size: int = 1, decode_lanes: int = 3, **kwargs):
    """Reference ``_segment_slices`` (:385-416).  Returns (uint32 [Z,Y,X] segmentation, image_embeddings).

    The plain case (untiled ``AutomaticMaskGenerator`` on a GPU, embeddings computed here) runs as a device pipeline
    (``_segment_slices_pipelined``: encoder batches, ``decode_lanes`` concurrent decode lanes, double-buffered label download; the
    embeddings stay on the device as with ``keep_on_device``); everything else - tiled generators, decoder-based generators, embeddings
    from / to a container, other ``generate`` arguments - takes the reference's loop below.  ``decode_lanes=0`` forces that loop."""
    assert data.ndim == 3
    if decode_lanes > 0 and _can_pipeline(data, predictor, segmentor, embedding_path, tile_shape, kwargs):
        res = _segment_slices_pipelined(data, predictor, segmentor, batch_size, decode_lanes, False, kwargs)
        if res is not None:
            return res
    if decode_lanes > 0 and _can_overlap_tiled(data, predictor, segmentor, embedding_path, tile_shape, halo):
        return _segment_slices_tiled_overlapped(data, predictor, segmentor, tile_shape, halo, batch_size, kwargs)
    image_embeddings = util.precompute_image_embeddings(predictor=predictor, input_=data, save_path=embedding_path, ndim=3,
                                                        tile_shape=tile_shape, halo=halo, verbose=verbose,
                                                        batch_size=batch_size, keep_on_device=tile_shape is None)
    offset = 0
    segmentation = np.zeros(data.shape, dtype="uint32")
    for i in range(segmentation.shape[0]):
        segmentor.initialize(data[i], image_embeddings=image_embeddings, verbose=False, i=i)
        seg = segmentor.generate(**kwargs)
        max_z = int(seg.max())
        if max_z == 0:
            continue
        seg[seg != 0] += offset
        offset = max_z + offset
        segmentation[i] = seg
    return segmentation, image_embeddings


def segment_slices_sharded(data: np.ndarray, predictor, segmentor, verbose: bool = False, batch_size: int = 1,
                           **kwargs) -> np.ndarray:
    """``segment_slices`` with the slices block-partitioned over the ranks of the default process group (weights
    replicated, no collective until the assembly).  Every rank returns the full uint32 [Z,Y,X] volume."""
    import torch.distributed as dist
    assert data.ndim == 3
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    start, stop = parallel.shard_range(data.shape[0], rank, world)
    dev = predictor.device if world > 1 and dist.get_backend() == "nccl" else "cpu"
    local = None
    if stop - start > 1 and _can_pipeline(data[start:stop], predictor, segmentor, None, None, kwargs):
        # the rank's block through the device pipeline, labels left in HBM for the gather (ids 1..K per item: the gather derives the
        # serial loop's offsets from the per-item max ids of ALL ranks)
        res = _segment_slices_pipelined(data[start:stop], predictor, segmentor, batch_size, 3, True, kwargs, offsets=False)
        if res is not None:
            local = res[0].to(dev)
    if local is None:
        host = np.zeros((stop - start,) + data.shape[1:], dtype="int32")
        if stop > start:
            emb = util.precompute_image_embeddings(predictor=predictor, input_=data[start:stop], ndim=3, verbose=verbose,
                                                   batch_size=batch_size, keep_on_device=True)
            for k in range(stop - start):
                segmentor.initialize(data[start + k], image_embeddings=emb, verbose=False, i=k)
                host[k] = segmentor.generate(**kwargs).astype("int32")
        local = torch.as_tensor(host, device=dev)
    out = parallel.gather_label_tiles(local, data.shape[0])
    return out.cpu().numpy().astype("uint32")


# ------------------------------------------------------------------------------------------ merge across z

def compute_edges_from_overlap(segmentation: np.ndarray, device=None):
    """``elf.tracking.tracking_utilities.compute_edges_from_overlap`` (called at reference :357): for every object of slice z and
    every object of slice z + 1 that it overlaps, an edge {"source", "target", "score"} with score = overlapping pixels / pixels of
    the source object (``overlapArraysNormalized``).  Counting runs on the device; returns (uv_ids int64 [E,2], scores float64 [E])
    sorted by (source, target)."""
    from . import ops
    dev = _lib.require_gpu(device)
    _check_ids_unique_per_slice(segmentation)
    vol = torch.as_tensor(np.ascontiguousarray(segmentation).astype(np.int32, copy=False)).to(dev)
    table = ops.slice_overlaps(vol)
    return edges_from_overlap_table(table)


def _check_ids_unique_per_slice(segmentation: np.ndarray) -> None:
    """The overlap table sums a source object's pixels per ID: an id that occurs in two slices (a volume that did not come out of
    ``segment_slices``, whose running offsets make every id slice-local) would have the sizes of two objects added up - silently wrong
    scores.  One pass over the per-slice id ranges; a full check only where two slices' ranges overlap."""
    seg = np.asarray(segmentation)
    if seg.ndim != 3 or seg.shape[0] < 2:
        return
    seen_max = 0
    for z in range(seg.shape[0]):
        ids = seg[z][seg[z] != 0]
        if ids.size == 0:
            continue
        lo, hi = int(ids.min()), int(ids.max())
        if lo <= seen_max:                       # ranges overlap: look at the ids themselves
            prev = np.unique(seg[:z][(seg[:z] >= lo) & (seg[:z] != 0)])
            both = np.intersect1d(prev, np.unique(ids))
            if both.size:
                raise ValueError(f"compute_edges_from_overlap: object ids must be unique across slices (id {int(both[0])} occurs in slice {z} "
                                 "and in an earlier one); relabel the slices with running offsets first, as segment_slices does")
        seen_max = max(seen_max, hi)


def edges_from_overlap_table(table: np.ndarray):
    """(source, target, pixels) rows incl. target 0 -> (uv_ids, scores): the source object's size is the sum of its row (its pixels
    over background included), edges to the background are dropped."""
    if len(table) == 0:
        return np.zeros((0, 2), dtype=np.int64), np.zeros((0,), dtype=np.float64)
    src, inv = np.unique(table[:, 0], return_inverse=True)
    size = np.bincount(inv, weights=table[:, 2].astype(np.float64))
    score = table[:, 2] / size[inv]
    fg = table[:, 1] != 0
    return table[fg, :2].astype(np.int64), score[fg]


def compute_edge_costs(probs: np.ndarray, beta: float = 0.5) -> np.ndarray:
    """``elf.segmentation.multicut.compute_edge_costs`` without weighting: log((1 - p) / p) + log((1 - beta) / beta), p clipped to
    [0.001, 0.999]."""
    p_min = 0.001
    p = (1.0 - 2 * p_min) * np.asarray(probs, dtype=np.float64) + p_min
    return np.log((1.0 - p) / p) + np.log((1.0 - beta) / beta)


def multicut_gaec(n_nodes: int, uv_ids: np.ndarray, costs: np.ndarray) -> np.ndarray:
    """Multicut by greedy additive edge contraction: repeatedly contract the edge with the largest positive cost, summing the costs
    of parallel edges, until no positive edge is left (positive = attractive, as in elf / nifty).  Returns consecutive node labels
    int64 [n_nodes] (label of node 0 first)."""
    import heapq
    adj = [dict() for _ in range(n_nodes)]
    for (u, v), c in zip(uv_ids.tolist(), costs.tolist()):
        if u == v:
            continue
        adj[u][v] = adj[u].get(v, 0.0) + c
        adj[v][u] = adj[v].get(u, 0.0) + c
    parent = list(range(n_nodes))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    heap = [(-c, min(u, v), max(u, v)) for u in range(n_nodes) for v, c in adj[u].items() if u < v and c > 0]
    heapq.heapify(heap)
    alive = [True] * n_nodes
    while heap:
        negc, u, v = heapq.heappop(heap)
        if not (alive[u] and alive[v]) or adj[u].get(v) != -negc:
            continue                                        # stale entry
        if len(adj[u]) < len(adj[v]):
            u, v = v, u                                     # merge the smaller neighbourhood (v) into u
        alive[v] = False
        parent[v] = u
        del adj[u][v]
        for w, c in adj[v].items():
            if w == u:
                continue
            del adj[w][v]
            nc = adj[u].get(w, 0.0) + c
            adj[u][w] = nc
            adj[w][u] = nc
            if nc > 0:
                heapq.heappush(heap, (-nc, min(u, w), max(u, w)))
        adj[v] = {}
    roots = np.array([find(i) for i in range(n_nodes)], dtype=np.int64)
    _, first = np.unique(roots, return_index=True)
    order = np.argsort(first)                               # labels in order of the smallest node id of each cluster
    lut = np.empty(len(order), dtype=np.int64)
    lut[order] = np.arange(len(order))
    return lut[np.searchsorted(np.sort(np.unique(roots)), roots)]


def multicut_energy(uv_ids: np.ndarray, costs: np.ndarray, labels: np.ndarray) -> float:
    """The multicut objective (nifty / elf convention): the sum of the costs of the CUT edges (positive = attractive: cutting it costs)."""
    uv = np.asarray(uv_ids, dtype=np.int64)
    cut = labels[uv[:, 0]] != labels[uv[:, 1]]
    return float(np.asarray(costs, dtype=np.float64)[cut].sum())


def multicut_refine(n_nodes: int, uv_ids: np.ndarray, costs: np.ndarray, labels: np.ndarray, max_sweeps: int = 50) -> np.ndarray:
    """Local search of the Kernighan-Lin kind on top of a multicut (the reference solves with elf's ``multicut_decomposition``, whose inner
    solver is Kernighan-Lin warm-started by greedy additive edge contraction - python-elf / nifty are absent, so the solver is restated
    as: GAEC, then this refinement): sweeps of (a) single-node moves - a node leaves its cluster for an adjacent cluster or for a new
    one of its own when that lowers the objective - and (b) joins of two adjacent clusters whose connecting costs sum to a positive
    value, until a sweep changes nothing.  Deterministic (nodes and clusters in ascending id order, strict improvements only), so the
    objective never increases over GAEC's; on forests GAEC is already optimal and nothing moves.  Returns consecutive labels in order of
    the smallest node id of each cluster."""
    uv = np.asarray(uv_ids, dtype=np.int64)
    w = np.asarray(costs, dtype=np.float64)
    lab = np.asarray(labels, dtype=np.int64).copy()
    adj = [dict() for _ in range(n_nodes)]
    for (u, v), c in zip(uv.tolist(), w.tolist()):
        if u != v:
            adj[u][v] = adj[u].get(v, 0.0) + c
            adj[v][u] = adj[v].get(u, 0.0) + c
    eps = 1e-12
    next_label = int(lab.max()) + 1 if n_nodes else 0
    for _ in range(max_sweeps):
        changed = False
        for u in range(n_nodes):                                         # (a) node moves
            if not adj[u]:
                continue
            to = {}
            for v, c in adj[u].items():
                to[int(lab[v])] = to.get(int(lab[v]), 0.0) + c
            stay = to.get(int(lab[u]), 0.0)                              # what cutting u out of its cluster would cost
            best, gain = None, eps
            for c_lab in sorted(to):
                if c_lab != lab[u] and to[c_lab] - stay > gain:
                    best, gain = c_lab, to[c_lab] - stay
            if -stay > gain:                                            # a cluster of its own
                best, gain = next_label, -stay
            if best is not None:
                if best == next_label:
                    next_label += 1
                lab[u] = best
                changed = True
        between = {}                                                     # (b) cluster joins
        for (u, v), c in zip(uv.tolist(), w.tolist()):
            a, b = int(lab[u]), int(lab[v])
            if a != b:
                key = (a, b) if a < b else (b, a)
                between[key] = between.get(key, 0.0) + c
        for (a, b) in sorted(between):
            if between[(a, b)] > eps:
                # (one join per sweep and pair; the sums of the other pairs are recomputed in the next sweep)
                lab[lab == b] = a
                changed = True
                break
        if not changed:
            break
    # a node move can take an articulation node out of a cluster: the rest keeps one label without being connected any more.  A multicut
    # is a partition into CONNECTED components, so clusters are re-cut by the connected components of the edges they keep.
    parent = list(range(n_nodes))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for u, v in uv.tolist():
        if u != v and lab[u] == lab[v]:
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
    lab = np.array([find(u) for u in range(n_nodes)], dtype=np.int64) if n_nodes else lab
    _, first = np.unique(lab, return_index=True)
    order = np.argsort(first)
    lut = np.empty(len(order), dtype=np.int64)
    lut[order] = np.arange(len(order))
    return lut[np.searchsorted(np.sort(np.unique(lab)), lab)]


def _relabel_sequential(seg: np.ndarray, offset: int = 1) -> np.ndarray:
    """``skimage.segmentation.relabel_sequential(seg, offset)[0]``: non-zero labels -> offset, offset + 1, ... in ascending order."""
    ids = np.unique(seg)
    ids = ids[ids != 0]
    lut = np.zeros(int(seg.max()) + 1, dtype=seg.dtype)
    lut[ids] = np.arange(offset, offset + len(ids), dtype=seg.dtype)
    return lut[seg]


def _preprocess_closing(slice_segmentation: np.ndarray, gap_closing: int) -> np.ndarray:
    """Reference :236-297: binary closing along z only; a closed object replaces the original ones unless it would merge more than
    one of them; slices within ``gap_closing`` of either end are only renumbered."""
    from scipy import ndimage
    structure = np.zeros((3, 1, 1))
    structure[:, 0, 0] = 1
    closed = ndimage.binary_closing(slice_segmentation > 0, iterations=gap_closing, structure=structure)
    out = np.zeros_like(slice_segmentation)
    n_slices = out.shape[0]
    offset = 1
    for z in range(n_slices):
        seg_z = slice_segmentation[z]
        if z < gap_closing or z >= n_slices - gap_closing:
            new = _relabel_sequential(seg_z, offset)
            offset = int(new.max()) + 1
            out[z] = new
            continue
        closed_z, n_closed = ndimage.label(closed[z], structure=np.ones((3, 3)))      # skimage.measure.label: full connectivity
        pairs = np.unique(np.stack([closed_z.reshape(-1), seg_z.reshape(-1).astype(np.int64)], axis=1), axis=0)
        pairs = pairs[(pairs[:, 0] != 0) & (pairs[:, 1] != 0)]
        ids_initial, ids_closed = [], []
        for cid in range(1, n_closed + 1):
            matched = pairs[pairs[:, 0] == cid, 1]
            if len(matched) > 1:
                ids_initial.extend(matched.tolist())
            else:
                ids_closed.append(cid)
        new = np.zeros_like(seg_z)
        cm = np.isin(closed_z, ids_closed)
        new[cm] = closed_z[cm]
        if ids_initial:
            im = np.isin(seg_z, ids_initial)
            new[im] = _relabel_sequential(seg_z[im], offset=int(new.max()) + 1)
        new = _relabel_sequential(new, offset)
        if new.max() > 0:
            offset = int(new.max()) + 1
        out[z] = new
    return out


def _filter_z_extent(segmentation: np.ndarray, min_z_extent: int) -> np.ndarray:
    """Reference :300-309: drop objects that span fewer than ``min_z_extent`` slices."""
    from scipy import ndimage
    drop = [i + 1 for i, sl in enumerate(ndimage.find_objects(segmentation)) if sl is not None and sl[0].stop - sl[0].start < min_z_extent]
    if drop:
        segmentation[np.isin(segmentation, drop)] = 0
    return segmentation


def merge_instance_segmentation_3d(slice_segmentation: np.ndarray, beta: float = 0.5, with_background: bool = True,
                                   gap_closing: Optional[int] = None, min_z_extent: Optional[int] = None, verbose: bool = False,
                                   device=None) -> np.ndarray:
    """Reference :312-382: objects of consecutive slices are nodes joined by their normalised overlap; edge costs
    ``compute_edge_costs(overlap)``, the multicut is solved on ``1 - costs`` (reference :365-373: a large overlap is attractive) and
    every slice object takes its cluster's label.  ``with_background``: edges that touch label 0 are maximally repulsive (none is
    emitted here: the background is not a node).  ``beta`` is accepted as in the reference, which hands it to the solver call only."""
    if gap_closing is not None and gap_closing > 0:
        slice_segmentation = _preprocess_closing(slice_segmentation, gap_closing)
    uv_ids, overlaps = compute_edges_from_overlap(slice_segmentation, device=device)
    if len(uv_ids) == 0:
        return slice_segmentation
    n_nodes = int(slice_segmentation.max()) + 1
    costs = compute_edge_costs(overlaps)
    if with_background:
        costs[(uv_ids == 0).any(axis=1)] = -8.0
    node_labels = multicut_gaec(n_nodes, uv_ids, 1.0 - costs)
    node_labels = multicut_refine(n_nodes, uv_ids, 1.0 - costs, node_labels)          # Kernighan-Lin style local search on GAEC's result
    if node_labels[0] != 0:                                 # keep the background at label 0
        node_labels = np.where(node_labels == node_labels[0], 0, np.where(node_labels < node_labels[0], node_labels + 1, node_labels))
    segmentation = node_labels[slice_segmentation].astype(slice_segmentation.dtype)
    if min_z_extent is not None and min_z_extent > 0:
        segmentation = _filter_z_extent(segmentation, min_z_extent)
    return segmentation


def automatic_3d_segmentation(volume: np.ndarray, predictor, segmentor, embedding_path=None, with_background: bool = True,
                              gap_closing: Optional[int] = None, min_z_extent: Optional[int] = None,
                              tile_shape: Optional[Tuple[int, int]] = None, halo: Optional[Tuple[int, int]] = None,
                              verbose: bool = False, return_embeddings: bool = False, batch_size: int = 1, **kwargs):
    """Reference :419-481: per-slice segmentation (``segment_slices``) merged across z (``merge_instance_segmentation_3d``)."""
    segmentation, image_embeddings = segment_slices(volume, predictor, segmentor, embedding_path=embedding_path, verbose=verbose,
                                                    tile_shape=tile_shape, halo=halo, batch_size=batch_size, **kwargs)
    segmentation = merge_instance_segmentation_3d(segmentation, beta=0.5, with_background=with_background, gap_closing=gap_closing,
                                                  min_z_extent=min_z_extent, verbose=verbose, device=predictor.device)
    return (segmentation, image_embeddings) if return_embeddings else segmentation


# ------------------------------------------------------------------------------------------ interactive 3-d projection

PROJECTION_MODES = ("box", "mask", "points", "points_and_mask", "single_point")
_PROJECTIONS = {                      # (use_box, use_mask, use_points, use_single_point)
    "mask": (True, True, False, False), "points": (False, False, True, False), "box": (True, False, False, False),
    "points_and_mask": (False, True, True, False), "single_point": (False, False, True, True),
}


def _validate_projection(projection):
    """Reference :48-72."""
    if isinstance(projection, str):
        if projection not in _PROJECTIONS:
            raise ValueError("Choose projection method from 'mask' / 'points' / 'box' / 'points_and_mask' / 'single_point'. "
                             f"You have passed the invalid option {projection}.")
        return _PROJECTIONS[projection]
    if isinstance(projection, dict):
        assert len(projection.keys()) == 3, "There should be three parameters assigned for the projection method."
        return projection["use_box"], projection["use_mask"], projection["use_points"], False
    raise ValueError(f"{projection} is not a supported projection method.")


def segment_mask_in_volume(segmentation: np.ndarray, predictor, image_embeddings, segmented_slices: np.ndarray,
                           stop_lower: bool, stop_upper: bool, iou_threshold: float, projection,
                           update_progress: Optional[callable] = None, box_extension: float = 0.0,
                           verbose: bool = False) -> Tuple[np.ndarray, Tuple[int, int]]:
    """Reference ``segment_mask_in_volume`` (:105-233).  ``segmentation`` [Z,H,W] holds the object (value 1) in the slices
    ``segmented_slices``; it is extended in place: downwards from the lowest and upwards from the highest annotated slice
    until the IoU between consecutive slices drops below ``iou_threshold`` (unless ``stop_lower`` / ``stop_upper``), and
    between annotated slices from both ends towards the middle (a single middle slice is prompted with the union of its
    two neighbours).  Returns the volume and the (lowest, highest) slice that now holds the object."""
    from .prompt_based_segmentation import segment_from_mask
    use_box, use_mask, use_points, use_single_point = _validate_projection(projection)
    if update_progress is None:
        def update_progress(*args):
            pass
    prompt_kw = dict(image_embeddings=image_embeddings, use_mask=use_mask, use_box=use_box, use_points=use_points,
                     box_extension=box_extension)

    def walk(z_start, z_stop, step, done, threshold=None):
        """Carry the mask from ``z_start`` towards ``z_stop``; returns the last slice written."""
        z = z_start + step
        while True:
            if verbose:
                print(f"Segment {z_start} to {z_stop}: segmenting slice {z}")
            previous = segmentation[z - step]
            seg_z, _, _ = segment_from_mask(predictor, previous, i=z, return_all=True, use_single_point=use_single_point,
                                            **prompt_kw)
            if threshold is not None:
                iou = util.compute_iou(previous, seg_z)
                if iou < threshold:
                    if verbose:
                        print(f"Segmentation stopped at slice {z} due to IOU {iou} < {threshold}.")
                    break
            segmentation[z] = seg_z
            z += step
            if done(z, z_stop):
                if verbose:
                    print(f"Segment {z_start} to {z_stop}: stop at slice {z}")
                break
            update_progress(1)
        return z - step

    def fill_between(z, below, above):
        """One slice prompted with the union of the object in two other slices."""
        union = np.logical_or(segmentation[below] == 1, segmentation[above] == 1)
        segmentation[z] = segment_from_mask(predictor, union, i=z, **prompt_kw)
        update_progress(1)

    z0, z1 = int(segmented_slices.min()), int(segmented_slices.max())
    last = segmentation.shape[0] - 1
    z_min = walk(z0, 0, -1, np.less, iou_threshold) if (z0 > 0 and not stop_lower) else z0
    z_max = walk(z1, last, 1, np.greater, iou_threshold) if (z1 < last and not stop_upper) else z1
    if z0 != z1:
        for z_start, z_stop in zip(segmented_slices[:-1], segmented_slices[1:]):
            gap = z_stop - z_start
            z_mid = int((z_start + z_stop) // 2)
            if gap == 1:
                continue
            if z_start == z0 and stop_lower:
                walk(z_stop, z_start, -1, np.less_equal)
            elif z_stop == z1 and stop_upper:
                walk(z_start, z_stop, 1, np.greater_equal)
            elif gap == 2:
                fill_between(z_start + 1, z_start, z_stop)
            else:
                walk(z_start, z_mid, 1, np.greater_equal if gap % 2 == 0 else np.greater)
                walk(z_stop, z_mid, -1, np.less_equal)
                if gap % 2 == 0:          # the middle slice is equally far from both ends: union of its neighbours
                    fill_between(z_mid, z_mid - 1, z_mid + 1)
    return segmentation, (z_min, z_max)
