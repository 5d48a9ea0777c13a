"""The prompt encoder and the mask decoder as stand-alone module calls - the reference calls them directly
(micro_sam/training/trainable_sam.py:96-106): ``sam.prompt_encoder(points, boxes, masks)`` and
``sam.mask_decoder(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output)``.
Compared with the oracle's prompt_encoder / mask_decoder (fp32 for the prompt encoder: exact arithmetic up to sin / cos;
HIP-like rounding mode for the decoder, tolerances of tests/test_gpu_model.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(vit_b_sd):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_tile
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    tile = synthetic_tile(5)
    predictor.set_image(util._to_image(tile))
    return dict(sd=vit_b_sd, predictor=predictor, feats=predictor.get_image_embedding())


def _prompts(P, Np, seed=0):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(P, Np, 2, generator=g) * 1000 + 10
    lbl = (torch.rand(P, Np, generator=g) > 0.3).to(torch.int)
    x0 = torch.rand(P, 2, generator=g) * 600 + 20
    boxes = torch.cat([x0, x0 + torch.rand(P, 2, generator=g) * 350 + 30], dim=1)
    masks = torch.randn(P, 1, 256, 256, generator=g) * 4
    return pts, lbl, boxes, masks


@pytest.mark.parametrize("kind", ["points", "boxes", "points+boxes", "masks", "points+masks"])
def test_prompt_encoder_forward(ctx, kind):
    from oracle import sam_ref as S
    sam = ctx["predictor"].model
    pts, lbl, boxes, masks = _prompts(6, 3, seed=len(kind))
    points = (pts, lbl) if "points" in kind else None
    bx = boxes if "boxes" in kind else None
    mk = masks if "masks" in kind else None
    sparse, dense = sam.prompt_encoder(None if points is None else (points[0].cuda(), points[1].cuda()),
                                       None if bx is None else bx.cuda(), None if mk is None else mk.cuda())
    with torch.no_grad():
        rs, rd = S.prompt_encoder(ctx["sd"], points, bx, mk)
    if points is None and bx is None:
        assert sparse.shape == (6, 0, 256)
    else:
        assert sparse.shape == rs.shape
        assert (sparse.cpu() - rs).abs().max().item() < 2e-4            # sinf / cosf of arguments up to ~ 2 pi * 4
    assert dense.shape == (6, 256, 64, 64)
    assert (dense.cpu() - rd).abs().max().item() < (2e-3 if mk is not None else 1e-6)
    pe = sam.prompt_encoder.get_dense_pe()
    assert (pe.cpu() - S.get_dense_pe(ctx["sd"])).abs().max().item() < 2e-4


@pytest.mark.parametrize("kind,multimask", [("points", True), ("boxes", False), ("points+masks", True)])
def test_mask_decoder_forward_matches_predict_torch_and_oracle(ctx, kind, multimask):
    """The module-call path (prompt_encoder -> mask_decoder) gives the fused predict path's result, and the oracle's."""
    from oracle import sam_ref as S
    sam, p = ctx["predictor"].model, ctx["predictor"]
    pts, lbl, boxes, masks = _prompts(5, 2, seed=7)
    points = (pts, lbl) if "points" in kind else None
    bx = boxes if "boxes" in kind else None
    mk = masks if "masks" in kind else None
    sparse, dense = sam.prompt_encoder(None if points is None else (points[0].cuda(), points[1].cuda()),
                                       None if bx is None else bx.cuda(), None if mk is None else mk.cuda())
    low, iou = sam.mask_decoder(image_embeddings=ctx["feats"], image_pe=sam.prompt_encoder.get_dense_pe(),
                                sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=multimask)
    low2, iou2 = sam.decode(ctx["feats"], None if points is None else pts.cuda(), None if points is None else lbl.cuda(),
                            None if bx is None else bx.cuda(), None if mk is None else mk.cuda(), multimask)
    # same kernels downstream of the token assembly.  With a mask prompt the per-prompt source stream is rounded to 16 bits from
    # emb + dense (module path) vs (emb + no_mask) + (conv - no_mask) (fused path): a few stream values round differently
    tol = 5e-3 if mk is not None else 1e-3
    assert (low - low2).abs().max().item() <= tol * low2.abs().max().item() + 1e-4
    assert (iou - iou2).abs().max().item() <= (1e-3 if mk is not None else 1e-4)
    with torch.no_grad():
        rs, rd = S.prompt_encoder(ctx["sd"], points, bx, mk)
        rl, ri = S.mask_decoder(ctx["sd"], ctx["feats"].cpu(), S.get_dense_pe(ctx["sd"]), rs, rd, multimask, precision="bf16")
    rng = (rl.max() - rl.min()).item()
    d = (low.cpu() - rl).abs()
    assert d.max().item() <= 0.03 * rng and d.mean().item() <= 0.006 * rng
    assert (iou.cpu() - ri).abs().max().item() <= 2e-3
    assert ((low.cpu() > 0) != (rl > 0)).float().mean().item() <= 0.01


def test_mask_decoder_rejects_foreign_pe(ctx):
    sam = ctx["predictor"].model
    sparse = torch.zeros(1, 2, 256, device="cuda")
    dense = sam.prompt_encoder.no_mask_embed.weight.detach().reshape(1, -1, 1, 1).expand(1, -1, 64, 64)
    with pytest.raises(NotImplementedError):
        sam.mask_decoder(ctx["feats"], torch.zeros(1, 256, 64, 64, device="cuda"), sparse, dense, True)


def test_batched_tiled_inference(vit_b_sd):
    """inference.batched_tiled_inference (reference micro_sam/inference.py:358-538): one tile == batched_inference on the image;
    2 x 2 tiles == the per-tile batched_inference records placed by their global_bbox; optimize_memory stitches per-tile NMS."""
    from micro_sam_amd import inference, util
    from micro_sam_amd.synthetic import synthetic_tile
    p = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    image = synthetic_tile(31, (1024, 1024))
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(40, 980, size=12), rng.uniform(40, 980, size=12)], axis=1)[:, None, :]       # [N,1,2] (x, y)
    lbl = np.ones((12, 1))
    one = inference.batched_tiled_inference(p, image, 8, points=pts, point_labels=lbl, tile_shape=(1024, 1024), halo=(0, 0),
                                            verbose_embeddings=False)
    # (tile-local records are merged through bbox / global_bbox windows whose extent is x2 - x1 of the INCLUSIVE box: the last
    # row / column of every mask is dropped, a quirk of the reference's merge that batched_inference's full-mask merge lacks)
    recs1 = inference.batched_inference(p, image, 8, points=pts, point_labels=lbl, verbose_embeddings=False,
                                        return_instance_segmentation=False)
    ref = util.mask_data_to_segmentation([{**r, "global_bbox": r["bbox"]} for r in recs1], shape=(1024, 1024), min_object_size=0)
    assert one.shape == (1024, 1024) and np.array_equal(one, ref)
    # 2 x 2 tiles with a halo: every prompt is decoded on the tile that contains it
    emb = util.precompute_image_embeddings(p, image, tile_shape=(512, 512), halo=(64, 64), verbose=False)
    recs = inference.batched_tiled_inference(p, None, 8, image_embeddings=emb, points=pts, point_labels=lbl,
                                             return_instance_segmentation=False)
    assert len(recs) == 12 and all("global_bbox" in r for r in recs)
    from micro_sam_amd.tiling import Blocking
    tiling = Blocking([0, 0], (1024, 1024), (512, 512))
    by_tile = {}
    for k in range(12):
        by_tile.setdefault(tiling.coordinates_to_block_id([int(round(pts[k, 0, 1])), int(round(pts[k, 0, 0]))]), []).append(k)
    n = 0
    for tile_id in sorted(by_tile):
        outer = tiling.get_block_with_halo(tile_id, [64, 64]).outer_block
        util.set_precomputed(p, emb, tile_id=tile_id)
        local = pts[by_tile[tile_id]] - np.array(outer.begin)[::-1]
        exp = inference.batched_inference(p, None, 8, points=local, point_labels=lbl[by_tile[tile_id]], return_instance_segmentation=False)
        for e in exp:
            r = recs[n]; n += 1
            assert torch.equal(torch.as_tensor(r["segmentation"]), torch.as_tensor(e["segmentation"])) and r["bbox"] == e["bbox"]
            assert r["global_bbox"] == [e["bbox"][0] + outer.begin[1], e["bbox"][1] + outer.begin[0], e["bbox"][2], e["bbox"][3]]
    seg = inference.batched_tiled_inference(p, None, 8, image_embeddings=emb, points=pts, point_labels=lbl)
    assert seg.shape == (1024, 1024) and seg.max() >= 1
    seg2 = inference.batched_tiled_inference(p, None, 8, image_embeddings=emb, points=pts, point_labels=lbl, optimize_memory=True,
                                             min_size=0)
    assert seg2.shape == (1024, 1024) and seg2.max() >= 1


def test_lora_surgery_merged_inference(vit_b_sd):
    """get_sam_model(peft_kwargs=...) (reference micro_sam/util.py:441-450, models/peft_sam.py:50-146): LoRA on q / v (+ mlp) of
    chosen blocks; the HIP encoder runs the merged weights, which equals a plain model carrying W + B A."""
    from micro_sam_amd import util
    from micro_sam_amd.models import peft_sam
    from micro_sam_amd.synthetic import synthetic_tile
    g = torch.Generator().manual_seed(0)
    layers, rank = [0, 2, 7], 4
    sd = dict(vit_b_sd)
    merged = dict(vit_b_sd)
    for i in layers:
        pre = f"image_encoder.blocks.{i}."
        w = sd.pop(pre + "attn.qkv.weight"); b = sd.pop(pre + "attn.qkv.bias")
        sd[pre + "attn.qkv.qkv_proj.weight"], sd[pre + "attn.qkv.qkv_proj.bias"] = w, b
        wm = w.clone()
        for m, sl in (("q", slice(0, 768)), ("v", slice(1536, 2304))):
            a = torch.randn(rank, 768, generator=g) * 0.05
            bb = torch.randn(768, rank, generator=g) * 0.05
            sd[pre + f"attn.qkv.w_a_linear_{m}.weight"], sd[pre + f"attn.qkv.w_b_linear_{m}.weight"] = a, bb
            wm[sl] += bb @ a
        merged[pre + "attn.qkv.weight"] = wm
        for k, (din, dout) in (("1", (768, 3072)), ("2", (3072, 768))):
            a = torch.randn(rank, din, generator=g) * 0.03
            bb = torch.randn(dout, rank, generator=g) * 0.03
            lw = sd.pop(pre + f"mlp.lin{k}.weight"); lb = sd.pop(pre + f"mlp.lin{k}.bias")
            sd[pre + f"mlp.mlp_layer.lin{k}.weight"], sd[pre + f"mlp.mlp_layer.lin{k}.bias"] = lw, lb
            sd[pre + f"mlp.w_a_linear_{k}.weight"], sd[pre + f"mlp.w_b_linear_{k}.weight"] = a, bb
            merged[pre + f"mlp.lin{k}.weight"] = lw + bb @ a
    p_lora = util.get_sam_model("vit_b", device="cuda", state_dict=sd,
                                peft_kwargs=dict(rank=rank, peft_module=peft_sam.LoRASurgery, attention_layers_to_update=layers,
                                                 update_matrices=["q", "v", "mlp"]))
    p_merged = util.get_sam_model("vit_b", device="cuda", state_dict=merged)
    p_plain = util.get_sam_model("vit_b", device="cuda", state_dict=dict(vit_b_sd))
    img = util._to_image(synthetic_tile(9))
    for p in (p_lora, p_merged, p_plain):
        p.set_image(img)
    assert torch.equal(p_lora.features, p_merged.features)
    assert (p_lora.features - p_plain.features).abs().max().item() > 1e-2          # the update is visible
    assert any("w_a_linear_q" in k for k in p_lora.model.state_dict())
    with pytest.raises(NotImplementedError):
        util.get_sam_model("vit_b", device="cuda", state_dict=dict(vit_b_sd), peft_kwargs=dict(rank=2, peft_module=torch.nn.Identity))


def test_precompute_state_driver(vit_b_sd, tmp_path):
    """precompute_state.precompute_state (reference precompute_state.py:227-278; VERDICT r4 missing #5): a folder of image files with a
    glob pattern -> one <name>.zarr per file with the embeddings and, with precompute_amg_state, amg_state.pickle next to them - what
    util.precompute_image_embeddings / cache_amg_state give when called by hand; and a single in-memory volume, state per slice."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    from micro_sam_amd import precompute_state as PS
    from micro_sam_amd import util
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from micro_sam_amd.synthetic import synthetic_tile
    src = tmp_path / "images"
    src.mkdir()
    tiles = [synthetic_tile(40 + k, (512, 512)) for k in range(2)]
    for k, t in enumerate(tiles):
        Image.fromarray(t).save(src / f"img_{k}.png")
    (src / "notes.txt").write_text("not an image")
    out = tmp_path / "state"
    PS.precompute_state(str(src), str(out), pattern="*.png", model_type="vit_b", precompute_amg_state=True, state_dict=vit_b_sd, device="cuda")
    assert sorted(p.name for p in out.iterdir()) == ["img_0.zarr", "img_1.zarr"]
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    for k, t in enumerate(tiles):
        emb = util.precompute_image_embeddings(predictor, t, str(out / f"img_{k}.zarr"), verbose=False)      # loads (signature matches)
        direct = util.precompute_image_embeddings(predictor, t, verbose=False)
        assert np.array_equal(np.asarray(emb["features"][:]), np.asarray(direct["features"]))
        assert (out / f"img_{k}.zarr" / "amg_state.pickle").exists()
        amg = PS.cache_amg_state(predictor, t, emb, str(out / f"img_{k}.zarr"), verbose=False)                # loads the pickle
        ref = AutomaticMaskGenerator(predictor)
        ref.initialize(t, direct)
        assert np.array_equal(amg.generate(), ref.generate())
    vol = np.stack(tiles)
    PS.precompute_state(vol, str(tmp_path / "vol"), model_type="vit_b", ndim=3, precompute_amg_state=True, state_dict=vit_b_sd, device="cuda")
    assert sorted(p.name for p in (tmp_path / "vol.zarr" / "amg_state").iterdir()) == ["state-0.pkl", "state-1.pkl"]
