"""Integer / byte post-processing parity: bit-exact against the CPU oracle (torch bilinear + reference formulas)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")

SIZES = [((1024, 1024), (1024, 1024)), ((1024, 768), (1024, 768)), ((1024, 1024), (512, 512)), ((683, 1024), (400, 600))]


@pytest.fixture(scope="module")
def low_res():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = torch.Generator().manual_seed(6)
    low = torch.randn(6, 256, 256, generator=g) * 3
    low = F.avg_pool2d(low[None], 5, stride=1, padding=2)[0] * 4
    low[4] = -5.0       # empty mask
    low[5] = 5.0        # full mask (RLE starts with a zero-length run)
    return low


@pytest.mark.parametrize("in_hw,out_hw", SIZES)
def test_postprocess_and_rle_bit_exact(low_res, in_hw, out_hw):
    from micro_sam_amd import ops
    from oracle import amg_ref as A
    from oracle import sam_ref as S
    ref_logits = S.postprocess_masks(low_res[None], in_hw, out_hw)[0]            # Sam.postprocess_masks on the CPU
    res = ops.postprocess_masks(low_res.cuda(), in_hw, out_hw, 0.0, 1.0, want_logits=True)
    assert bool((res["logits"].cpu() == ref_logits).all()), "bilinear resampling must be bit-exact"
    m = ref_logits > 0.0
    counts_ref = torch.stack([(ref_logits > 1.0).sum((1, 2)), (ref_logits > -1.0).sum((1, 2)), m.sum((1, 2))], 1).int()
    assert res["counts"].cpu().tolist() == counts_ref.tolist()                   # stability numerator / denominator, area
    assert res["boxes"].cpu().tolist() == A.batched_mask_to_box(m).tolist()
    assert bool((ops.unpack_bits(res["bits"], out_hw[0]).cpu() == m).all())
    counts, offsets = ops.rle_encode(res["bits"], out_hw[0], out_hw[1])
    assert ops.rles_to_list(counts, offsets, out_hw[0], out_hw[1]) == A.mask_to_rle(m)
    # stability score as the reference computes it
    stab = (res["counts"][:, 0] / res["counts"][:, 1]).cpu()
    ref_stab = A.calculate_stability_score(ref_logits, 0.0, 1.0)
    assert torch.equal(torch.nan_to_num(stab, nan=-1.0), torch.nan_to_num(ref_stab, nan=-1.0))


def test_postprocess_of_fp16_low_res_logits_is_the_fp32_arithmetic_on_the_widened_values():
    """msam_postprocess_masks16 with MSAM_F16 input (round 4: the AMG path keeps its low-res logits in 16 bits between up_fused_kernel
    and the post-processing): every output equals the fp32 kernel's on the same values widened to fp32, both resampling paths."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import ops
    g = torch.Generator().manual_seed(23)
    low16 = (torch.randn(9, 256, 256, generator=g) * 6).to(torch.float16).cuda()
    for in_hw, out_hw in (((1024, 1024), (1024, 1024)), ((683, 1024), (517, 775))):
        a = ops.postprocess_masks(low16, in_hw, out_hw, 0.0, 1.0, want_logits=True)
        b = ops.postprocess_masks(low16.float(), in_hw, out_hw, 0.0, 1.0, want_logits=True)
        for k in ("counts", "boxes", "bits", "logits"):
            assert torch.equal(a[k], b[k]), (k, in_hw, out_hw)


def test_postprocess_decided_words_are_exact():
    """Object-like logits (|v| of 10..40 away from the boundary): most 64-column x 32-row words are decided from the range of their
    ten low-res rows without interpolation (postprocess_kernel, "decided words"); bits, counts and boxes must equal the
    reference's per-pixel arithmetic, including masks whose stability thresholds (+-1) run through flat regions."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import ops
    from oracle import amg_ref as A
    from oracle import sam_ref as S
    g = torch.Generator().manual_seed(16)
    yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
    low = torch.empty(8, 256, 256)
    for i in range(8):
        f = torch.zeros(256, 256)
        for _ in range(1 + 2 * i):
            cy, cx, r = (torch.rand(3, generator=g) * torch.tensor([256.0, 256.0, 30.0]) + torch.tensor([0.0, 0.0, 4.0])).tolist()
            f = torch.maximum(f, torch.clamp(1.5 - torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2) / r, 0.0, 1.0))
        low[i] = f * 45.0 - 14.0 + torch.randn(256, 256, generator=g) * 0.3
    low[6] = low[6].clamp(-0.995, 0.995)          # a plateau just inside the +-1 stability band: never "decided"
    low[7, :128] = 1.005                          # within 0.01 of the upper threshold: the exact path must run
    ref_logits = S.postprocess_masks(low[None], (1024, 1024), (1024, 1024))[0]
    res = ops.postprocess_masks(low.cuda(), (1024, 1024), (1024, 1024), 0.0, 1.0)
    m = ref_logits > 0.0
    counts_ref = torch.stack([(ref_logits > 1.0).sum((1, 2)), (ref_logits > -1.0).sum((1, 2)), m.sum((1, 2))], 1).int()
    assert res["counts"].cpu().tolist() == counts_ref.tolist()
    assert res["boxes"].cpu().tolist() == A.batched_mask_to_box(m).tolist()
    assert bool((ops.unpack_bits(res["bits"], 1024).cpu() == m).all())


def test_vendored_api_on_noisy_masks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import _vendored
    from oracle import amg_ref as A
    g = torch.Generator().manual_seed(7)
    m = torch.rand(3, 300, 200, generator=g) > 0.5
    m[0, 0, 0] = True
    assert _vendored.mask_to_rle_pytorch(m.cuda()) == A.mask_to_rle(m)
    assert _vendored.batched_mask_to_box(m.cuda()).cpu().tolist() == A.batched_mask_to_box(m).tolist()
    # reference known answer: test/test_vendored.py:12-25
    k = torch.zeros(10, 10, dtype=torch.bool); k[7:9, 3:5] = True
    assert _vendored.batched_mask_to_box(k.cuda()).cpu().tolist() == [3, 7, 4, 8]


def test_rle_round_trip_at_full_size():
    """Size-independent property at BASELINE size: decode(encode(mask)) == mask, sum(counts) == H*W."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import amg_utils, ops
    g = torch.Generator().manual_seed(8)
    low = F.avg_pool2d(torch.randn(1, 64, 256, 256, generator=g), 7, 1, 3)[0] * 20
    res = ops.postprocess_masks(low.cuda(), (1024, 1024), (1024, 1024))
    counts, offsets = ops.rle_encode(res["bits"], 1024, 1024)
    rles = ops.rles_to_list(counts, offsets, 1024, 1024, as_list=False)
    masks = ops.unpack_bits(res["bits"], 1024).cpu().numpy()
    for i, r in enumerate(rles):
        assert int(r["counts"].sum()) == 1024 * 1024
        assert (amg_utils.rle_to_mask(r) == masks[i]).all()
        assert amg_utils.area_from_rle(r) == int(res["counts"][i, 2])


@pytest.mark.parametrize("crop_box,size", [([0, 0, 64, 40], (40, 64)), ([37, 5, 137, 70], (131, 200)),
                                           ([320, 320, 720, 600], (600, 720)), ([3, 33, 35, 65], (97, 40))])
def test_uncrop_bits_matches_pad(crop_box, size):
    """msam_uncrop_bits == segment_anything uncrop_masks (zero padding) on the unpacked masks, for row offsets that are
    not multiples of the 32-row word."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import ops
    from micro_sam_amd._vendored import pack_bits
    x0, y0, x1, y1 = crop_box
    h, w = size
    g = torch.Generator().manual_seed(x0 * 7 + y0)
    masks = torch.rand(5, y1 - y0, x1 - x0, generator=g) > 0.5
    bits = pack_bits(masks.cuda())
    out = ops.uncrop_bits(bits, crop_box, h, w)
    assert tuple(out.shape) == (5, (h + 31) // 32, w)
    ref = torch.zeros(5, h, w, dtype=torch.bool)
    ref[:, y0:y1, x0:x1] = masks
    assert torch.equal(ops.unpack_bits(out, h).cpu(), ref)
    assert torch.equal(out.cpu(), pack_bits(ref.cuda()).cpu())


@pytest.mark.parametrize("dtype,shape", [("uint8", (1024, 1024)), ("uint16", (300, 200)), ("float32", (64, 80, 3)),
                                         ("uint8", (128, 96, 2)), ("float32", (50, 60, 5)), ("uint8", (40, 40, 1))])
def test_to_image_device_is_bit_identical(dtype, shape):
    """util._to_image on the device (msam_to_image) == the host formula of the reference (micro_sam/util.py:618-651)."""
    import warnings
    from micro_sam_amd import util
    rng = np.random.default_rng(abs(hash((dtype, shape))) % 1000)
    if dtype == "float32":
        x = (rng.normal(0, 30, size=shape) + 5).astype(np.float32)
    else:
        x = rng.integers(3, 250 if dtype == "uint8" else 60000, size=shape).astype(dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = util._to_image(x)
        t = torch.from_numpy(x).cuda()                                   # uint16 takes the kernel's own uint16 path
        got = util.to_image_device(t).cpu().numpy()
        if dtype == "uint16":                                            # ... and the float32 path agrees (exact conversion)
            assert np.array_equal(util.to_image_device(torch.from_numpy(x.astype(np.float32)).cuda()).cpu().numpy(), got)
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.array_equal(got, ref)
    const = np.full(shape[:2], 7, dtype=np.uint8)                       # max == min: everything maps to 0, no division blow-up
    assert np.array_equal(util.to_image_device(torch.as_tensor(const).cuda()).cpu().numpy(), util._to_image(const))


def test_raw_tile_fast_path_gives_the_same_embedding():
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    p = util.get_sam_model("vit_b", device="cuda", state_dict=synthetic_state_dict("vit_b", 0))
    tiles = [synthetic_tile(s) for s in (11, 12)]
    fast, osz, isz = util._compute_embeddings_batched_raw(p, tiles)
    slow, osz2, isz2 = util._compute_embeddings_batched(p, [util._to_image(t) for t in tiles])
    assert torch.equal(fast, slow) and list(osz) == [tuple(o) for o in osz2] and list(isz) == list(isz2)
    small = [synthetic_tile(s, (512, 512)) for s in (1, 2)]             # needs the PIL resize: host path, same function
    a, _, _ = util._compute_embeddings_batched_raw(p, small)
    b, _, _ = util._compute_embeddings_batched(p, [util._to_image(t) for t in small])
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape,new", [((896, 896, 3), (1024, 1024)), ((512, 512, 3), (1024, 1024)), ((600, 450, 3), (1024, 768)),
                                       ((1400, 1100, 3), (1024, 805)), ((768, 1024, 3), (768, 1024)), ((301, 517, 3), (596, 1024))])
def test_resize_on_the_device_is_pillow(shape, new):
    """ops.resize_bilinear_u8 (msam_resample_u8, the two fixed-point passes of Pillow's BILINEAR resize) == Pillow, bit for bit; and a
    tile that needs the resize gives the same embedding through the raw-tile device path as through the host path (PIL)."""
    _gpu()
    from PIL import Image
    from micro_sam_amd import ops
    rng = np.random.default_rng(shape[0] + shape[1])
    imgs = rng.integers(0, 256, size=(2,) + shape, dtype=np.uint8)
    got = ops.resize_bilinear_u8(torch.as_tensor(imgs).cuda(), new[0], new[1]).cpu().numpy()
    for b in range(2):
        ref = np.array(Image.fromarray(imgs[b]).resize((new[1], new[0]), Image.BILINEAR))
        assert got[b].shape == ref.shape and np.array_equal(got[b], ref)


def test_raw_tile_path_with_resize_matches_the_host_path(vit_b_sd):
    _gpu()
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_tile
    p = util.get_sam_model("vit_b", device="cuda", state_dict=vit_b_sd)
    tile = synthetic_tile(7, (896, 640))                                  # long side 896 -> Pillow resize to 1024 x 731
    f_raw, osz, isz = util._compute_embeddings_batched_raw(p, [tile])
    assert osz == [(896, 640)] and isz == [(1024, 731)]
    f_host, _, isz_h = util._compute_embeddings_batched(p, [util._to_image(tile)])
    assert isz_h == isz and torch.equal(f_raw, f_host)
