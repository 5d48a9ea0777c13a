"""Embedding cache on disk (reference micro_sam/util.py:684-747 writers, :1038-1094 signature, :1184-1196 open modes;
shape expectations of the reference's test/test_util.py:123-246): zarr v2 layout written by ``micro_sam_amd.zarr_store``
and the cache logic of ``util.precompute_image_embeddings(save_path=...)`` - host logic only, the encoder is replaced by
a deterministic stand-in so nothing here needs a GPU (the GPU round trip is tests/test_gpu_model.py::test_zarr_cache_gpu).
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from micro_sam_amd import util, zarr_store
from micro_sam_amd.transforms import ResizeLongestSide


class _Encoder:
    def __init__(self):
        self.calls = 0

    def forward_u8(self, batch):
        self.calls += batch.shape[0]
        m = batch.float().mean(dim=(1, 2, 3))                         # depends on the pixels
        base = torch.arange(256 * 64 * 64, dtype=torch.float32).reshape(1, 256, 64, 64) / (256 * 64 * 64)
        return base + m[:, None, None, None] + 1.0                    # never all-zero


class _Model:
    def __init__(self):
        self.image_encoder = _Encoder()


class _Predictor:
    """Duck-typed ``SamPredictor`` on the CPU."""

    def __init__(self, model_type="vit_b"):
        self.model = _Model()
        self.device = torch.device("cpu")
        self.transform = ResizeLongestSide(1024)
        self.model_type, self.model_name, self._hash = model_type, model_type, "xxh128:test"
        self.reset_image()

    def reset_image(self):
        self.features = self.original_size = self.input_size = None
        self.is_image_set = False

    def set_image(self, image):
        resized = self.transform.apply_image(image)
        self.features = self.model.image_encoder.forward_u8(torch.as_tensor(np.ascontiguousarray(resized))[None])
        self.original_size, self.input_size = image.shape[:2], tuple(resized.shape[:2])
        self.is_image_set = True

    def get_image_embedding(self):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        return self.features


def test_store_layout_is_zarr_v2(tmp_path):
    p = str(tmp_path / "e.zarr")
    f = zarr_store.open(p, mode="a")
    a = f.create_dataset("features", shape=(3, 1, 4, 5, 5), chunks=(1, 1, 4, 5, 5), dtype="float32")
    x = np.random.default_rng(0).random((1, 4, 5, 5), dtype=np.float32)
    a[1] = x
    assert json.load(open(os.path.join(p, ".zgroup"))) == {"zarr_format": 2}
    meta = json.load(open(os.path.join(p, "features", ".zarray")))
    assert meta == {"zarr_format": 2, "shape": [3, 1, 4, 5, 5], "chunks": [1, 1, 4, 5, 5], "dtype": "<f4", "compressor": None,
                    "fill_value": 0, "order": "C", "filters": None, "dimension_separator": "."}
    # chunk file = raw little-endian C-order bytes of the chunk, named by the dotted chunk index (zarr spec v2)
    assert sorted(n for n in os.listdir(os.path.join(p, "features")) if not n.startswith(".")) == ["1.0.0.0.0"]
    assert open(os.path.join(p, "features", "1.0.0.0.0"), "rb").read() == x.tobytes()
    assert np.array_equal(a[1], x) and np.count_nonzero(a[0]) == 0 and a[:].shape == (3, 1, 4, 5, 5) and a.ndim == 5
    f.attrs["input_size"] = (10, 12)
    f.attrs["tile_shape"] = None
    assert json.load(open(os.path.join(p, ".zattrs"))) == {"input_size": [10, 12], "tile_shape": None}
    r = zarr_store.open(p, mode="r")
    assert np.array_equal(r["features"][1], x)
    with pytest.raises(PermissionError):
        r["features"][0] = x
    with pytest.raises(KeyError):
        r["nope"]


def test_store_partial_chunks_and_codecs(tmp_path):
    import zlib
    f = zarr_store.open(str(tmp_path / "c.zarr"))
    b = f.create_dataset("odd", shape=(5, 7), chunks=(2, 3), dtype="int32", compressor="zlib")
    v = np.arange(35, dtype="int32").reshape(5, 7)
    b[...] = v
    assert np.array_equal(b[:], v) and np.array_equal(b[1:4, 2:6], v[1:4, 2:6]) and b[4, 6] == 34 and b[-1, 0] == 28
    b[1:3, 1:5] = 7
    v[1:3, 1:5] = 7
    assert np.array_equal(b[:], v)
    raw = open(os.path.join(str(tmp_path / "c.zarr"), "odd", "2.2"), "rb").read()          # edge chunks are full-size
    assert np.array_equal(np.frombuffer(zlib.decompress(raw), dtype="<i4").reshape(2, 3)[:1, :1], v[4:, 6:])
    # a container written with zarr-python's default codec cannot be decoded with the stdlib: loud error
    meta_path = os.path.join(str(tmp_path / "c.zarr"), "odd", ".zarray")
    meta = json.load(open(meta_path))
    meta["compressor"] = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}
    json.dump(meta, open(meta_path, "w"))
    with pytest.raises(RuntimeError, match="blosc"):
        f["odd"][:]


def test_cache_2d_roundtrip_and_signature(tmp_path):
    rng = np.random.default_rng(1)
    image = rng.integers(0, 255, (96, 128), dtype=np.uint8)
    pred = _Predictor()
    path = str(tmp_path / "emb.zarr")
    e1 = util.precompute_image_embeddings(pred, image, save_path=path, verbose=False)
    assert e1["features"].shape == (1, 256, 64, 64) and e1["original_size"] == (96, 128) and e1["input_size"] == (768, 1024)
    attrs = json.load(open(os.path.join(path, ".zattrs")))
    assert set(attrs) == {"data_signature", "tile_shape", "halo", "model_type", "model_name", "micro_sam_version",
                          "model_hash", "input_size", "original_size"}
    import hashlib
    assert attrs["data_signature"] == hashlib.sha1(image.tobytes()).hexdigest() and attrs["input_size"] == [768, 1024]
    n = pred.model.image_encoder.calls
    pred2 = _Predictor()
    e2 = util.precompute_image_embeddings(pred2, image, save_path=path, verbose=False)       # loaded, not recomputed
    assert pred2.model.image_encoder.calls == 0 and pred.model.image_encoder.calls == n
    assert np.array_equal(np.asarray(e2["features"]), np.asarray(e1["features"]))
    assert pred2.is_image_set and pred2.original_size == (96, 128) and tuple(pred2.features.shape) == (1, 256, 64, 64)
    # other data / other tiling / other model type: RuntimeError (reference util.py:1088-1091)
    with pytest.raises(RuntimeError, match="data_signature"):
        util.precompute_image_embeddings(_Predictor(), image[::-1].copy(), save_path=path, verbose=False)
    with pytest.raises(RuntimeError, match="tile_shape"):
        util.precompute_image_embeddings(_Predictor(), image, save_path=path, tile_shape=(64, 64), halo=(8, 8), verbose=False)
    with pytest.raises(RuntimeError, match="model_type"):
        util.precompute_image_embeddings(_Predictor("vit_l"), image, save_path=path, verbose=False)
    # model hash / version mismatch only warns (reference util.py:1080-1086)
    p3 = _Predictor()
    p3._hash = "xxh128:other"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        util.precompute_image_embeddings(p3, image, save_path=path, verbose=False)
    assert any("model_hash" in str(x.message) for x in w)


def test_cache_3d_partial_resume_and_lazy_loading(tmp_path):
    rng = np.random.default_rng(2)
    vol = rng.integers(0, 255, (5, 64, 64), dtype=np.uint8)
    path = str(tmp_path / "vol.zarr")
    pred = _Predictor()
    ref = util.precompute_image_embeddings(pred, vol, verbose=False, batch_size=2)             # in memory
    assert ref["features"].shape == (5, 1, 256, 64, 64)
    # simulate an interrupted run: slices 0 and 1 on disk, no signature yet
    f = zarr_store.open(path)
    ds = f.create_dataset("features", shape=(5, 1, 256, 64, 64), chunks=(1, 1, 256, 64, 64), dtype="float32")
    ds[0] = ref["features"][0]
    ds[1] = ref["features"][1]
    pred2 = _Predictor()
    e = util.precompute_image_embeddings(pred2, vol, save_path=path, verbose=False, batch_size=2)
    assert pred2.model.image_encoder.calls == 3                                               # only the missing slices
    assert np.array_equal(e["features"], ref["features"]) and e["input_size"] == (1024, 1024) and e["original_size"] == (64, 64)
    pred3 = _Predictor()
    lazy = util.precompute_image_embeddings(pred3, vol, save_path=path, verbose=False, lazy_loading=True)
    assert isinstance(lazy["features"], zarr_store.Array) and pred3.model.image_encoder.calls == 0
    assert lazy["features"].chunks == (1, 1, 256, 64, 64)
    util.set_precomputed(pred3, lazy, i=3)
    assert np.array_equal(pred3.features.numpy(), ref["features"][3])
    with pytest.raises(ValueError):
        util.set_precomputed(pred3, lazy)                                                     # 3-d data needs an index
    # wrong shape of a partial container
    bad = str(tmp_path / "bad.zarr")
    zarr_store.open(bad).create_dataset("features", shape=(4, 1, 256, 64, 64), chunks=(1, 1, 256, 64, 64), dtype="float32")
    with pytest.raises(RuntimeError, match="Invalid partial features"):
        util.precompute_image_embeddings(_Predictor(), vol, save_path=bad, verbose=False)


def test_cache_tiled_2d_and_3d(tmp_path):
    rng = np.random.default_rng(3)
    image = rng.integers(0, 255, (200, 260), dtype=np.uint8)
    path = str(tmp_path / "tiled.zarr")
    mem = util.precompute_image_embeddings(_Predictor(), image, tile_shape=(128, 128), halo=(16, 16), verbose=False, batch_size=3)
    e = util.precompute_image_embeddings(_Predictor(), image, save_path=path, tile_shape=(128, 128), halo=(16, 16),
                                         verbose=False, batch_size=3)
    assert e["input_size"] is None and e["original_size"] is None and len(e["features"]) == 6
    pred = _Predictor()
    c = util.precompute_image_embeddings(pred, image, save_path=path, tile_shape=(128, 128), halo=(16, 16), verbose=False)
    assert pred.model.image_encoder.calls == 0 and isinstance(c["features"], zarr_store.Group)
    feats = c["features"]
    assert feats.attrs["shape"] == [200, 260] and feats.attrs["tile_shape"] == [128, 128] and feats.attrs["halo"] == [16, 16]
    assert sorted(feats.keys(), key=int) == [str(t) for t in range(6)]
    for t in range(6):
        assert np.array_equal(feats[str(t)][:], mem["features"][t][:].numpy())
        assert tuple(feats[str(t)].attrs["original_size"]) == mem["features"][t].attrs["original_size"]
        assert tuple(feats[str(t)].attrs["input_size"]) == mem["features"][t].attrs["input_size"]
    util.set_precomputed(pred, c, tile_id=4)
    assert pred.original_size == mem["features"][4].attrs["original_size"]
    # masked + 3-d
    vol = rng.integers(0, 255, (3, 200, 260), dtype=np.uint8)
    mask = np.zeros(vol.shape, dtype=bool)
    mask[0, :50, :50] = True
    mask[2, 150:, 200:] = True
    path3 = str(tmp_path / "tiled3.zarr")
    m3 = util.precompute_image_embeddings(_Predictor(), vol, tile_shape=(128, 128), halo=(16, 16), verbose=False, mask=mask)
    util.precompute_image_embeddings(_Predictor(), vol, save_path=path3, tile_shape=(128, 128), halo=(16, 16), verbose=False,
                                     mask=mask, batch_size=2)
    c3 = util.precompute_image_embeddings(_Predictor(), vol, save_path=path3, tile_shape=(128, 128), halo=(16, 16), verbose=False)
    tim = c3["features"].attrs["tiles_in_mask"]
    assert tim == m3["features"].attrs["tiles_in_mask"] == {"0": [0], "1": [], "2": [4, 5]}  # outer blocks (halo) decide
    assert c3["features"]["5"].shape == (3, 1, 256, 64, 64)
    assert np.array_equal(c3["features"]["5"][2], m3["features"][5][2].numpy())
    assert np.count_nonzero(c3["features"]["5"][0]) == 0                                      # slice without mask: fill value
