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
    # an unknown codec is a loud error
    meta_path = os.path.join(str(tmp_path / "c.zarr"), "odd", ".zarray")
    meta = json.load(open(meta_path))
    meta["compressor"] = {"id": "snappy"}
    json.dump(meta, open(meta_path, "w"))
    with pytest.raises(RuntimeError, match="snappy"):
        f["odd"][:]


# ---------------------------------------------------------------------------------------------------------------
# caches written by zarr-python itself (the reference uses zarr's default compressors, micro_sam/util.py:685-707):
# Blosc frames under zarr 2, the v3 layout with zstd under zarr 3.  zarr / numcodecs are not in this image: the frames
# and stores below are assembled from the format descriptions, with REAL LZ4 / zstd streams from pyarrow's codecs.
# ---------------------------------------------------------------------------------------------------------------

def _compress(piece: bytes, codec: str) -> bytes:
    import zlib
    import pyarrow as pa
    if codec == "zlib":
        return zlib.compress(piece, 5)
    return pa.compress(piece, codec={"lz4": "lz4_raw", "zstd": "zstd"}[codec], asbytes=True)


def _blosc_frame(data: bytes, typesize=4, blocksize=None, shuffle=True, codec="lz4", dont_split=False, memcpy=False) -> bytes:
    """c-blosc 1 frame: header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), block start table, blocks of
    1 or `typesize` splits, each split = int32 compressed size + stream (or the raw bytes when that is not smaller)."""
    import struct
    nbytes = len(data)
    blocksize = blocksize or max(nbytes, 1)
    flags = (1 if shuffle else 0) | (0x10 if dont_split else 0) | ({"lz4": 1, "zlib": 3, "zstd": 4}[codec] << 5)
    if memcpy:
        return bytes([2, 1, flags | 2, typesize]) + struct.pack("<iii", nbytes, blocksize, 16 + nbytes) + data
    nblocks = -(-nbytes // blocksize)
    bodies = []
    for b in range(nblocks):
        chunk = data[b * blocksize:(b + 1) * blocksize]
        leftover = len(chunk) != blocksize
        if shuffle and typesize > 1:
            n = len(chunk) // typesize
            chunk = np.frombuffer(chunk[:n * typesize], np.uint8).reshape(n, typesize).T.tobytes() + chunk[n * typesize:]
        split = (not dont_split) and typesize <= 16 and blocksize // typesize >= 128 and not leftover
        nsplits = typesize if split else 1
        part = len(chunk) // nsplits
        body = b""
        for k in range(nsplits):
            piece = chunk[k * part:(k + 1) * part]
            comp = _compress(piece, codec)
            if len(comp) >= len(piece):
                comp = piece
            body += struct.pack("<i", len(comp)) + comp
        bodies.append(body)
    starts, pos = [], 16 + 4 * nblocks
    for body in bodies:
        starts.append(pos)
        pos += len(body)
    return bytes([2, 1, flags, typesize]) + struct.pack("<iii", nbytes, blocksize, pos) + struct.pack(f"<{nblocks}i", *starts) \
        + b"".join(bodies)


def _embedding_like(shape, seed=0):
    """fp32 values with structure (what shuffling + LZ4 actually compresses: repeated exponent bytes) and some constant runs."""
    g = np.random.default_rng(seed)
    x = g.normal(0, 1, shape).astype("float32")
    x.reshape(-1)[: x.size // 8] = 0.25
    return x


def test_lz4_block_decoder_on_real_lz4_blocks():
    import pyarrow as pa
    from micro_sam_amd import zarr_codecs as ZC
    g = np.random.default_rng(0)
    cases = [b"", b"a", b"abcabcabc" * 200, bytes(g.integers(0, 4, 70000, dtype=np.uint8)), bytes(g.integers(0, 256, 30000, dtype=np.uint8)),
             bytes(300000), (np.arange(60000) // 7).astype(np.int32).tobytes(), _embedding_like((64, 64)).tobytes()]
    for c in cases:
        comp = pa.compress(c, codec="lz4_raw", asbytes=True)
        assert ZC.lz4_block_decompress(comp, len(c)) == c
    with pytest.raises(ValueError):
        ZC.lz4_block_decompress(pa.compress(cases[2], codec="lz4_raw", asbytes=True), len(cases[2]) - 1)
    with pytest.raises(ValueError):
        ZC.lz4_block_decompress(b"\x10a\x05\x00", 20)                       # match offset beyond the output


@pytest.mark.parametrize("codec", ["lz4", "zlib", "zstd"])
def test_blosc_frames(codec):
    from micro_sam_amd import zarr_codecs as ZC
    x = _embedding_like((4, 64, 64)).tobytes()                                # 65536 bytes
    variants = [dict(), dict(blocksize=16384), dict(blocksize=20000),          # one block; 4 blocks; 3 blocks + a leftover block
                dict(shuffle=False), dict(dont_split=True), dict(typesize=1), dict(typesize=8, blocksize=32768),
                dict(blocksize=400), dict(memcpy=True)]                        # blocks too small to split (400 / 4 < 128)
    for kw in variants:
        frame = _blosc_frame(x, codec=codec, **kw)
        assert ZC.blosc_decompress(frame) == x, kw
    assert ZC.blosc_decompress(_blosc_frame(x[:1001], codec=codec, blocksize=512)) == x[:1001]      # tail that is no whole element
    assert len(_blosc_frame(x, codec=codec)) < len(x)                         # the frames above really are compressed
    with pytest.raises(ValueError):
        ZC.blosc_decompress(_blosc_frame(x, codec=codec)[:-3])
    bad = bytearray(_blosc_frame(x, codec=codec)); bad[2] = (bad[2] & 0x1F) | (2 << 5)        # snappy
    with pytest.raises(RuntimeError, match="snappy"):
        ZC.blosc_decompress(bytes(bad))


def test_v2_container_with_zarr_pythons_default_blosc_chunks(tmp_path):
    """A v2 array as zarr-python 2 writes it by default: Blosc(cname lz4, clevel 5, byte shuffle), one frame per chunk."""
    p = str(tmp_path / "ref2.zarr")
    os.makedirs(os.path.join(p, "features"))
    json.dump({"zarr_format": 2}, open(os.path.join(p, ".zgroup"), "w"))
    json.dump({"input_size": [768, 1024], "original_size": [96, 128]}, open(os.path.join(p, ".zattrs"), "w"))
    x = _embedding_like((3, 1, 8, 16, 16), seed=3)
    json.dump({"zarr_format": 2, "shape": [3, 1, 8, 16, 16], "chunks": [1, 1, 8, 16, 16], "dtype": "<f4", "fill_value": 0.0,
               "order": "C", "filters": None,
               "compressor": {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}},
              open(os.path.join(p, "features", ".zarray"), "w"))
    for z in (0, 2):                                                           # slice 1 was never written: fill value
        open(os.path.join(p, "features", f"{z}.0.0.0.0"), "wb").write(_blosc_frame(x[z].tobytes()))
    f = zarr_store.open(p, mode="r")
    a = f["features"]
    assert a.shape == (3, 1, 8, 16, 16) and a.dtype == np.float32 and f.attrs["input_size"] == [768, 1024]
    assert np.array_equal(a[0], x[0]) and np.array_equal(a[2, 0, 3:5], x[2, 0, 3:5]) and not a[1].any()
    # numcodecs' plain Zstd and LZ4 codecs
    import struct
    import pyarrow as pa
    for cid, enc in (("zstd", lambda b: pa.compress(b, codec="zstd", asbytes=True)),
                     ("lz4", lambda b: struct.pack("<I", len(b)) + pa.compress(b, codec="lz4_raw", asbytes=True))):
        d = os.path.join(p, cid)
        os.makedirs(d)
        json.dump({"zarr_format": 2, "shape": [8, 16], "chunks": [8, 16], "dtype": "<f4", "fill_value": 0.0, "order": "C",
                   "filters": None, "compressor": {"id": cid, "level": 1}}, open(os.path.join(d, ".zarray"), "w"))
        open(os.path.join(d, "0.0"), "wb").write(enc(x[0, 0, 0, :8].tobytes()))
        assert np.array_equal(f[cid][:], x[0, 0, 0, :8])


def _write_v3_store(path, x, attrs, codecs, sep="/"):
    import pyarrow as pa
    os.makedirs(os.path.join(path, "features"))
    json.dump({"zarr_format": 3, "node_type": "group", "attributes": attrs}, open(os.path.join(path, "zarr.json"), "w"))
    meta = {"zarr_format": 3, "node_type": "array", "shape": list(x.shape), "data_type": "float32",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [1] + list(x.shape[1:])}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": sep}}, "fill_value": 0.0,
            "codecs": codecs, "attributes": {"note": "per-array attributes"}, "storage_transformers": [], "dimension_names": None}
    json.dump(meta, open(os.path.join(path, "features", "zarr.json"), "w"))
    for z in range(x.shape[0]):
        buf = x[z].tobytes()
        for c in codecs[1:]:
            if c["name"] == "zstd":
                buf = pa.compress(buf, codec="zstd", asbytes=True)
            elif c["name"] == "gzip":
                import gzip
                buf = gzip.compress(buf)
            elif c["name"] == "blosc":
                buf = _blosc_frame(buf, codec="zstd")
            elif c["name"] == "crc32c":
                buf = buf + b"\0\0\0\0"
        idx = [str(z)] + ["0"] * (x.ndim - 1)
        cp = os.path.join(path, "features", *(["c"] + idx)) if sep == "/" else os.path.join(path, "features", sep.join(["c"] + idx))
        os.makedirs(os.path.dirname(cp), exist_ok=True)
        open(cp, "wb").write(buf)


@pytest.mark.parametrize("codecs,sep", [
    ([{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "zstd", "configuration": {"level": 0, "checksum": False}}], "/"),
    ([{"name": "bytes", "configuration": {"endian": "little"}}], "/"),
    ([{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "gzip", "configuration": {"level": 5}}, {"name": "crc32c"}], "."),
    ([{"name": "bytes", "configuration": {"endian": "little"}},
      {"name": "blosc", "configuration": {"cname": "zstd", "clevel": 5, "shuffle": "shuffle", "typesize": 4, "blocksize": 0}}], "/"),
])
def test_zarr_v3_container_read_only(tmp_path, codecs, sep):
    """The layout zarr-python 3 writes by default (zarr.json nodes, attributes inside, chunks under c/, bytes + zstd)."""
    p = str(tmp_path / "ref3.zarr")
    x = _embedding_like((3, 1, 8, 16, 16), seed=5)
    _write_v3_store(p, x, {"input_size": [768, 1024], "original_size": [96, 128], "tile_shape": None}, codecs, sep)
    for mode in ("r", "a"):
        f = zarr_store.open(p, mode=mode)
        assert "features" in f and "nope" not in f and list(f) == ["features"] and len(f) == 1
        assert f.attrs["input_size"] == [768, 1024] and f.attrs.get("tile_shape") is None and "halo" not in f.attrs
        a = f["features"]
        assert a.shape == (3, 1, 8, 16, 16) and a.chunks == (1, 1, 8, 16, 16) and a.ndim == 5 and a.dtype == np.float32
        assert np.array_equal(a[:], x) and np.array_equal(a[1, 0, 2:4, 5], x[1, 0, 2:4, 5]) and a.attrs["note"].startswith("per-array")
        with pytest.raises(PermissionError):
            a[0] = x[0]
        with pytest.raises(PermissionError):
            f.attrs["x"] = 1
        with pytest.raises(PermissionError):
            f.create_dataset("more", shape=(1,), dtype="float32")
        with pytest.raises(KeyError):
            f["nope"]
    with pytest.raises(RuntimeError, match="v3"):
        zarr_store.open(p, mode="w")
    # an unsupported codec is named
    meta = json.load(open(os.path.join(p, "features", "zarr.json")))
    meta["codecs"] = [{"name": "sharding_indexed", "configuration": {}}]
    json.dump(meta, open(os.path.join(p, "features", "zarr.json"), "w"))
    with pytest.raises(RuntimeError, match="sharding_indexed"):
        zarr_store.open(p, mode="r")["features"]


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


def test_precompute_loads_a_cache_written_by_zarr_python_3(tmp_path):
    """precompute_image_embeddings(save_path=...) on a v3 container with the reference's signature attrs: the embedding is
    loaded (no encoder call); the reference's own version / hash strings only warn (reference util.py:1080-1086); other data
    raises; an incomplete v3 cache cannot be completed here."""
    rng = np.random.default_rng(2)
    image = rng.integers(0, 255, (96, 128), dtype=np.uint8)
    feats = _embedding_like((1, 256, 64, 64), seed=9)
    sig = util._get_embedding_signature(image, _Predictor(), None, None)
    sig.update({"micro_sam_version": "1.7.1", "model_hash": "xxh128:0123456789abcdef", "input_size": [768, 1024],
                "original_size": [96, 128]})
    p = str(tmp_path / "by_reference.zarr")
    _write_v3_store(p, feats, sig, [{"name": "bytes", "configuration": {"endian": "little"}},
                                    {"name": "zstd", "configuration": {"level": 0, "checksum": False}}])
    pred = _Predictor()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        emb = util.precompute_image_embeddings(pred, image, save_path=p, verbose=False)
    assert pred.model.image_encoder.calls == 0 and np.array_equal(np.asarray(emb["features"]), feats)
    assert emb["input_size"] == (768, 1024) and emb["original_size"] == (96, 128) and pred.is_image_set
    assert {k for x in w for k in ("micro_sam_version", "model_hash") if k in str(x.message)} == {"micro_sam_version", "model_hash"}
    with pytest.raises(RuntimeError, match="data_signature"):
        util.precompute_image_embeddings(_Predictor(), image[::-1].copy(), save_path=p, verbose=False)
    # a v3 container without a finished embedding: nothing can be written into it
    q = str(tmp_path / "empty3.zarr")
    os.makedirs(q)
    json.dump({"zarr_format": 3, "node_type": "group", "attributes": {}}, open(os.path.join(q, "zarr.json"), "w"))
    with pytest.raises(PermissionError, match="read-only"):
        util.precompute_image_embeddings(_Predictor(), image, save_path=q, verbose=False)
