"""Minimal zarr **v2** directory store for the embedding cache (``save_path`` of ``precompute_image_embeddings``).

The reference keeps embeddings in a zarr container (``micro_sam/util.py:684-747`` dataset creation / ``_write_batch``,
``:1038-1094`` signature attrs, ``:1184-1196`` open modes).  ``zarr`` itself is not available in this image, so this
module writes / reads the on-disk format directly (zarr storage specification v2: a directory per group with
``.zgroup`` / ``.zattrs``, a directory per array with ``.zarray`` / ``.zattrs`` and one file per chunk named by the
``.``-joined chunk index).  Containers written here open unchanged with ``zarr.open`` (zarr-python 2 and 3 both read
the v2 layout), which is the contract between a precompute box and an annotation laptop.

Only the subset of the zarr API that micro_sam touches is provided: ``open(path, mode)``, ``Group.require_group /
create_dataset / attrs / __contains__ / __getitem__``, ``Array.shape / chunks / dtype / ndim / attrs`` and basic
indexing (ints and unit-step slices).  Chunks are written uncompressed by default (fp32 embeddings do not compress and
the writer has to keep up with > 100 tiles/s); ``compressor="zlib"`` is available.

Reading also covers what the reference itself leaves on disk (it creates its datasets with zarr's defaults,
``micro_sam/util.py:685-707``): v2 containers whose chunks are Blosc frames (zarr-python 2's default: LZ4 + byte shuffle) or
``zstd`` / ``lz4`` / stdlib-codec streams (``zarr_codecs``), and - read-only - the zarr **v3** layout zarr-python 3 writes
by default (``zarr.json`` per node with the attributes inside, chunk files under ``c/``, codec pipeline ``bytes`` +
``zstd`` / ``gzip`` / ``blosc``; sharding and transposing codecs raise).  Both decoders are restated from the format
specifications; neither zarr nor numcodecs is in this image to produce reference files (parity unpinned, see
``zarr_codecs``).
"""
from __future__ import annotations

import io
import json
import os
import threading
from typing import Any, Dict, Iterator, Optional, Sequence, Tuple

import numpy as np


def _jsonable(v: Any) -> Any:
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, np.ndarray):
        return _jsonable(v.tolist())
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, range):
        return list(v)
    return v


def _atomic_write(path: str, data: bytes) -> None:
    tmp = f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"
    with io.open(tmp, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


class Attributes:
    """``.zattrs``: a JSON object, re-read on access and rewritten on every update (like zarr's ``Attributes``).
    Sequences come back as lists (JSON), exactly as with zarr - callers that need tuples convert."""

    def __init__(self, path: str, read_only: bool = False) -> None:
        self._path = path
        self._read_only = read_only
        self._lock = threading.Lock()

    def _load(self) -> Dict[str, Any]:
        try:
            with io.open(self._path, "r") as fh:
                return json.load(fh)
        except FileNotFoundError:
            return {}

    def asdict(self) -> Dict[str, Any]:
        return self._load()

    def __getitem__(self, key: str) -> Any:
        return self._load()[key]

    def get(self, key: str, default: Any = None) -> Any:
        return self._load().get(key, default)

    def __contains__(self, key: object) -> bool:
        return key in self._load()

    def __iter__(self) -> Iterator[str]:
        return iter(self._load())

    def __len__(self) -> int:
        return len(self._load())

    def keys(self):
        return self._load().keys()

    def items(self):
        return self._load().items()

    def update(self, other: Dict[str, Any]) -> None:
        if self._read_only:
            raise PermissionError("zarr container opened read-only")
        with self._lock:
            d = self._load()
            d.update({str(k): _jsonable(v) for k, v in other.items()})
            _atomic_write(self._path, json.dumps(d, indent=4).encode())

    def __setitem__(self, key: str, value: Any) -> None:
        self.update({key: value})


def _decode(raw: bytes, compressor: Optional[Dict[str, Any]], nbytes: Optional[int] = None) -> bytes:
    """Chunk file -> chunk bytes for a numcodecs compressor config (v2 ``.zarray`` "compressor")."""
    if compressor is None:
        return raw
    cid = compressor.get("id")
    if cid == "zlib":
        import zlib
        return zlib.decompress(raw)
    if cid == "gzip":
        import gzip
        return gzip.decompress(raw)
    if cid == "bz2":
        import bz2
        return bz2.decompress(raw)
    if cid == "lzma":
        import lzma
        return lzma.decompress(raw)
    if cid == "blosc":
        from .zarr_codecs import blosc_decompress
        return blosc_decompress(raw)
    if cid == "zstd":
        from .zarr_codecs import zstd_decompress
        if nbytes is None:
            raise RuntimeError("micro_sam_amd.zarr_store: zstd chunks need the chunk size")
        return zstd_decompress(raw, nbytes)
    if cid == "lz4":                                    # numcodecs.LZ4: little-endian uint32 size + one LZ4 block
        import struct
        from .zarr_codecs import lz4_block_decompress
        return lz4_block_decompress(raw[4:], struct.unpack_from("<I", raw, 0)[0])
    raise RuntimeError(f"micro_sam_amd.zarr_store: chunks compressed with '{cid}' cannot be read here (supported: blosc with "
                       "lz4 / zlib / zstd, zstd, lz4, zlib, gzip, bz2, lzma and uncompressed chunks); recompute the "
                       "embeddings or re-save them with one of these codecs")


def _encode(raw: bytes, compressor: Optional[Dict[str, Any]]) -> bytes:
    if compressor is None:
        return raw
    if compressor.get("id") == "zlib":
        import zlib
        return zlib.compress(raw, int(compressor.get("level", 1)))
    raise RuntimeError(f"micro_sam_amd.zarr_store: writing with compressor {compressor} is not supported")


class Array:
    """One zarr v2 array (C order, no filters)."""

    def __init__(self, path: str, read_only: bool = False) -> None:
        self._path = path
        self._read_only = read_only
        with io.open(os.path.join(path, ".zarray"), "r") as fh:
            meta = json.load(fh)
        if meta.get("zarr_format") != 2:
            raise RuntimeError(f"{path}: unsupported zarr_format {meta.get('zarr_format')}")
        if meta.get("order", "C") != "C" or meta.get("filters"):
            raise RuntimeError(f"{path}: only C-order arrays without filters are supported")
        self.shape: Tuple[int, ...] = tuple(int(s) for s in meta["shape"])
        self.chunks: Tuple[int, ...] = tuple(int(s) for s in meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self._compressor = meta.get("compressor")
        fv = meta.get("fill_value", 0)
        self._fill = 0 if fv is None else (float(fv) if isinstance(fv, str) else fv)   # "NaN" / "Infinity" strings
        self._sep = meta.get("dimension_separator", ".")
        self.attrs = Attributes(os.path.join(path, ".zattrs"), read_only)

    # -- geometry
    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def name(self) -> str:
        return os.path.basename(self._path)

    def __len__(self) -> int:
        return self.shape[0]

    def _chunk_path(self, cidx: Sequence[int]) -> str:
        if self._sep == "/":
            return os.path.join(self._path, *[str(c) for c in cidx])
        return os.path.join(self._path, ".".join(str(c) for c in cidx))

    def _normalise(self, key) -> Tuple[Tuple[Tuple[int, int], ...], Tuple[bool, ...]]:
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            pos = key.index(Ellipsis)
            key = key[:pos] + (slice(None),) * (self.ndim - (len(key) - 1)) + key[pos + 1:]
        if len(key) > self.ndim:
            raise IndexError(f"too many indices for a {self.ndim}-d array")
        key = key + (slice(None),) * (self.ndim - len(key))
        ranges, squeeze = [], []
        for k, n in zip(key, self.shape):
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError(f"index {k} out of range for axis of length {n}")
                ranges.append((k, k + 1)); squeeze.append(True)
            elif isinstance(k, slice):
                start, stop, step = k.indices(n)
                if step != 1:
                    raise IndexError("only unit-step slices are supported")
                ranges.append((start, max(stop, start))); squeeze.append(False)
            else:
                raise IndexError(f"unsupported index {k!r}")
        return tuple(ranges), tuple(squeeze)

    def _chunk_ranges(self, ranges):
        """Iterate (chunk index, slices inside the chunk, slices inside the selection)."""
        per_axis = []
        for (lo, hi), c in zip(ranges, self.chunks):
            items = []
            if hi > lo:
                for ci in range(lo // c, (hi - 1) // c + 1):
                    a, b = max(lo, ci * c), min(hi, (ci + 1) * c)
                    items.append((ci, slice(a - ci * c, b - ci * c), slice(a - lo, b - lo)))
            per_axis.append(items)
        idx = [0] * len(per_axis)
        if any(len(p) == 0 for p in per_axis):
            return
        while True:
            sel = [per_axis[d][idx[d]] for d in range(len(per_axis))]
            yield tuple(s[0] for s in sel), tuple(s[1] for s in sel), tuple(s[2] for s in sel)
            d = len(per_axis) - 1
            while d >= 0:
                idx[d] += 1
                if idx[d] < len(per_axis[d]):
                    break
                idx[d] = 0
                d -= 1
            if d < 0:
                return

    def _read_chunk(self, cidx) -> Optional[np.ndarray]:
        try:
            with io.open(self._chunk_path(cidx), "rb") as fh:
                raw = fh.read()
        except FileNotFoundError:
            return None
        buf = _decode(raw, self._compressor, int(np.prod(self.chunks)) * self.dtype.itemsize)
        return np.frombuffer(buf, dtype=self.dtype).reshape(self.chunks)

    def __getitem__(self, key) -> np.ndarray:
        ranges, squeeze = self._normalise(key)
        out = np.empty(tuple(hi - lo for lo, hi in ranges), dtype=self.dtype)
        for cidx, in_chunk, in_sel in self._chunk_ranges(ranges):
            chunk = self._read_chunk(cidx)
            out[in_sel] = self._fill if chunk is None else chunk[in_chunk]
        return out.reshape(tuple(s for s, q in zip(out.shape, squeeze) if not q))

    def __setitem__(self, key, value) -> None:
        if self._read_only:
            raise PermissionError("zarr container opened read-only")
        ranges, squeeze = self._normalise(key)
        sel_shape = tuple(hi - lo for lo, hi in ranges)
        value = np.asarray(value, dtype=self.dtype)
        kept = tuple(s for s, q in zip(sel_shape, squeeze) if not q)
        value = np.broadcast_to(value.reshape(value.shape[-len(kept):] if value.ndim > len(kept) and
                                              int(np.prod(value.shape)) == int(np.prod(kept)) else value.shape), kept)
        value = value.reshape(sel_shape)
        for cidx, in_chunk, in_sel in self._chunk_ranges(ranges):
            part = value[in_sel]
            if part.shape == self.chunks:
                chunk = np.ascontiguousarray(part)
            else:
                chunk = self._read_chunk(cidx)
                chunk = np.full(self.chunks, self._fill, dtype=self.dtype) if chunk is None else chunk.copy()
                chunk[in_chunk] = part
            path = self._chunk_path(cidx)
            if self._sep == "/":
                os.makedirs(os.path.dirname(path), exist_ok=True)
            _atomic_write(path, _encode(chunk.tobytes(), self._compressor))

    def chunk_initialized(self, cidx: Sequence[int]) -> bool:
        return os.path.exists(self._chunk_path(cidx))


class Group:
    def __init__(self, path: str, read_only: bool = False) -> None:
        self._path = path
        self._read_only = read_only
        self.attrs = Attributes(os.path.join(path, ".zattrs"), read_only)

    @property
    def path(self) -> str:
        return self._path

    def _child(self, name: str) -> str:
        return os.path.join(self._path, str(name))

    def __contains__(self, name: object) -> bool:
        p = self._child(str(name))
        return os.path.exists(os.path.join(p, ".zarray")) or os.path.exists(os.path.join(p, ".zgroup"))

    def __getitem__(self, name: str):
        p = self._child(name)
        if os.path.exists(os.path.join(p, ".zarray")):
            return Array(p, self._read_only)
        if os.path.exists(os.path.join(p, ".zgroup")):
            return Group(p, self._read_only)
        raise KeyError(name)

    def keys(self):
        return sorted(n for n in os.listdir(self._path) if n in self)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self) -> int:
        return len(self.keys())

    def require_group(self, name: str) -> "Group":
        p = self._child(name)
        if os.path.exists(os.path.join(p, ".zarray")):
            raise RuntimeError(f"{p} is an array, not a group")
        if not os.path.exists(os.path.join(p, ".zgroup")):
            if self._read_only:
                raise PermissionError("zarr container opened read-only")
            _init_group(p)
        return Group(p, self._read_only)

    def create_dataset(self, name: str, data: Optional[np.ndarray] = None, shape: Optional[Sequence[int]] = None,
                       chunks: Optional[Sequence[int]] = None, dtype: Any = None, compressor: Optional[str] = None,
                       fill_value: Any = 0) -> Array:
        """``group.create_dataset`` (zarr 2) / ``create_array`` (zarr 3) as used at reference util.py:685-707."""
        if self._read_only:
            raise PermissionError("zarr container opened read-only")
        if name in self:
            raise RuntimeError(f"dataset {name} exists already in {self._path}")
        if data is not None:
            data = np.asarray(data)
            shape = data.shape if shape is None else tuple(shape)
            dtype = data.dtype if dtype is None else dtype
        if shape is None or dtype is None:
            raise ValueError("create_dataset needs data or shape + dtype")
        shape = tuple(int(s) for s in shape)
        chunks = shape if chunks is None else tuple(int(c) for c in chunks)
        chunks = tuple(max(c, 1) for c in chunks)
        dt = np.dtype(dtype)
        p = self._child(name)
        os.makedirs(p, exist_ok=True)
        meta = {"zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": dt.str,
                "compressor": None if compressor is None else {"id": compressor, "level": 1},
                "fill_value": fill_value, "order": "C", "filters": None, "dimension_separator": "."}
        _atomic_write(os.path.join(p, ".zarray"), json.dumps(meta, indent=4).encode())
        arr = Array(p)
        if data is not None:
            arr[...] = data
        return arr

    create_array = create_dataset


# ------------------------------------------------------------------------------------------ zarr v3, read-only

class StaticAttributes(dict):
    """Attributes of a v3 node (they live inside ``zarr.json``); read-only here."""

    def asdict(self) -> Dict[str, Any]:
        return dict(self)

    def update(self, *args, **kwargs) -> None:      # noqa: D102
        raise PermissionError("zarr v3 containers are read-only in micro_sam_amd.zarr_store")

    __setitem__ = update


_V3_DTYPES = {"bool": "?", "int8": "i1", "int16": "i2", "int32": "i4", "int64": "i8", "uint8": "u1", "uint16": "u2",
              "uint32": "u4", "uint64": "u8", "float16": "f2", "float32": "f4", "float64": "f8", "complex64": "c8",
              "complex128": "c16"}


def _read_node_v3(path: str) -> Dict[str, Any]:
    with io.open(os.path.join(path, "zarr.json"), "r") as fh:
        meta = json.load(fh)
    if meta.get("zarr_format") != 3:
        raise RuntimeError(f"{path}: unsupported zarr_format {meta.get('zarr_format')}")
    return meta


class ArrayV3(Array):
    """One zarr v3 array (regular chunk grid; codecs ``bytes`` followed by any of ``zstd`` / ``gzip`` / ``blosc`` /
    ``crc32c``), read-only.  Indexing is inherited from the v2 array."""

    def __init__(self, path: str) -> None:    # noqa: D107 - does not call the v2 constructor on purpose
        self._path, self._read_only = path, True
        meta = _read_node_v3(path)
        if meta.get("node_type") != "array":
            raise RuntimeError(f"{path} is not an array")
        grid = meta["chunk_grid"]
        if grid.get("name") != "regular":
            raise RuntimeError(f"{path}: chunk grid '{grid.get('name')}' is not supported")
        self.shape = tuple(int(v) for v in meta["shape"])
        self.chunks = tuple(int(v) for v in grid["configuration"]["chunk_shape"])
        dt = meta["data_type"]
        if not isinstance(dt, str) or dt not in _V3_DTYPES:
            raise RuntimeError(f"{path}: data type {dt!r} is not supported")
        endian, self._byte_codecs = "<", []
        for codec in meta.get("codecs", []):
            name, conf = codec.get("name"), codec.get("configuration") or {}
            if name == "bytes":
                endian = ">" if conf.get("endian", "little") == "big" else "<"
            elif name in ("zstd", "gzip", "blosc", "crc32c"):
                self._byte_codecs.append((name, conf))
            else:
                raise RuntimeError(f"{path}: codec '{name}' is not supported (bytes, zstd, gzip, blosc, crc32c are)")
        self.dtype = np.dtype(endian + _V3_DTYPES[dt])
        fv = meta.get("fill_value", 0)
        self._fill = 0 if fv is None else (float(fv.replace("Infinity", "inf")) if isinstance(fv, str) else fv)
        enc = meta.get("chunk_key_encoding") or {"name": "default"}
        conf = enc.get("configuration") or {}
        self._v2_keys = enc.get("name") == "v2"
        self._sep = conf.get("separator", "." if self._v2_keys else "/")
        self._compressor = None
        self.attrs = StaticAttributes(meta.get("attributes") or {})

    def _chunk_path(self, cidx: Sequence[int]) -> str:
        parts = [str(c) for c in cidx]
        if self._v2_keys:
            return os.path.join(self._path, *(parts if self._sep == "/" else [".".join(parts)]))
        return os.path.join(self._path, *(["c"] + parts if self._sep == "/" else [self._sep.join(["c"] + parts)]))

    def _read_chunk(self, cidx) -> Optional[np.ndarray]:
        try:
            with io.open(self._chunk_path(cidx), "rb") as fh:
                buf = fh.read()
        except FileNotFoundError:
            return None
        nbytes = int(np.prod(self.chunks)) * self.dtype.itemsize
        for name, conf in reversed(self._byte_codecs):              # decoding runs the pipeline backwards
            if name == "crc32c":
                buf = buf[:-4]                                       # trailing checksum (not verified)
            elif name == "gzip":
                import gzip
                buf = gzip.decompress(buf)
            elif name == "zstd":
                from .zarr_codecs import zstd_decompress
                buf = zstd_decompress(buf, nbytes)
            else:
                from .zarr_codecs import blosc_decompress
                buf = blosc_decompress(buf)
        return np.frombuffer(buf, dtype=self.dtype).reshape(self.chunks)

    def __setitem__(self, key, value) -> None:
        raise PermissionError("zarr v3 containers are read-only in micro_sam_amd.zarr_store")


class GroupV3:
    """A zarr v3 group, read-only: what ``precompute_image_embeddings`` needs to LOAD a cache that zarr-python 3 wrote."""

    def __init__(self, path: str) -> None:
        self._path = path
        meta = _read_node_v3(path)
        if meta.get("node_type") != "group":
            raise RuntimeError(f"{path} is not a group")
        self.attrs = StaticAttributes(meta.get("attributes") or {})

    @property
    def path(self) -> str:
        return self._path

    def __contains__(self, name: object) -> bool:
        return os.path.exists(os.path.join(self._path, str(name), "zarr.json"))

    def __getitem__(self, name: str):
        p = os.path.join(self._path, str(name))
        if name not in self:
            raise KeyError(name)
        return ArrayV3(p) if _read_node_v3(p).get("node_type") == "array" else GroupV3(p)

    def keys(self):
        return sorted(n for n in os.listdir(self._path) if n in self)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self) -> int:
        return len(self.keys())

    def require_group(self, name: str) -> "GroupV3":
        if name in self:
            return self[name]
        raise PermissionError("zarr v3 containers are read-only in micro_sam_amd.zarr_store (the cache is incomplete: "
                              "recompute it into a new save_path)")

    def create_dataset(self, *args, **kwargs):
        raise PermissionError("zarr v3 containers are read-only in micro_sam_amd.zarr_store (the cache is incomplete: "
                              "recompute it into a new save_path)")

    create_array = create_dataset


def _init_group(path: str) -> None:
    os.makedirs(path, exist_ok=True)
    _atomic_write(os.path.join(path, ".zgroup"), json.dumps({"zarr_format": 2}, indent=4).encode())


def open(path, mode: str = "a") -> Group:   # noqa: A001 - mirrors zarr.open
    """``zarr.open(path, mode)`` for a directory container holding a root group (modes "r", "a", "w")."""
    path = os.fspath(path)
    if mode not in ("r", "a", "w"):
        raise ValueError(f"unsupported mode {mode!r}")
    if os.path.exists(os.path.join(path, "zarr.json")):
        if mode == "w":
            raise RuntimeError(f"{path} is a zarr v3 container; micro_sam_amd.zarr_store writes the v2 layout only")
        return GroupV3(path)                        # read-only, whatever the mode: a complete cache is only read
    exists = os.path.exists(os.path.join(path, ".zgroup"))
    if mode == "r":
        if not exists:
            raise FileNotFoundError(f"{path} is not a zarr v2 group")
        return Group(path, read_only=True)
    if mode == "w" and os.path.isdir(path):
        import shutil
        shutil.rmtree(path)
        exists = False
    if not exists:
        if os.path.isdir(path) and os.listdir(path):
            raise RuntimeError(f"{path} exists and is not a zarr v2 group")
        _init_group(path)
    return Group(path)
