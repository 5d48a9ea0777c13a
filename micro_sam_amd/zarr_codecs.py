"""Chunk decoders for embedding caches that zarr-python wrote with its default compressors (host side, read path only).

The reference creates its datasets with zarr's defaults (``micro_sam/util.py:685-707``): Blosc (LZ4, byte shuffle) under
zarr-python 2, Zstandard under zarr-python 3.  Neither ``numcodecs`` nor ``blosc`` / ``lz4`` / ``zstandard`` is available
here, so the two container formats are decoded directly:

* ``blosc_decompress``: the Blosc 1 frame (c-blosc ``blosc.c``: 16-byte header, block start table, per block 1 or
  ``typesize`` splits each prefixed with its compressed size, optional byte un-shuffle) around LZ4 / zlib / zstd streams;
* ``lz4_block_decompress``: the LZ4 block format (token, literal run, little-endian match offset, match run);
* ``zstd_decompress``: through pyarrow's bundled zstd when pyarrow is importable (optional).

PARITY UNPINNED against the real libraries' writers (they are not in this image): the LZ4 decoder is tested on blocks
produced by pyarrow's LZ4 (``lz4_raw``), the Blosc frame on frames assembled by the test suite from the format description
above (tests/test_zarr_cache_host.py).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

BLOSC_MAX_SPLITS = 16          # c-blosc: MAX_SPLITS
BLOSC_MIN_BUFFERSIZE = 128     # c-blosc: MIN_BUFFERSIZE
_BLOSC_CODECS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}


def lz4_block_decompress(src: bytes, out_size: int) -> bytes:
    """One LZ4 block -> exactly ``out_size`` bytes."""
    src = memoryview(src)
    out = bytearray(out_size)
    n, i, o = len(src), 0, 0
    while i < n:
        token = src[i]; i += 1
        run = token >> 4
        if run == 15:
            while True:
                b = src[i]; i += 1
                run += b
                if b != 255:
                    break
        if run:
            if i + run > n or o + run > out_size:
                raise ValueError("corrupt LZ4 block (literal run)")
            out[o:o + run] = src[i:i + run]
            i += run; o += run
        if i >= n:                                     # the last sequence holds literals only
            break
        offset = src[i] | (src[i + 1] << 8); i += 2
        if offset == 0 or offset > o:
            raise ValueError("corrupt LZ4 block (match offset)")
        length = token & 15
        if length == 15:
            while True:
                b = src[i]; i += 1
                length += b
                if b != 255:
                    break
        length += 4
        if o + length > out_size:
            raise ValueError("corrupt LZ4 block (match run)")
        start = o - offset
        if offset >= length:
            out[o:o + length] = out[start:start + length]
        else:                                          # overlapping match = the last `offset` bytes repeated
            pattern = bytes(out[start:o])
            reps = -(-length // offset)
            out[o:o + length] = (pattern * reps)[:length]
        o += length
    if o != out_size:
        raise ValueError(f"corrupt LZ4 block: {o} bytes instead of {out_size}")
    return bytes(out)


def zstd_decompress(src: bytes, out_size: int) -> bytes:
    try:
        import pyarrow as pa
    except ImportError as exc:
        raise RuntimeError("micro_sam_amd.zarr_codecs: zstd-compressed chunks need pyarrow (its bundled zstd) to be read") from exc
    return pa.decompress(bytes(src), decompressed_size=int(out_size), codec="zstd", asbytes=True)


def _unshuffle(buf: bytes, typesize: int) -> bytes:
    """Inverse of Blosc's byte shuffle on one block: byte plane j of the elements back to byte j of each element; the
    tail that does not fill an element stays where it is."""
    n = len(buf) // typesize
    body = np.frombuffer(buf, dtype=np.uint8, count=n * typesize).reshape(typesize, n).T
    return np.ascontiguousarray(body).tobytes() + buf[n * typesize:]


def blosc_decompress(src: bytes) -> bytes:
    """A Blosc 1 frame (what ``numcodecs.Blosc`` writes) -> the original bytes."""
    if len(src) < 16:
        raise ValueError("corrupt Blosc frame (header)")
    _version, _versionlz, flags, typesize = src[0], src[1], src[2], src[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<iii", src, 4)
    if cbytes != len(src):
        raise ValueError(f"corrupt Blosc frame: header says {cbytes} bytes, chunk has {len(src)}")
    if flags & 0x02:                                   # BLOSC_MEMCPYED
        return bytes(src[16:16 + nbytes])
    if flags & 0x04:
        raise RuntimeError("micro_sam_amd.zarr_codecs: bit-shuffled Blosc frames are not supported")
    codec = _BLOSC_CODECS.get((flags & 0xE0) >> 5)
    if codec not in ("lz4", "zlib", "zstd"):
        raise RuntimeError(f"micro_sam_amd.zarr_codecs: Blosc frames compressed with '{codec}' are not supported")
    shuffled = bool(flags & 0x01) and typesize > 1
    dont_split = bool(flags & 0x10)
    nblocks = -(-nbytes // blocksize) if nbytes else 0
    starts = struct.unpack_from(f"<{nblocks}i", src, 16)
    out = []
    for b in range(nblocks):
        this = min(blocksize, nbytes - b * blocksize)
        leftover = this != blocksize
        split = (not dont_split) and typesize <= BLOSC_MAX_SPLITS and (blocksize // typesize) >= BLOSC_MIN_BUFFERSIZE \
            and not leftover
        nsplits = typesize if split else 1
        part = this // nsplits
        pos = starts[b]
        pieces = []
        for _ in range(nsplits):
            (csize,) = struct.unpack_from("<i", src, pos); pos += 4
            chunk = src[pos:pos + csize]; pos += csize
            if csize == part:                          # stored
                pieces.append(bytes(chunk))
            elif codec == "lz4":
                pieces.append(lz4_block_decompress(chunk, part))
            elif codec == "zlib":
                pieces.append(zlib.decompress(bytes(chunk)))
            else:
                pieces.append(zstd_decompress(chunk, part))
        block = b"".join(pieces)
        if len(block) != this:
            raise ValueError("corrupt Blosc frame (block size)")
        out.append(_unshuffle(block, typesize) if shuffled else block)
    return b"".join(out)
