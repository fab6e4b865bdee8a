"""Minimal baseline-TIFF reader / writer for the single-channel scientific images the reference trains on
(float32 .tif files read by MONAI ``LoadImage`` -> tifffile; ``vae_scripts/README.md:259``).  tifffile is not
installed in this environment, so the subset that those files use is implemented here with the struct module:
classic TIFF (magic 42), little or big endian, strips that are uncompressed or zlib/deflate-compressed (codes 8 and
32946, no predictor), one sample per pixel, 8/16/32-bit unsigned or signed integers and 32/64-bit floats.  Tiled,
LZW/JPEG-compressed, predictor-coded, multi-sample or BigTIFF files raise ``ValueError`` naming the unsupported
feature."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 8: ("h", 2), 9: ("i", 4),
          11: ("f", 4), 12: ("d", 8), 16: ("Q", 8)}


def _values(buf: bytes, bo: str, typ: int, count: int, value_field: bytes):
    fmt, size = _TYPES[typ]
    total = size * count
    raw = value_field[:total] if total <= 4 else None
    return fmt, size, total, raw


def read_tiff(path: str) -> np.ndarray:
    """Return the first image of ``path`` as a 2-D numpy array (rows x columns) in its stored dtype."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    magic, ifd = struct.unpack(bo + "HI", buf[2:8])
    if magic == 43:
        raise ValueError(f"{path}: BigTIFF is not supported")
    if magic != 42:
        raise ValueError(f"{path}: bad TIFF magic {magic}")
    (n,) = struct.unpack(bo + "H", buf[ifd:ifd + 2])
    tags = {}
    for i in range(n):
        e = buf[ifd + 2 + 12 * i: ifd + 14 + 12 * i]
        tag, typ, count = struct.unpack(bo + "HHI", e[:8])
        if typ not in _TYPES:
            continue
        fmt, size, total, raw = _values(buf, bo, typ, count, e[8:12])
        if raw is None:
            (off,) = struct.unpack(bo + "I", e[8:12])
            raw = buf[off:off + total]
        if typ == 5:
            tags[tag] = struct.unpack(bo + "I" * (2 * count), raw)
        elif typ == 2:
            tags[tag] = raw
        else:
            tags[tag] = struct.unpack(bo + fmt * count, raw)
    need = (256, 257, 273)
    for t in need:
        if t not in tags:
            raise ValueError(f"{path}: missing TIFF tag {t}" + (" (tiled TIFFs are not supported)" if 322 in tags else ""))
    width, height = tags[256][0], tags[257][0]
    comp = tags.get(259, (1,))[0]
    if comp not in (1, 8, 32946):
        raise ValueError(f"{path}: compressed TIFF (compression={comp}) is not supported (only none and deflate)")
    if comp != 1 and tags.get(317, (1,))[0] != 1:
        raise ValueError(f"{path}: deflate with predictor {tags[317][0]} is not supported")
    if tags.get(277, (1,))[0] != 1:
        raise ValueError(f"{path}: {tags[277][0]} samples per pixel; only single-channel images are supported")
    bits = tags.get(258, (1,))[0]
    sfmt = tags.get(339, (1,))[0]
    kind = {1: "u", 2: "i", 3: "f"}.get(sfmt)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise ValueError(f"{path}: unsupported sample format {sfmt} / {bits} bits")
    dtype = np.dtype(f"{bo}{kind}{bits // 8}")
    offsets = tags[273]
    counts = tags.get(279)
    rows_per_strip = tags.get(278, (height,))[0]
    out = np.empty((height, width), dtype=dtype.newbyteorder("="))
    row = 0
    for si, off in enumerate(offsets):
        rows = min(rows_per_strip, height - row)
        nbytes = rows * width * dtype.itemsize
        if comp == 1:
            if counts is not None and counts[si] < nbytes:
                raise ValueError(f"{path}: strip {si} is shorter than its rows")
            out[row:row + rows] = np.frombuffer(buf, dtype=dtype, count=rows * width, offset=off).reshape(rows, width)
        else:
            if counts is None:
                raise ValueError(f"{path}: compressed strips need StripByteCounts")
            raw = zlib.decompress(buf[off:off + counts[si]])
            if len(raw) < nbytes:
                raise ValueError(f"{path}: strip {si} inflates to {len(raw)} bytes, {nbytes} expected")
            out[row:row + rows] = np.frombuffer(raw, dtype=dtype, count=rows * width).reshape(rows, width)
        row += rows
        if row >= height:
            break
    if row < height:
        raise ValueError(f"{path}: strips cover {row} of {height} rows")
    return out


def write_tiff(path: str, image: np.ndarray, rows_per_strip: int | None = None, big_endian: bool = False,
               deflate: bool = False) -> None:
    """Write a 2-D array as a single-strip (or multi-strip) classic TIFF, uncompressed or deflate-compressed."""
    a = np.ascontiguousarray(image)
    if a.ndim != 2:
        raise ValueError("write_tiff: 2-D arrays only")
    kind = {"u": 1, "i": 2, "f": 3}.get(a.dtype.kind)
    if kind is None:
        raise ValueError(f"write_tiff: dtype {a.dtype}")
    bo = ">" if big_endian else "<"
    a = a.astype(a.dtype.newbyteorder(bo))
    h, w = a.shape
    rps = rows_per_strip or h
    nstrips = (h + rps - 1) // rps
    strips = [a[i * rps:(i + 1) * rps].tobytes() for i in range(nstrips)]
    if deflate:
        strips = [zlib.compress(b_, 6) for b_ in strips]
    data = b"".join(strips)
    strip_bytes = [len(b_) for b_ in strips]
    data_off = 8
    strip_offs = [data_off + sum(strip_bytes[:i]) for i in range(nstrips)]
    extra_off = data_off + len(data)
    extra = b""

    def entry(tag, typ, vals):
        nonlocal extra
        fmt, size = _TYPES[typ]
        raw = struct.pack(bo + fmt * len(vals), *vals)
        if len(raw) <= 4:
            field = raw.ljust(4, b"\0")
        else:
            field = struct.pack(bo + "I", extra_off + len(extra))
            extra += raw
        return struct.pack(bo + "HHI", tag, typ, len(vals)) + field

    entries = [entry(256, 4, [w]), entry(257, 4, [h]), entry(258, 3, [a.itemsize * 8]), entry(259, 3, [8 if deflate else 1]),
               entry(262, 3, [1]), entry(273, 4, strip_offs), entry(277, 3, [1]), entry(278, 4, [rps]),
               entry(279, 4, strip_bytes), entry(339, 3, [kind])]
    ifd_off = extra_off + len(extra)
    ifd = struct.pack(bo + "H", len(entries)) + b"".join(entries) + struct.pack(bo + "I", 0)
    with open(path, "wb") as f:
        f.write((b"MM" if big_endian else b"II") + struct.pack(bo + "HI", 42, ifd_off))
        f.write(data)
        f.write(extra)
        f.write(ifd)
