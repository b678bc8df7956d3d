"""Tiny PNG writer for the tests (zlib + the five PNG row filters), so that the C++ reader in include/xfeat/image_io.h is
exercised on every filter type and colour type without an imaging library."""
import struct
import zlib

import numpy as np


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def write_png(path, img: np.ndarray, filters=None):
    """img: uint8 [H,W] (gray), [H,W,2] (gray+alpha), [H,W,3] (RGB) or [H,W,4] (RGBA); filters: per-row filter types (0..4)"""
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = img.reshape(h, w * ch).astype(np.uint8)
    raw = bytearray()
    prev = np.zeros(w * ch, np.uint8)
    for y in range(h):
        cur = rows[y]
        ft = (filters[y % len(filters)] if filters else 0)
        a = np.concatenate([np.zeros(ch, np.uint8), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.uint8), prev[:-ch]])
        if ft == 0: enc = cur
        elif ft == 1: enc = cur - a
        elif ft == 2: enc = cur - prev
        elif ft == 3: enc = cur - ((a.astype(np.int32) + prev) >> 1).astype(np.uint8)
        else: enc = cur - _paeth(a, prev, c)
        raw.append(ft); raw += enc.astype(np.uint8).tobytes()
        prev = cur

    def chunk(ty, data):
        return struct.pack(">I", len(data)) + ty + data + struct.pack(">I", zlib.crc32(ty + data) & 0xffffffff)
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2                                   # two IDAT chunks: the reader must concatenate them
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                + chunk(b"tEXt", b"Comment\x00test") + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


def opencv_gray(img: np.ndarray, rgb_flag: int) -> np.ndarray:
    """what the reference feeds the extractor for a colour file: imread -> B,G,R memory order, then COLOR_RGB2GRAY (Camera.RGB = 1)
    or COLOR_BGR2GRAY, OpenCV's 8-bit fixed point (4899, 9617, 1868, >> 14 with rounding)"""
    if img.ndim == 2:
        return img
    if img.shape[2] == 2:
        return img[:, :, 0]
    r, g, b = (img[:, :, k].astype(np.int64) for k in range(3))
    c0, c1, c2 = b, g, r
    y = (c0 * 4899 + c1 * 9617 + c2 * 1868) if rgb_flag else (c0 * 1868 + c1 * 9617 + c2 * 4899)
    return ((y + (1 << 13)) >> 14).astype(np.uint8)
