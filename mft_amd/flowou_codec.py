"""The reference's on-disk cache entry, ``<left>--<right>.flowouX16.pkl``
(``write_flowou_X16`` / ``read_flowou_X16``, MFT/utils/io.py:495-563): a pickled dict

    {'flow_x' | 'flow_y' | 'occlusion' | 'sigma':
        {'data': PNG file as a uint8 array, 'min': float32, 'max': float32}}

where each PNG is an 8-bit 3-channel image holding one uint16-quantised channel:
cv2 channel order (B, G, R) = (0, high byte, low byte), i.e. (R, G, B) = (low, high, 0)
in the file.

Split of the work here: min/max + quantisation and its inverse run on the MI355X
(``mftx_quantize_u16`` / ``mftx_dequantize_u16``) so only uint16 planes cross PCIe; the
PNG container is host work -- zlib from the standard library, scanline
reconstruction in C (``mftx_png_unfilter``) -- since cv2 does not exist in this
environment.  Files written here are plain PNGs (filter type 0) that ``cv2.imdecode``
reads; files written by the reference (any filter type, non-interlaced 8-bit RGB) are
read here.
"""
from __future__ import annotations

import ctypes as C
import pickle
import struct
import zlib

import numpy as np
import torch

from . import _lib, ops

CHANNELS = ("flow_x", "flow_y", "occlusion", "sigma")
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def png_encode_rgb8(img, level=1):
    """(H, W, 3) uint8 RGB -> PNG file bytes (non-interlaced, filter type 0 on every row)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W, ch = img.shape
    assert ch == 3
    raw = np.zeros((H, 1 + 3 * W), np.uint8)
    raw[:, 1:] = img.reshape(H, 3 * W)
    ihdr = struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)
    return _PNG_SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")


def png_decode_rgb8(buf):
    """PNG file bytes (8-bit RGB, non-interlaced) -> (H, W, 3) uint8 RGB."""
    buf = bytes(memoryview(np.ascontiguousarray(buf)).cast("B")) if not isinstance(buf, (bytes, bytearray)) else bytes(buf)
    if buf[:8] != _PNG_SIG:
        raise ValueError("not a PNG file")
    pos, idat, ihdr = 8, [], None
    while pos < len(buf):
        (length,), tag = struct.unpack(">I", buf[pos:pos + 4]), buf[pos + 4:pos + 8]
        body = buf[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", buf[pos + 8 + length:pos + 12 + length])
        if zlib.crc32(tag + body) & 0xFFFFFFFF != crc:
            raise ValueError(f"PNG chunk {tag!r}: bad CRC")
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + length
    if ihdr is None:
        raise ValueError("PNG without IHDR")
    W, H, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or ctype != 2 or interlace != 0:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}): "
                         "flowouX16 planes are 8-bit RGB, non-interlaced")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).copy()
    if raw.size != H * (1 + 3 * W):
        raise ValueError("PNG data size mismatch")
    _lib.check(_lib.load().mftx_png_unfilter(raw.ctypes.data_as(C.c_void_p), H, 3 * W, 3), "mftx_png_unfilter")
    return raw[: H * 3 * W].reshape(H, W, 3)


def _u16_to_png(u16):
    """(H, W) uint16 -> PNG of (R, G, B) = (low byte, high byte, 0) == cv2's (B, G, R) = (0, high, low)."""
    rgb = np.zeros(u16.shape + (3,), np.uint8)
    rgb[..., 0] = u16 & 0xFF
    rgb[..., 1] = u16 >> 8
    return np.frombuffer(png_encode_rgb8(rgb), np.uint8)


def _png_to_u16(buf):
    rgb = png_decode_rgb8(buf)
    return (rgb[..., 1].astype(np.uint16) << 8) | rgb[..., 0]


def pack_flowou_X16(path, channels):
    """Host half of the writer: ``channels`` = 4 x (uint16 (H, W) array, min, max) in CHANNELS order."""
    result = {name: {"data": _u16_to_png(np.asarray(q, np.uint16)), "min": np.float32(lo), "max": np.float32(hi)}
              for name, (q, lo, hi) in zip(CHANNELS, channels)}
    with open(path, "wb") as fout:
        pickle.dump(result, fout)


def unpack_flowou_X16(path):
    """Host half of the reader: -> 4 x (uint16 (H, W) array, min, max) in CHANNELS order."""
    with open(path, "rb") as fin:
        data = pickle.load(fin)
    return [(_png_to_u16(data[name]["data"]), np.float32(data[name]["min"]), np.float32(data[name]["max"]))
            for name in CHANNELS]


def write_flowou_X16(path, flow, occlusions, uncertainty):
    """flow (2, H, W), occlusions (1, H, W), uncertainty (1, H, W): float32 DEVICE tensors."""
    planes = (flow[0], flow[1], occlusions[0], uncertainty[0])
    enc = [ops.quantize_u16(p) for p in planes]                 # 8 launches, no sync yet
    channels = []
    for q, lohi in enc:
        lo, hi = lohi.cpu().numpy()
        channels.append((q.cpu().numpy(), lo, hi))
    pack_flowou_X16(path, channels)


def read_flowou_X16(path, device="cuda"):
    """-> flow (2, H, W), occlusions (1, H, W), uncertainty (1, H, W): float32 tensors on ``device``."""
    fx, fy, occl, sigma = (ops.dequantize_u16(torch.from_numpy(q).to(device), float(lo), float(hi))
                           for q, lo, hi in unpack_flowou_X16(path))
    return torch.stack([fx, fy]), occl[None], sigma[None]
