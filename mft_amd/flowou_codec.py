"""The reference's on-disk cache entry, ``<left>--<right>.flowouX16.pkl``
(``write_flowou_X16`` / ``read_flowou_X16``, MFT/utils/io.py:495-563): a pickled dict

    {'flow_x' | 'flow_y' | 'occlusion' | 'sigma':
        {'data': PNG file as a uint8 array, 'min': float32, 'max': float32}}

where each PNG is an 8-bit 3-channel image holding one uint16-quantised channel:
cv2 channel order (B, G, R) = (0, high byte, low byte), i.e. (R, G, B) = (low, high, 0)
in the file.

(The ``.flowou.png`` fixed-point and ``.flowouX32`` variants of the reference are at the end of this module.)

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


_COLOUR_TYPE = {3: 2, 4: 6}                     # channels -> PNG colour type (RGB, RGBA)


def png_encode(img, level=1):
    """(H, W, 3 | 4) uint8 or uint16 array in FILE channel order (R, G, B[, A]) -> PNG file bytes
    (non-interlaced, filter type 0 on every row, 16-bit samples big-endian)."""
    img = np.ascontiguousarray(img)
    H, W, ch = img.shape
    assert ch in _COLOUR_TYPE and img.dtype in (np.uint8, np.uint16)
    depth = 8 * img.dtype.itemsize
    body = img.astype(">u2").view(np.uint8) if depth == 16 else img
    row = ch * W * (depth // 8)
    raw = np.zeros((H, 1 + row), np.uint8)
    raw[:, 1:] = body.reshape(H, row)
    ihdr = struct.pack(">IIBBBBB", W, H, depth, _COLOUR_TYPE[ch], 0, 0, 0)
    return _PNG_SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")


def png_decode(buf):
    """PNG file bytes (8- or 16-bit RGB / RGBA, non-interlaced, any scanline filters) -> (H, W, C) uint8 / uint16
    array in FILE channel order.  Scanline reconstruction runs in C (``mftx_png_unfilter``)."""
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
    ch = {2: 3, 6: 4}.get(ctype)
    if depth not in (8, 16) or ch is None or interlace != 0:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}): "
                         "flow-cache planes are 8/16-bit RGB(A), non-interlaced")
    bpp = ch * depth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).copy()
    if raw.size != H * (1 + bpp * W):
        raise ValueError("PNG data size mismatch")
    _lib.check(_lib.load().mftx_png_unfilter(raw.ctypes.data_as(C.c_void_p), H, bpp * W, bpp), "mftx_png_unfilter")
    data = raw[: H * bpp * W]
    if depth == 16:
        return data.view(">u2").astype(np.uint16).reshape(H, W, ch)
    return data.reshape(H, W, ch)


def png_encode_rgb8(img, level=1):
    """(H, W, 3) uint8 RGB -> PNG file bytes."""
    assert img.shape[2] == 3 and img.dtype == np.uint8
    return png_encode(img, level)


def png_decode_rgb8(buf):
    """PNG file bytes (8-bit RGB, non-interlaced) -> (H, W, 3) uint8 RGB."""
    img = png_decode(buf)
    if img.dtype != np.uint8 or img.shape[2] != 3:
        raise ValueError("unsupported PNG: flowouX16 planes are 8-bit RGB, non-interlaced")
    return img


def cv2_imencode_png(arr):
    """What ``cv2.imencode('.png', arr)[1]`` holds for a 3- / 4-channel uint8 / uint16 array in cv2's (B, G, R[, A])
    channel order: the file stores (R, G, B[, A])."""
    arr = np.asarray(arr)
    order = [2, 1, 0] + ([3] if arr.shape[2] == 4 else [])
    return np.frombuffer(png_encode(arr[..., order]), np.uint8)


def cv2_imdecode_png(buf):
    """``cv2.imdecode(buf, cv2.IMREAD_UNCHANGED)`` for such a PNG: back to (B, G, R[, A])."""
    img = png_decode(buf)
    order = [2, 1, 0] + ([3] if img.shape[2] == 4 else [])
    return np.ascontiguousarray(img[..., order])


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


# ---------------------------------------------------------------------------------------------------------
# The reference's other two cache-entry formats (MFT/utils/io.py:174-219 dispatches on the first suffix).
# They are cold paths (the tracker's cache writes .flowouX16): host-side numpy, evaluated exactly as the
# reference evaluates them (same dtypes, same operation order), PNG container as above.
# ---------------------------------------------------------------------------------------------------------
FLOWOU_IO_FLOW_MULTIPLIER = 2 ** 5          # MFT/utils/io.py:170-172
FLOWOU_IO_OCCLUSION_MULTIPLIER = 2 ** 15
FLOWOU_IO_UNCERTAINTY_MULTIPLIER = 2 ** 9


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def write_flowou1_png(path, flow, occlusions, uncertainty):
    """``.flowou.png`` (MFT/utils/io.py:222-259): one 16-bit 4-channel PNG, fixed point -- flow 2**15 + 32 x (|x| < 1024),
    occlusion 2**15 x clipped to [0, 1], sigma 2**9 x clipped to [0, 127]; values are truncated by astype(uint16)."""
    from pathlib import Path
    path = Path(path)
    assert path.suffixes == ['.flowou', '.png']
    path.parent.mkdir(parents=True, exist_ok=True)
    flow, occlusions, uncertainty = _np(flow), _np(occlusions), _np(uncertainty)
    f = np.transpose(flow, (1, 2, 0))
    assert np.all(np.abs(f) < 2 ** 15 / FLOWOU_IO_FLOW_MULTIPLIER), "out-of-range values - cannot be written"
    f = 2 ** 15 + FLOWOU_IO_FLOW_MULTIPLIER * f
    o = FLOWOU_IO_OCCLUSION_MULTIPLIER * np.transpose(np.clip(occlusions, 0, 1), (1, 2, 0))
    u = np.transpose(np.clip(uncertainty, 0, 127), (1, 2, 0))
    assert np.all(u >= 0) and np.all(u < 2 ** 16 / FLOWOU_IO_UNCERTAINTY_MULTIPLIER)
    u = FLOWOU_IO_UNCERTAINTY_MULTIPLIER * u
    data = np.concatenate([f, o, u], axis=2).astype(np.uint16)      # what the reference hands to cv2.imwrite
    with open(path, "wb") as fout:
        fout.write(cv2_imencode_png(data).tobytes())


def read_flowou1_png(path):
    """-> flow (2, H, W), occlusions (1, H, W), uncertainty (1, H, W) float32 numpy (MFT/utils/io.py:262-291)."""
    from pathlib import Path
    assert Path(path).suffixes == ['.flowou', '.png']
    with open(path, "rb") as fin:
        data = cv2_imdecode_png(fin.read())
    data = np.transpose(data, (2, 0, 1))
    flow = (data[:2].astype(np.float32) - 2 ** 15) / FLOWOU_IO_FLOW_MULTIPLIER
    occl = data[2:3].astype(np.float32) / FLOWOU_IO_OCCLUSION_MULTIPLIER
    unc = data[3:4].astype(np.float32) / FLOWOU_IO_UNCERTAINTY_MULTIPLIER
    return flow, occl, unc


def write_flowou_X32(path, flow, occlusions, uncertainty):
    """``.flowouX32.pkl`` (MFT/utils/io.py:372-411): like X16 with 32-bit quantisation (truncating) in four uint8 planes."""
    import pickle

    def encode_channel(xs):
        f_xs = np.float32(xs)
        lb, ub = np.amin(f_xs), np.amax(f_xs)
        xs_01 = np.zeros_like(f_xs) if np.abs(ub - lb) < 1e-8 else (f_xs - lb) / (ub - lb)
        with np.errstate(invalid="ignore"):
            q = np.uint32(xs_01 * (2 ** 32 - 1))
        planes = np.dstack((np.uint8((q & 0xFF000000) >> 24), np.uint8((q & 0x00FF0000) >> 16),
                            np.uint8((q & 0x0000FF00) >> 8), np.uint8(q & 0x000000FF)))
        return {'data': cv2_imencode_png(planes), 'min': lb, 'max': ub}

    flow, occlusions, uncertainty = _np(flow), _np(occlusions), _np(uncertainty)
    result = {'flow_x': encode_channel(flow[0]), 'flow_y': encode_channel(flow[1]),
              'occlusion': encode_channel(occlusions[0]), 'sigma': encode_channel(uncertainty[0])}
    with open(path, 'wb') as fout:
        pickle.dump(result, fout)


def read_flowou_X32(path):
    """-> flow (2, H, W), occlusions (1, H, W), uncertainty (1, H, W) float32 numpy (MFT/utils/io.py:414-443)."""
    import pickle

    def decode_channel(d):
        b4, b3, b2, b1 = np.dsplit(np.uint32(cv2_imdecode_png(d['data'])), 4)
        q = ((b4 << 24) | (b3 << 16) | (b2 << 8) | b1)[..., 0]
        return (np.float32(q) / (2 ** 32 - 1)) * (d['max'] - d['min']) + d['min']

    with open(path, 'rb') as fin:
        data = pickle.load(fin)
    flow = np.stack((decode_channel(data['flow_x']), decode_channel(data['flow_y'])), axis=0)
    return flow, decode_channel(data['occlusion'])[None], decode_channel(data['sigma'])[None]


def write_flowou(path, flow, occlusions, uncertainty):
    """Dispatch on the first suffix like MFT/utils/io.py:174-197."""
    from pathlib import Path
    suf = Path(path).suffixes[0]
    if suf == '.flowou':
        write_flowou1_png(path, flow, occlusions, uncertainty)
    elif suf == '.flowouX16':
        dev = flow.device if isinstance(flow, torch.Tensor) and flow.is_cuda else "cuda"
        write_flowou_X16(path, *(torch.as_tensor(t).to(dev, torch.float32) for t in (flow, occlusions, uncertainty)))
    elif suf == '.flowouX32':
        write_flowou_X32(path, flow, occlusions, uncertainty)
    else:
        raise ValueError(f"Incorrect flowou path suffix: {Path(path).suffixes}")


def read_flowou(path):
    """-> flow (2, H, W), occlusions (1, H, W), uncertainty (1, H, W) as float32 numpy arrays (MFT/utils/io.py:200-219)."""
    from pathlib import Path
    suf = Path(path).suffixes[0]
    if suf == '.flowou':
        return read_flowou1_png(path)
    if suf == '.flowouX16':
        return tuple(t.cpu().numpy() for t in read_flowou_X16(path))
    if suf == '.flowouX32':
        return read_flowou_X32(path)
    raise ValueError(f"Incorrect flowou path suffix: {Path(path).suffixes}")
