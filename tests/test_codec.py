"""Flow-cache codec: the ``.flowouX16.pkl`` container on the host (no GPU
needed: pickle, PNG via zlib, scanline reconstruction in C from libmftx.so) and, marked
gpu, the quantisation kernels against the oracle, bit for bit."""
import pickle
import struct
import zlib

import numpy as np
import pytest
import torch

from oracle import mft_oracle as O
from mft_amd import flowou_codec as fc


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def png_with_filters(img, filters):
    """Independent test-side PNG writer: row y is stored with filter type filters[y % len]."""
    H, W, _ = img.shape
    bpp, rb = 3, 3 * W
    flat = img.reshape(H, rb).astype(np.int32)
    raw = bytearray()
    for y in range(H):
        ft = filters[y % len(filters)]
        cur = flat[y]
        prev = flat[y - 1] if y else np.zeros(rb, np.int32)
        out = np.zeros(rb, np.int32)
        for i in range(rb):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) // 2, paeth(int(a), int(b), int(c))][ft]
            out[i] = (cur[i] - pred) & 0xFF
        raw.append(ft)
        raw += bytes(out.astype(np.uint8))
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2                      # two IDAT chunks, plus an ancillary chunk to skip

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) +
            chunk(b"tEXt", b"Software\x00test") + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("filters", [[0], [1], [2], [3], [4], [0, 1, 2, 3, 4], [4, 3, 1]])
def test_png_decode_every_filter_type(filters):
    rng = np.random.default_rng(len(filters) * 7 + filters[0])
    img = rng.integers(0, 256, size=(13, 17, 3), dtype=np.uint8)
    img[3:6] = 200                                      # runs, so that Up/Paeth predictors matter
    out = fc.png_decode_rgb8(png_with_filters(img, filters))
    assert out.dtype == np.uint8 and np.array_equal(out, img)
    # the same bytes handed over like cv2.imencode returns them (uint8 array, also as a column)
    assert np.array_equal(fc.png_decode_rgb8(np.frombuffer(png_with_filters(img, filters), np.uint8)[:, None]), img)


def test_png_roundtrip_and_errors():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(32, 48, 3), dtype=np.uint8)
    buf = fc.png_encode_rgb8(img)
    assert buf[:8] == b"\x89PNG\r\n\x1a\n" and np.array_equal(fc.png_decode_rgb8(buf), img)
    with pytest.raises(ValueError):
        fc.png_decode_rgb8(b"JFIF" + buf)
    bad = bytearray(buf); bad[40] ^= 0xFF
    with pytest.raises(ValueError):
        fc.png_decode_rgb8(bytes(bad))
    gray = bytearray(buf); gray[8 + 8 + 9] = 0             # colour type 0: CRC no longer matches either
    with pytest.raises(ValueError):
        fc.png_decode_rgb8(bytes(gray))


def test_container_layout_matches_reference_writer(tmp_path):
    """What write_flowou_X16 pickles (MFT/utils/io.py:495-533): four {'data','min','max'} dicts,
    'data' a PNG of cv2 (B, G, R) = (0, high byte, low byte)."""
    rng = np.random.default_rng(3)
    planes = [rng.normal(0, s, size=(24, 40)).astype(np.float32) for s in (5.0, 3.0, 0.3, 2.0)]
    chans = [O.quantize_u16(p) for p in planes]
    path = tmp_path / "3--7.flowouX16.pkl"
    fc.pack_flowou_X16(path, chans)
    with open(path, "rb") as f:
        d = pickle.load(f)
    assert list(d) == ["flow_x", "flow_y", "occlusion", "sigma"]
    for name, (q, lo, hi) in zip(d, chans):
        e = d[name]
        assert sorted(e) == ["data", "max", "min"] and e["data"].dtype == np.uint8 and e["data"].ndim == 1
        assert e["min"].dtype == np.float32 and e["min"] == lo and e["max"] == hi
        bgr = fc.png_decode_rgb8(e["data"])[..., ::-1]          # what cv2.imdecode would return
        assert np.array_equal(bgr, O.u16_to_bgr(q)) and np.array_equal(O.bgr_to_u16(bgr), q)
    back = fc.unpack_flowou_X16(path)
    for (q, lo, hi), (q2, lo2, hi2) in zip(chans, back):
        assert np.array_equal(q, q2) and lo == lo2 and hi == hi2


def test_reading_a_reference_style_file(tmp_path):
    """An entry whose PNGs use row filters (as libpng / cv2 write them) and a column-shaped buffer."""
    rng = np.random.default_rng(4)
    chans = [O.quantize_u16(rng.normal(0, 2, size=(16, 20)).astype(np.float32)) for _ in range(4)]
    d = {}
    for name, (q, lo, hi) in zip(fc.CHANNELS, chans):
        png = png_with_filters(np.ascontiguousarray(O.u16_to_bgr(q)[..., ::-1]), [1, 4, 2])
        d[name] = {"data": np.frombuffer(png, np.uint8)[:, None], "min": lo, "max": hi}
    path = tmp_path / "0--1.flowouX16.pkl"
    with open(path, "wb") as f:
        pickle.dump(d, f)
    for (q, lo, hi), (q2, lo2, hi2) in zip(chans, fc.unpack_flowou_X16(path)):
        assert np.array_equal(q, q2) and lo == lo2 and hi == hi2


def test_oracle_quantisation_properties():
    rng = np.random.default_rng(5)
    x = rng.normal(0, 7, size=(50, 60)).astype(np.float32)
    q, lo, hi = O.quantize_u16(x)
    assert q.dtype == np.uint16 and q.min() == 0 and q.max() == 65535 and lo == x.min() and hi == x.max()
    back = O.dequantize_u16(q, lo, hi)
    assert back.dtype == np.float32 and np.abs(back - x).max() <= (hi - lo) / 65535 * 0.5 + 1e-5
    q0, lo0, hi0 = O.quantize_u16(np.full((4, 4), 2.5, np.float32))
    assert not q0.any() and np.all(O.dequantize_u16(q0, lo0, hi0) == 2.5)


# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,scale", [((512, 512), 30.0), ((67, 93), 1.0), ((1, 5), 1e-3), ((1088, 1920), 400.0)])
def test_quantize_dequantize_bit_exact_vs_oracle(shape, scale):
    from mft_amd import ops
    rng = np.random.default_rng(shape[0] + shape[1])
    x = (rng.normal(0, scale, size=shape) + rng.uniform(-scale, scale)).astype(np.float32)
    q, lohi = ops.quantize_u16(torch.from_numpy(x).cuda())
    oq, lo, hi = O.quantize_u16(x)
    assert tuple(lohi.cpu().numpy()) == (lo, hi)
    assert np.array_equal(q.cpu().numpy(), oq)
    back = ops.dequantize_u16(q, float(lo), float(hi)).cpu().numpy()
    assert np.array_equal(back, O.dequantize_u16(oq, lo, hi))


@pytest.mark.gpu
def test_quantize_flat_channel_and_errors():
    from mft_amd import ops
    from mft_amd._lib import MftxError
    x = torch.full((8, 12), -3.25, device="cuda")
    q, lohi = ops.quantize_u16(x)
    assert not q.cpu().numpy().any() and lohi.tolist() == [-3.25, -3.25]
    assert torch.equal(ops.dequantize_u16(q, -3.25, -3.25), x)
    with pytest.raises(MftxError):
        ops.quantize_u16(torch.empty(0, device="cuda"))
    with pytest.raises(MftxError):
        ops.quantize_u16(torch.zeros(4, 4))


@pytest.mark.gpu
def test_flow_cache_disk_tier_on_device(tmp_path):
    """Spill to disk -> read back: quantisation error bounded by half a step of each channel's
    range, bit-identical to the oracle's decode of the same file, reference file naming."""
    from mft_amd.io import FlowCache
    g = torch.Generator().manual_seed(1)
    H, W = 96, 128
    val = (torch.randn(2, H, W, generator=g) * 9, torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g) * 4 + 0.1)
    val = tuple(t.cuda() for t in val)
    c = FlowCache(tmp_path / "d", max_RAM_MB=0, max_GPU_RAM_MB=0)
    c.write(4, 9, *val)
    assert [p.name for p in (tmp_path / "d").iterdir()] == ["4--9.flowouX16.pkl"]
    got = c.read(4, 9)
    want = [O.dequantize_u16(q, lo, hi) for q, lo, hi in fc.unpack_flowou_X16(tmp_path / "d" / "4--9.flowouX16.pkl")]
    planes = [got[0][0], got[0][1], got[1][0], got[2][0]]
    src = [val[0][0], val[0][1], val[1][0], val[2][0]]
    for p, w, s in zip(planes, want, src):
        assert p.is_cuda and np.array_equal(p.cpu().numpy(), w)
        step = float(s.max() - s.min()) / 65535
        assert float((p - s).abs().max()) <= 0.5 * step + 1e-5


@pytest.mark.gpu
def test_result_write_read_flowouX16(tmp_path):
    from mft_amd.results import FlowOUTrackingResult
    g = torch.Generator().manual_seed(2)
    r = FlowOUTrackingResult(torch.randn(2, 40, 56, generator=g) * 5, torch.rand(1, 40, 56, generator=g),
                             torch.rand(1, 40, 56, generator=g) + 0.5)
    p = tmp_path / "0--12.flowouX16.pkl"
    r.write(p)
    back = FlowOUTrackingResult.read(p)
    assert not back.flow.is_cuda and float((back.flow - r.flow).abs().max()) < 1e-3
    assert float((back.occlusion - r.occlusion).abs().max()) < 2e-5 and float((back.sigma - r.sigma).abs().max()) < 2e-5
    r.write(tmp_path / "x.flowou.png")                       # the fixed-point variant: 1/32 px flow steps
    back = FlowOUTrackingResult.read(tmp_path / "x.flowou.png")
    assert float((back.flow - r.flow).abs().max()) <= 1 / 32 and float((back.occlusion - r.occlusion).abs().max()) <= 2 ** -15


def test_flowou_png_and_X32_vs_reference_golden(golden_dir, tmp_path):
    """``.flowou.png`` (16-bit fixed point) and ``.flowouX32.pkl`` against what the REFERENCE's own writers handed to
    cv2 and its readers returned (tests/golden/codec.npz, captured under a cv2 stub): planes and decoded arrays
    bit for bit, through real files and the product's own PNG coder."""
    import golden_inputs as gi
    g = np.load(golden_dir / "codec.npz")
    d = gi.codec_inputs()
    # .flowou.png
    p1 = tmp_path / "sub" / "3--5.flowou.png"
    fc.write_flowou(p1, d["flow"], d["occl"], d["sigma"])
    with open(p1, "rb") as fh:
        assert np.array_equal(fc.cv2_imdecode_png(fh.read()), g["png16_bgra"])       # what cv2.imread would return
    f, o, s_ = fc.read_flowou(p1)
    assert f.dtype == np.float32
    assert np.array_equal(f, g["png16_dec_flow"]) and np.array_equal(o, g["png16_dec_occl"])
    assert np.array_equal(s_, g["png16_dec_sigma"])
    with pytest.raises(AssertionError):
        fc.write_flowou(tmp_path / "x.flowou.png", d["flow"] * 1000, d["occl"], d["sigma"])   # |flow| >= 1024
    # .flowouX32.pkl
    p2 = tmp_path / "3--5.flowouX32.pkl"
    with np.errstate(invalid="ignore"):
        fc.write_flowou(p2, d["flow"], d["occl"], d["sigma"])
    with open(p2, "rb") as fh:
        pk = pickle.load(fh)
    for i, name in enumerate(fc.CHANNELS):
        assert np.array_equal(fc.cv2_imdecode_png(pk[name]["data"]), g["x32_bgra"][i]), name
        assert np.float32(pk[name]["min"]) == g["x32_lohi"][i, 0] and np.float32(pk[name]["max"]) == g["x32_lohi"][i, 1]
    f, o, s_ = fc.read_flowou(p2)
    assert np.array_equal(f, g["x32_dec_flow"]) and np.array_equal(o, g["x32_dec_occl"])
    assert np.array_equal(s_, g["x32_dec_sigma"])
    with pytest.raises(ValueError):
        fc.read_flowou(tmp_path / "a.flowou2.png")
    # FlowOUTrackingResult.write / read dispatch on the suffix (MFT/results.py:61-72)
    from mft_amd.results import FlowOUTrackingResult
    r = FlowOUTrackingResult(torch.from_numpy(d["flow"]), torch.from_numpy(d["occl"]), torch.from_numpy(d["sigma"]))
    r.write(tmp_path / "r.flowou.png")
    back = FlowOUTrackingResult.read(tmp_path / "r.flowou.png")
    assert np.array_equal(back.flow.numpy(), g["png16_dec_flow"])


def test_png_16bit_rgba_every_filter_type():
    """16-bit RGBA rows (bpp = 8) through the C scanline reconstruction, all filter types."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, size=(9, 11, 4)).astype(np.uint16)
    H, W, _ = img.shape
    be = img.astype(">u2").view(np.uint8).reshape(H, W * 8).astype(np.int32)
    bpp, rb = 8, W * 8
    raw = bytearray()
    for y in range(H):
        ft = y % 5
        cur, prev = be[y], (be[y - 1] if y else np.zeros(rb, np.int32))
        out = np.zeros(rb, np.int32)
        for i in range(rb):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) // 2, paeth(int(a), int(b), int(c))][ft]
            out[i] = (cur[i] - pred) & 0xFF
        raw.append(ft)
        raw += bytes(out.astype(np.uint8))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 16, 6, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))
    assert np.array_equal(fc.png_decode(png), img)


@pytest.mark.gpu
def test_device_codec_vs_reference_golden(golden_dir, tmp_path):
    """mftx_quantize_u16 / mftx_dequantize_u16 and the container against what the REFERENCE's own
    write_flowou_X16 / read_flowou_X16 produced for the same inputs (tests/golden/codec.npz, captured in
    the build container under a cv2 stub): uint16 planes, (min, max) and decoded floats, bit for bit."""
    import golden_inputs as gi
    from mft_amd import ops
    g = np.load(golden_dir / "codec.npz")
    d = gi.codec_inputs()
    flow, occl, sigma = (torch.from_numpy(d[k]).cuda() for k in ("flow", "occl", "sigma"))
    planes = [flow[0], flow[1], occl[0], sigma[0]]          # views at odd offsets: 37 * 53 floats apart
    dec = [g["dec_flow"][0], g["dec_flow"][1], g["dec_occl"][0], g["dec_sigma"][0]]
    for i, x in enumerate(planes):
        q, lohi = ops.quantize_u16(x)
        want_u16 = O.bgr_to_u16(g["bgr"][i])
        assert np.array_equal(q.cpu().numpy(), want_u16), i
        assert np.array_equal(lohi.cpu().numpy(), g["lohi"][i]), i
        back = ops.dequantize_u16(q, float(g["lohi"][i, 0]), float(g["lohi"][i, 1]))
        assert np.array_equal(back.cpu().numpy(), dec[i]), i
    # the file round trip through the product's writer / reader gives the reference's decoded arrays
    path = tmp_path / "3--5.flowouX16.pkl"
    fc.write_flowou_X16(path, flow, occl, sigma)
    f2, o2, s2 = fc.read_flowou_X16(path)
    assert np.array_equal(f2.cpu().numpy(), g["dec_flow"]) and np.array_equal(o2.cpu().numpy(), g["dec_occl"])
    assert np.array_equal(s2.cpu().numpy(), g["dec_sigma"])
    # and the PNG planes inside are the (B, G, R) planes the reference hands to cv2.imencode
    with open(path, "rb") as fh:
        pk = pickle.load(fh)
    for i, name in enumerate(fc.CHANNELS):
        assert np.array_equal(fc.png_decode_rgb8(pk[name]["data"])[..., ::-1], g["bgr"][i]), name


@pytest.mark.gpu
def test_codec_on_unaligned_batched_views(tmp_path):
    """Planes sliced out of a batched engine output start at k * H * W * 4 bytes: with H * W % 4 != 0 they
    are not 16-byte aligned.  The cache's disk tier and result.write must take them."""
    from mft_amd import ops
    from mft_amd.io import FlowCache
    H, W, P = 125, 187, 3
    gen = torch.Generator().manual_seed(5)
    flow = torch.randn(P, 2, H, W, generator=gen).cuda()
    occl = torch.rand(P, 1, H, W, generator=gen).cuda()
    sigma = torch.rand(P, 1, H, W, generator=gen).cuda()
    assert flow[1][1].data_ptr() % 16 != 0
    for i in range(P):
        for x in (flow[i][0], flow[i][1], occl[i][0], sigma[i][0]):
            q, lohi = ops.quantize_u16(x)
            rq, lo, hi = O.quantize_u16(x.cpu().numpy())
            assert np.array_equal(q.cpu().numpy(), rq) and float(lohi[0]) == lo and float(lohi[1]) == hi
    cache = FlowCache(tmp_path / "c", max_RAM_MB=0, max_GPU_RAM_MB=0)
    cache.write(4, 9, flow[1], occl[1], sigma[1])                       # straight to the disk tier
    f, o, s_ = cache.read(4, 9)
    assert f is not None and (f - flow[1]).abs().max() < (flow[1].max() - flow[1].min()) / 65535
