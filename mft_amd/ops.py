"""Torch-tensor front end of the C ABI (device memory + streams are PyTorch's;
all arithmetic happens in libmftx's HIP kernels).

Every function checks device / dtype / contiguity, passes raw device pointers
and the current HIP stream, and raises ``MftxError`` on a non-zero return --
same contract as the reference's one native op
(``MFT/RAFT/alt_cuda_corr/correlation.cpp:19-33``: CHECK_CUDA, CHECK_CONTIGUOUS,
RuntimeError).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, MftxError, check

ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _device_visible(t):
    """A tensor a kernel may address: device memory, or pinned host memory (mapped into the device's address space)."""
    return isinstance(t, torch.Tensor) and (t.is_cuda or t.is_pinned())


def copy_bytes(src: torch.Tensor, dst: torch.Tensor):
    """dst <- src by a copy KERNEL on the current stream (``mftx_copy_bytes``): either side may be pinned host memory.  No
    SDMA queue is involved, so uploads and downloads enqueued this way do not serialise behind each other."""
    if not (_device_visible(src) and _device_visible(dst)):
        raise MftxError("copy_bytes: tensors must be on the device or in pinned host memory")
    if not (src.is_contiguous() and dst.is_contiguous()) or src.numel() * src.element_size() != dst.numel() * dst.element_size():
        raise MftxError("copy_bytes: contiguous tensors of the same size in bytes")
    check(_lib.load().mftx_copy_bytes(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), _stream()), "mftx_copy_bytes")
    return dst


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MftxError(f"{name} must be a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise MftxError(f"{name} must be {dtype}")
    if not t.is_contiguous():
        raise MftxError(f"{name} must be contiguous")
    return t.data_ptr()


# ---------------------------------------------------------------------------
# weight packing (host plumbing, once per checkpoint)
# ---------------------------------------------------------------------------

def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> [round_up(Cout,128), kh*kw, round_up(Cin,32)], zero
    padded: the K axis (tap, cin) is contiguous per output channel, which is what
    the NT implicit-GEMM kernel streams (csrc/conv_gemm.hip)."""
    cout, cin, kh, kw = w.shape
    n_pad = -(-cout // 128) * 128
    c_pad = -(-cin // 32) * 32
    out = torch.zeros(n_pad, kh * kw, c_pad, dtype=torch.float32, device=w.device)
    out[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    return out.contiguous()


ARITH_F32, ARITH_SPLIT = 0, 1


class SplitRangeError(MftxError):
    """An operand of the split arithmetic is not below 65504 in magnitude (or not finite)."""


def count_not_below(x: torch.Tensor, limit: float) -> int:
    """Number of elements of a float32 device tensor with !(|x| < limit) -- NaN counts; limit = inf counts the
    non-finite ones (``mftx_count_not_below``).  Synchronises (reads one counter back)."""
    lib = _lib.load()
    if x.numel() == 0:
        return 0
    x = x if x.is_contiguous() else x.contiguous()
    cnt = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(lib.mftx_count_not_below(_chk(x, "x"), x.numel(), float(limit), cnt.data_ptr(), _stream()), "mftx_count_not_below")
    return int(cnt.item())


def split_weights(wpk: torch.Tensor, check_range: bool = True) -> torch.Tensor:
    """Packed fp32 conv weights -> the split-fp16 form (same shape and size) that ``conv2d(arith=ARITH_SPLIT)``
    and the refinement engine's split arithmetic stream (csrc/conv_gemm.hip: split_weights_kernel).
    check_range (weights: always; it costs one host sync at load): values that are not below 65504 in magnitude have no
    finite fp16 high half -- ``SplitRangeError`` instead of NaN products later."""
    lib = _lib.load()
    if check_range:
        bad = count_not_below(wpk, _lib.SPLIT_LIMIT)
        if bad:
            raise SplitRangeError(f"split arithmetic: {bad} of {wpk.numel()} values are not below {_lib.SPLIT_LIMIT:g} "
                                  "in magnitude (fp16 range of the high halves); use arith='fp32'")
    out = torch.empty_like(wpk)
    check(lib.mftx_split_weights(_chk(wpk, "wpk"), out.data_ptr(), wpk.numel(), _stream()), "mftx_split_weights")
    return out


def split_activations(x: torch.Tensor) -> torch.Tensor:
    """fp32 [M, C] (C % 8 == 0) -> the split form the engine stores GEMM inputs in: every 8 channels of a row as
    [hi x 8 | lo x 8] fp16 (same shape and dtype as a container).  The same kernel as ``split_weights``; no range check
    (activations out of range surface as NaN, see ``mftx_count_not_below``)."""
    return split_weights(x, check_range=False)


def unsplit_activations(x: torch.Tensor) -> torch.Tensor:
    """Inverse of ``split_activations`` (exact: hi + lo / 2048 is representable in fp32)."""
    m, c = x.shape
    h = x.contiguous().view(torch.float16).reshape(m, c // 8, 2, 8).float()
    return (h[:, :, 0] + h[:, :, 1] * (1.0 / 2048.0)).reshape(m, c)


def pack_raft_weights(sd: dict, device) -> list:
    """The 30 tensors ``mftx_raft_create`` expects, in WeightSlot order
    (csrc/raft_engine.hip).  z|r gates and the two OU heads are fused into single
    GEMMs by stacking / block-placing their weights (exact: the extra terms are
    multiplications by zero weights)."""
    g = lambda k: sd[k].to(device=device, dtype=torch.float32)  # noqa: E731
    u, o = "update_block.", "occlusion_block."
    out = []

    def conv(name):
        out.append(pack_conv_weight(g(name + ".weight")))
        out.append(g(name + ".bias").contiguous())

    conv(u + "encoder.convc1")
    conv(u + "encoder.convc2")
    # convf1: [128, 2, 7, 7] -> [(ky, kx, c), co] for the direct kernel
    out.append(g(u + "encoder.convf1.weight").permute(2, 3, 1, 0).reshape(98, 128).contiguous())
    out.append(g(u + "encoder.convf1.bias").contiguous())
    conv(u + "encoder.convf2")
    conv(u + "encoder.conv")
    # GRU gates: input channels are [h(128) | inp(128) | motion(128)].  The inp columns are split off
    # (they are evaluated once per pair, csrc/raft_engine.hip); the remaining [h | motion] columns form
    # the per-iteration GEMM.
    dyn = list(range(0, 128)) + list(range(256, 384))
    for sfx in ("1", "2"):
        wzr = torch.cat([g(u + f"gru.convz{sfx}.weight"), g(u + f"gru.convr{sfx}.weight")], 0)
        out.append(pack_conv_weight(wzr[:, dyn].contiguous()))
        out.append(pack_conv_weight(wzr[:, 128:256].contiguous()))
        out.append(torch.cat([g(u + f"gru.convz{sfx}.bias"), g(u + f"gru.convr{sfx}.bias")]).contiguous())
        wq = g(u + f"gru.convq{sfx}.weight")
        out.append(pack_conv_weight(wq[:, dyn].contiguous()))
        out.append(pack_conv_weight(wq[:, 128:256].contiguous()))
        out.append(g(u + f"gru.convq{sfx}.bias").contiguous())
    conv(u + "flow_head.conv1")
    conv(u + "flow_head.conv2")
    conv(u + "mask.0")
    conv(u + "mask.2")
    w1 = torch.cat([g(o + "occl_head.conv1.weight"), g(o + "uncertainty_head.conv1.weight")], 0)
    out.append(pack_conv_weight(w1))
    out.append(torch.cat([g(o + "occl_head.conv1.bias"), g(o + "uncertainty_head.conv1.bias")]).contiguous())
    w2 = torch.zeros(3, 256, 3, 3, dtype=torch.float32, device=device)
    w2[0:2, 0:128] = g(o + "occl_head.conv2.weight")
    w2[2:3, 128:256] = g(o + "uncertainty_head.conv2.weight")
    out.append(pack_conv_weight(w2))
    out.append(torch.cat([g(o + "occl_head.conv2.bias"), g(o + "uncertainty_head.conv2.bias")]).contiguous())
    assert len(out) == _lib.NUM_RAFT_WEIGHTS
    return out


_ENC_CONVS = ["conv1", "layer1.0.conv1", "layer1.0.conv2", "layer1.1.conv1", "layer1.1.conv2",
              "layer2.0.conv1", "layer2.0.conv2", "layer2.0.downsample.0", "layer2.1.conv1", "layer2.1.conv2",
              "layer3.0.conv1", "layer3.0.conv2", "layer3.0.downsample.0", "layer3.1.conv1", "layer3.1.conv2",
              "conv2"]


def _bn_name(conv_name):
    """BatchNorm that follows a conv of BasicEncoder (core/extractor.py:6-62,118-166)."""
    if conv_name == "conv1":
        return "norm1"
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    if conv_name == "conv2":
        return None
    blk, c = conv_name.rsplit(".", 1)
    return f"{blk}.norm{c[-1]}"


def pack_encoder_weights(sd: dict, prefix: str, batch_norm: bool, device) -> list:
    """(packed weight, bias) per conv of a BasicEncoder, in EncConv order
    (csrc/encoder.hip).  Eval-mode batch norm (cnet) is folded into the conv:
    BN(conv_w(x) + b) = conv_{w*s}(x) + (b*s + t), s = gamma/sqrt(var+eps), t = beta - mean*s.
    The 7x7 stem is repacked as 7 row taps over 28-float (7 pixels x RGB0) windows; the cnet
    head is split into its tanh (net) and relu (inp) halves (core/raft.py:146-149)."""
    g = lambda k: sd[f"{prefix}.{k}"].to(device=device, dtype=torch.float32)  # noqa: E731
    out = []
    for name in _ENC_CONVS:
        w, b = g(name + ".weight"), g(name + ".bias")
        bn = _bn_name(name) if batch_norm else None
        if bn is not None:
            s_ = g(bn + ".weight") / torch.sqrt(g(bn + ".running_var") + 1e-5)
            t_ = g(bn + ".bias") - g(bn + ".running_mean") * s_
            w = w * s_.reshape(-1, 1, 1, 1)
            b = b * s_ + t_
        if name == "conv1":                       # [64,3,7,7] -> [128][ky][kx*4 + c], c = 3 zero
            w4 = torch.zeros(64, 7, 7, 4, dtype=torch.float32, device=device)
            w4[..., :3] = w.permute(0, 2, 3, 1)
            pk = torch.zeros(128, 7, 32, dtype=torch.float32, device=device)
            pk[:64, :, :28] = w4.reshape(64, 7, 28)
            out += [pk.contiguous(), b.contiguous()]
        elif name == "conv2" and batch_norm:       # cnet head: net | inp halves
            out += [pack_conv_weight(w[:128].contiguous()), b[:128].contiguous(),
                    pack_conv_weight(w[128:].contiguous()), b[128:].contiguous()]
        else:
            out += [pack_conv_weight(w.contiguous()), b.contiguous()]
    return out


class EncoderEngine:
    """Handle on the native encoder runtime (``mftx_encoder_*``): fnet or cnet."""

    def __init__(self, state_dict: dict, prefix: str, instance_norm: bool, device, arith=None, graph=True):
        """graph: replay the layers between pre-processing and head as a hipGraph (``mftx_encoder_set_graph``)."""
        lib = _lib.load()
        self.device = torch.device(device)
        self.instance_norm = instance_norm
        self.weights = pack_encoder_weights(state_dict, prefix, not instance_norm, self.device)
        arr, self._keep = _lib.ptr_array([_chk(t, "weight") for t in self.weights])
        handle = C.c_void_p()
        check(lib.mftx_encoder_create(arr, len(self.weights), int(instance_norm), C.byref(handle)),
              "mftx_encoder_create")
        self._h = handle
        self._ws = None
        if not graph:
            check(lib.mftx_encoder_set_graph(self._h, 0), "mftx_encoder_set_graph")
        self.arith = ARITH_SPLIT if arith is None else int(arith)
        if self.arith == ARITH_SPLIT:           # split-fp16 products: the convolutions stream split weights
            self.split = [split_weights(t) for t in self.weights[0::2]]
            sarr, self._keep_split = _lib.ptr_array([t.data_ptr() for t in self.split])
            check(lib.mftx_encoder_set_split_weights(self._h, sarr, len(self.split)), "mftx_encoder_set_split_weights")
        elif self.arith != ARITH_F32:
            raise MftxError(f"unknown arithmetic {arith!r}")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mftx_encoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def forward(self, img_u8: torch.Tensor, out=None):
        """img_u8: uint8 [H0, W0, 3] BGR on the device -> pixel-major maps at 1/8 resolution.
        out: optional pre-allocated (out0, out1) to write into (contiguous [h*w, C] views)."""
        lib = _lib.load()
        H0, W0 = img_u8.shape[:2]
        h, w = -(-H0 // 8), -(-W0 // 8)
        need = lib.mftx_encoder_workspace_bytes(H0, W0)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is not None:
            outs = out
            want = ((h * w, 256), None) if self.instance_norm else ((h * w, 128), (h * w, 128))
            for t, shp in zip(outs, want):
                if shp is not None and (t is None or tuple(t.shape) != shp):
                    raise MftxError(f"encoder output must be {shp}")
        elif self.instance_norm:
            outs = (torch.empty(h * w, 256, dtype=torch.float32, device=self.device), None)
        else:
            outs = (torch.empty(h * w, 128, dtype=torch.float32, device=self.device),
                    torch.empty(h * w, 128, dtype=torch.float32, device=self.device))
        check(lib.mftx_encoder_forward(self._h, _chk(img_u8, "img", torch.uint8), H0, W0, _chk(outs[0], "out0"),
                                       _chk(outs[1], "out1") if outs[1] is not None else None,
                                       self._ws.data_ptr(), self._ws.numel(), _stream()), "mftx_encoder_forward")
        return outs


# ---------------------------------------------------------------------------
# per-op wrappers
# ---------------------------------------------------------------------------

def pyramid_layout(h: int, w: int):
    """(stride[4] floats per query cell, (hb0, wb0, hb1, wb1) block grids) of the correlation pyramid
    (include/mftx.h, mftx_corr_pyramid)."""
    stride, grid = (C.c_longlong * 4)(), (C.c_int * 4)()
    check(_lib.load().mftx_corr_pyramid_layout(h, w, stride, grid), "mftx_corr_pyramid_layout")
    return list(stride), tuple(grid)


def _level_index(l: int, h: int, w: int, device):
    """Flat offset inside a query's level-l slice of every (y, x) of the level, row-major order."""
    stride, (hb0, wb0, hb1, wb1) = pyramid_layout(h, w)
    hl, wl = h >> l, w >> l
    ys, xs = torch.meshgrid(torch.arange(hl, device=device), torch.arange(wl, device=device), indexing="ij")
    if l < 2:
        wb = (wb0, wb1)[l]
        idx = ((ys >> 2) * wb + (xs >> 3)) * 32 + (ys & 3) * 8 + (xs & 7)
    else:
        idx = ys * wl + xs
    return idx.reshape(-1), stride[l]


def unblock_level(t: torch.Tensor, l: int, h: int, w: int) -> torch.Tensor:
    """Level l as mftx_corr_pyramid stores it [P, h*w, stride_l] -> row-major [P, h*w, h_l*w_l]
    (the reference's layout; tests and tools only)."""
    idx, _ = _level_index(l, h, w, t.device)
    return t[..., idx].contiguous()


def block_level(t: torch.Tensor, l: int, h: int, w: int) -> torch.Tensor:
    """Inverse of ``unblock_level``: row-major [P, h*w, h_l*w_l] -> the stored layout, padding zero."""
    idx, stride = _level_index(l, h, w, t.device)
    out = torch.zeros(t.shape[:-1] + (stride,), dtype=t.dtype, device=t.device)
    out[..., idx] = t
    return out


def corr_pyramid(f1: torch.Tensor, f2: torch.Tensor, h: int, w: int, arith: int = ARITH_F32):
    """f1, f2: pixel-major [P, h*w, C] -> 4 levels [P, h*w, stride_l] in the stored layout
    (``unblock_level`` gives the reference's row-major view).  ``arith``: ARITH_F32 (fp32 MFMA) or ARITH_SPLIT
    (split-fp16 products, what the refinement engine runs by default)."""
    lib = _lib.load()
    P, N, Cc = f1.shape
    assert N == h * w and f2.shape == f1.shape
    stride, _ = pyramid_layout(h, w)
    lv = [torch.empty(P, N, stride[l], dtype=torch.float32, device=f1.device) for l in range(4)]
    if arith == ARITH_SPLIT:
        scratch = torch.empty_like(f2)
        check(lib.mftx_corr_pyramid_split(_chk(f1, "f1"), _chk(f2, "f2"), P, Cc, h, w,
                                          *[t.data_ptr() for t in lv], scratch.data_ptr(), _stream()), "mftx_corr_pyramid_split")
        return lv
    if arith != ARITH_F32:
        raise MftxError("corr_pyramid: arith must be ARITH_F32 or ARITH_SPLIT")
    check(lib.mftx_corr_pyramid(_chk(f1, "f1"), _chk(f2, "f2"), P, Cc, h, w,
                                *[t.data_ptr() for t in lv], _stream()), "mftx_corr_pyramid")
    return lv


def corr_lookup(lv, coords: torch.Tensor, h: int, w: int, r: int = 4):
    """lv: the 4 levels in the stored layout; coords pixel-major [P, h*w, 2] (x, y) -> [P, h*w, 324]."""
    lib = _lib.load()
    P = coords.shape[0]
    stride, _ = pyramid_layout(h, w)
    for l, t in enumerate(lv):
        if tuple(t.shape) != (P, h * w, stride[l]):
            raise MftxError(f"corr_lookup: level {l} must be [{P}, {h * w}, {stride[l]}]")
    out = torch.empty(P, h * w, 324, dtype=torch.float32, device=coords.device)
    check(lib.mftx_corr_lookup(*[_chk(t, "level") for t in lv], _chk(coords, "coords"), P, h, w, r,
                               out.data_ptr(), 324, _stream()), "mftx_corr_lookup")
    return out


def pack_lookup_convc1_weights(wpk: torch.Tensor) -> torch.Tensor:
    """convc1's weight in the ``pack_conv_weight`` form [256, 1, 352] -> the fragment stream of the fused
    lookup + convc1 kernel (``mftx_pack_lookup_convc1_weights``; opaque bytes)."""
    lib = _lib.load()
    if wpk.dim() != 3 or wpk.shape[0] != 256 or wpk.shape[1] != 1 or wpk.shape[2] < 324:
        raise MftxError("pack_lookup_convc1_weights: expected the packed convc1 weight [256, 1, >= 324]")
    out = torch.empty(_lib.LOOKUP_CONVC1_WEIGHT_BYTES, dtype=torch.uint8, device=wpk.device)
    check(lib.mftx_pack_lookup_convc1_weights(_chk(wpk, "wpk"), wpk.shape[2], out.data_ptr(), _stream()),
          "mftx_pack_lookup_convc1_weights")
    return out


def pack_tile_conv_weights(wpk: torch.Tensor, N: int, cin: int, check_range: bool = True) -> torch.Tensor:
    """A layer's weight in the ``pack_conv_weight`` form [>= N, taps, cin_pad] -> the weight stream of the tile-resident
    conv kernel (``mftx_pack_tile_conv_weights``; opaque bytes, N * taps * cin * 4 of them).  Operands of the split
    arithmetic: with ``check_range`` a value not below 65504 raises ``SplitRangeError``."""
    lib = _lib.load()
    if wpk.dim() != 3 or wpk.shape[0] < N or wpk.shape[2] < cin:
        raise MftxError("pack_tile_conv_weights: expected the packed weight [>= N, taps, >= cin]")
    if check_range:
        bad = count_not_below(wpk, _lib.SPLIT_LIMIT)
        if bad:
            raise SplitRangeError(f"pack_tile_conv_weights: {bad} weights are not below {_lib.SPLIT_LIMIT} in magnitude")
    taps = wpk.shape[1]
    out = torch.empty(N * taps * cin * 4, dtype=torch.uint8, device=wpk.device)
    check(lib.mftx_pack_tile_conv_weights(_chk(wpk, "wpk"), N, taps, cin, wpk.shape[2], out.data_ptr(), _stream()),
          "mftx_pack_tile_conv_weights")
    return out


def tile_conv2d(x: torch.Tensor, wtile: torch.Tensor, bias, P, h, w, N, kh, kw, act=None, x2=None, addend=None,
                out_split=False, out=None):
    """``conv2d(..., arith=ARITH_SPLIT, a_split=True)`` on the tile-resident kernel (``mftx_tile_conv2d``): x (and x2)
    [P*h*w, 128] in split form, wtile from ``pack_tile_conv_weights`` -> [P*h*w, N] (fp32, or split form)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(P * h * w, N, dtype=torch.float32, device=x.device)
    d = ConvDesc()
    d.a0, d.lda0, d.c0 = _chk(x, "x"), x.shape[1], x.shape[1]
    if x2 is not None:
        d.a1, d.lda1, d.c1 = _chk(x2, "x2"), x2.shape[1], x2.shape[1]
    else:
        d.a1, d.lda1, d.c1 = None, 0, 0
    d.wpk = None
    d.bias = _chk(bias, "bias") if bias is not None else None
    d.out, d.ldo = _chk(out, "out"), out.shape[1]
    d.P, d.h, d.w, d.N, d.kh, d.kw = P, h, w, N, kh, kw
    d.act, d.out_scale = ACT[act], 1.0
    d.addend, d.ld_addend = (_chk(addend, "addend"), addend.shape[1]) if addend is not None else (None, 0)
    d.arith, d.a_split, d.out_split = ARITH_SPLIT, 1, int(bool(out_split))
    check(lib.mftx_tile_conv2d(C.byref(d), _chk(wtile, "wtile", torch.uint8), _stream()), "mftx_tile_conv2d")
    return out


def gru_half(h_split, motion_split, wzr_tile, wq_tile, pre_zr, pre_q, hf, P, h, w, vertical=False):
    """One SepConvGRU pass as one kernel (``mftx_gru_half``): h_split / motion_split [M, 128] in split form, the gates' weights
    as ``pack_tile_conv_weights`` streams (cin = 256), pre_zr [M, 256] / pre_q [M, 128] the pre-activation addends, hf [M, 128]
    the fp32 h -> (new h in fp32 [M, 128], new h in split form [M, 128], the kernel's scratch [M, 128]: the z gate's
    sums before the context part and the sigmoid)."""
    M = P * h * w
    z = torch.empty(M, 128, dtype=torch.float32, device=hf.device)
    h_out = torch.empty(M, 128, dtype=torch.float32, device=hf.device)
    hf_out = torch.empty(M, 128, dtype=torch.float32, device=hf.device)
    check(_lib.load().mftx_gru_half(_chk(h_split, "h"), h_split.shape[1], _chk(motion_split, "motion"), motion_split.shape[1],
                                    _chk(wzr_tile, "wzr", torch.uint8), _chk(wq_tile, "wq", torch.uint8), _chk(pre_zr, "pre_zr"),
                                    _chk(pre_q, "pre_q"), _chk(z, "z"), _chk(hf, "hf"), _chk(hf_out, "hf_out"), _chk(h_out, "h_out"), 128, P, h, w,
                                    1 if vertical else 0, _stream()), "mftx_gru_half")
    return hf_out, h_out, z


def pack_flow_head_weights(w2pk: torch.Tensor, check_range: bool = True) -> torch.Tensor:
    """flow_head.conv2's weight in the ``pack_conv_weight`` form [>= 2, 9, 256] -> the projection matrix of the fused
    flow head (``mftx_pack_flow_head_weights``; opaque bytes)."""
    lib = _lib.load()
    if w2pk.dim() != 3 or w2pk.shape[0] < 2 or tuple(w2pk.shape[1:]) != (9, 256):
        raise MftxError("pack_flow_head_weights: expected the packed weight [>= 2, 9, 256]")
    if check_range and count_not_below(w2pk, _lib.SPLIT_LIMIT):
        raise SplitRangeError(f"pack_flow_head_weights: weights not below {_lib.SPLIT_LIMIT} in magnitude")
    out = torch.empty(_lib.FLOW_HEAD_WEIGHT_BYTES, dtype=torch.uint8, device=w2pk.device)
    check(lib.mftx_pack_flow_head_weights(_chk(w2pk, "w2pk"), out.data_ptr(), _stream()), "mftx_pack_flow_head_weights")
    return out


def pack_ou_heads_weights(w1pk: torch.Tensor, w2pk: torch.Tensor, check_range: bool = True):
    """The occlusion + uncertainty heads' layers in the ``pack_conv_weight`` form -- first layers stacked [>= 256, 9, cin_pad >= 712],
    second layers block-diagonal [>= 3, 9, 256] -> (wtile, wproj) of ``mftx_ou_heads`` (opaque bytes)."""
    lib = _lib.load()
    if w1pk.dim() != 3 or w1pk.shape[0] < 256 or w1pk.shape[1] != 9 or w1pk.shape[2] < 712:
        raise MftxError("pack_ou_heads_weights: expected the packed first layers [>= 256, 9, >= 712]")
    if w2pk.dim() != 3 or w2pk.shape[0] < 3 or tuple(w2pk.shape[1:]) != (9, 256):
        raise MftxError("pack_ou_heads_weights: expected the packed second layers [>= 3, 9, 256]")
    if check_range and (count_not_below(w1pk, _lib.SPLIT_LIMIT) or count_not_below(w2pk, _lib.SPLIT_LIMIT)):
        raise SplitRangeError(f"pack_ou_heads_weights: weights not below {_lib.SPLIT_LIMIT} in magnitude")
    wtile = torch.empty(_lib.OU_HEADS_WTILE_BYTES, dtype=torch.uint8, device=w1pk.device)
    wproj = torch.empty(_lib.FLOW_HEAD_WEIGHT_BYTES, dtype=torch.uint8, device=w1pk.device)
    check(lib.mftx_pack_ou_heads_weights(_chk(w1pk, "w1pk"), w1pk.shape[2], _chk(w2pk, "w2pk"), wtile.data_ptr(), wproj.data_ptr(), _stream()),
          "mftx_pack_ou_heads_weights")
    return wtile, wproj


def ou_heads(a_split: torch.Tensor, h: int, w: int, wtile: torch.Tensor, b1, wproj: torch.Tensor, b2):
    """Both layers of the occlusion and uncertainty heads (``mftx_ou_heads``): a_split [P*h*w, 712] split-form rows ->
    [P*h*w, 4] = occlusion logits 0, 1, log-variance, (unused)."""
    lib = _lib.load()
    M = a_split.shape[0]
    P = M // (h * w)
    T = torch.empty(M, 27, dtype=torch.float32, device=a_split.device)
    out = torch.zeros(M, 4, dtype=torch.float32, device=a_split.device)
    check(lib.mftx_ou_heads(_chk(a_split, "a"), a_split.shape[1], P, h, w, _chk(wtile, "wtile", torch.uint8), _chk(b1, "b1"),
                            _chk(wproj, "wproj", torch.uint8), _chk(b2, "b2"), T.data_ptr(), out.data_ptr(), 4, _stream()), "mftx_ou_heads")
    return out


def flow_head(hsplit: torch.Tensor, h: int, w: int, wtile: torch.Tensor, b1, wproj: torch.Tensor, b2, coords=None):
    """delta = conv2(relu(conv1(h))) of the flow head without materialising the 256 hidden channels (``mftx_flow_head``):
    hsplit [P*h*w, >= 128] split-form rows -> delta [P*h*w, 2]; coords (optional, [P*h*w, 2]) += delta in place."""
    lib = _lib.load()
    M = hsplit.shape[0]
    P = M // (h * w)
    T = torch.empty(M, 18, dtype=torch.float32, device=hsplit.device)
    delta = torch.empty(M, 2, dtype=torch.float32, device=hsplit.device)
    check(lib.mftx_flow_head(_chk(hsplit, "hsplit"), hsplit.stride(0), P, h, w, _chk(wtile, "wtile", torch.uint8), _chk(b1, "b1"),
                             _chk(wproj, "wproj", torch.uint8), _chk(b2, "b2"), T.data_ptr(), delta.data_ptr(),
                             _chk(coords, "coords") if coords is not None else None, _stream()), "mftx_flow_head")
    return delta


def pack_flow_branch_weights(w98: torch.Tensor, w2pk: torch.Tensor, check_range: bool = True) -> torch.Tensor:
    """convf1's weight as [98 = (ky, kx, c), 128] and convf2's in the ``pack_conv_weight`` form [>= 64, 9, 128] -> the
    fragment streams of the fused flow-branch kernel (``mftx_pack_flow_branch_weights``; opaque bytes).  The weights are
    operands of the split arithmetic: with ``check_range`` a value not below 65504 raises ``SplitRangeError``."""
    lib = _lib.load()
    if tuple(w98.shape) != (98, 128) or w2pk.dim() != 3 or w2pk.shape[0] < 64 or tuple(w2pk.shape[1:]) != (9, 128):
        raise MftxError("pack_flow_branch_weights: expected convf1 as [98, 128] and convf2 as [>= 64, 9, 128]")
    if check_range:
        bad = count_not_below(w98, _lib.SPLIT_LIMIT) + count_not_below(w2pk, _lib.SPLIT_LIMIT)
        if bad:
            raise SplitRangeError(f"pack_flow_branch_weights: {bad} weights are not below {_lib.SPLIT_LIMIT} in magnitude")
    out = torch.empty(_lib.FLOW_BRANCH_WEIGHT_BYTES, dtype=torch.uint8, device=w98.device)
    check(lib.mftx_pack_flow_branch_weights(_chk(w98, "w98"), _chk(w2pk, "w2pk"), out.data_ptr(), _stream()),
          "mftx_pack_flow_branch_weights")
    return out


def flow_branch(coords: torch.Tensor, h: int, w: int, wflow: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor,
                out: torch.Tensor = None, hx: torch.Tensor = None) -> torch.Tensor:
    """relu(convf2(relu(convf1(coords - grid)))) without materialising convf1's 128 channels (``mftx_flow_branch``;
    split-fp16 arithmetic): coords [P, h*w, 2], wflow from ``pack_flow_branch_weights`` -> [P*h*w, 64] in SPLIT form
    (``unsplit_activations`` decodes it).  hx (optional): [P*h*w, 384] split-form rows whose channels 382, 383 receive
    the flow itself."""
    lib = _lib.load()
    P = coords.shape[0]
    if out is None:
        out = torch.empty(P * h * w, 64, dtype=torch.float32, device=coords.device)
    check(lib.mftx_flow_branch(_chk(coords, "coords"), P, h, w, _chk(wflow, "wflow", torch.uint8), _chk(b1, "b1"),
                               _chk(b2, "b2"), out.data_ptr(), out.stride(0),
                               hx.data_ptr() if hx is not None else None, hx.stride(0) if hx is not None else 0, _stream()),
          "mftx_flow_branch")
    return out


def corr_lookup_convc1(lv, coords: torch.Tensor, h: int, w: int, wfused: torch.Tensor, bias: torch.Tensor,
                       out_split: bool = False):
    """relu(convc1(lookup(coords)) + bias) without materialising the lookup (``mftx_corr_lookup_convc1``; split-fp16
    arithmetic): lv as ``corr_lookup``, coords [P, h*w, 2], wfused from ``pack_lookup_convc1_weights`` -> [P*h*w, 256]
    (fp32, or the split form when out_split)."""
    lib = _lib.load()
    P = coords.shape[0]
    stride, _ = pyramid_layout(h, w)
    for l, t in enumerate(lv):
        if tuple(t.shape) != (P, h * w, stride[l]):
            raise MftxError(f"corr_lookup_convc1: level {l} must be [{P}, {h * w}, {stride[l]}]")
    out = torch.empty(P * h * w, 256, dtype=torch.float32, device=coords.device)
    check(lib.mftx_corr_lookup_convc1(*[_chk(t, "level") for t in lv], _chk(coords, "coords"), P, h, w,
                                      _chk(wfused, "wfused", torch.uint8), _chk(bias, "bias"), out.data_ptr(), 256,
                                      int(bool(out_split)), _stream()), "mftx_corr_lookup_convc1")
    return out


def fmap_pyramid(f2: torch.Tensor, h: int, w: int):
    """f2 pixel-major [P, h*w, C] -> [f2, pooled level 1, 2, 3] ([P, h_l*w_l, C]); the feature pyramid of the
    on-demand correlation (AlternateCorrBlock, core/corr.py:78-82)."""
    lib = _lib.load()
    P, N, Cc = f2.shape
    assert N == h * w
    lv = [torch.empty(P, (h >> l) * (w >> l), Cc, dtype=torch.float32, device=f2.device) for l in (1, 2, 3)]
    check(lib.mftx_fmap_pyramid(_chk(f2, "f2"), P, Cc, h, w, *[t.data_ptr() for t in lv], _stream()), "mftx_fmap_pyramid")
    return [f2] + lv


def corr_lookup_ondemand(f1: torch.Tensor, f2_levels, coords: torch.Tensor, h: int, w: int, r: int = 4):
    """f1 [P, h*w, 256], f2_levels from ``fmap_pyramid``, coords [P, h*w, 2] -> [P, h*w, 324] (as ``corr_lookup``)."""
    lib = _lib.load()
    P, N, Cc = f1.shape
    out = torch.empty(P, N, 324, dtype=torch.float32, device=f1.device)
    check(lib.mftx_corr_lookup_ondemand(_chk(f1, "f1"), *[_chk(t, "f2 level") for t in f2_levels], _chk(coords, "coords"),
                                        P, Cc, h, w, r, out.data_ptr(), 324, _stream()), "mftx_corr_lookup_ondemand")
    return out


def conv2d(x: torch.Tensor, wpk: torch.Tensor, bias, P, h, w, N, kh, kw, act=None, out_scale=1.0, x2=None,
           addend=None, stride=0, hin=0, win=0, pad_y=0, pad_x=0, residual_mode=0, arith=ARITH_F32, a_split=False,
           out_split=False, out=None, tile=None):
    """x: pixel-major [P*h*w, C0] (optionally concatenated with x2 [P*h*w, C1]) ->
    [P*h*w, N].  arith = ARITH_SPLIT: wpk is the output of ``split_weights``.  tile: force the workgroup tile shape
    (``mftx_conv2d_tile``; tests and micro-benchmarks -- every shape gives the same bits)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(P * h * w, N, dtype=torch.float32, device=x.device)
    d = ConvDesc()
    d.a0, d.lda0, d.c0 = _chk(x, "x"), x.shape[1], x.shape[1]
    if x2 is not None:
        d.a1, d.lda1, d.c1 = _chk(x2, "x2"), x2.shape[1], x2.shape[1]
    else:
        d.a1, d.lda1, d.c1 = None, 0, 0
    d.wpk = _chk(wpk, "wpk")
    d.bias = _chk(bias, "bias") if bias is not None else None
    d.out, d.ldo = _chk(out, "out"), out.shape[1]
    d.P, d.h, d.w, d.N, d.kh, d.kw = P, h, w, N, kh, kw
    d.act, d.out_scale = ACT[act], out_scale
    d.addend, d.ld_addend = (_chk(addend, "addend"), addend.shape[1]) if addend is not None else (None, 0)
    d.stride, d.hin, d.win, d.pad_y, d.pad_x, d.residual_mode = stride, hin, win, pad_y, pad_x, residual_mode
    d.arith = arith
    d.a_split, d.out_split = int(bool(a_split)), int(bool(out_split))      # operands in split form (see split_activations)
    if tile is None:
        check(lib.mftx_conv2d(C.byref(d), _stream()), "mftx_conv2d")
    else:
        check(lib.mftx_conv2d_tile(C.byref(d), int(tile), _stream()), "mftx_conv2d_tile")
    return out


def convex_upsample(flow_lr, ou, mask, P, h, w, pads=(0, 0, 0, 0), want_packed=False):
    """flow_lr [M,2], ou [M,ld>=3], mask [M,576] -> flow [P,2,H0,W0], occl, sigma [P,1,H0,W0]
    (+ packed [P,H0,W0,4] = (fx, fy, occl, sigma) per pixel)."""
    lib = _lib.load()
    pl, pr, pt, pb = pads
    H0, W0 = 8 * h - pt - pb, 8 * w - pl - pr
    dev = flow_lr.device
    flow = torch.empty(P, 2, H0, W0, dtype=torch.float32, device=dev)
    occl = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev)
    sigma = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev)
    packed = torch.empty(P, H0, W0, 4, dtype=torch.float32, device=dev) if want_packed else None
    check(lib.mftx_convex_upsample(_chk(flow_lr, "flow_lr"), _chk(ou, "ou"), ou.shape[1], _chk(mask, "mask"),
                                   P, h, w, pl, pr, pt, pb, flow.data_ptr(), occl.data_ptr(), sigma.data_ptr(),
                                   packed.data_ptr() if want_packed else None, _stream()), "mftx_convex_upsample")
    return (flow, occl, sigma, packed) if want_packed else (flow, occl, sigma)


def _planes(res, H, W):
    flow, occl, sigma = res
    if flow.shape != (2, H, W) or occl.shape != (1, H, W) or sigma.shape != (1, H, W):
        raise MftxError("FlowOU planes must be [2,H,W], [1,H,W], [1,H,W]")
    return _chk(flow, "flow"), _chk(occl, "occlusion"), _chk(sigma, "sigma")


def _new_result(H, W, device):
    return (torch.empty(2, H, W, dtype=torch.float32, device=device),
            torch.empty(1, H, W, dtype=torch.float32, device=device),
            torch.empty(1, H, W, dtype=torch.float32, device=device))


def chain(L, R):
    """chain_results: L, R = (flow[2,H,W], occl[1,H,W], sigma[1,H,W]) -> same triple."""
    lib = _lib.load()
    _, H, W = L[0].shape
    out = _new_result(H, W, L[0].device)
    check(lib.mftx_chain(*_planes(L, H, W), *_planes(R, H, W), H, W, *[t.data_ptr() for t in out], _stream()),
          "mftx_chain")
    return out


def warp_backward(flow, img):
    """flow [2,H,W], img [C,H,W] -> img sampled at grid + flow, [C,H,W]."""
    lib = _lib.load()
    Cc, H, W = img.shape
    if flow.shape != (2, H, W):
        raise MftxError("warp_backward: flow must be [2,H,W] matching img")
    out = torch.empty_like(img)
    check(lib.mftx_warp_backward(_chk(flow, "flow"), _chk(img, "img"), Cc, H, W, out.data_ptr(), _stream()),
          "mftx_warp_backward")
    return out


def select(cands, thr, want_chosen=False):
    """cands: list of chained triples ordered [inf, 1, 2, ...]."""
    lib = _lib.load()
    K = len(cands)
    _, H, W = cands[0][0].shape
    cols = list(zip(*[_planes(c, H, W) for c in cands]))
    arrs = [_lib.ptr_array(list(c)) for c in cols]
    out = _new_result(H, W, cands[0][0].device)
    chosen = torch.empty(H, W, dtype=torch.int8, device=out[0].device) if want_chosen else None
    check(lib.mftx_select(K, arrs[0][0], arrs[1][0], arrs[2][0], float(thr), H, W, *[t.data_ptr() for t in out],
                          chosen.data_ptr() if want_chosen else None, _stream()), "mftx_select")
    return out + (chosen,)


def chain_select(Ls, Rs, thr, want_chosen=False):
    """Fused chain + select over K (L, R) pairs ordered [inf, 1, 2, ...]."""
    lib = _lib.load()
    K = len(Ls)
    assert len(Rs) == K
    _, H, W = Ls[0][0].shape
    lcols = list(zip(*[_planes(c, H, W) for c in Ls]))
    rcols = list(zip(*[_planes(c, H, W) for c in Rs]))
    arrs = [_lib.ptr_array(list(c)) for c in lcols + rcols]
    out = _new_result(H, W, Ls[0][0].device)
    chosen = torch.empty(H, W, dtype=torch.int8, device=out[0].device) if want_chosen else None
    check(lib.mftx_chain_select(K, *[a[0] for a in arrs], float(thr), H, W, *[t.data_ptr() for t in out],
                                chosen.data_ptr() if want_chosen else None, _stream()), "mftx_chain_select")
    return out + (chosen,)


def chain_select_packed(Ls, Rs, thr, want_chosen=False):
    """Fused chain + select with the right operands in the packed per-pixel format: Ls = K x (flow[2,H,W],
    occl[1,H,W], sigma[1,H,W]), Rs = K x [H,W,4] (fx, fy, occl, sigma)."""
    lib = _lib.load()
    K = len(Ls)
    assert len(Rs) == K
    _, H, W = Ls[0][0].shape
    lcols = list(zip(*[_planes(c, H, W) for c in Ls]))
    for r in Rs:
        if tuple(r.shape) != (H, W, 4):
            raise MftxError("packed FlowOU must be [H, W, 4]")
    arrs = [_lib.ptr_array(list(c)) for c in lcols] + [_lib.ptr_array([_chk(r, "packed") for r in Rs])]
    out = _new_result(H, W, Ls[0][0].device)
    chosen = torch.empty(H, W, dtype=torch.int8, device=out[0].device) if want_chosen else None
    check(lib.mftx_chain_select_packed(K, *[a[0] for a in arrs], float(thr), H, W, *[t.data_ptr() for t in out],
                                       chosen.data_ptr() if want_chosen else None, _stream()), "mftx_chain_select_packed")
    return out + (chosen,)


_quant_ws = {}


def quantize_u16(x):
    """One channel of a cache entry -> (uint16 tensor of x's shape, lohi float32[2] = (min, max)),
    both on the device (``compress_channel`` of MFT/utils/io.py:495-506).  No host sync."""
    lib = _lib.load()
    if isinstance(x, torch.Tensor) and x.is_cuda:
        x = x.contiguous()
        if x.data_ptr() % 16:        # a plane sliced out of a batched result: the kernels load 16 bytes per lane
            x = x.clone()
    _chk(x, "x")
    if x.numel() == 0:
        raise MftxError("quantize_u16: empty channel")
    ws = _quant_ws.get(x.device)
    if ws is None:
        ws = _quant_ws[x.device] = torch.empty(lib.mftx_quantize_workspace_bytes(), dtype=torch.uint8, device=x.device)
    q = torch.empty(x.shape, dtype=torch.uint16, device=x.device)
    lohi = torch.empty(2, dtype=torch.float32, device=x.device)
    check(lib.mftx_quantize_u16(x.data_ptr(), x.numel(), q.data_ptr(), lohi.data_ptr(), ws.data_ptr(), ws.numel(),
                                _stream()), "mftx_quantize_u16")
    return q, lohi


def dequantize_u16(q, lo, hi):
    """uint16 device tensor + the channel's (min, max) -> float32 (``decompress_channel``,
    MFT/utils/io.py:548-551)."""
    lib = _lib.load()
    if isinstance(q, torch.Tensor) and q.is_cuda:
        q = q.contiguous()
        if q.data_ptr() % 16:
            q = q.clone()
    _chk(q, "q", torch.uint16)
    x = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    check(lib.mftx_dequantize_u16(q.data_ptr(), q.numel(), float(lo), float(hi), x.data_ptr(), _stream()),
          "mftx_dequantize_u16")
    return x


class RaftEngine:
    """Handle on the native refinement runtime (``mftx_raft_*``)."""

    # WeightSlot indices (csrc/raft_engine.hip) of the weights that feed GEMM layers
    GEMM_SLOTS = (0, 2, 6, 8, 10, 11, 13, 14, 16, 17, 19, 20, 22, 26, 28, 30)

    OPTIONS = {"fork": 0, "presplit": 1, "group": 2, "fuse_lookup": 3, "graph": 4, "fuse_flow": 5, "tile_conv": 6, "fuse_head": 7, "tile_volume": 8, "fuse_gru": 9, "tile_cells": 10, "fuse_ou": 11, "tile_conv2p": 12}      # MFTX_RAFT_OPT_*
    # WeightSlot -> (N, cin) of the layers with a tile-resident kernel (csrc/tile_conv.hip): GRU gates (per-iteration and
    # context parts, both passes), flow head and mask head first layers
    # ... and (round 6) the motion encoder's convc2 (N = 192) and conv (126 of 128), 3 x 3 over 256 channels: two channel passes
    TILE_SLOTS = {2: (192, 256), 8: (128, 256), 10: (256, 256), 11: (256, 128), 13: (128, 256), 14: (128, 128), 16: (256, 256),
                  17: (256, 128), 19: (128, 256), 20: (128, 128), 22: (256, 128), 26: (256, 128)}

    def __init__(self, state_dict: dict, device, ondemand_corr=False, arith=ARITH_SPLIT, options=None):
        """options: {"fork" | "presplit" | "group" | "fuse_lookup" | "graph" | "fuse_flow" | "tile_conv" | "fuse_head" | "tile_volume": int} scheduling options of this handle
        (``mftx_raft_set_option``; defaults are the measured best)."""
        lib = _lib.load()
        self.device = torch.device(device)
        self.weights = pack_raft_weights(state_dict, self.device)   # keep alive: the engine holds raw pointers
        arr, self._keep = _lib.ptr_array([_chk(t, "weight") for t in self.weights])
        handle = C.c_void_p()
        check(lib.mftx_raft_create(arr, len(self.weights), C.byref(handle)), "mftx_raft_create")
        self._h = handle
        self._ws = None
        self.arith = int(arith)
        if self.arith == ARITH_SPLIT:          # split-fp16 products: the GEMM layers stream split weights
            self.split = [split_weights(t) if i in self.GEMM_SLOTS else None for i, t in enumerate(self.weights)]
            sarr, self._keep_split = _lib.ptr_array([t.data_ptr() if t is not None else None for t in self.split])
            check(lib.mftx_raft_set_split_weights(self._h, sarr, len(self.split)), "mftx_raft_set_split_weights")
            # lookup + convc1 as one kernel (the 324 features never leave the CU): convc1's weights as its fragment stream
            self.wfused = pack_lookup_convc1_weights(self.weights[0])
            check(lib.mftx_raft_set_lookup_fused(self._h, self.wfused.data_ptr()), "mftx_raft_set_lookup_fused")
            # the flow branch (convf1 -> convf2) as one kernel: both weights as its fragment streams
            self.wflow = pack_flow_branch_weights(self.weights[4], self.weights[6])
            check(lib.mftx_raft_set_flow_fused(self._h, self.wflow.data_ptr()), "mftx_raft_set_flow_fused")
            # the layers whose input tile fits a CU's LDS: weights as the streams of the tile-resident kernel
            self.wtile = [pack_tile_conv_weights(t, *self.TILE_SLOTS[i]) if i in self.TILE_SLOTS else None
                          for i, t in enumerate(self.weights)]
            tarr, self._keep_tile = _lib.ptr_array([t.data_ptr() if t is not None else None for t in self.wtile])
            check(lib.mftx_raft_set_tile_weights(self._h, tarr, len(self.wtile)), "mftx_raft_set_tile_weights")
            # ... and the flow head's last layer as the projection epilogue of its first (the 256 hidden channels stay in LDS)
            self.wproj = pack_flow_head_weights(self.weights[24])
            check(lib.mftx_raft_set_flow_head(self._h, self.wproj.data_ptr()), "mftx_raft_set_flow_head")
            # ... and the occlusion + uncertainty heads as one tile-resident kernel (five channel passes, projection epilogue)
            self.wou, self.wouproj = pack_ou_heads_weights(self.weights[30], self.weights[32])
            check(lib.mftx_raft_set_ou_heads(self._h, self.wou.data_ptr(), self.wouproj.data_ptr()), "mftx_raft_set_ou_heads")
        elif self.arith != ARITH_F32:
            raise MftxError(f"unknown arithmetic {arith!r}")
        # device-side count of non-finite output pixels, incremented by the last kernel of every refinement (no host sync;
        # read by nonfinite_count() when the caller synchronises anyway)
        # (16 bytes, the counter in word 0: the size mftx_copy_bytes moves, see nonfinite_snapshot)
        self._nonfinite = torch.zeros(4, dtype=torch.int32, device=self.device)
        check(lib.mftx_raft_set_nonfinite_counter(self._h, self._nonfinite.data_ptr()), "mftx_raft_set_nonfinite_counter")
        options = dict(options or {})
        self._gather = bool(options.pop("gather", 1))     # (Python-side: per-pair map lists instead of stacked batch tensors, A/B)
        for k, v in options.items():
            self.set_option(k, v)
        self.ondemand_corr = bool(ondemand_corr)
        if self.ondemand_corr:                 # raft_params.alternate_corr: no stored correlation volume
            check(lib.mftx_raft_set_ondemand(self._h, 1), "mftx_raft_set_ondemand")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mftx_raft_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def debug_refine(self, fmap1, fmap2, net, inp, h, w, iters, pads=(0, 0, 0, 0), flow_init=None):
        """``refine`` with the debug payload of ``RAFT.forward(vis_debug=True)`` (core/raft.py:159-176, 255-257):
        -> (flow, occl, sigma), {'costvolume_pyramid': 4 x [P*h*w, 1, h_l, w_l], 'coords_left': [P, 2, h, w],
        'iterations': (iters + 1) x {'coords': [P, 2, h, w]}}, everything on the CPU like the reference's."""
        lib = _lib.load()
        P = fmap1.shape[0]
        M = P * h * w
        trace = torch.empty(iters + 1, M, 2, dtype=torch.float32, device=self.device)
        check(lib.mftx_raft_set_coords_trace(self._h, trace.data_ptr()), "mftx_raft_set_coords_trace")
        try:
            out = self.refine(fmap1, fmap2, net, inp, h, w, iters, pads=pads, flow_init=flow_init)
        finally:
            check(lib.mftx_raft_set_coords_trace(self._h, None), "mftx_raft_set_coords_trace")
        if self.ondemand_corr:
            pyramid = None                      # (alternate_corr keeps no volume, as in the reference)
        else:
            stride, _ = pyramid_layout(h, w)
            pyramid = []
            for l in range(4):
                lvl = self.region(f"lvl{l}", P, h, w, stride[l]).reshape(P, h * w, stride[l])
                pyramid.append(unblock_level(lvl, l, h, w).reshape(M, 1, h >> l, w >> l).cpu())
        to_map = lambda t: t.reshape(P, h, w, 2).permute(0, 3, 1, 2).contiguous().cpu()      # noqa: E731
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        debug = {"costvolume_pyramid": pyramid,
                 "coords_left": torch.stack([xs, ys])[None].repeat(P, 1, 1, 1),
                 "iterations": [{"coords": to_map(trace[i])} for i in range(iters + 1)]}
        return out, debug

    def graph_stats(self):
        """(graphs captured, graph launches) of this handle (``mftx_raft_graph_stats``)."""
        c, r = C.c_ulonglong(), C.c_ulonglong()
        check(_lib.load().mftx_raft_graph_stats(self._h, C.byref(c), C.byref(r)), "mftx_raft_graph_stats")
        return int(c.value), int(r.value)

    def nonfinite_count(self, reset=False):
        """Output pixels with a non-finite flow / occlusion / sigma since the last reset (one 4-byte read: synchronises)."""
        n = int(self._nonfinite[0].item())
        if reset and n:
            self._nonfinite.zero_()
        return n

    def nonfinite_snapshot(self, host_words):
        """Enqueue, on the current stream, a copy of the counter into ``host_words`` (4 int32 of PINNED host memory, the count in
        word 0) -- no host wait: whoever later waits for an event recorded behind this call may read the word."""
        copy_bytes(self._nonfinite, host_words)

    def set_option(self, name, value):
        if name not in self.OPTIONS:
            raise MftxError(f"unknown engine option {name!r} (known: {', '.join(sorted(self.OPTIONS))}, gather)")
        check(_lib.load().mftx_raft_set_option(self._h, self.OPTIONS[name], int(value)), "mftx_raft_set_option")
        if name == "tile_conv":
            self._tile_conv = int(value)
        if name == "tile_volume":
            self._tile_volume = int(value)

    def workspace(self, P, h, w):
        need = _lib.load().mftx_raft_workspace_bytes_for(self._h, P, h, w)
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None:      # graphs captured on the old buffer would keep pointing into freed memory
                check(_lib.load().mftx_raft_clear_graphs(self._h), "mftx_raft_clear_graphs")
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    REGIONS = ("lvl0", "lvl1", "lvl2", "lvl3", "coords1", "corr", "cor1", "corflo", "flo1", "hx", "z", "rh", "fh",
               "delta", "mask", "ouin", "ouh", "ou", "flow_lr")

    #: regions stored in split form when the engine runs the split arithmetic (they feed GEMMs), and their row strides
    SPLIT_REGIONS = {"cor1": 256, "corflo": 256, "flo1": 128, "hx": 384, "rh": 128, "ouin": 712}

    def region(self, name, P, h, w, cols):
        """A workspace region as [P*h*w, cols] fp32 (parity tests): a view, or -- for the regions the split arithmetic
        keeps in split form -- the decoded values."""
        offs = (C.c_size_t * 19)()
        check(_lib.load().mftx_raft_workspace_layout_for(self._h, P, h, w, offs, 19), "mftx_raft_workspace_layout_for")
        off = offs[self.REGIONS.index(name)]
        M = P * h * w
        if self.arith == ARITH_SPLIT and name in self.SPLIT_REGIONS:
            ld = self.SPLIT_REGIONS[name]
            raw = self._ws[off: off + 4 * M * ld].view(torch.float32).reshape(M, ld)
            return unsplit_activations(raw)[:, :cols]
        return self._ws[off: off + 4 * M * cols].view(torch.float32).reshape(M, cols)

    MAX_GATHER = 16

    def can_gather(self, P):
        """``refine`` accepts LISTS of per-pair maps (no stacking into batch tensors) -- the split arithmetic with the
        stored, tile-resident correlation volume, up to 16 pairs."""
        return (self.arith == ARITH_SPLIT and not self.ondemand_corr and P <= self.MAX_GATHER and
                getattr(self, "_tile_volume", 1) != 0 and getattr(self, "_gather", True))

    def refine(self, fmap1, fmap2, net, inp, h, w, iters, pads=(0, 0, 0, 0), want_flow_lr=False, flow_init=None,
               packed=None, planar=True):
        """fmap1/fmap2 [P, h*w, 256], net/inp [P, h*w, 128] pixel-major ->
        flow [P,2,H0,W0], occl [P,1,H0,W0], sigma [P,1,H0,W0] (+ flow_lr [P,h*w,2]).
        flow_init: optional [P, h*w, 2] initial flow at 1/8 resolution (core/raft.py:153-154).
        packed: optional pre-allocated [P,H0,W0,4] that also receives (fx, fy, occl, sigma) per pixel;
        planar=False (with packed): only the packed result is written, flow = occl = sigma = None."""
        lib = _lib.load()
        gathered = isinstance(fmap1, (list, tuple))      # per-pair maps [h*w, 256] / [h*w, 128] (mftx_raft_refine_gather)
        P = len(fmap1) if gathered else fmap1.shape[0]
        pl, pr, pt, pb = pads
        H0, W0 = 8 * h - pt - pb, 8 * w - pl - pr
        dev = self.device
        if not planar and packed is None:
            raise MftxError("refine: planar=False needs a packed output")
        flow = torch.empty(P, 2, H0, W0, dtype=torch.float32, device=dev) if planar else None
        occl = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev) if planar else None
        sigma = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev) if planar else None
        flow_lr = torch.empty(P, h * w, 2, dtype=torch.float32, device=dev) if want_flow_lr else None
        if flow_init is not None and tuple(flow_init.shape) != (P, h * w, 2):
            raise MftxError("flow_init must be [P, h*w, 2]")
        if packed is not None and tuple(packed.shape) != (P, H0, W0, 4):
            raise MftxError("packed must be [P, H0, W0, 4]")
        ws = self.workspace(P, h, w)
        if gathered:
            if not self.can_gather(P) or not (len(fmap2) == len(net) == len(inp) == P):
                raise MftxError("refine: per-pair map lists need the split arithmetic with the tile-resident volume and P <= 16")
            arrs = [_lib.ptr_array([_chk(t, "map") for t in lst]) for lst in (fmap1, fmap2, net, inp)]
            check(lib.mftx_raft_refine_gather(self._h, P, h, w, iters, arrs[0][0], arrs[1][0], arrs[2][0], arrs[3][0],
                                              _chk(flow_init, "flow_init") if flow_init is not None else None,
                                              pl, pr, pt, pb,
                                              flow.data_ptr() if planar else None, occl.data_ptr() if planar else None,
                                              sigma.data_ptr() if planar else None,
                                              _chk(packed, "packed") if packed is not None else None,
                                              flow_lr.data_ptr() if want_flow_lr else None,
                                              ws.data_ptr(), ws.numel(), _stream()), "mftx_raft_refine_gather")
            return (flow, occl, sigma, flow_lr) if want_flow_lr else (flow, occl, sigma)
        check(lib.mftx_raft_refine(self._h, P, h, w, iters, _chk(fmap1, "fmap1"), _chk(fmap2, "fmap2"),
                                   _chk(net, "net"), _chk(inp, "inp"),
                                   _chk(flow_init, "flow_init") if flow_init is not None else None,
                                   pl, pr, pt, pb,
                                   flow.data_ptr() if planar else None, occl.data_ptr() if planar else None,
                                   sigma.data_ptr() if planar else None,
                                   _chk(packed, "packed") if packed is not None else None,
                                   flow_lr.data_ptr() if want_flow_lr else None,
                                   ws.data_ptr(), ws.numel(), _stream()), "mftx_raft_refine")
        return (flow, occl, sigma, flow_lr) if want_flow_lr else (flow, occl, sigma)
