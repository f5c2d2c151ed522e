"""Parity of every libmftx kernel (called through the C ABI) against the CPU
oracle and the reference-generated golden vectors.  Needs an MI355X."""
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi
from oracle import mft_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def pm(x):
    """[1,C,h,w] (cpu) -> pixel-major [h*w, C] on the GPU."""
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous().to(DEV)


def from_pm(x, h, w):
    """pixel-major [h*w, C] (gpu) -> [1,C,h,w] cpu."""
    return x.reshape(h, w, -1).permute(2, 0, 1)[None].cpu()


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def ops_mod():
    from mft_amd import ops
    return ops


@pytest.fixture(scope="module")
def gold(golden_dir):
    return dict(np.load(golden_dir / "ops.npz"))


@pytest.fixture(scope="module")
def inp():
    return {k: T(v) for k, v in gi.ops_inputs().items()}


def maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


# ---------------------------------------------------------------------------
# a4/a5/a6
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("arith", [0, 1])
def test_corr_pyramid_vs_golden(ops_mod, gold, inp, arith):
    """(arith 1: the split-fp16 volume the refinement engine builds by default, mftx_corr_pyramid_split)"""
    h, w = gi.OPS_H, gi.OPS_W
    lv = ops_mod.corr_pyramid(pm(inp["fmap1"])[None], pm(inp["fmap2"])[None], h, w, arith=arith)
    rows = gold["pyr_rows_idx"]
    for l, t in enumerate(lv):
        t = ops_mod.unblock_level(t, l, h, w)           # stored (blocked) layout -> the reference's row-major maps
        got = t[0].cpu()[rows].reshape(gold[f"pyr{l}_rows"].shape)
        assert maxerr(got, T(gold[f"pyr{l}_rows"])) < 3e-5, l
        cs = gi.checksum(t.cpu().numpy())
        assert np.allclose(cs, gold[f"pyr{l}_checksum"], rtol=1e-5, atol=1e-2)


def test_corr_pool_bit_exact_vs_oracle(ops_mod, inp):
    """Pooling uses ATen's summation order: levels 1..3 must equal avg-pooling
    the kernel's own level 0 on the CPU bit for bit."""
    h, w = gi.OPS_H, gi.OPS_W
    lv = ops_mod.corr_pyramid(pm(inp["fmap1"])[None], pm(inp["fmap2"])[None], h, w)
    lv = [ops_mod.unblock_level(t, l, h, w) for l, t in enumerate(lv)]
    v = lv[0][0].cpu().reshape(h * w, 1, h, w)
    for l in range(1, 4):
        v = F.avg_pool2d(v, 2, stride=2)
        assert torch.equal(v.reshape(h * w, -1), lv[l][0].cpu()), l


@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("h,w,P", [(17, 23, 2), (64, 64, 1), (20, 20, 1), (33, 50, 1)])
def test_corr_pyramid_fused_pooling_sizes(ops_mod, h, w, P, arith):
    """The volume GEMM's pooling epilogue at ragged sizes (partial super-blocks, floored levels, level strides
    that are not multiples of 4): level 0 against an fp64 product, levels 1..3 bit for bit against avg_pool2d
    of the level below; the padding of the blocked level 0 is zero."""
    g = torch.Generator().manual_seed(h * 100 + w)
    f1 = torch.randn(P, h * w, 256, generator=g).to(DEV)
    f2 = torch.randn(P, h * w, 256, generator=g).to(DEV)
    raw = ops_mod.corr_pyramid(f1, f2, h, w, arith=arith)
    lv = [ops_mod.unblock_level(t, l, h, w) for l, t in enumerate(raw)]
    ref0 = torch.einsum("pic,pjc->pij", f1.double(), f2.double()) / 16.0
    assert maxerr(lv[0], ref0) < (2e-4 if arith == 0 else 2e-5)       # (split products: closer to fp64 than fp32 MFMA chains)
    assert torch.equal(ops_mod.block_level(lv[0], 0, h, w), raw[0])          # padding cells are zero
    for p in range(P):
        v = lv[0][p].cpu().reshape(h * w, 1, h, w)
        for l in range(1, 4):
            v = F.avg_pool2d(v, 2, stride=2)
            assert torch.equal(v.reshape(h * w, -1), lv[l][p].cpu()), (p, l)


def test_corr_lookup_vs_golden(ops_mod, gold, inp):
    h, w = gi.OPS_H, gi.OPS_W
    lv = ops_mod.corr_pyramid(pm(inp["fmap1"])[None], pm(inp["fmap2"])[None], h, w)
    coords = pm(inp["coords1"])[None]
    out = ops_mod.corr_lookup(lv, coords, h, w)
    assert maxerr(from_pm(out[0], h, w), T(gold["lookup"])) < 5e-5


def test_corr_lookup_batched_and_odd_size(ops_mod):
    """P=3 pairs at an odd 1/8 grid (17x23: pyramid sizes floor), vs the oracle."""
    g = torch.Generator().manual_seed(3)
    h, w, P = 17, 23, 3
    f1 = torch.randn(P, 256, h, w, generator=g)
    f2 = torch.randn(P, 256, h, w, generator=g)
    coords = O.pixel_grid(h, w)[None] + 3 * torch.randn(P, 2, h, w, generator=g)
    a = torch.stack([pm(f1[i:i + 1]) for i in range(P)])
    b = torch.stack([pm(f2[i:i + 1]) for i in range(P)])
    lv = ops_mod.corr_pyramid(a, b, h, w)
    out = ops_mod.corr_lookup(lv, torch.stack([pm(coords[i:i + 1]) for i in range(P)]), h, w)
    for i in range(P):
        pyr = O.corr_pyramid(O.corr_volume(f1[i:i + 1], f2[i:i + 1]))
        for l in range(4):
            assert maxerr(ops_mod.unblock_level(lv[l], l, h, w)[i].cpu(), pyr[l].reshape(h * w, -1)) < 5e-5
        ref = O.corr_lookup(pyr, coords[i:i + 1])
        assert maxerr(from_pm(out[i], h, w), ref) < 1e-4


@pytest.mark.parametrize("h,w,P,spread", [(16, 24, 1, 3.0), (17, 23, 3, 6.0), (64, 64, 2, 40.0)])
def test_corr_lookup_ondemand_vs_materialised(ops_mod, h, w, P, spread):
    """a17 (AlternateCorrBlock / alt_cuda_corr): the on-demand lookup must give the materialised lookup's output --
    same layout, fp32-rounding apart -- and the pooled feature pyramid is avg_pool2d bit for bit."""
    g = torch.Generator().manual_seed(h + w)
    f1 = torch.randn(P, 256, h, w, generator=g)
    f2 = torch.randn(P, 256, h, w, generator=g)
    coords = O.pixel_grid(h, w)[None] + spread * torch.randn(P, 2, h, w, generator=g)
    coords[0, :, 0, 0] = torch.tensor([5.0, 3.0])                     # exactly integral
    a = torch.stack([pm(f1[i:i + 1]) for i in range(P)])
    b = torch.stack([pm(f2[i:i + 1]) for i in range(P)])
    cpm = torch.stack([pm(coords[i:i + 1]) for i in range(P)])
    f2lv = ops_mod.fmap_pyramid(b, h, w)
    ref = f2
    for l in range(1, 4):
        ref = F.avg_pool2d(ref, 2, stride=2)
        assert torch.equal(f2lv[l].cpu(), ref.permute(0, 2, 3, 1).reshape(P, -1, 256)), l
    od = ops_mod.corr_lookup_ondemand(a, f2lv, cpm, h, w)
    mat = ops_mod.corr_lookup(ops_mod.corr_pyramid(a, b, h, w), cpm, h, w)
    assert maxerr(od, mat) < 2e-4
    for i in range(min(P, 2)):
        want = O.corr_lookup_ondemand(f1[i:i + 1], f2[i:i + 1], coords[i:i + 1])
        assert maxerr(from_pm(od[i], h, w), want) < 2e-4


# ---------------------------------------------------------------------------
# a7-a9/a11: conv kernel
# ---------------------------------------------------------------------------

CONV_CASES = [
    # cin, cout, kh, kw, act, scale, P, h, w
    (324, 256, 1, 1, "relu", 1.0, 1, 16, 24),
    (256, 192, 3, 3, "relu", 1.0, 2, 16, 24),
    (256, 126, 3, 3, "relu", 1.0, 1, 17, 23),
    (384, 256, 1, 5, "sigmoid", 1.0, 1, 16, 24),
    (384, 128, 5, 1, "tanh", 1.0, 3, 16, 24),
    (256, 2, 3, 3, None, 1.0, 1, 16, 24),
    (256, 576, 1, 1, None, 0.25, 1, 16, 24),
    (712, 256, 3, 3, "relu", 1.0, 1, 16, 24),
    (128, 64, 3, 3, "relu", 1.0, 7, 32, 32),   # large-M tile path
]


@pytest.mark.parametrize("cin,cout,kh,kw,act,scale,P,h,w", CONV_CASES)
def test_conv2d_vs_torch(ops_mod, cin, cout, kh, kw, act, scale, P, h, w):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(P, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    ref = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref) * scale
    xp = x.permute(0, 2, 3, 1).reshape(P * h * w, cin).contiguous().to(DEV)
    out = ops_mod.conv2d(xp, ops_mod.pack_conv_weight(wt.to(DEV)), b.to(DEV), P, h, w, cout, kh, kw, act=act,
                         out_scale=scale)
    got = out.reshape(P, h, w, cout).permute(0, 3, 1, 2).cpu()
    assert maxerr(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))


# ---------------------------------------------------------------------------
# split-fp16 arithmetic of the conv GEMM (mftx_conv_desc.arith = MFTX_ARITH_SPLIT)
# ---------------------------------------------------------------------------

def test_split_weights_format(ops_mod):
    """mftx_split_weights: every 8 consecutive floats -> [8 fp16 high halves | 8 fp16 low halves x 2048], and
    hi + lo / 2048 reproduces the fp32 value to half an fp32 ulp -- also for magnitudes far from 1."""
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(128, 5 * 64, generator=g) * torch.exp(4 * torch.randn(128, 5 * 64, generator=g))).clamp(-6e4, 6e4).to(DEV)
    sp = ops_mod.split_weights(w)
    assert sp.shape == w.shape and sp.dtype == torch.float32
    halves = sp.view(torch.float16).reshape(128, -1, 2, 8)                 # per 8 k: hi x 8, lo x 8
    hi, lo = halves[:, :, 0].float(), halves[:, :, 1].float()
    w8 = w.reshape(128, -1, 8)
    assert torch.equal(hi, w8.half().float())                               # hi = round-to-nearest fp16
    assert torch.equal(lo, ((w8 - hi) * 2048).half().float())
    rec = hi.double() + lo.double() / 2048
    # half an fp32 ulp; below ~1e-7 the low half is itself a subnormal fp16 (spacing 2^-24 / 2048): absolute floor
    assert bool(((rec - w8.double()).abs() <= 2.0 ** -23 * w8.double().abs() + 2.0 ** -36).all())


@pytest.mark.parametrize("cin,cout,kh,kw,act,scale,P,h,w", [c for c in CONV_CASES if c[1] > 4])
def test_conv2d_split_arith_vs_fp64(ops_mod, cin, cout, kh, kw, act, scale, P, h, w):
    """The split arithmetic is an fp32-grade product: against an fp64 convolution its error is no larger than the
    fp32-MFMA kernel's on the same data (operands spread over several decades), and both meet the fp32 tolerance
    of test_conv2d_vs_torch."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(P, cin, h, w, generator=g) * torch.exp(torch.randn(P, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    fn = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act]
    ref = fn(F.conv2d(x.double(), wt.double(), b.double(), padding=(kh // 2, kw // 2))) * scale
    xp = x.permute(0, 2, 3, 1).reshape(P * h * w, cin).contiguous().to(DEV)
    wp = ops_mod.pack_conv_weight(wt.to(DEV))
    err = {}
    for name, ar, wk in (("fp32", ops_mod.ARITH_F32, wp), ("split", ops_mod.ARITH_SPLIT, ops_mod.split_weights(wp))):
        out = ops_mod.conv2d(xp, wk, b.to(DEV), P, h, w, cout, kh, kw, act=act, out_scale=scale, arith=ar)
        got = out.reshape(P, h, w, cout).permute(0, 3, 1, 2).cpu().double()
        err[name] = (got - ref).abs()
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float(err["split"].max()) < tol and float(err["fp32"].max()) < tol
    assert float(err["split"].pow(2).mean().sqrt()) <= 1.05 * float(err["fp32"].pow(2).mean().sqrt()) + 1e-12


@pytest.mark.parametrize("band", ["wide", "tiny", "near_limit"])
def test_conv2d_split_arith_operand_range(ops_mod, band):
    """The split arithmetic over the whole operand range it is specified for: magnitudes from 1e-8 up to the fp16 limit
    (wide), entirely inside the fp16-subnormal band |x| < 6.1e-5 (tiny: the halves lose bits there -- the ABSOLUTE error
    of an operand stays <= 2^-36), and within 1 % of 65504 (near_limit).  Error against an fp64 evaluation, relative to
    the sum of the products' magnitudes, next to the fp32-MFMA kernel's on the same data."""
    g = torch.Generator().manual_seed(17)
    P, h, w, cin, cout = 2, 24, 32, 256, 128
    M = P * h * w
    sign = lambda shape: torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0)     # noqa: E731
    if band == "wide":
        x = sign((M, cin)) * torch.exp(torch.empty(M, cin).uniform_(np.log(1e-8), np.log(6.0e4), generator=g))
        x[::7, ::5] = sign((len(x[::7]), len(x[0, ::5]))) * torch.empty(len(x[::7]), len(x[0, ::5])).uniform_(64850.0, 65503.0, generator=g)
    elif band == "tiny":
        x = sign((M, cin)) * torch.exp(torch.empty(M, cin).uniform_(np.log(1e-8), np.log(6.0e-5), generator=g))
    else:
        x = sign((M, cin)) * torch.empty(M, cin).uniform_(64850.0, 65503.0, generator=g)
    wt = sign((cout, cin, 3, 3)) * torch.exp(torch.empty(cout, cin, 3, 3).uniform_(np.log(1e-6), np.log(1e2), generator=g))
    bias = torch.zeros(cout)
    xd, wp = x.to(DEV).contiguous(), ops_mod.pack_conv_weight(wt.to(DEV))
    xi = x.reshape(P, h, w, cin).permute(0, 3, 1, 2).double()
    ref = F.conv2d(xi, wt.double(), padding=1).permute(0, 2, 3, 1).reshape(M, cout)
    mag = F.conv2d(xi.abs(), wt.double().abs(), padding=1).permute(0, 2, 3, 1).reshape(M, cout)
    err = {}
    for name, ar, wk in (("fp32", ops_mod.ARITH_F32, wp), ("split", ops_mod.ARITH_SPLIT, ops_mod.split_weights(wp))):
        y = ops_mod.conv2d(xd, wk, bias.to(DEV), P, h, w, cout, 3, 3, arith=ar).cpu().double()
        assert bool(torch.isfinite(y).all()), name
        err[name] = float(((y - ref).abs() / mag).max())
    # relative to sum |a b|: an fp32 dot product of 2304 terms is good to ~1e-6; the split products to the same grade
    assert err["split"] < (2e-5 if band == "tiny" else 1e-6), err
    if band != "tiny":
        assert err["split"] <= 2.0 * err["fp32"] + 1e-8, err
    # the same operands handed over in split form (what the engine stores): same bits as splitting in registers
    y_reg = ops_mod.conv2d(xd, ops_mod.split_weights(wp), bias.to(DEV), P, h, w, cout, 3, 3, arith=ops_mod.ARITH_SPLIT)
    y_pre = ops_mod.conv2d(ops_mod.split_activations(xd), ops_mod.split_weights(wp), bias.to(DEV), P, h, w, cout, 3, 3,
                           arith=ops_mod.ARITH_SPLIT, a_split=True)
    assert torch.equal(y_reg, y_pre)


def _flow_config(**raft_kw):
    from mft_amd.config import Config
    fc, rp = Config(), Config()
    fc.model, fc.synthetic_weights_seed, fc.flow_iters = None, 7, 2
    rp.occlusion_module, rp.small, rp.mixed_precision = "separate_with_uncertainty", False, False
    for k, v in raft_kw.items():
        setattr(rp, k, v)
    fc.raft_params = rp
    return fc


def test_split_weights_range_guard(ops_mod, weights_np):
    """A weight that is not below 65504 in magnitude has no finite fp16 high half: split_weights refuses it
    (SplitRangeError), RaftEngine with it, and the plugin falls back to the fp32 arithmetic with a logged reason."""
    from mft_amd.raft import RAFTWrapper
    w = torch.zeros(128, 1, 64, device=DEV)
    w[3, 0, 5] = 65503.0
    ops_mod.split_weights(w)                                  # the largest fp16 itself is fine
    assert ops_mod.count_not_below(w, 65504.0) == 0
    for bad in (65504.0, -7.0e4, float("inf"), float("nan")):
        w[3, 0, 5] = bad
        assert ops_mod.count_not_below(w, 65504.0) == 1
        with pytest.raises(ops_mod.SplitRangeError):
            ops_mod.split_weights(w)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights_np.items()}
    sd["update_block.gru.convz1.weight"][5, 7, 0, 2] = 1.0e5
    with pytest.raises(ops_mod.SplitRangeError):
        ops_mod.RaftEngine({k: v.to(DEV) for k, v in sd.items()}, DEV)
    fc = _flow_config()
    wrapper = RAFTWrapper(fc, state_dict=sd)
    assert wrapper.arith == "fp32"                            # refused the split arithmetic, kept working
    img = np.random.default_rng(0).integers(0, 255, (128, 192, 3), dtype=np.uint8)
    flow, extra = wrapper.compute_flow(img, img, mode="flow")
    assert flow.shape == (2, 128, 192)
    assert RAFTWrapper(fc, state_dict={k: torch.from_numpy(v) for k, v in weights_np.items()}).arith == "split"


def test_split_arith_out_of_range_is_nan(ops_mod, weights_np):
    """Activations are not clamped.  One that leaves the fp16 range (|x| >= 65520: hi = inf, residual = -inf) makes every
    product it takes part in NaN: the outputs of its receptive field are NaN -- never a finite wrong value -- and
    raft_params.check_finite turns that into an exception.  The fp32 arithmetic has no such limit."""
    from mft_amd.raft import RAFTWrapper
    g = torch.Generator().manual_seed(2)
    P, h, w = 1, 16, 24
    x = torch.randn(P * h * w, 128, generator=g)
    x[100, 17] = 7.0e4
    wt = ops_mod.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(DEV))
    y = ops_mod.conv2d(x.to(DEV), ops_mod.split_weights(wt), None, P, h, w, 128, 3, 3, arith=ops_mod.ARITH_SPLIT).cpu()
    y32 = ops_mod.conv2d(x.to(DEV), wt, None, P, h, w, 128, 3, 3).cpu()
    assert bool(torch.isfinite(y32).all())
    bad = ~torch.isfinite(y).all(1)
    yy, xx = 100 // w, 100 % w
    hood = torch.zeros(h, w, dtype=torch.bool)
    hood[max(yy - 1, 0):yy + 2, max(xx - 1, 0):xx + 2] = True
    assert torch.equal(bad.reshape(h, w), hood)               # exactly the 3 x 3 receptive field
    assert torch.allclose(y[~bad], y32[~bad], rtol=1e-4, atol=1e-4)
    # the engine: features scaled so that the correlation volume leaves the range (<f1, f2> / 16 ~ 1e6)
    sd = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    mk = _flow_config
    wrapper = RAFTWrapper(mk(), state_dict=sd)
    eng = wrapper.engine
    N = 16 * 24
    f = torch.full((1, N, 256), 300.0, device=DEV)
    net = torch.zeros(1, N, 128, device=DEV)
    flow, occl, sigma = eng.refine(f, f, net, net, 16, 24, 2)[:3]
    assert not bool(torch.isfinite(flow).any())               # NaN everywhere, not a finite wrong flow
    assert ops_mod.count_not_below(flow.reshape(-1), float("inf")) == flow.numel()
    checked = RAFTWrapper(mk(check_finite=True), state_dict=sd)
    from mft_amd.raft import FrameFeatures
    ff = FrameFeatures(f[0], net[0], net[0], 16, 24, (0, 0, 0, 0), (128, 192))
    checked._frames = {"a": ff, "b": ff}
    img = np.zeros((128, 192, 3), np.uint8)
    with pytest.raises(FloatingPointError, match="fp16 range"):
        checked.compute_pairs([("a", img, "b", img)])


@pytest.mark.parametrize("cin,cout,kh,kw,P,h,w,x2c", [(256, 192, 3, 3, 2, 16, 24, 0), (128, 128, 1, 5, 7, 32, 32, 128),
                                                    (328, 256, 1, 1, 1, 17, 23, 0), (256, 126, 3, 3, 3, 33, 47, 0),
                                                    (712, 256, 3, 3, 1, 16, 24, 0)])
def test_conv2d_split_form_operands(ops_mod, cin, cout, kh, kw, P, h, w, x2c):
    """Split-form operands (the engine's storage of GEMM inputs): a pre-split A gives the same bits as A split in
    registers, and a split-form output is the fp32 output's split, bit for bit -- the N = 126 tail leaves the rest
    of its 8-channel group alone."""
    g = torch.Generator().manual_seed(cin + cout)
    M = P * h * w
    x = (torch.randn(M, cin, generator=g) * torch.exp(torch.randn(M, cin, generator=g))).to(DEV)
    x2 = torch.randn(M, x2c, generator=g).to(DEV) if x2c else None
    wt = ops_mod.split_weights(ops_mod.pack_conv_weight((torch.randn(cout, cin + x2c, kh, kw, generator=g) * 0.05).to(DEV)))
    b = torch.randn(cout, generator=g).to(DEV)
    ref = ops_mod.conv2d(x, wt, b, P, h, w, cout, kh, kw, act="relu", arith=ops_mod.ARITH_SPLIT, x2=x2)
    xs = ops_mod.split_activations(x)
    assert torch.equal(ops_mod.unsplit_activations(xs), ops_mod.unsplit_activations(ops_mod.split_activations(ops_mod.unsplit_activations(xs))))
    got = ops_mod.conv2d(xs, wt, b, P, h, w, cout, kh, kw, act="relu", arith=ops_mod.ARITH_SPLIT, a_split=True,
                         x2=ops_mod.split_activations(x2) if x2c else None)
    assert torch.equal(got, ref)
    if cout % 8 == 0:
        outs = ops_mod.conv2d(xs, wt, b, P, h, w, cout, kh, kw, act="relu", arith=ops_mod.ARITH_SPLIT, a_split=True,
                              x2=ops_mod.split_activations(x2) if x2c else None, out_split=True)
        assert torch.equal(outs, ops_mod.split_activations(ref))
    else:                                   # N = 126 into a 128-wide split buffer: channels 126, 127 are not touched
        buf = ops_mod.split_activations(torch.full((M, 128), 7.0, device=DEV))
        ops_mod.conv2d(xs, wt, b, P, h, w, cout, kh, kw, act="relu", arith=ops_mod.ARITH_SPLIT, a_split=True,
                       out_split=True, out=buf)
        dec = ops_mod.unsplit_activations(buf)
        assert torch.equal(dec[:, :cout], ops_mod.unsplit_activations(ops_mod.split_activations(
            torch.cat([ref, torch.zeros(M, 2, device=DEV)], 1)))[:, :cout])
        assert bool((dec[:, cout:] == 7.0).all())


@pytest.mark.parametrize("N,ldo,act", [(126, 128, "relu"), (125, 128, None), (62, 64, "relu"), (254, 256, "tanh"), (190, 192, "relu")])
def test_conv2d_split_ragged_n_into_wider_rows(ops_mod, N, ldo, act):
    """N % 4 != 0 with 16-byte aligned output rows (the motion encoder's 126 channels next to the 2 flow channels): the
    straight-line epilogue stores the last column group value by value -- fp32 and split-form outputs, every tile
    shape the layer sizes select; the columns past N keep their contents."""
    g = torch.Generator().manual_seed(N)
    P, h, w, cin = 2, 40, 56, 64
    M = P * h * w
    x = torch.randn(M, cin, generator=g).to(DEV)
    wt = (torch.randn(N, cin, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    ref = F.conv2d(x.double().reshape(P, h, w, cin).permute(0, 3, 1, 2), wt.double(), b.double(), padding=1)
    ref = {"relu": torch.relu, "tanh": torch.tanh, None: lambda t: t}[act](ref).permute(0, 2, 3, 1).reshape(M, N)
    ws = ops_mod.split_weights(ops_mod.pack_conv_weight(wt))
    xs = ops_mod.split_activations(x)
    out = torch.full((M, ldo), 7.0, device=DEV)
    ops_mod.conv2d(xs, ws, b, P, h, w, N, 3, 3, act=act, arith=ops_mod.ARITH_SPLIT, a_split=True, out=out)
    assert maxerr(out[:, :N].double(), ref) < 2e-5
    assert bool((out[:, N:] == 7.0).all())
    if ldo % 8 == 0:
        outs = ops_mod.split_activations(torch.full((M, ldo), 7.0, device=DEV))
        ops_mod.conv2d(xs, ws, b, P, h, w, N, 3, 3, act=act, arith=ops_mod.ARITH_SPLIT, a_split=True, out_split=True, out=outs)
        dec = ops_mod.unsplit_activations(outs)
        assert torch.equal(dec[:, :N], ops_mod.unsplit_activations(ops_mod.split_activations(
            torch.cat([out[:, :N], torch.zeros(M, ldo - N, device=DEV)], 1)))[:, :N])
        assert bool((dec[:, N:] == 7.0).all())


@pytest.mark.parametrize("cin", [32, 64, 96, 128, 160])
@pytest.mark.parametrize("cout,P", [(256, 7), (128, 7), (192, 7), (64, 7), (256, 1)])
def test_conv2d_split_short_k(ops_mod, cin, cout, P):
    """One to five K chunks -- fewer than, as many as, and more than the chunks of the LDS ring (2 to 4 by tile shape): the
    peeled K loop's three parts in every combination, on the tile shapes the sizes select (128 x 256, 128 x 128 of eight
    waves, 128 x 192, 64 x 64), A split in registers and stored split."""
    g = torch.Generator().manual_seed(cin + cout + P)
    h = w = 64
    M = P * h * w
    x = torch.randn(M, cin, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    ref = x.double() @ wt.double().reshape(cout, cin).t() + b.double()
    ws = ops_mod.split_weights(ops_mod.pack_conv_weight(wt))
    y = ops_mod.conv2d(x, ws, b, P, h, w, cout, 1, 1, arith=ops_mod.ARITH_SPLIT)
    assert maxerr(y.double(), ref) < 1e-5
    y2 = ops_mod.conv2d(ops_mod.split_activations(x), ws, b, P, h, w, cout, 1, 1, arith=ops_mod.ARITH_SPLIT, a_split=True)
    assert torch.equal(y2, y)


def test_conv2d_split_arith_errors(ops_mod):
    from mft_amd._lib import MftxError
    x = torch.zeros(16 * 24, 256, device=DEV)
    w2 = ops_mod.pack_conv_weight(torch.zeros(2, 256, 3, 3, device=DEV))
    with pytest.raises(MftxError):                      # N <= 4 runs on the VALU kernel: fp32 weights only
        ops_mod.conv2d(x, ops_mod.split_weights(w2), None, 1, 16, 24, 2, 3, 3, arith=ops_mod.ARITH_SPLIT)
    with pytest.raises(MftxError):
        ops_mod.conv2d(x, ops_mod.pack_conv_weight(torch.zeros(64, 256, 3, 3, device=DEV)), None, 1, 16, 24, 64, 3, 3, arith=7)
    with pytest.raises(MftxError):
        ops_mod.split_weights(torch.zeros(3, 5, device=DEV))       # not a multiple of 8 floats


def test_conv2d_two_segments(ops_mod):
    g = torch.Generator().manual_seed(5)
    P, h, w = 2, 16, 24
    a = torch.randn(P, 128, h, w, generator=g)
    bb = torch.randn(P, 256, h, w, generator=g)
    wt = torch.randn(128, 384, 1, 5, generator=g) * 0.03
    bias = torch.randn(128, generator=g) * 0.1
    ref = torch.tanh(F.conv2d(torch.cat([a, bb], 1), wt, bias, padding=(0, 2)))
    ap = a.permute(0, 2, 3, 1).reshape(-1, 128).contiguous().to(DEV)
    bp = bb.permute(0, 2, 3, 1).reshape(-1, 256).contiguous().to(DEV)
    out = ops_mod.conv2d(ap, ops_mod.pack_conv_weight(wt.to(DEV)), bias.to(DEV), P, h, w, 128, 1, 5, act="tanh",
                         x2=bp)
    assert maxerr(out.reshape(P, h, w, 128).permute(0, 3, 1, 2).cpu(), ref) < 2e-5


def test_conv2d_argument_errors(ops_mod):
    from mft_amd._lib import MftxError
    x = torch.zeros(16 * 24, 130, device=DEV)
    wt = ops_mod.pack_conv_weight(torch.zeros(8, 130, 3, 3, device=DEV))
    with pytest.raises(MftxError):
        ops_mod.conv2d(x, wt, None, 1, 16, 24, 8, 3, 3)          # 130 channels: not a multiple of 4
    with pytest.raises(MftxError):
        ops_mod.conv2d(torch.zeros(16 * 24, 128), wt, None, 1, 16, 24, 8, 3, 3)   # CPU tensor


# ---------------------------------------------------------------------------
# a10 + a12
# ---------------------------------------------------------------------------

def _run_update_block(ops_mod, weights_np, inp_d, mode="f32"):
    """One engine iteration from the per-op fixture's (net, inp, corr, flow): not reachable through
    mftx_raft_refine (it computes corr itself), so the layers are chained through mftx_conv2d exactly
    as csrc/raft_engine.hip chains them.  mode: "f32" (fp32 MFMA), "split" (split-fp16 products, activations split in
    registers) or "presplit" (activations handed over and returned in split form wherever the engine stores them so:
    channel counts in whole groups of 8); the N <= 4 output layers run on the fp32 VALU kernel in every mode, as in
    the engine."""
    sd = {k: torch.from_numpy(v).to(DEV) for k, v in weights_np.items()}
    h, w = gi.OPS_H, gi.OPS_W

    def conv(x, name, k, act, x2=None, out_scale=1.0):
        wt = sd[name + ".weight"]
        kh, kw = wt.shape[2:]
        N = wt.shape[0]
        pk = ops_mod.pack_conv_weight(wt)
        bias = sd[name + ".bias"].contiguous()
        if mode == "f32" or N <= 4:
            return ops_mod.conv2d(x, pk, bias, 1, h, w, N, kh, kw, act=act, x2=x2, out_scale=out_scale)
        pk = ops_mod.split_weights(pk)
        groups = x.shape[1] % 8 == 0 and (x2 is None or x2.shape[1] % 8 == 0)
        if mode == "presplit" and groups:
            osp = N % 8 == 0
            y = ops_mod.conv2d(ops_mod.split_activations(x), pk, bias, 1, h, w, N, kh, kw, act=act, out_scale=out_scale,
                               x2=None if x2 is None else ops_mod.split_activations(x2), arith=ops_mod.ARITH_SPLIT,
                               a_split=True, out_split=osp)
            return ops_mod.unsplit_activations(y).contiguous() if osp else y
        return ops_mod.conv2d(x, pk, bias, 1, h, w, N, kh, kw, act=act, x2=x2, out_scale=out_scale, arith=ops_mod.ARITH_SPLIT)
    return sd, conv, h, w


@pytest.mark.parametrize("mode", ["f32", "split", "presplit"])
def test_update_block_and_ou_heads_vs_golden(ops_mod, gold, inp, weights_np, mode):
    """a7-a9 + a11 per op against the reference's own update block / occlusion block outputs
    (gold ub_* / ou_*): the layers run through mftx_conv2d in the engine's order, the gate algebra in torch -- in the fp32
    MFMA arithmetic and in the DEFAULT product arithmetic (split fp16), with fp32 and with split-form operands."""
    sd, conv, h, w = _run_update_block(ops_mod, weights_np, inp, mode)
    u, o = "update_block.", "occlusion_block."
    corr, flow, net, ctx = pm(inp["corr"]), pm(inp["flow"]), pm(inp["net"]), pm(inp["inp"])
    cor = conv(conv(corr, u + "encoder.convc1", 1, "relu"), u + "encoder.convc2", 3, "relu")
    # convf1 has 2 input channels: pad to 4 for the 16-byte rule (zero weights for the padding)
    wf1 = sd[u + "encoder.convf1.weight"]
    wf1p = torch.zeros(128, 4, 7, 7, device=DEV)
    wf1p[:, :2] = wf1
    flow4 = torch.cat([flow, torch.zeros_like(flow)], 1).contiguous()
    wf1pk = ops_mod.pack_conv_weight(wf1p)
    flo = ops_mod.conv2d(flow4, wf1pk if mode == "f32" else ops_mod.split_weights(wf1pk), sd[u + "encoder.convf1.bias"].contiguous(),
                         1, h, w, 128, 7, 7, act="relu", arith=ops_mod.ARITH_F32 if mode == "f32" else ops_mod.ARITH_SPLIT)
    flo = conv(flo, u + "encoder.convf2", 3, "relu")
    mot = conv(torch.cat([cor, flo], 1).contiguous(), u + "encoder.conv", 3, "relu")
    motion = torch.cat([mot, flow], 1).contiguous()
    assert maxerr(from_pm(motion, h, w), T(gold["ub_motion"])) < 2e-4
    x = torch.cat([ctx, motion], 1).contiguous()
    hcur = net
    for sfx in ("1", "2"):
        hx = torch.cat([hcur, x], 1).contiguous()
        z = conv(hx, u + f"gru.convz{sfx}", 5, "sigmoid")
        r = conv(hx, u + f"gru.convr{sfx}", 5, "sigmoid")
        q = conv(torch.cat([r * hcur, x], 1).contiguous(), u + f"gru.convq{sfx}", 5, "tanh")
        hcur = ((1 - z) * hcur + z * q).contiguous()
    assert maxerr(from_pm(hcur, h, w), T(gold["ub_net"])) < 2e-4
    delta = conv(conv(hcur, u + "flow_head.conv1", 3, "relu"), u + "flow_head.conv2", 3, None)
    assert maxerr(from_pm(delta, h, w), T(gold["ub_delta"])) < 2e-4
    mask = conv(conv(hcur, u + "mask.0", 3, "relu"), u + "mask.2", 1, None, out_scale=0.25)
    assert maxerr(from_pm(mask, h, w), T(gold["ub_mask"])) < 2e-4
    # a11: OU heads on the fixture's own inputs (core/update.py:196-214)
    ou_in = torch.cat([pm(inp["net"]), pm(inp["inp"]), pm(inp["corr"]), pm(inp["flow"]), pm(inp["delta_flow"]),
                       pm(inp["motion"])], 1).contiguous()
    assert ou_in.shape[1] == 712
    occl = conv(conv(ou_in, o + "occl_head.conv1", 3, "relu"), o + "occl_head.conv2", 3, None)
    unc = conv(conv(ou_in, o + "uncertainty_head.conv1", 3, "relu"), o + "uncertainty_head.conv2", 3, None)
    assert maxerr(from_pm(occl, h, w), T(gold["ou_occl"])) < 2e-4
    assert maxerr(from_pm(unc, h, w), T(gold["ou_unc"])) < 2e-4


def test_convex_upsample_vs_golden(ops_mod, gold, inp):
    h, w = gi.OPS_H, gi.OPS_W
    ou = torch.cat([pm(inp["occl_lr"]), pm(inp["unc_lr"]), torch.zeros(h * w, 1, device=DEV)], 1).contiguous()
    flow, occl, sigma = ops_mod.convex_upsample(pm(inp["flow"]), ou, pm(inp["mask"]), 1, h, w)
    assert maxerr(flow.cpu(), T(gold["up_flow"])) < 2e-5
    ref_occl = torch.softmax(T(gold["up_occl"]), dim=1)[:, 1:2]
    ref_sigma = torch.sqrt(torch.exp(T(gold["up_unc"])))
    assert maxerr(occl.cpu(), ref_occl) < 1e-5
    assert maxerr(sigma.cpu(), ref_sigma) < 1e-5 * float(ref_sigma.max())
    # cropped output == crop of the full output (InputPadder.unpad)
    f2, o2, s2 = ops_mod.convex_upsample(pm(inp["flow"]), ou, pm(inp["mask"]), 1, h, w, pads=(2, 3, 1, 2))
    assert torch.equal(f2, flow[..., 1:8 * h - 2, 2:8 * w - 3])
    assert torch.equal(o2, occl[..., 1:8 * h - 2, 2:8 * w - 3])


# ---------------------------------------------------------------------------
# a14 / a15
# ---------------------------------------------------------------------------

def dev3(t):
    return tuple(T(x).to(DEV) for x in t)


def test_chain_vs_golden(ops_mod, golden_dir):
    g = np.load(golden_dir / "sequence_stub.npz")
    flow, occ, sig = ops_mod.chain(dev3(gi.stub_flowou(0, 7)), dev3(gi.stub_flowou(7, 9)))
    assert maxerr(flow.cpu(), T(g["chain_flow"])) < 2e-5
    assert maxerr(occ.cpu(), T(g["chain_occl"])) < 1e-5
    assert maxerr(sig.cpu(), T(g["chain_sigma"])) < 1e-5


def test_warp_backward_and_chain_method(ops_mod):
    from mft_amd.results import FlowOUTrackingResult
    L = FlowOUTrackingResult(*dev3(gi.stub_flowou(0, 7)))
    R = dev3(gi.stub_flowou(7, 9))
    ref = O.chain(tuple(T(x) for x in gi.stub_flowou(0, 7)), tuple(T(x) for x in gi.stub_flowou(7, 9)))
    assert maxerr(L.chain(R[0]).cpu(), ref[0]) < 2e-5
    warped = L.warp_backward(torch.cat([R[1], R[2]], 0))
    occ = torch.maximum(L.occlusion, warped[0:1])
    assert maxerr(occ.cpu(), ref[1]) < 1e-5
    assert bool(L.invalid_mask().any())


def _candidates():
    pairs = [(0, 9), (8, 9), (7, 9), (5, 9), (1, 9), (3, 9), (6, 9)]
    H, W = gi.SEQ_H, gi.SEQ_W
    Ls, Rs = [], []
    for a, b in pairs:
        if a == 0:
            Ls.append((np.zeros((2, H, W), np.float32), np.zeros((1, H, W), np.float32), np.zeros((1, H, W), np.float32)))
        else:
            Ls.append(gi.stub_flowou(0, a))
        Rs.append(gi.stub_flowou(a, b))
    return Ls, Rs


def test_select_vs_oracle_and_fused_bitwise(ops_mod):
    Ls, Rs = _candidates()
    thr = 0.02
    dL, dR = [dev3(l) for l in Ls], [dev3(r) for r in Rs]
    chained = [ops_mod.chain(l, r) for l, r in zip(dL, dR)]
    f1, o1, s1, c1 = ops_mod.select(chained, thr, want_chosen=True)
    f2, o2, s2, c2 = ops_mod.chain_select(dL, dR, thr, want_chosen=True)
    # fused == chain -> select, bit for bit (the multi-GPU path relies on it)
    assert torch.equal(f1, f2) and torch.equal(o1, o2) and torch.equal(s1, s2) and torch.equal(c1, c2)
    # selection semantics: identical to the oracle given identical candidates
    of, oo, os_, oi = O.select([tuple(t.cpu() for t in c) for c in chained], thr)
    assert torch.equal(c1.cpu().long(), oi)
    assert torch.equal(f1.cpu(), of) and torch.equal(o1.cpu(), oo) and torch.equal(s1.cpu(), os_)
    assert len(torch.unique(c1)) >= 4                     # several candidates actually win
    assert bool((oo == 1).any())                          # out-of-image flows marked occluded
    # the chained candidates themselves agree with the oracle
    for c, l, r in zip(chained, Ls, Rs):
        ref = O.chain(tuple(T(x) for x in l), tuple(T(x) for x in r))
        assert maxerr(c[0].cpu(), ref[0]) < 3e-5 and maxerr(c[2].cpu(), ref[2]) < 1e-5


def _rand_results(K, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    Ls, Rs = [], []
    for k in range(K):
        Ls.append((torch.randn(2, H, W, generator=g).mul(3).to(DEV), torch.rand(1, H, W, generator=g).mul(0.05).to(DEV),
                   torch.rand(1, H, W, generator=g).add(0.1).to(DEV)))
        Rs.append((torch.randn(2, H, W, generator=g).mul(3).to(DEV), torch.rand(1, H, W, generator=g).mul(0.05).to(DEV),
                   torch.rand(1, H, W, generator=g).add(0.1).to(DEV)))
    return Ls, Rs


def test_chain_select_packed_and_vec4_bitwise(ops_mod):
    """Three forms of the fused chain + select must agree bit for bit: one pixel per thread (operands at
    4-byte-misaligned addresses force it), four pixels per thread with planar right operands, and four pixels per
    thread with PACKED right operands (the tracker's path); all against the oracle."""
    K, H, W = 5, 64, 96
    Ls, Rs = _rand_results(K, H, W, 3)
    thr = 0.02
    v4 = ops_mod.chain_select(Ls, Rs, thr, want_chosen=True)
    packed = [torch.cat([f, o, s_], 0).permute(1, 2, 0).contiguous() for f, o, s_ in Rs]
    pk = ops_mod.chain_select_packed(Ls, packed, thr, want_chosen=True)

    def misaligned(t):                       # same values, base address 4 bytes off a 16-byte boundary
        buf = torch.empty(t.numel() + 4, dtype=t.dtype, device=t.device)
        v = buf[1: 1 + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    Lm = [tuple(misaligned(t) for t in L) for L in Ls]
    sc = ops_mod.chain_select(Lm, Rs, thr, want_chosen=True)
    for a, b, c in zip(v4, pk, sc):
        assert torch.equal(a, b) and torch.equal(a, c)
    want = O.select([O.chain(tuple(t.cpu() for t in L), tuple(t.cpu() for t in R)) for L, R in zip(Ls, Rs)], thr)
    assert (v4[3].cpu().long() == want[3]).float().mean() > 0.999
    same = v4[3].cpu().long() == want[3]
    assert (v4[0].cpu() - want[0]).abs().max(0).values[same].max() < 2e-4
    # select alone: vec4 == scalar
    cands = [ops_mod.chain(L, R) for L, R in zip(Ls, Rs)]
    s4 = ops_mod.select(cands, thr, want_chosen=True)
    s1 = ops_mod.select([tuple(misaligned(t) for t in c) for c in cands], thr, want_chosen=True)
    for a, b, c in zip(s4, s1, v4):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_convex_upsample_packed_equals_planar(ops_mod, inp):
    h, w = gi.OPS_H, gi.OPS_W
    ou = torch.cat([pm(inp["occl_lr"]), pm(inp["unc_lr"]), torch.zeros(h * w, 1, device=DEV)], 1).contiguous()
    for pads in ((0, 0, 0, 0), (2, 3, 1, 2)):
        flow, occl, sigma, packed = ops_mod.convex_upsample(pm(inp["flow"]), ou, pm(inp["mask"]), 1, h, w, pads=pads,
                                                            want_packed=True)
        ref = torch.cat([flow[0], occl[0], sigma[0]], 0).permute(1, 2, 0)
        assert torch.equal(packed[0], ref)
        plain = ops_mod.convex_upsample(pm(inp["flow"]), ou, pm(inp["mask"]), 1, h, w, pads=pads)
        assert torch.equal(plain[0], flow) and torch.equal(plain[2], sigma)


def test_select_all_occluded_picks_first(ops_mod):
    H, W = 8, 16
    cands = []
    for k in range(3):
        cands.append((torch.full((2, H, W), float(k), device=DEV), torch.ones(1, H, W, device=DEV),
                      torch.full((1, H, W), 1.0 / (k + 1), device=DEV)))
    f, o, s, c = ops_mod.select(cands, 0.02, want_chosen=True)
    assert int(c.abs().sum()) == 0 and float(f.abs().sum()) == 0.0
    # ties keep the first index
    cands = [(torch.full((2, H, W), float(k), device=DEV), torch.zeros(1, H, W, device=DEV),
              torch.full((1, H, W), 0.5, device=DEV)) for k in range(3)]
    _, _, _, c = ops_mod.select(cands, 0.02, want_chosen=True)
    assert int(c.abs().sum()) == 0


def test_chain_select_arg_errors(ops_mod):
    from mft_amd._lib import MftxError
    H, W = 8, 16
    good = (torch.zeros(2, H, W, device=DEV), torch.zeros(1, H, W, device=DEV), torch.zeros(1, H, W, device=DEV))
    with pytest.raises(MftxError):
        ops_mod.chain_select([good] * 17, [good] * 17, 0.02)          # K > 16
    bad = (torch.zeros(2, H, W + 1, device=DEV), good[1], good[2])
    with pytest.raises(MftxError):
        ops_mod.chain(good, bad)


# ---------------------------------------------------------------------------
# full-size (512x512 -> 64x64 grid, BASELINE config 2) size-independent properties
# ---------------------------------------------------------------------------

def test_fullsize_properties(ops_mod):
    g = torch.Generator().manual_seed(11)
    h = w = 64
    f1 = torch.randn(2, h * w, 256, generator=g).to(DEV)
    f2 = torch.randn(2, h * w, 256, generator=g).to(DEV)
    lv = ops_mod.corr_pyramid(f1, f2, h, w)
    lvT = ops_mod.corr_pyramid(f2, f1, h, w)
    l0, l0T = ops_mod.unblock_level(lv[0], 0, h, w), ops_mod.unblock_level(lvT[0], 0, h, w)
    # V(f1,f2)[i][j] == V(f2,f1)[j][i] bitwise (same fp32 reduction order)
    assert torch.equal(l0[0], l0T[0].t())
    # spot-check level 0 against an fp64 dot product
    i = torch.tensor([0, 17, 4095]); j = torch.tensor([5, 2048, 4000])
    ref = (f1[1, i].double() * f2[1, j].double()).sum(1) / 16
    assert maxerr(l0[1][i, j].cpu(), ref.cpu()) < 2e-5
    # lookup at integer coordinates reads the volume itself (centre tap, all levels' level-0 part)
    grid = O.pixel_grid(h, w).permute(1, 2, 0).reshape(1, h * w, 2).repeat(2, 1, 1).contiguous().to(DEV)
    out = ops_mod.corr_lookup(lv, grid, h, w)
    centre = 4 * 9 + 4
    diag = torch.arange(h * w, device=DEV)
    assert maxerr(out[0][:, centre].cpu(), l0[0][diag, diag].cpu()) < 1e-6
    # chain with an identity left result returns the right result
    H = W = 512
    R = (torch.randn(2, H, W, generator=g).to(DEV), torch.rand(1, H, W, generator=g).to(DEV),
         torch.rand(1, H, W, generator=g).to(DEV))
    I = (torch.zeros(2, H, W, device=DEV), torch.zeros(1, H, W, device=DEV), torch.zeros(1, H, W, device=DEV))
    c = ops_mod.chain(I, R)
    # (white-noise R: the reference's normalise/un-normalise round trip moves the
    # sample point by ~3e-5 px, times a gradient of a few units per pixel)
    assert maxerr(c[0].cpu(), R[0].cpu()) < 1e-3 and maxerr(c[1].cpu(), R[1].cpu()) < 2e-4
    # convex upsampling of a constant field is that constant (x8 for flow) in the interior
    M = h * w
    const = torch.tensor([1.5, -2.0], device=DEV).repeat(M, 1).contiguous()
    ou = torch.zeros(M, 4, device=DEV)
    mask = torch.randn(M, 576, generator=g).to(DEV)
    flow, occl, sigma = ops_mod.convex_upsample(const, ou, mask, 1, h, w)
    assert maxerr(flow[0, 0, 8:-8, 8:-8].cpu(), torch.full((496, 496), 12.0)) < 1e-4
    assert maxerr(occl.cpu(), torch.full_like(occl.cpu(), 0.5)) < 1e-6


def test_conv2d_addend_splits_linear_conv(ops_mod):
    """conv([a | b]) == conv_a(a) + conv_b(b): the engine evaluates the `inp`
    part of the GRU gates once and feeds it back as an epilogue addend."""
    g = torch.Generator().manual_seed(21)
    P, h, w = 2, 16, 24
    a = torch.randn(P, 128, h, w, generator=g)
    b = torch.randn(P, 128, h, w, generator=g)
    wt = torch.randn(256, 256, 1, 5, generator=g) * 0.03
    bias = torch.randn(256, generator=g) * 0.1
    ref = torch.sigmoid(F.conv2d(torch.cat([a, b], 1), wt, bias, padding=(0, 2)))
    ap = a.permute(0, 2, 3, 1).reshape(-1, 128).contiguous().to(DEV)
    bp = b.permute(0, 2, 3, 1).reshape(-1, 128).contiguous().to(DEV)
    pre = ops_mod.conv2d(bp, ops_mod.pack_conv_weight(wt[:, 128:].contiguous().to(DEV)), bias.to(DEV), P, h, w, 256,
                         1, 5)
    out = ops_mod.conv2d(ap, ops_mod.pack_conv_weight(wt[:, :128].contiguous().to(DEV)), None, P, h, w, 256, 1, 5,
                         act="sigmoid", addend=pre)
    assert maxerr(out.reshape(P, h, w, 256).permute(0, 3, 1, 2).cpu(), ref) < 2e-5


@pytest.mark.parametrize("cin,cout,k,stride,h,w", [(64, 96, 3, 2, 32, 48), (64, 96, 1, 2, 32, 48), (96, 128, 3, 2, 17, 23)])
def test_conv2d_strided_vs_torch(ops_mod, cin, cout, k, stride, h, w):
    """Stride-2 convolutions of the encoders (core/extractor.py:6-62): input grid h x w,
    output grid floor((h + 2p - k)/2) + 1."""
    g = torch.Generator().manual_seed(31 + k)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.05
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, b, stride=stride, padding=k // 2)
    ho, wo = ref.shape[-2:]
    xp = x.permute(0, 2, 3, 1).reshape(h * w, cin).contiguous().to(DEV)
    out = ops_mod.conv2d(xp, ops_mod.pack_conv_weight(wt.to(DEV)), b.to(DEV), 1, ho, wo, cout, k, k, stride=stride,
                         hin=h, win=w)
    assert maxerr(out.reshape(1, ho, wo, cout).permute(0, 3, 1, 2).cpu(), ref) < 2e-5
    # residual epilogue: relu(relu(conv + b) + res)
    res = torch.randn(1, cout, ho, wo, generator=g)
    out = ops_mod.conv2d(xp, ops_mod.pack_conv_weight(wt.to(DEV)), b.to(DEV), 1, ho, wo, cout, k, k, act="relu",
                         stride=stride, hin=h, win=w, residual_mode=1,
                         addend=res.permute(0, 2, 3, 1).reshape(ho * wo, cout).contiguous().to(DEV))
    assert maxerr(out.reshape(1, ho, wo, cout).permute(0, 3, 1, 2).cpu(), torch.relu(torch.relu(ref) + res)) < 2e-5


@pytest.mark.parametrize("size", [(128, 192), (125, 187)])
def test_native_encoders_vs_oracle(ops_mod, gold, weights_np, weights_cpu, size):
    """fnet (instance norm) and cnet (folded batch norm) on the MFMA conv kernel vs the
    oracle's encoders; the 128x192 case is also pinned to the reference-generated golden."""
    from mft_amd.synth import SyntheticVideo
    H, W = size
    vid = SyntheticVideo(H, W, n_frames=4, seed=3)
    sd = {k: T(v).to(DEV) for k, v in weights_np.items()}
    fe = ops_mod.EncoderEngine(sd, "fnet", True, DEV)
    ce = ops_mod.EncoderEngine(sd, "cnet", False, DEV)
    img = T(vid[0]).to(DEV)
    fmap, _ = fe.forward(img)
    net, inp = ce.forward(img)
    x = O.normalise_image(O.preprocess(vid[0]))
    rf = O.encoder(x, weights_cpu, "fnet", "instance")
    rc = O.encoder(x, weights_cpu, "cnet", "batch")
    h, w = rf.shape[-2:]
    assert maxerr(from_pm(fmap, h, w), rf) < 1e-4
    assert maxerr(from_pm(net, h, w), torch.tanh(rc[:, :128])) < 1e-4
    assert maxerr(from_pm(inp, h, w), torch.relu(rc[:, 128:])) < 1e-4
    if size == (128, 192):
        assert maxerr(from_pm(fmap, h, w), T(gold["fnet"])) < 1e-4
        assert maxerr(torch.cat([from_pm(net, h, w), from_pm(inp, h, w)], 1),
                      torch.cat([torch.tanh(T(gold["cnet"])[:, :128]), torch.relu(T(gold["cnet"])[:, 128:])], 1)) < 1e-4


def test_raft_engine_argument_errors(ops_mod, weights_np):
    """Error behaviour of the engine entry points: negative codes surface as MftxError
    (the reference's native op raises RuntimeError through TORCH_CHECK)."""
    import ctypes as C
    from mft_amd import _lib
    from mft_amd._lib import MftxError
    lib = _lib.load()
    sd = {k: T(v).to(DEV) for k, v in weights_np.items()}
    eng = ops_mod.RaftEngine(sd, DEV)
    h = w = 16
    f = torch.zeros(1, h * w, 256, device=DEV)
    n = torch.zeros(1, h * w, 128, device=DEV)
    out = [torch.zeros(1, c, 8 * h, 8 * w, device=DEV) for c in (2, 1, 1)]
    ws = torch.zeros(1024, dtype=torch.uint8, device=DEV)            # far too small
    # raw C ABI: (handle, P, h, w, iters, fmap1, fmap2, net, inp, flow_init, 4 pads, flow, occl, sigma, packed, flow_lr, ws, bytes, stream)
    rc = lib.mftx_raft_refine(eng._h, 1, h, w, 2, f.data_ptr(), f.data_ptr(), n.data_ptr(), n.data_ptr(), None, 0, 0, 0, 0,
                              out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -3 and b"workspace" in lib.mftx_last_error_string()
    with pytest.raises(MftxError):
        eng.refine(f, f, n, n, 8, 8, 2)                               # grid too small for 4 pyramid levels
    with pytest.raises(MftxError):
        eng.refine(f, f, n, n, h, w, 0)                               # iters < 1
    with pytest.raises(MftxError):
        eng.refine(f, f, n, n, h, w, 2, pads=(4, 4, 0, 0))            # padding >= 8
    with pytest.raises(MftxError):
        eng.refine(f.cpu(), f, n, n, h, w, 2)                         # CPU tensor
    bad = C.c_void_p()
    assert lib.mftx_raft_create(None, 34, C.byref(bad)) == -1
    assert lib.mftx_raft_refine(None, 1, h, w, 2, f.data_ptr(), f.data_ptr(), n.data_ptr(), n.data_ptr(), None, 0, 0, 0, 0,
                                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None, None, ws.data_ptr(),
                                ws.numel(), None) == -4
    # split weights: every GEMM slot must be given; NULL switches back to fp32 MFMA
    sp, keep = _lib.ptr_array([t.data_ptr() if t is not None and i != 22 else None for i, t in enumerate(eng.split)])
    assert lib.mftx_raft_set_split_weights(eng._h, sp, 34) == -1 and lib.mftx_raft_arith(eng._h) == 1     # slot 22 (flow head) missing: refused, mode kept
    assert lib.mftx_raft_set_split_weights(eng._h, sp, 33) == -1
    assert lib.mftx_raft_set_split_weights(eng._h, None, 0) == 0 and lib.mftx_raft_arith(eng._h) == 0
    sp, keep = _lib.ptr_array([t.data_ptr() if t is not None else None for t in eng.split])
    assert lib.mftx_raft_set_split_weights(eng._h, sp, 34) == 0 and lib.mftx_raft_arith(eng._h) == 1
    # the happy path still works after the failures
    flow, occl, sigma = eng.refine(f, f, n, n, h, w, 2)
    assert bool(torch.isfinite(flow).all()) and float(occl.min()) >= 0 and float(sigma.min()) >= 0


# ---------------------------------------------------------------------------
# lookup fused into convc1 (mftx_corr_lookup_convc1): core/corr.py:30-51 + core/update.py:152-153
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("h,w,P,spread", [(16, 24, 1, 3.0), (17, 23, 3, 6.0), (33, 50, 2, 12.0), (64, 64, 7, 40.0), (46, 62, 5, 500.0)])
def test_lookup_convc1_fused_vs_separate(ops_mod, h, w, P, spread):
    """relu(convc1(lookup)) as ONE kernel -- the 324 features live in LDS only -- against the two kernels it replaces
    (mftx_corr_lookup -> mftx_conv2d in the split arithmetic) and against an fp64 evaluation of the same products:
    odd 1/8 grids (floor-sized pyramid levels), tiles of every size (4 .. 64 cells), ragged last tiles, windows that
    leave the level (zeros), wild coordinates."""
    g = torch.Generator().manual_seed(h * 100 + w)
    N = h * w
    f1 = torch.randn(P, N, 256, generator=g).to(DEV)
    f2 = (f1.cpu() + 0.5 * torch.randn(P, N, 256, generator=g)).to(DEV)
    lv = ops_mod.corr_pyramid(f1, f2, h, w, arith=ops_mod.ARITH_SPLIT)
    grid = torch.stack([pm(O.pixel_grid(h, w)[None])] * P)
    coords = (grid.cpu() + spread * torch.randn(P, N, 2, generator=g)).to(DEV).contiguous()
    wt = torch.randn(256, 324, 1, 1, generator=g) * 0.05
    bias = torch.randn(256, generator=g).to(DEV)
    wpk = ops_mod.pack_conv_weight(wt.to(DEV))
    feats = ops_mod.corr_lookup(lv, coords, h, w).reshape(P * N, 324)
    apart = ops_mod.conv2d(feats, ops_mod.split_weights(wpk), bias, P, h, w, 256, 1, 1, act="relu", arith=ops_mod.ARITH_SPLIT)
    wf = ops_mod.pack_lookup_convc1_weights(wpk)
    fused = ops_mod.corr_lookup_convc1(lv, coords, h, w, wf, bias)
    ref64 = torch.relu(feats.double().cpu() @ wt.reshape(256, 324).double().T + bias.double().cpu())
    scale = float(ref64.abs().max())
    e_fused, e_apart = maxerr(fused.cpu(), ref64) / scale, maxerr(apart.cpu(), ref64) / scale
    assert e_fused < 2e-6 and e_fused < 4 * e_apart + 1e-7, (e_fused, e_apart)
    # split-form output (what convc2 reads in the engine): the same values, bit for bit
    split = ops_mod.corr_lookup_convc1(lv, coords, h, w, wf, bias, out_split=True)
    enc = ops_mod.split_activations(fused)
    assert torch.equal(split.view(torch.int32), enc.view(torch.int32))


@pytest.mark.parametrize("h,w,P", [(24, 40, 5), (64, 64, 7)])
def test_lookup_convc1_fused_batch_and_tile_invariance(ops_mod, h, w, P):
    """A cell's row does not depend on the batch it is computed in (tile size, position in the tile, workgroup) -- nor on whether its
    tile is a workgroup's FIRST or a later one: 7 x 4096 cells are two tiles per workgroup, and round 6 had the conversion's
    bilinear blend contract differently in the prologue's and the loop's inlined copies (1-ulp differences in every second tile;
    the blend is spelled out in fused multiply-adds since: lf_blend4)."""
    g = torch.Generator().manual_seed(9)
    N = h * w
    f1 = torch.randn(P, N, 256, generator=g).to(DEV)
    f2 = torch.randn(P, N, 256, generator=g).to(DEV)
    lv = ops_mod.corr_pyramid(f1, f2, h, w, arith=ops_mod.ARITH_SPLIT)
    coords = (torch.stack([pm(O.pixel_grid(h, w)[None])] * P).cpu() + 5 * torch.randn(P, N, 2, generator=g)).to(DEV).contiguous()
    wpk = ops_mod.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(DEV))
    bias = torch.randn(256, generator=g).to(DEV)
    wf = ops_mod.pack_lookup_convc1_weights(wpk)
    whole = ops_mod.corr_lookup_convc1(lv, coords, h, w, wf, bias).reshape(P, N, 256)
    for i in range(P):
        one = ops_mod.corr_lookup_convc1([t[i:i + 1].contiguous() for t in lv], coords[i:i + 1].contiguous(), h, w, wf, bias)
        assert torch.equal(one.reshape(N, 256), whole[i]), i


def test_lookup_convc1_independent_of_stale_lds(ops_mod):
    """The fused kernel must not depend on what LDS held before it ran (it once did: a stale address-table entry of an
    unused cell could be a misaligned offset, a misaligned dword of finite floats can be a NaN, and a zero-weight dummy
    sample spread it over a row -- on the first launches of a fresh process only).  Here a GEMM over NaN operands leaves
    every CU's LDS full of NaNs right before each launch; tools/lf_stress.py does the same with an in-kernel poison."""
    g = torch.Generator().manual_seed(77)
    nan_x = torch.full((7 * 64 * 64, 256), float("nan"), device=DEV)
    nan_w = ops_mod.split_weights(ops_mod.pack_conv_weight(torch.zeros(256, 256, 3, 3, device=DEV)), check_range=False)
    for P, h, w in ((2, 33, 50), (1, 16, 24), (7, 64, 64)):
        N = h * w
        f1 = torch.randn(P, N, 256, generator=g).to(DEV)
        f2 = torch.randn(P, N, 256, generator=g).to(DEV)
        lv = ops_mod.corr_pyramid(f1, f2, h, w, arith=ops_mod.ARITH_SPLIT)
        coords = (torch.stack([pm(O.pixel_grid(h, w)[None])] * P).cpu() + 8 * torch.randn(P, N, 2, generator=g)).to(DEV).contiguous()
        wpk = ops_mod.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(DEV))
        bias = torch.randn(256, generator=g).to(DEV)
        wf = ops_mod.pack_lookup_convc1_weights(wpk)
        ref = None
        for rep in range(4):
            ops_mod.conv2d(nan_x, nan_w, None, 7, 64, 64, 256, 3, 3, arith=ops_mod.ARITH_SPLIT)      # NaNs into every LDS
            out = ops_mod.corr_lookup_convc1(lv, coords, h, w, wf, bias)
            assert bool(torch.isfinite(out).all())
            ref = out.clone() if ref is None else ref
            assert torch.equal(out, ref), (P, h, w, rep)


def test_lookup_convc1_argument_errors(ops_mod):
    h, w = 16, 16
    stride, _ = ops_mod.pyramid_layout(h, w)
    lv = [torch.zeros(1, h * w, stride[l], device=DEV) for l in range(4)]
    coords = torch.zeros(1, h * w, 2, device=DEV)
    wf = torch.zeros(393216, dtype=torch.uint8, device=DEV)
    bias = torch.zeros(256, device=DEV)
    with pytest.raises(ops_mod.MftxError):
        ops_mod.corr_lookup_convc1(lv[:3] + [lv[3][:, :, :1].contiguous()], coords, h, w, wf, bias)
    with pytest.raises(ops_mod.MftxError):
        ops_mod.pack_lookup_convc1_weights(torch.zeros(256, 1, 320, device=DEV))
    out = ops_mod.corr_lookup_convc1(lv, coords, h, w, wf, bias)
    assert float(out.abs().max()) == 0.0



def _engine_outputs(options, P=3, h=24, w=40, iters=4):
    """One refinement of the RAFT engine on seeded random features with the given per-handle options."""
    from mft_amd import ops
    from mft_amd.weights import make_weights
    sd = {k: torch.from_numpy(v).cuda() for k, v in make_weights(7).items()}
    eng = ops.RaftEngine(sd, "cuda", options=options)
    g = torch.Generator().manual_seed(11)
    f1 = torch.randn(P, h * w, 256, generator=g).cuda()
    f2 = (f1.cpu() + 0.3 * torch.randn(P, h * w, 256, generator=g)).cuda()
    net = torch.tanh(torch.randn(P, h * w, 128, generator=g)).cuda()
    inp = torch.relu(torch.randn(P, h * w, 128, generator=g)).cuda()
    flow, occl, sigma = eng.refine(f1, f2, net, inp, h, w, iters)[:3]
    return np.concatenate([t.cpu().numpy().ravel() for t in (flow, occl, sigma)])


def test_engine_register_split_bitwise():
    """The refinement engine with fp32 activations split in the GEMMs' registers (option presplit = 0: other kernels,
    other epilogue variants, GRU state in fp32 only) agrees with the default engine (activations stored in split form by
    their producers) to the last digits; with the flow branch in order on one stream (fork = 0) and with the grouped
    launches kept apart (group = 0) bit for bit.  Options are per handle (mftx_raft_set_option), not process state."""
    base = {"fuse_lookup": 0, "fuse_flow": 0, "tile_conv": 0, "fuse_head": 0}
    outs = {tag: _engine_outputs(dict(base, **extra)) for tag, extra in
            (("default", {}), ("nopresplit", {"presplit": 0}), ("nofork", {"fork": 0}), ("nogroup", {"group": 0, "fork": 0}))}
    assert np.isfinite(outs["default"]).all()
    assert np.array_equal(outs["nofork"], outs["default"])
    assert np.array_equal(outs["nogroup"], outs["default"])
    # (not bit for bit: the OU heads' gather reads h back from its split form, hi + lo / 2048, which re-splits to another
    # pair of halves in rare rounding ties -- differences of an ulp in their inputs)
    assert np.abs(outs["nopresplit"] - outs["default"]).max() < 2e-5


def test_engine_fused_lookup_matches_unfused():
    """The engine with lookup + convc1 as ONE kernel (the default) against the same engine with the two kept apart: the
    fused kernel sums convc1's 324 products in another order (level by level, 16 at a time), so the iterations drift
    apart by fp32 rounding only."""
    fused, apart = _engine_outputs({}), _engine_outputs({"fuse_lookup": 0})
    assert np.isfinite(fused).all()
    n = 3 * 2 * 192 * 320
    d = (fused[:n] - apart[:n]).reshape(3, 2, -1)
    epe = np.sqrt((d ** 2).sum(1)).mean()
    assert epe < 1e-4, epe
    assert np.abs(fused[n:] - apart[n:]).max() < 1e-3


def _flow_branch_case(ops_mod, P, h, w, spread, seed=3):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(128, 2, 7, 7, generator=g) * 0.1
    b1 = torch.randn(128, generator=g) * 0.1
    w2 = torch.randn(64, 128, 3, 3, generator=g) * 0.05
    b2 = torch.randn(64, generator=g) * 0.1
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xs, ys], -1).reshape(1, h * w, 2)
    coords = grid + spread * torch.randn(P, h * w, 2, generator=g)
    w98 = w1.permute(2, 3, 1, 0).reshape(98, 128).contiguous().cuda()
    wflow = ops_mod.pack_flow_branch_weights(w98, ops_mod.pack_conv_weight(w2.cuda()))
    return coords, grid, (w1, b1, w2, b2), wflow


@pytest.mark.parametrize("P,h,w,spread", [(1, 8, 16, 2.0), (2, 21, 37, 5.0), (1, 64, 64, 1.0), (3, 17, 16, 40.0), (1, 9, 1, 3.0)])
def test_flow_branch_fused_vs_fp64(ops_mod, P, h, w, spread):
    """convf1 -> convf2 as one kernel (mftx_flow_branch) against the two convolutions in fp64 (core/update.py:154-156):
    whole and ragged tiles, images smaller than a tile, the zero padding of BOTH layers (convf2 sees zeros, not convf1
    of the padding, outside the image), and the flow written to the tail of the GRU input."""
    coords, grid, (w1, b1, w2, b2), wflow = _flow_branch_case(ops_mod, P, h, w, spread)
    hx = torch.zeros(P * h * w, 384, device=DEV)
    out = ops_mod.flow_branch(coords.cuda(), h, w, wflow, b1.cuda(), b2.cuda(), hx=hx)
    got = ops_mod.unsplit_activations(out).cpu().double()
    flow = (coords - grid).double().reshape(P, h, w, 2).permute(0, 3, 1, 2)
    f1 = torch.relu(torch.nn.functional.conv2d(flow, w1.double(), b1.double(), padding=3))
    ref = torch.relu(torch.nn.functional.conv2d(f1, w2.double(), b2.double(), padding=1)).permute(0, 2, 3, 1).reshape(P * h * w, 64)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-6 * max(scale, 1.0), (float((got - ref).abs().max()), scale)
    want = torch.zeros(P * h * w, 384)
    want[:, 382:384] = (coords - grid).reshape(P * h * w, 2)
    assert torch.equal(hx.cpu(), ops_mod.split_activations(want.cuda()).cpu()), "flow tail of hx (and nothing else of it)"


def test_flow_branch_batch_invariance_and_stale_lds(ops_mod):
    """A pair's result does not depend on the batch it is part of, nor on what the workgroup's LDS held before."""
    h, w = 24, 40
    coords, grid, (w1, b1, w2, b2), wflow = _flow_branch_case(ops_mod, 3, h, w, 4.0)
    c = coords.cuda()
    full = ops_mod.flow_branch(c, h, w, wflow, b1.cuda(), b2.cuda())
    for k in range(3):
        one = ops_mod.flow_branch(c[k:k + 1].contiguous(), h, w, wflow, b1.cuda(), b2.cuda())
        assert torch.equal(one, full[k * h * w:(k + 1) * h * w]), k
    for _ in range(5):      # other kernels in between leave other bytes in LDS
        torch.randn(1 << 20, device=DEV).sort()
        assert torch.equal(ops_mod.flow_branch(c, h, w, wflow, b1.cuda(), b2.cuda()), full)


def test_flow_branch_out_of_range_is_nan_and_argument_errors(ops_mod):
    h, w = 16, 32
    coords, grid, (w1, b1, w2, b2), wflow = _flow_branch_case(ops_mod, 1, h, w, 1.0)
    c = coords.clone()
    c[0, 5 * w + 7, 0] += 1e5                                   # a flow beyond the fp16 range of the high halves
    out = ops_mod.unsplit_activations(ops_mod.flow_branch(c.cuda(), h, w, wflow, b1.cuda(), b2.cuda())).cpu().reshape(h, w, 64)
    ok = ops_mod.unsplit_activations(ops_mod.flow_branch(coords.cuda(), h, w, wflow, b1.cuda(), b2.cuda())).cpu().reshape(h, w, 64)
    bad = ~torch.isfinite(out).all(-1)
    assert bad[5, 7] and bad.sum() <= 9 * 9                       # NaN inside the receptive field (7 x 7, then 3 x 3) ...
    ys, xs = torch.nonzero(bad, as_tuple=True)
    assert int((ys - 5).abs().max()) <= 4 and int((xs - 7).abs().max()) <= 4
    assert torch.equal(out[~bad], ok[~bad])                       # ... and the same bits everywhere else: never a finite wrong value
    with pytest.raises(ops_mod.MftxError):
        ops_mod.pack_flow_branch_weights(torch.zeros(98, 64, device=DEV), torch.zeros(128, 9, 128, device=DEV))
    with pytest.raises(ops_mod.SplitRangeError):
        ops_mod.pack_flow_branch_weights(torch.full((98, 128), 7e4, device=DEV), torch.zeros(128, 9, 128, device=DEV))
    with pytest.raises(ops_mod.MftxError):                        # a split-form output needs 32-byte rows
        ops_mod.flow_branch(coords.cuda(), h, w, wflow, b1.cuda(), b2.cuda(), out=torch.zeros(h * w, 68, device=DEV))


def test_engine_fused_flow_matches_unfused():
    """The engine with the flow branch as ONE kernel (the default) against the same engine with convf1 (fp32 VALU) and
    convf2 (split GEMM) kept apart: other summation orders, split products where convf1 had fp32 ones -- fp32 rounding."""
    fused, apart = _engine_outputs({}), _engine_outputs({"fuse_flow": 0})
    assert np.isfinite(fused).all()
    n = 3 * 2 * 192 * 320
    d = (fused[:n] - apart[:n]).reshape(3, 2, -1)
    epe = np.sqrt((d ** 2).sum(1)).mean()
    assert epe < 1e-4, epe
    assert np.abs(fused[n:] - apart[n:]).max() < 1e-3


@pytest.mark.parametrize("cin,cout,kh,kw,P,h,w,act", [
    (128, 256, 3, 3, 2, 21, 37, "relu"), (128, 256, 3, 3, 1, 64, 64, "relu"), (128, 128, 3, 3, 1, 9, 5, None),
    (256, 256, 1, 5, 2, 13, 70, None), (256, 256, 5, 1, 1, 70, 13, "relu"), (256, 128, 1, 5, 1, 24, 40, None),
    (256, 128, 5, 1, 3, 40, 24, None), (128, 256, 1, 5, 1, 16, 33, None), (128, 128, 5, 1, 1, 33, 16, "relu")])
def test_tile_conv_vs_conv_gemm_and_fp64(ops_mod, cin, cout, kh, kw, P, h, w, act):
    """The tile-resident kernel (mftx_tile_conv2d) against the ring-buffered GEMM on the same split-form operands and
    against fp64: whole and ragged tiles, images smaller than a tile, both channel segments, bias + addend, fp32 and
    split-form outputs."""
    g = torch.Generator().manual_seed(cin + cout + kh)
    M = P * h * w
    x = torch.randn(M, cin, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    add = torch.randn(M, cout, generator=g)
    wpk = ops_mod.pack_conv_weight(wt.cuda())
    xs = ops_mod.split_activations(x.cuda())
    x1, x2 = (xs, None) if cin == 128 else (xs[:, :128].contiguous(), xs[:, 128:].contiguous())
    wtile = ops_mod.pack_tile_conv_weights(wpk, cout, cin)
    got = ops_mod.tile_conv2d(x1, wtile, b.cuda(), P, h, w, cout, kh, kw, act=act, x2=x2, addend=add.cuda())
    ref = ops_mod.conv2d(x1, ops_mod.split_weights(wpk), b.cuda(), P, h, w, cout, kh, kw, act=act, x2=x2, addend=add.cuda(),
                         arith=1, a_split=True)
    xi = x.double().reshape(P, h, w, cin).permute(0, 3, 1, 2)
    r64 = torch.nn.functional.conv2d(xi, wt.double(), b.double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(M, cout) + add.double()
    if act == "relu":
        r64 = torch.relu(r64)
    scale = float(r64.abs().max())
    e_tile, e_ring = float((got.cpu().double() - r64).abs().max()), float((ref.cpu().double() - r64).abs().max())
    assert e_tile < 3e-6 * scale and e_tile < 2 * e_ring + 1e-6 * scale, (e_tile, e_ring, scale)
    gs = ops_mod.tile_conv2d(x1, wtile, b.cuda(), P, h, w, cout, kh, kw, act=act, x2=x2, addend=add.cuda(), out_split=True)
    assert torch.equal(gs, ops_mod.split_activations(got))


def test_tile_conv_batch_invariance_and_argument_errors(ops_mod):
    g = torch.Generator().manual_seed(2)
    P, h, w = 3, 24, 40
    x = ops_mod.split_activations(torch.randn(P * h * w, 128, generator=g).cuda())
    wpk = ops_mod.pack_conv_weight((torch.randn(256, 128, 3, 3, generator=g) * 0.05).cuda())
    wtile = ops_mod.pack_tile_conv_weights(wpk, 256, 128)
    full = ops_mod.tile_conv2d(x, wtile, None, P, h, w, 256, 3, 3, act="relu")
    for k in range(P):
        one = ops_mod.tile_conv2d(x[k * h * w:(k + 1) * h * w].contiguous(), wtile, None, 1, h, w, 256, 3, 3, act="relu")
        assert torch.equal(one, full[k * h * w:(k + 1) * h * w]), k
    with pytest.raises(ops_mod.MftxError):          # 3 x 3 over 256 channels does not fit a CU's LDS: no kernel
        ops_mod.tile_conv2d(x, wtile, None, P, h, w, 256, 3, 3, x2=x)
    with pytest.raises(ops_mod.MftxError):
        ops_mod.tile_conv2d(x, wtile, None, P, h, w, 192, 3, 3)
    with pytest.raises(ops_mod.SplitRangeError):
        ops_mod.pack_tile_conv_weights(torch.full((256, 9, 128), 1e5, device=DEV), 256, 128)


@pytest.mark.parametrize("cout,P,h,w", [(192, 1, 24, 40), (126, 1, 24, 40), (192, 2, 33, 47), (126, 3, 9, 70), (192, 7, 64, 64), (126, 7, 64, 64)])
def test_tile_conv_two_pass_vs_conv_gemm_and_fp64(ops_mod, cout, P, h, w):
    """Round 6: 3 x 3 over 256 channels tile-resident in TWO channel passes (convc2 256 -> 192 and conv 256 -> 126 of the motion
    encoder, core/update.py:152-160; csrc/tile_conv.hip: tile_conv2p_kernel) against the ring-buffered GEMM on the same split-form
    operands and against fp64: whole and ragged tiles, images smaller than a tile, the K-split column tiles of N = 192, and conv's
    ragged last channel group -- channels 126, 127 of its 128-wide output rows are NOT written (the flow lives there)."""
    g = torch.Generator().manual_seed(cout + h)
    M = P * h * w
    x = torch.randn(M, 256, generator=g)
    wt = torch.randn(cout, 256, 3, 3, generator=g) * 0.04
    b = torch.randn(cout, generator=g)
    wpk = ops_mod.pack_conv_weight(wt.cuda())
    xs = ops_mod.split_activations(x.cuda())
    N = 192 if cout == 192 else 128
    wtile = ops_mod.pack_tile_conv_weights(wpk, N, 256)
    sentinel = ops_mod.split_activations(torch.full((M, N), 7.25, device=DEV))
    got_s = ops_mod.tile_conv2d(xs, wtile, b.cuda(), P, h, w, cout, 3, 3, act="relu", out_split=True, out=sentinel.clone())
    got = ops_mod.unsplit_activations(got_s)
    ref = ops_mod.unsplit_activations(ops_mod.conv2d(xs, ops_mod.split_weights(wpk), b.cuda(), P, h, w, cout, 3, 3, act="relu", arith=1, a_split=True,
                                                     out_split=True, out=sentinel.clone()))
    xi = x.double().reshape(P, h, w, 256).permute(0, 3, 1, 2)
    r64 = torch.relu(torch.nn.functional.conv2d(xi, wt.double(), b.double(), padding=1)).permute(0, 2, 3, 1).reshape(M, cout)
    scale = float(r64.abs().max())
    e_tile = float((got[:, :cout].cpu().double() - r64).abs().max())
    e_ring = float((ref[:, :cout].cpu().double() - r64).abs().max())
    assert e_tile < 3e-6 * scale and e_tile < 2 * e_ring + 1e-6 * scale, (e_tile, e_ring, scale)
    if cout < N:
        assert torch.equal(got_s.view(torch.int32).reshape(M, -1, 8)[:, -1, 3::4], sentinel.view(torch.int32).reshape(M, -1, 8)[:, -1, 3::4])   # channels 126, 127: both halves untouched
        assert bool((got[:, cout:] == 7.25).all())
    # a pair's bits do not depend on the batch (nor, with it, on the cells per tile the launcher picks)
    one = ops_mod.tile_conv2d(xs[:h * w].contiguous(), wtile, b.cuda(), 1, h, w, cout, 3, 3, act="relu", out_split=True, out=sentinel[:h * w].clone())
    assert torch.equal(one, got_s[:h * w])


def test_engine_two_pass_tile_conv_matches_ring_gemm():
    """The engine with convc2 and conv on the two-pass tile-resident kernel (the default where the tile-resident layers run) against
    the same engine with the two layers on the ring-buffered GEMM: fp32 rounding of their K sums."""
    two, ring = _engine_outputs({"tile_conv": 2}), _engine_outputs({"tile_conv": 2, "tile_conv2p": 0})
    assert np.isfinite(two).all() and not np.array_equal(two, ring)
    n = 3 * 2 * 192 * 320
    d = (two[:n] - ring[:n]).reshape(3, 2, -1)
    assert np.sqrt((d ** 2).sum(1)).mean() < 1e-4
    assert np.abs(two[n:] - ring[n:]).max() < 1e-3


def test_engine_tile_conv_matches_ring_gemm():
    """The engine with the GRU gates, their context parts and the flow / mask heads' first layers on the tile-resident
    kernel (the default) against the same engine with every layer on the ring-buffered GEMM: fp32 rounding of the K sums."""
    tiled, ring = _engine_outputs({"tile_conv": 2}), _engine_outputs({"tile_conv": 0})      # (2: whatever the batch -- this one is small)
    assert np.isfinite(tiled).all()
    n = 3 * 2 * 192 * 320
    d = (tiled[:n] - ring[:n]).reshape(3, 2, -1)
    epe = np.sqrt((d ** 2).sum(1)).mean()
    assert epe < 1e-4, epe
    assert np.abs(tiled[n:] - ring[n:]).max() < 1e-3


@pytest.mark.parametrize("P,h,w", [(3, 24, 40), (1, 72, 80), (1, 33, 140), (2, 135, 16), (1, 64, 64), (1, 135, 240)])
def test_engine_fused_gru_bitwise(P, h, w):
    """Each SepConvGRU pass as ONE kernel (mftx_gru_half: the tile loaded once, r * h formed in LDS in place, h ping-ponging
    between two buffers) against the two tile-resident launches it replaces: every output is the same sequence of products
    and sums -- the same bits.  Sizes: one tile across (nothing recomputed), maps wider / taller than a tile (R tiles that
    overlap by the gates' 2-cell halo, both tile shapes), ragged edges, and MORE TILES THAN CUs (135 x 240: workgroups of a
    second round start when their neighbours of the first have already written the new state -- the round-4 bug this size
    caught: the fp32 copy of h was updated in place)."""
    fused, apart = _engine_outputs({"tile_conv": 2}, P, h, w, 3), _engine_outputs({"tile_conv": 2, "fuse_gru": 0}, P, h, w, 3)
    assert np.isfinite(fused).all()
    assert np.array_equal(fused, apart)


@pytest.mark.parametrize("P,h,w", [(3, 24, 40), (1, 72, 80), (2, 33, 47), (1, 135, 240)])
def test_engine_tile_cells_bitwise(P, h, w):
    """The tile-resident kernels with 128, 64 or 32 cells per tile (round 4: the small-batch path -- one pair per rank of a
    sharded frame, ramp-up frames, 256 x 256 videos): a smaller tile is the same kernel with fewer MFMA row tiles per wave,
    every output the same sequence of products and sums.  Same bits, so the engine may pick the tile by the batch."""
    base = _engine_outputs({"tile_conv": 2, "tile_cells": 128}, P, h, w, 3)
    assert np.isfinite(base).all()
    for cells in (64, 32, 0):
        assert np.array_equal(_engine_outputs({"tile_conv": 2, "tile_cells": cells}, P, h, w, 3), base), cells
    assert np.array_equal(_engine_outputs({"tile_conv": 2, "tile_cells": 64, "fuse_gru": 0, "fuse_head": 0}, P, h, w, 3),
                          _engine_outputs({"tile_conv": 2, "tile_cells": 128, "fuse_gru": 0, "fuse_head": 0}, P, h, w, 3))


@pytest.mark.parametrize("P,h,w,vertical", [(1, 16, 24, False), (1, 16, 24, True), (2, 9, 150, False), (1, 150, 7, True), (1, 64, 64, False)])
def test_gru_half_vs_fp64(ops_mod, P, h, w, vertical):
    """mftx_gru_half against the gate algebra of core/update.py:108-123 in fp64: the z gate (from the scratch buffer), the new h (fp32 copy and split form)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5 + h + w)
    M = P * h * w
    hf = torch.tanh(torch.randn(M, 128, generator=g))
    mo = torch.relu(torch.randn(M, 128, generator=g))
    kh, kw = (5, 1) if vertical else (1, 5)
    wzr = torch.randn(256, 256, kh, kw, generator=g) * 0.04
    wq = torch.randn(128, 256, kh, kw, generator=g) * 0.04
    pre_zr = torch.randn(M, 256, generator=g) * 0.5
    pre_q = torch.randn(M, 128, generator=g) * 0.5
    pack = lambda wt, n: ops_mod.pack_tile_conv_weights(ops_mod.pack_conv_weight(wt.cuda()), n, 256)      # noqa: E731
    hf_dev = hf.cuda()
    hf_new, h_out, z = ops_mod.gru_half(ops_mod.split_activations(hf_dev), ops_mod.split_activations(mo.cuda()), pack(wzr, 256), pack(wq, 128),
                                pre_zr.cuda(), pre_q.cuda(), hf_dev, P, h, w, vertical=vertical)
    to_map = lambda t: t.double().reshape(P, h, w, -1).permute(0, 3, 1, 2)      # noqa: E731
    to_rows = lambda t: t.permute(0, 2, 3, 1).reshape(M, -1)                    # noqa: E731
    pad = (kh // 2, kw // 2)
    hd, md = to_map(hf), to_map(mo)
    zr = torch.sigmoid(F.conv2d(torch.cat([hd, md], 1), wzr.double(), padding=pad) + to_map(pre_zr))
    zz, rr = zr[:, :128], zr[:, 128:]
    q = torch.tanh(F.conv2d(torch.cat([rr * hd, md], 1), wq.double(), padding=pad) + to_map(pre_q))
    want = (1 - zz) * hd + zz * q
    # (z is the kernel's scratch: the z gate's sums BEFORE the context part and the sigmoid -- the blend applies both)
    assert (torch.sigmoid(z.cpu().double() + pre_zr[:, :128].double()) - to_rows(zz)).abs().max() < 2e-6
    assert (hf_new.cpu().double() - to_rows(want)).abs().max() < 5e-6
    assert torch.equal(hf_dev.cpu(), hf)                                        # the input state is left alone
    assert torch.equal(ops_mod.unsplit_activations(h_out), ops_mod.unsplit_activations(ops_mod.split_activations(hf_new)))
    with pytest.raises(ops_mod.MftxError):         # h is read from one buffer and written to another
        from mft_amd import _lib
        hs = ops_mod.split_activations(hf_dev)
        ops_mod.check(_lib.load().mftx_gru_half(hs.data_ptr(), 128, hs.data_ptr(), 128, pack(wzr, 256).data_ptr(), pack(wq, 128).data_ptr(),
                                                pre_zr.cuda().data_ptr(), pre_q.cuda().data_ptr(), z.data_ptr(), hf_dev.data_ptr(), hf_new.data_ptr(),
                                                hs.data_ptr(), 128, P, h, w, 0, None), "mftx_gru_half")


@pytest.mark.parametrize("P,h,w", [(1, 16, 24), (2, 21, 37), (1, 64, 64), (1, 5, 3)])
def test_ou_heads_fused_vs_fp64(ops_mod, P, h, w):
    """Both layers of the occlusion and uncertainty heads as one tile-resident kernel (five channel passes over the 712-channel
    input, projection epilogue) + the stencil sum (mftx_ou_heads) against the four convolutions in fp64 (core/update.py:177-214)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(17 + h)
    M = P * h * w
    x = torch.randn(M, 712, generator=g) * torch.exp(0.5 * torch.randn(M, 712, generator=g))
    wo1, wu1 = torch.randn(128, 712, 3, 3, generator=g) * 0.02, torch.randn(128, 712, 3, 3, generator=g) * 0.02
    bo1, bu1 = torch.randn(128, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
    wo2, wu2 = torch.randn(2, 128, 3, 3, generator=g) * 0.05, torch.randn(1, 128, 3, 3, generator=g) * 0.05
    bo2, bu2 = torch.randn(2, generator=g), torch.randn(1, generator=g)
    w1 = torch.cat([wo1, wu1], 0)
    w2 = torch.zeros(3, 256, 3, 3)
    w2[0:2, 0:128] = wo2
    w2[2:3, 128:256] = wu2
    wtile, wproj = ops_mod.pack_ou_heads_weights(ops_mod.pack_conv_weight(w1.cuda()), ops_mod.pack_conv_weight(w2.cuda()))
    got = ops_mod.ou_heads(ops_mod.split_activations(x.cuda()), h, w, wtile, torch.cat([bo1, bu1]).cuda(), wproj, torch.cat([bo2, bu2]).cuda())
    xm = x.double().reshape(P, h, w, 712).permute(0, 3, 1, 2)
    occ = F.conv2d(torch.relu(F.conv2d(xm, wo1.double(), bo1.double(), padding=1)), wo2.double(), bo2.double(), padding=1)
    unc = F.conv2d(torch.relu(F.conv2d(xm, wu1.double(), bu1.double(), padding=1)), wu2.double(), bu2.double(), padding=1)
    want = torch.cat([occ, unc], 1).permute(0, 2, 3, 1).reshape(M, 3)
    scale = max(float(want.abs().max()), 1.0)
    assert float((got[:, :3].cpu().double() - want).abs().max()) < 5e-6 * scale
    # ... and against the two launches it replaces (ring-buffered GEMM + small-N kernel): fp32 rounding of the K sums
    hid = ops_mod.conv2d(ops_mod.split_activations(x.cuda()), ops_mod.split_weights(ops_mod.pack_conv_weight(w1.cuda())), torch.cat([bo1, bu1]).cuda(),
                         P, h, w, 256, 3, 3, act="relu", arith=ops_mod.ARITH_SPLIT, a_split=True)
    two = ops_mod.conv2d(hid, ops_mod.pack_conv_weight(w2.cuda()), torch.cat([bo2, bu2]).cuda(), P, h, w, 3, 3, 3)
    assert float((got[:, :3] - two[:, :3]).abs().max()) < 5e-6 * scale


def test_engine_fused_ou_heads_matches_two_kernels():
    """The engine with the occlusion / uncertainty heads as one tile-resident kernel (the default where the tile-resident layers
    run) against the 712 -> 256 GEMM + small-N kernel: the flow is untouched (the heads come after the last update), occlusion and
    sigma agree to fp32 rounding."""
    fused, apart = _engine_outputs({"tile_conv": 2}), _engine_outputs({"tile_conv": 2, "fuse_ou": 0})
    assert np.isfinite(fused).all()
    n = 3 * 2 * 192 * 320
    assert np.array_equal(fused[:n], apart[:n])
    assert np.abs(fused[n:] - apart[n:]).max() < 1e-4
    cells = [_engine_outputs({"tile_conv": 2, "tile_cells": c}) for c in (64, 32)]
    assert np.array_equal(cells[0], fused) and np.array_equal(cells[1], fused)
    # the heads' 712-channel input gathered from its parts by the kernel's loader (the default) or materialised first (2).  Not bit for
    # bit: ou_gather decodes the split-form state (hi + lo / 2048) and splits it again, which lands on another pair of halves in rare
    # rounding ties; the loader copies the halves as they are -- an ulp in a few inputs
    for args in ((3, 24, 40, 4), (1, 33, 47, 2)):
        mat, gat = _engine_outputs({"tile_conv": 2, "fuse_ou": 2}, *args), _engine_outputs({"tile_conv": 2}, *args)
        nn = args[0] * 2 * 64 * args[1] * args[2]
        assert np.array_equal(mat[:nn], gat[:nn])            # the flow (flow_lr written by the loader) is the same
        assert np.abs(mat[nn:] - gat[nn:]).max() < 1e-5


@pytest.mark.parametrize("P,h,w", [(1, 8, 16), (2, 21, 37), (1, 64, 64), (1, 3, 5)])
def test_flow_head_fused_vs_fp64(ops_mod, P, h, w):
    """Both layers of the flow head as the tile-resident kernel with the projection epilogue + the stencil sum
    (mftx_flow_head) against the two convolutions in fp64 (core/update.py:6-14), and coords += delta."""
    g = torch.Generator().manual_seed(9)
    M = P * h * w
    x = torch.randn(M, 128, generator=g)
    w1, b1 = torch.randn(256, 128, 3, 3, generator=g) * 0.05, torch.randn(256, generator=g) * 0.1
    w2, b2 = torch.randn(2, 256, 3, 3, generator=g) * 0.05, torch.randn(2, generator=g) * 0.1
    coords = torch.randn(M, 2, generator=g)
    wtile = ops_mod.pack_tile_conv_weights(ops_mod.pack_conv_weight(w1.cuda()), 256, 128)
    wproj = ops_mod.pack_flow_head_weights(ops_mod.pack_conv_weight(w2.cuda()))
    cd = coords.cuda()
    delta = ops_mod.flow_head(ops_mod.split_activations(x.cuda()), h, w, wtile, b1.cuda(), wproj, b2.cuda(), coords=cd)
    xi = x.double().reshape(P, h, w, 128).permute(0, 3, 1, 2)
    f1 = torch.relu(torch.nn.functional.conv2d(xi, w1.double(), b1.double(), padding=1))
    ref = torch.nn.functional.conv2d(f1, w2.double(), b2.double(), padding=1).permute(0, 2, 3, 1).reshape(M, 2)
    scale = float(ref.abs().max())
    assert float((delta.cpu().double() - ref).abs().max()) < 3e-6 * max(scale, 1.0)
    assert torch.equal(cd.cpu(), coords + delta.cpu())


def test_engine_fused_flow_head_matches_two_kernels():
    """The engine with the flow head as one tile-resident kernel + stencil sum (the default) against the engine with its
    second layer as the small-N kernel on a materialised first layer: other summation order, fp32 rounding."""
    fused, apart = _engine_outputs({"tile_conv": 2}), _engine_outputs({"tile_conv": 2, "fuse_head": 0})
    assert np.isfinite(fused).all()
    n = 3 * 2 * 192 * 320
    d = (fused[:n] - apart[:n]).reshape(3, 2, -1)
    assert np.sqrt((d ** 2).sum(1)).mean() < 1e-4
    assert np.abs(fused[n:] - apart[n:]).max() < 1e-3


def test_engine_tile_resident_volume_bitwise():
    """The correlation volume by the tile-resident kernel (csrc/volume_tile.hip, the default) and by the ring-buffered
    GEMM: the same sequence of products and sums per output -- the engine's results are the same bits."""
    tiled, ring = _engine_outputs({}), _engine_outputs({"tile_volume": 0})
    assert np.isfinite(tiled).all()
    assert np.array_equal(tiled, ring)


def test_engine_deferred_flow_head_update_bitwise():
    """An iteration's coordinate update applied by the NEXT iteration's flow-branch kernel (into a second buffer; the
    default when flow branch and flow head are both fused) against the same update by a kernel of its own (fuse_head = 2):
    the same sums in the same order -- the same bits."""
    merged, own = _engine_outputs({"tile_conv": 2}), _engine_outputs({"tile_conv": 2, "fuse_head": 2})
    assert np.isfinite(merged).all()
    assert np.array_equal(merged, own)


def test_engine_refine_gather_bitwise():
    """mftx_raft_refine_gather -- the pairs' maps through per-pair pointers, a shared second map split once -- against
    mftx_raft_refine on the stacked maps: the same kernels on the same values, the same bits (shared and distinct right
    frames); and its refusals."""
    from mft_amd import ops
    from mft_amd.weights import make_weights
    sd = {k: torch.from_numpy(v).cuda() for k, v in make_weights(7).items()}
    eng = ops.RaftEngine(sd, "cuda")
    g = torch.Generator().manual_seed(11)
    P, h, w = 3, 24, 40
    f1 = [torch.randn(h * w, 256, generator=g).cuda() for _ in range(P)]
    f2 = [(f1[0].cpu() + 0.3 * torch.randn(h * w, 256, generator=g)).cuda() for _ in range(P)]
    net = [torch.tanh(torch.randn(h * w, 128, generator=g)).cuda() for _ in range(P)]
    inp = [torch.relu(torch.randn(h * w, 128, generator=g)).cuda() for _ in range(P)]
    for rights in (f2, [f2[0]] * P):
        a = eng.refine(torch.stack(f1), torch.stack(rights), torch.stack(net), torch.stack(inp), h, w, 3)
        b = eng.refine(f1, rights, net, inp, h, w, 3)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    eng.set_option("tile_volume", 0)
    assert not eng.can_gather(P)
    with pytest.raises(ops.MftxError):
        eng.refine(f1, f2, net, inp, h, w, 3)


def _tile_layers(arith, tile, ref=False):
    """A few conv GEMMs with the tile shape forced (mftx_conv2d_tile), outputs concatenated."""
    from mft_amd import ops
    g = torch.Generator().manual_seed(5)
    outs = []
    layers = [(256, 128, 1, 5, 1, 64, 64, "tanh"), (256, 126, 3, 3, 2, 33, 47, "relu"),
              (128, 64, 3, 3, 1, 64, 64, "relu"), (324, 256, 1, 1, 1, 17, 23, None)]
    if arith == 2:      # channel counts in whole 8-channel groups; N = 256 wide, two input segments, M not a multiple of any tile
        layers = [(256, 128, 1, 5, 1, 64, 64, "tanh"), (256, 126, 3, 3, 2, 33, 47, "relu"),
                  (384, 256, 5, 1, 2, 40, 56, "relu"), (328, 256, 1, 1, 1, 17, 23, None), (192, 384, 3, 3, 1, 24, 31, "relu")]
    for (cin, cout, kh, kw, P, h, w, act) in layers:
        x = torch.randn(P * h * w, cin, generator=g).cuda()
        wt = ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) * 0.05).cuda())
        b = torch.randn(cout, generator=g).cuda()
        if arith:
            wt = ops.split_weights(wt)
        if arith == 2:
            x2 = None
            if cin == 384:                                   # as the GRU gates: [h | x] from two tensors
                x, x2 = x[:, :128].contiguous(), x[:, 128:].contiguous()
            y = ops.conv2d(ops.split_activations(x), wt, b, P, h, w, cout, kh, kw, act=act, arith=1, a_split=True,
                           x2=None if x2 is None else ops.split_activations(x2), tile=tile)
            if ref:                                          # the same layer with A split in registers: same bits
                r = ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act=act, arith=1, x2=x2, tile=tile)
                assert torch.equal(y, r), ((cin, cout, kh, kw), float((y - r).abs().max()))
            outs.append(y.cpu().numpy().ravel())
            continue
        outs.append(ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act=act, arith=arith, tile=tile).cpu().numpy().ravel())
    return np.concatenate(outs)


def test_conv_tile_shapes_bitwise():
    """Every tile shape -- 128x128, 128x64, 64x64 on v_mfma_f32_32x32x2_f32 and the 32x32 tile of
    v_mfma_f32_16x16x4_f32 waves used for very small M x N -- gives the same bits: the reduction order
    is a property of the kernel, not of the tiling (the 1-vs-N GPU equality relies on it)."""
    outs = {tile: _tile_layers(0, tile) for tile in (0, 1, 2, 5)}
    for tile in (0, 1, 5):
        assert np.array_equal(outs[tile], outs[2]), tile
    # the same for the split-arithmetic kernels: 128x128 (four and eight waves), 64x64, 128x256, 128x192, rings of two to
    # four chunks
    outs = {tile: _tile_layers(1, tile) for tile in (0, 6, 9, 10, 14)}
    for tile in (6, 9, 10, 14):
        assert np.array_equal(outs[tile], outs[0]), tile
    # and for a pre-split A operand
    # ... including the warp-specialised 128 x 128 tile (11: consumer waves multiply, producer waves stage)
    outs = {tile: _tile_layers(2, tile, ref=(tile == 0)) for tile in (0, 6, 9, 10, 11, 14)}
    for tile in (6, 9, 10, 11, 14):
        assert np.array_equal(outs[tile], outs[0]), tile
    # the measurement-only shapes (224 x 128, 112 x 256 on 16-row MFMAs) are not part of the product build
    from mft_amd.ops import MftxError
    with pytest.raises(MftxError, match="measurement-only"):
        _tile_layers(2, 13)


def test_engine_graph_replay_bitwise():
    """mftx_raft_refine replays its workspace-only launch sequence as a hipGraph from the third call with a (shape,
    workspace) key on (option graph, default on): first call plain, second captured, later ones replayed -- same bits as
    plain launches every time, with fresh inputs each call (the graph holds addresses, not values)."""
    from mft_amd import ops
    from mft_amd.weights import make_weights
    sd = {k: torch.from_numpy(v).cuda() for k, v in make_weights(7).items()}
    plain, graphed = ops.RaftEngine(sd, "cuda", options={"graph": 0}), ops.RaftEngine(sd, "cuda")
    g = torch.Generator().manual_seed(21)
    P, h, w = 2, 24, 40
    for stream in (torch.cuda.Stream(), torch.cuda.default_stream()):
      # (the legacy default stream cannot be captured: the engine moves such calls onto a private stream, ordered by events)
      graphed = ops.RaftEngine(sd, "cuda")
      s = stream
      with torch.cuda.stream(s):
        for call in range(5):
            f1 = torch.randn(P, h * w, 256, generator=g).cuda()
            f2 = (f1.cpu() + 0.3 * torch.randn(P, h * w, 256, generator=g)).cuda()
            net = torch.tanh(torch.randn(P, h * w, 128, generator=g)).cuda()
            inp = torch.relu(torch.randn(P, h * w, 128, generator=g)).cuda()
            a = plain.refine(f1, f2, net, inp, h, w, 3)
            b = graphed.refine(f1, f2, net, inp, h, w, 3)
            for x, y in zip(a, b):
                assert torch.equal(x, y), call
        # another shape gets its own graph; the first one stays valid
        f = torch.randn(1, 16 * 24, 256, generator=g).cuda()
        n = torch.zeros(1, 16 * 24, 128).cuda()
        for call in range(3):
            assert all(torch.equal(x, y) for x, y in zip(plain.refine(f, f, n, n, 16, 24, 2), graphed.refine(f, f, n, n, 16, 24, 2)))
      s.synchronize()
      captures, replays = graphed.graph_stats()
      assert captures == 2 and replays == 4, (captures, replays)
    assert plain.graph_stats() == (0, 0)


def test_encoder_graph_replay_bitwise(weights_np):
    from mft_amd import ops
    sd = {k: torch.from_numpy(v).cuda() for k, v in weights_np.items()}
    rng = np.random.default_rng(3)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for prefix, inorm in (("fnet", True), ("cnet", False)):
            plain = ops.EncoderEngine(sd, prefix, inorm, "cuda", graph=False)
            graphed = ops.EncoderEngine(sd, prefix, inorm, "cuda")
            for call in range(4):
                img = torch.from_numpy(rng.integers(0, 255, (125, 187, 3), dtype=np.uint8)).cuda()
                a, b = plain.forward(img), graphed.forward(img)
                for x, y in zip(a, b):
                    assert (x is None and y is None) or torch.equal(x, y), (prefix, call)
    s.synchronize()


def test_round4_entry_points_argument_errors(ops_mod):
    """Error behaviour of the round-4 C-ABI entry points, called raw: null pointers, misaligned split-form rows, in-place state,
    bad sizes and unknown options are refused with a negative code and a message; nothing is launched."""
    from mft_amd import _lib
    lib = _lib.load()
    M, h, w = 16 * 16, 16, 16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, 128, generator=g).to(DEV)
    xs = ops_mod.split_activations(x)
    wz = ops_mod.pack_tile_conv_weights(ops_mod.pack_conv_weight((torch.randn(256, 256, 1, 5, generator=g) * 0.05).to(DEV)), 256, 256)
    wq = ops_mod.pack_tile_conv_weights(ops_mod.pack_conv_weight((torch.randn(128, 256, 1, 5, generator=g) * 0.05).to(DEV)), 128, 256)
    pz, pq = torch.zeros(M, 256, device=DEV), torch.zeros(M, 128, device=DEV)
    z, hf2, ho = torch.empty(M, 128, device=DEV), torch.empty(M, 128, device=DEV), torch.empty(M, 128, device=DEV)
    args = [xs.data_ptr(), 128, xs.data_ptr(), 128, wz.data_ptr(), wq.data_ptr(), pz.data_ptr(), pq.data_ptr(), z.data_ptr(), x.data_ptr(),
            hf2.data_ptr(), ho.data_ptr(), 128, 1, h, w, 0, None]
    assert lib.mftx_gru_half(*args) == 0
    for i, bad, code in ((0, None, -1), (4, None, -1), (9, hf2.data_ptr(), -1), (0, xs.data_ptr() + 16, -2), (1, 130, -2), (13, 0, -1), (16, 2, -1)):
        a = list(args)
        a[i] = bad
        assert lib.mftx_gru_half(*a) == code, (i, lib.mftx_last_error_string())
    # mftx_ou_heads
    a712 = ops_mod.split_activations(torch.randn(M, 712, generator=g).to(DEV))
    w1 = ops_mod.pack_conv_weight((torch.randn(256, 712, 3, 3, generator=g) * 0.02).to(DEV))
    w2 = ops_mod.pack_conv_weight((torch.randn(3, 256, 3, 3, generator=g) * 0.05).to(DEV))
    wt, wp = ops_mod.pack_ou_heads_weights(w1, w2)
    b1, b2 = torch.zeros(256, device=DEV), torch.zeros(3, device=DEV)
    T_, out = torch.empty(M, 27, device=DEV), torch.zeros(M, 4, device=DEV)
    ou = [a712.data_ptr(), 712, 1, h, w, wt.data_ptr(), b1.data_ptr(), wp.data_ptr(), b2.data_ptr(), T_.data_ptr(), out.data_ptr(), 4, None]
    assert lib.mftx_ou_heads(*ou) == 0
    for i, bad, code in ((0, None, -1), (5, None, -1), (11, 2, -1), (1, 704, -2), (0, a712.data_ptr() + 8, -2), (2, 0, -1)):
        a = list(ou)
        a[i] = bad
        assert lib.mftx_ou_heads(*a) == code, (i, lib.mftx_last_error_string())
    with pytest.raises(ops_mod.MftxError):
        ops_mod.pack_ou_heads_weights(w1[:, :, :704].contiguous(), w2)
    assert lib.mftx_pack_ou_heads_weights(w1.data_ptr(), 700, w2.data_ptr(), wt.data_ptr(), wp.data_ptr(), None) == -1
    # engine options and setters
    from mft_amd.weights import make_weights
    eng = ops_mod.RaftEngine({k: torch.from_numpy(v).to(DEV) for k, v in make_weights(3).items()}, DEV)
    with pytest.raises(ops_mod.MftxError, match="unknown engine option"):
        eng.set_option("fuse_everything", 1)
    assert lib.mftx_raft_set_option(eng._h, 99, 1) == -1
    assert lib.mftx_raft_set_ou_heads(eng._h, wt.data_ptr(), None) == -1
    assert lib.mftx_raft_set_nonfinite_counter(eng._h, z.data_ptr() + 2) == -2
    assert lib.mftx_raft_clear_graphs(None) == -4 and lib.mftx_raft_clear_graphs(eng._h) == 0
    assert lib.mftx_tile_conv_fills_chip(0, 64, 64) == 0 and lib.mftx_tile_conv_fills_chip(7, 64, 64) == 1


@pytest.mark.timeout(600)
def test_lookup_convc1_wide_variant():
    """Round 6: the fused lookup that gathers in 16-byte pieces (csrc/lookup_convc1_wide.hip -> mft_amd/libmftx_lfwide.so, built by
    __graft_entry__.build(); NOT the default: profiles/r6d_lookup_wide_gather.txt) stays parity-green: the fused-lookup tests -- vs the
    two kernels it replaces and fp64 at odd grids / ragged tiles / wild coordinates (levels with a width that is not a multiple of 4
    take its dword path), batch and tile invariance, stale LDS, the engine with it against the engine without, update-block goldens,
    one compute_flow golden -- in a process that loads the variant library."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    lib = repo / "mft_amd" / "libmftx_lfwide.so"
    assert lib.exists(), "run __graft_entry__.build() (make -C mft_amd/csrc lfwide)"
    env = dict(os.environ, MFTX_LIB=str(lib))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          str(repo / "tests" / "test_gpu_kernels.py"), str(repo / "tests" / "test_gpu_e2e.py"),
                          "-k", "(lookup_convc1 and not wide_variant) or engine_fused_lookup or update_block_and_ou_heads or compute_flow_vs_golden"],
                         capture_output=True, text=True, env=env, timeout=550)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1500:]
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 12, out.stdout[-500:]
