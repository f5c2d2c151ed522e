"""Randomised parity of the HBM-bound kernels against the oracle over ragged shapes and awkward
values (hypothesis, fixed seed database off): chain / select / chain_select, the correlation lookup and
the convex upsampler.  Sizes are kept small -- the point is shape and value coverage (odd sizes, flows
that leave the frame, ties, everything occluded, thresholds hit exactly), not throughput."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import mft_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
SET = settings(max_examples=25, deadline=None, derandomize=True,
               suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def rand_result(rng, H, W, spread):
    flow = (rng.standard_normal((2, H, W)) * spread).astype(np.float32)
    occl = rng.choice([0.0, 0.01, 0.02, 0.020000001, 0.5, 1.0], size=(1, H, W)).astype(np.float32)
    sigma = rng.choice([0.25, 0.5, 0.5, 1.0, 3.0], size=(1, H, W)).astype(np.float32)     # ties on purpose
    return flow, occl, sigma


def dev3(t):
    return tuple(torch.from_numpy(x).to(DEV) for x in t)


def cpu3(t):
    return tuple(torch.from_numpy(x) for x in t)


@SET
@given(H=st.integers(2, 41), W=st.integers(2, 53), K=st.integers(1, 7), seed=st.integers(0, 10_000),
       spread=st.sampled_from([0.3, 3.0, 40.0]))
def test_chain_select_random(H, W, K, seed, spread):
    from mft_amd import ops
    rng = np.random.default_rng(seed)
    Ls = [rand_result(rng, H, W, spread) for _ in range(K)]
    Rs = [rand_result(rng, H, W, spread) for _ in range(K)]
    thr = 0.02
    chained = [ops.chain(dev3(l), dev3(r)) for l, r in zip(Ls, Rs)]
    for c, l, r in zip(chained, Ls, Rs):
        ref = O.chain(cpu3(l), cpu3(r))
        scale = 1.0 + spread
        assert (c[0].cpu() - ref[0]).abs().max() < 2e-5 * scale
        assert (c[1].cpu() - ref[1]).abs().max() < 1e-6 and (c[2].cpu() - ref[2]).abs().max() < 2e-6
    # selection is exact given identical candidates, fused == unfused bit for bit
    f1, o1, s1, c1 = ops.select(chained, thr, want_chosen=True)
    of, oo, os_, oi = O.select([tuple(t.cpu() for t in c) for c in chained], thr)
    assert torch.equal(c1.cpu().long(), oi) and torch.equal(f1.cpu(), of)
    assert torch.equal(o1.cpu(), oo) and torch.equal(s1.cpu(), os_)
    f2, o2, s2, c2 = ops.chain_select([dev3(l) for l in Ls], [dev3(r) for r in Rs], thr, want_chosen=True)
    assert torch.equal(f1, f2) and torch.equal(o1, o2) and torch.equal(s1, s2) and torch.equal(c1, c2)


@SET
@given(h=st.integers(16, 25), w=st.integers(16, 29), P=st.integers(1, 3), seed=st.integers(0, 10_000),
       spread=st.sampled_from([0.5, 6.0, 60.0]))
def test_lookup_random(h, w, P, seed, spread):
    # (h, w >= 16: a pyramid level of size 1 makes the reference's own coordinate normalisation divide by zero)
    from mft_amd import ops
    rng = np.random.default_rng(seed)
    N = h * w
    vol = torch.from_numpy(rng.standard_normal((P, N, h, w)).astype(np.float32))
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords = np.stack([xs, ys])[None] + (rng.standard_normal((P, 2, h, w)) * spread).astype(np.float32)
    coords[:, :, 0, 0] = np.array([xs[0, 0], ys[0, 0]])[None]          # exactly integral
    coords = torch.from_numpy(coords.astype(np.float32))
    got = []
    levels = None
    for p in range(P):
        pyr = O.corr_pyramid(vol[p].reshape(N, 1, h, w))
        ref = O.corr_lookup(pyr, coords[p:p + 1])                       # [1, 324, h, w]
        got.append(ref[0].permute(1, 2, 0).reshape(N, 324))
        lv = [l.reshape(N, -1) for l in pyr]
        levels = [lv] if levels is None else levels + [lv]
    lv_dev = [ops.block_level(torch.stack([levels[p][l] for p in range(P)]).to(DEV), l, h, w).contiguous()
              for l in range(4)]                                        # row-major maps -> the stored pyramid layout
    cpm = coords.permute(0, 2, 3, 1).reshape(P, N, 2).contiguous().to(DEV)
    out = ops.corr_lookup(lv_dev, cpm, h, w).cpu()
    ref = torch.stack(got)
    assert (out.reshape(P, N, 324) - ref).abs().max() < 2e-4


@SET
@given(h=st.integers(2, 9), w=st.integers(2, 11), pads=st.tuples(st.integers(0, 3), st.integers(0, 4), st.integers(0, 3),
                                                                 st.integers(0, 4)), seed=st.integers(0, 10_000))
def test_convex_upsample_random(h, w, pads, seed):
    from mft_amd import ops
    rng = np.random.default_rng(seed)
    N = h * w
    flow = rng.standard_normal((1, 2, h, w)).astype(np.float32) * 3
    ou = rng.standard_normal((1, 3, h, w)).astype(np.float32)
    mask = rng.standard_normal((1, 576, h, w)).astype(np.float32) * 2
    T = torch.from_numpy
    pm = lambda x: T(x)[0].permute(1, 2, 0).reshape(N, -1).contiguous().to(DEV)     # noqa: E731
    ou4 = torch.zeros(N, 4, device=DEV)
    ou4[:, :3] = pm(ou)
    f, o, s = ops.convex_upsample(pm(flow), ou4, pm(mask), 1, h, w, pads=pads)
    pl, pr, pt, pb = pads
    H0, W0 = 8 * h - pt - pb, 8 * w - pl - pr
    rf = O.convex_upsample(T(flow), T(mask), 8.0)[0, :, pt:pt + H0, pl:pl + W0]
    ro = torch.softmax(O.convex_upsample(T(ou[:, :2]), T(mask), 1.0), dim=1)[0, 1:2, pt:pt + H0, pl:pl + W0]
    rs = torch.sqrt(torch.exp(O.convex_upsample(T(ou[:, 2:3]), T(mask), 1.0)))[0, :, pt:pt + H0, pl:pl + W0]
    assert f.shape == (1, 2, H0, W0)
    assert (f[0].cpu() - rf).abs().max() < 1e-4 and (o[0].cpu() - ro).abs().max() < 1e-5
    assert ((s[0].cpu() - rs).abs() / rs).max() < 1e-4


@settings(max_examples=120, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(c0=st.integers(1, 96).map(lambda v: 4 * v), two=st.booleans(), c1=st.integers(1, 48).map(lambda v: 4 * v),
       cout=st.integers(1, 300), k=st.sampled_from([(1, 1), (3, 3), (1, 5), (5, 1), (7, 1), (3, 1)]),
       P=st.integers(1, 3), h=st.integers(3, 19), w=st.integers(3, 23),
       act=st.sampled_from([None, "relu", "sigmoid", "tanh"]), with_addend=st.booleans(), seed=st.integers(0, 10_000))
def test_conv2d_random(c0, two, c1, cout, k, P, h, w, act, with_addend, seed):
    """The implicit-GEMM conv over ragged channel counts (K chunks partly zero-filled), one or two input
    segments (the first a multiple of 32 channels, as the C ABI demands), every kernel shape of the update
    block and encoders, all epilogues, M and N tails -- against torch's conv2d."""
    import torch.nn.functional as F
    from mft_amd import ops
    kh, kw = k
    if two:
        c0 = max(32, (c0 // 32) * 32)
    cin = c0 + (c1 if two else 0)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(P, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    lin = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    add = torch.randn(P, cout, h, w, generator=g) if with_addend else None
    if add is not None:
        lin = lin + add
    ref = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](lin)
    pm = lambda t: t.permute(0, 2, 3, 1).reshape(P * h * w, -1).contiguous().to(DEV)      # noqa: E731
    xa = pm(x[:, :c0])
    xb = pm(x[:, c0:]) if two else None
    out = ops.conv2d(xa, ops.pack_conv_weight(wt.to(DEV)), b.to(DEV), P, h, w, cout, kh, kw, act=act, x2=xb,
                     addend=pm(add) if add is not None else None)
    got = out.reshape(P, h, w, cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() < 3e-5 * max(1.0, float(lin.abs().max()))


# ---- round 3: the tile-resident kernels over ragged shapes (tiles cut by the image border, images smaller than a tile,
# odd widths, several pairs) against fp64 / the ring-buffered kernels

@SET
@given(h=st.integers(1, 27), w=st.integers(1, 41), P=st.integers(1, 3), seed=st.integers(0, 10_000),
       spread=st.sampled_from([0.5, 6.0, 60.0]))
def test_flow_branch_random(h, w, P, seed, spread):
    from mft_amd import ops
    g = torch.Generator().manual_seed(seed)
    w1, b1 = torch.randn(128, 2, 7, 7, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
    w2, b2 = torch.randn(64, 128, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xs, ys], -1).reshape(1, h * w, 2)
    coords = grid + spread * torch.randn(P, h * w, 2, generator=g)
    wflow = ops.pack_flow_branch_weights(w1.permute(2, 3, 1, 0).reshape(98, 128).contiguous().to(DEV), ops.pack_conv_weight(w2.to(DEV)))
    got = ops.unsplit_activations(ops.flow_branch(coords.to(DEV), h, w, wflow, b1.to(DEV), b2.to(DEV))).cpu().double()
    flow = (coords - grid).double().reshape(P, h, w, 2).permute(0, 3, 1, 2)
    f1 = torch.relu(torch.nn.functional.conv2d(flow, w1.double(), b1.double(), padding=3))
    ref = torch.relu(torch.nn.functional.conv2d(f1, w2.double(), b2.double(), padding=1)).permute(0, 2, 3, 1).reshape(P * h * w, 64)
    assert float((got - ref).abs().max()) < 2e-6 * max(float(ref.abs().max()), 1.0)


@SET
@given(shape=st.sampled_from([(128, 256, 3, 3), (128, 128, 3, 3), (256, 256, 1, 5), (256, 128, 1, 5), (256, 256, 5, 1),
                              (256, 128, 5, 1), (128, 256, 5, 1), (128, 128, 1, 5)]),
       h=st.integers(1, 37), w=st.integers(1, 37), P=st.integers(1, 2), act=st.sampled_from([None, "relu"]),
       with_addend=st.booleans(), seed=st.integers(0, 10_000))
def test_tile_conv_random(shape, h, w, P, act, with_addend, seed):
    from mft_amd import ops
    cin, cout, kh, kw = shape
    g = torch.Generator().manual_seed(seed)
    M = P * h * w
    x = torch.randn(M, cin, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    add = torch.randn(M, cout, generator=g) if with_addend else None
    wpk = ops.pack_conv_weight(wt.to(DEV))
    xs = ops.split_activations(x.to(DEV))
    x1, x2 = (xs, None) if cin == 128 else (xs[:, :128].contiguous(), xs[:, 128:].contiguous())
    got = ops.tile_conv2d(x1, ops.pack_tile_conv_weights(wpk, cout, cin), b.to(DEV), P, h, w, cout, kh, kw, act=act, x2=x2,
                          addend=None if add is None else add.to(DEV))
    xi = x.double().reshape(P, h, w, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xi, wt.double(), b.double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(M, cout)
    if add is not None:
        ref = ref + add.double()
    if act == "relu":
        ref = torch.relu(ref)
    assert float((got.cpu().double() - ref).abs().max()) < 3e-6 * max(float(ref.abs().max()), 1.0)


@SET
@given(h=st.integers(8, 29), w=st.integers(8, 37), P=st.integers(1, 2), seed=st.integers(0, 10_000))
def test_corr_pyramid_tile_resident_random(h, w, P, seed):
    """Level 0 against fp64, the pooled levels bit for bit against avg_pool2d of the stored level 0 (ATen's order), over
    shapes whose super-blocks are cut by the map's border and whose pooled sizes are odd."""
    from mft_amd import ops
    g = torch.Generator().manual_seed(seed)
    f1, f2 = torch.randn(P, h * w, 256, generator=g), torch.randn(P, h * w, 256, generator=g)
    lv = ops.corr_pyramid(f1.to(DEV), f2.to(DEV), h, w, arith=1)
    stride, _ = ops.pyramid_layout(h, w)
    l0 = ops.unblock_level(lv[0], 0, h, w).reshape(P * h * w, 1, h, w).cpu()
    ref = (f1.double() @ f2.double().transpose(1, 2) / 16).reshape(P * h * w, 1, h, w)
    assert float((l0.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    cur = l0
    for l in (1, 2, 3):
        cur = torch.nn.functional.avg_pool2d(cur, 2, 2)
        hl, wl = h >> l, w >> l
        if l == 1:
            got = ops.unblock_level(lv[1], 1, h, w).reshape(P * h * w, 1, hl, wl).cpu()
        else:
            got = lv[l].reshape(P * h * w, -1)[:, :hl * wl].reshape(P * h * w, 1, hl, wl).cpu()
        assert torch.equal(got, cur), l


@SET
@given(h=st.integers(1, 70), w=st.integers(1, 70), P=st.integers(1, 2), vertical=st.booleans(), seed=st.integers(0, 10_000))
def test_gru_half_random(h, w, P, vertical, seed):
    """One SepConvGRU pass as one kernel (mftx_gru_half) against the gate algebra of core/update.py:108-123 in fp64, over ragged
    maps: narrower / shorter than a tile, wider / taller (tiles that overlap by the candidate's halo), one cell."""
    import torch.nn.functional as F
    from mft_amd import ops
    g = torch.Generator().manual_seed(seed)
    M = P * h * w
    hf = torch.tanh(torch.randn(M, 128, generator=g))
    mo = torch.relu(torch.randn(M, 128, generator=g))
    kh, kw = (5, 1) if vertical else (1, 5)
    wzr = torch.randn(256, 256, kh, kw, generator=g) * 0.04
    wq = torch.randn(128, 256, kh, kw, generator=g) * 0.04
    pre_zr = torch.randn(M, 256, generator=g) * 0.5
    pre_q = torch.randn(M, 128, generator=g) * 0.5
    pack = lambda wt, n: ops.pack_tile_conv_weights(ops.pack_conv_weight(wt.to(DEV)), n, 256)      # noqa: E731
    hf_dev = hf.to(DEV)
    hf_new, h_out, z = ops.gru_half(ops.split_activations(hf_dev), ops.split_activations(mo.to(DEV)), pack(wzr, 256), pack(wq, 128),
                                    pre_zr.to(DEV), pre_q.to(DEV), hf_dev, P, h, w, vertical=vertical)
    to_map = lambda t: t.double().reshape(P, h, w, -1).permute(0, 3, 1, 2)      # noqa: E731
    to_rows = lambda t: t.permute(0, 2, 3, 1).reshape(M, -1)                    # noqa: E731
    pad = (kh // 2, kw // 2)
    hd, md = to_map(hf), to_map(mo)
    zr = torch.sigmoid(F.conv2d(torch.cat([hd, md], 1), wzr.double(), padding=pad) + to_map(pre_zr))
    zz, rr = zr[:, :128], zr[:, 128:]
    q = torch.tanh(F.conv2d(torch.cat([rr * hd, md], 1), wq.double(), padding=pad) + to_map(pre_q))
    want = (1 - zz) * hd + zz * q
    assert (torch.sigmoid(z.cpu().double() + pre_zr[:, :128].double()) - to_rows(zz)).abs().max() < 2e-6      # (z: scratch, the gate's sums)
    assert (hf_new.cpu().double() - to_rows(want)).abs().max() < 5e-6
    assert torch.equal(ops.unsplit_activations(h_out), ops.unsplit_activations(ops.split_activations(hf_new)))
