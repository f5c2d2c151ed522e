"""CPU ORACLE for the MFT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``mft_amd``) never does: it runs
on hand-written HIP kernels and fails loudly when they are missing.

This is a from-scratch restatement (torch CPU tensor ops + explicit index
arithmetic, fp32 throughout) of the reference algorithm; every function cites
the reference lines it follows (paths relative to ``/root/reference``).

Parity pin: the reference ships no tests and no golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tools/make_goldens.py`` (which
imports ``/root/reference`` read-only) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against them.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

F32 = torch.float32


# ---------------------------------------------------------------------------
# sampling primitives
# ---------------------------------------------------------------------------

def pixel_grid(H, W, device=None):
    """G[0,y,x]=x, G[1,y,x]=y  (core/utils/utils.py:115-118;
    MFT/utils/geom_utils.py:429-452 builds the same from idx % W, idx // W)."""
    ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device),
                            indexing="ij")
    return torch.stack([xs, ys], dim=0).to(F32)


# pixel coord -> normalised [-1,1] -> pixel coord, in fp32, exactly as the
# reference does before/inside ``F.grid_sample(align_corners=True)``: the two
# call sites normalise slightly differently, ATen un-normalises with
# ((g+1)/2)*(size-1).

def _norm_utils(p, size):
    # core/utils/utils.py:102-104: 2*x/(W-1) - 1
    g = 2 * p / (size - 1) - 1
    return ((g + 1) / 2) * (size - 1)


def _norm_interp(p, size):
    # MFT/utils/interpolation.py:69-72: x * float32(2/(W-1)) - 1
    scale = torch.tensor(np.float32(2 / (size - 1)))
    g = p * scale - 1
    return ((g + 1) / 2) * (size - 1)


def bilinear_zeros(img, ix, iy):
    """Bilinear sample with zero padding, align_corners=True pixel coords.

    img: (C, H, W); ix, iy: (...) fp32 pixel coordinates (already un-normalised).
    Returns (C, ...).  Semantics of ``F.grid_sample(mode='bilinear',
    padding_mode='zeros', align_corners=True)``: the 4 integer neighbours of
    (ix, iy) contribute with weights (1-fx)(1-fy).. and any neighbour outside
    [0,W-1]x[0,H-1] contributes 0.
    """
    C, H, W = img.shape
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    x0 = x0.to(torch.int64)
    y0 = y0.to(torch.int64)
    flat = img.reshape(C, H * W)
    out = torch.zeros((C,) + tuple(ix.shape), dtype=F32, device=img.device)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx = x0 + dx
            yy = y0 + dy
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(-1)
            v = flat[:, idx].reshape((C,) + tuple(ix.shape))
            out = out + v * (wx * wy * ok.to(F32))
    return out


# ---------------------------------------------------------------------------
# a1: pre/post-processing (MFT/raft.py:39-62, core/utils/utils.py:7-24)
# ---------------------------------------------------------------------------

def pad_amounts(H0, W0):
    """InputPadder(mode='sintel'): (left, right, top, bottom) replicate pads."""
    ph = (((H0 // 8) + 1) * 8 - H0) % 8
    pw = (((W0 // 8) + 1) * 8 - W0) % 8
    return pw // 2, pw - pw // 2, ph // 2, ph - ph // 2


def preprocess(img_bgr):
    """uint8 BGR HxWx3 -> padded fp32 RGB [1,3,H,W] in [0,255] (MFT/raft.py:41-48)."""
    rgb = torch.from_numpy(np.ascontiguousarray(img_bgr[:, :, ::-1])).permute(2, 0, 1)[None].to(F32)
    l, r, t, b = pad_amounts(*img_bgr.shape[:2])
    return F.pad(rgb, [l, r, t, b], mode="replicate")


def unpad(x, H0, W0):
    l, r, t, b = pad_amounts(H0, W0)
    H, W = x.shape[-2:]
    return x[..., t:H - b, l:W - r]


# ---------------------------------------------------------------------------
# a3: encoders (core/extractor.py:6-62,118-195; core/raft.py:62-63)
# ---------------------------------------------------------------------------

def _norm(x, sd, name, kind):
    if kind == "instance":  # nn.InstanceNorm2d: eps 1e-5, no affine, no running stats
        return F.instance_norm(x, eps=1e-5)
    # nn.BatchNorm2d in eval mode
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5)


def _res_block(x, sd, p, kind, stride):
    y = F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=stride, padding=1)
    y = F.relu(_norm(y, sd, p + ".norm1", kind))
    y = F.conv2d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    y = F.relu(_norm(y, sd, p + ".norm2", kind))
    if stride != 1:
        x = F.conv2d(x, sd[p + ".downsample.0.weight"], sd[p + ".downsample.0.bias"], stride=stride)
        x = _norm(x, sd, p + ".downsample.1" if kind == "batch" else p + ".norm3", kind)
    return F.relu(x + y)


def encoder(x, sd, prefix, kind):
    """BasicEncoder.forward: 7x7/2 conv, norm, relu, 3 stages of 2 residual
    blocks (64 s1, 96 s2, 128 s2), 1x1 conv to 256 (core/extractor.py:168-195)."""
    x = F.conv2d(x, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"], stride=2, padding=3)
    x = F.relu(_norm(x, sd, prefix + ".norm1", kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _res_block(x, sd, f"{prefix}.layer{li}.0", kind, stride)
        x = _res_block(x, sd, f"{prefix}.layer{li}.1", kind, 1)
    return F.conv2d(x, sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"])


def normalise_image(x):
    return 2 * (x / 255.0) - 1.0  # core/raft.py:122-124


def features(sd, image):
    """fnet(image) -> [1,256,h,w]; per-sample instance norm makes the batched
    call of core/raft.py:134 equal to two single-image calls."""
    return encoder(normalise_image(image), sd, "fnet", "instance")


def context(sd, image):
    """cnet(image) -> (net=tanh(first 128), inp=relu(last 128)) (core/raft.py:146-149)."""
    c = encoder(normalise_image(image), sd, "cnet", "batch")
    return torch.tanh(c[:, :128]), torch.relu(c[:, 128:])


# ---------------------------------------------------------------------------
# a4-a6: correlation volume, pyramid, lookup (core/corr.py)
# ---------------------------------------------------------------------------

def corr_volume(fmap1, fmap2):
    """V[i,j] = sum_c f1[c,i] f2[c,j] / sqrt(C) -> [N,1,h2,w2] (core/corr.py:53-69); fmap2 may be a pooled
    (smaller) map, as in the on-demand variant."""
    B, C, h, w = fmap1.shape
    assert B == 1
    h2, w2 = fmap2.shape[-2:]
    a = fmap1.reshape(C, h * w)
    b = fmap2.reshape(C, h2 * w2)
    v = torch.matmul(a.t(), b) / torch.sqrt(torch.tensor(float(C)))
    return v.reshape(h * w, 1, h2, w2)


def corr_pyramid(vol, levels=4):
    """3x avg_pool2d(2,2) over the target dims, floor sizes (core/corr.py:22-28)."""
    pyr = [vol]
    for _ in range(levels - 1):
        v = pyr[-1]
        H2, W2 = v.shape[-2] // 2, v.shape[-1] // 2
        v = v[..., :2 * H2, :2 * W2]
        v = (v[..., 0::2, 0::2] + v[..., 0::2, 1::2] + v[..., 1::2, 0::2] + v[..., 1::2, 1::2]) * 0.25
        pyr.append(v)
    return pyr


def corr_lookup(pyr, coords, r=4):
    """coords [1,2,h,w] (x,y) -> [1, L*(2r+1)^2, h, w]; channel
    l*81 + a*9 + b samples level l at (cx/2^l + (a-r), cy/2^l + (b-r))
    (core/corr.py:30-51: meshgrid(dy,dx) stacked then added to (x,y), so the
    first window index offsets x)."""
    _, _, h, w = coords.shape
    N = h * w
    cx = coords[0, 0].reshape(N, 1, 1)
    cy = coords[0, 1].reshape(N, 1, 1)
    d = torch.linspace(-r, r, 2 * r + 1)
    da = d.reshape(1, -1, 1)   # first window index -> added to x
    db = d.reshape(1, 1, -1)   # second window index -> added to y
    outs = []
    for l, v in enumerate(pyr):
        Hl, Wl = v.shape[-2:]
        px = cx / 2 ** l + da + 0 * db
        py = cy / 2 ** l + db + 0 * da
        ix = _norm_utils(px, Wl)
        iy = _norm_utils(py, Hl)
        vol = v.reshape(N, Hl * Wl)
        # per-query-pixel sampling: gather from row i of the volume
        x0 = torch.floor(ix); y0 = torch.floor(iy)
        fx = ix - x0; fy = iy - y0
        x0 = x0.to(torch.int64); y0 = y0.to(torch.int64)
        acc = torch.zeros_like(ix)
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xx = x0 + dx; yy = y0 + dy
                ok = (xx >= 0) & (xx <= Wl - 1) & (yy >= 0) & (yy <= Hl - 1)
                idx = (yy.clamp(0, Hl - 1) * Wl + xx.clamp(0, Wl - 1)).reshape(N, -1)
                val = torch.gather(vol, 1, idx).reshape(ix.shape)
                acc = acc + val * (wx * wy * ok.to(F32))
        outs.append(acc.reshape(N, -1))
    out = torch.cat(outs, dim=1)                       # [N, 324]
    return out.t().reshape(1, -1, h, w).contiguous()


def corr_lookup_ondemand(fmap1, fmap2, coords, r=4, levels=4):
    """AlternateCorrBlock (core/corr.py:72-100) + alt_cuda_corr's forward kernel
    (alt_cuda_corr/correlation_kernel.cu:18-119): the pyramid is built on the SECOND feature map
    (F.avg_pool2d, corr.py:80-82; fmap1 stays at level 0, corr.py:91), every level's window is
    <f1[cell], f2_l[tap]> / sqrt(C) sampled bilinearly at coords / 2^l with zeros outside; channel
    l*81 + ix*9 + iy like CorrBlock (kernel lines 99-102: index iy + rd*ix, ix offsets x).
    "Parity unpinned" against the CUDA op itself (it cannot be built here: no nvcc); pinned through the
    equivalence with CorrBlock the reference relies on (tests: vs the golden lookup)."""
    pyr = []
    f2 = fmap2
    for l in range(levels):
        pyr.append(corr_volume(fmap1, f2))           # [N, 1, h_l, w_l] = <f1, f2_l> / sqrt(C)
        f2 = F.avg_pool2d(f2, 2, stride=2)
    return corr_lookup(pyr, coords, r)


# ---------------------------------------------------------------------------
# a7-a9: update block (core/update.py:96-160,216-238)
# ---------------------------------------------------------------------------

def _conv(x, sd, name, padding):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def motion_encoder(sd, flow, corr):
    e = "update_block.encoder"
    cor = F.relu(_conv(corr, sd, e + ".convc1", 0))
    cor = F.relu(_conv(cor, sd, e + ".convc2", 1))
    flo = F.relu(_conv(flow, sd, e + ".convf1", 3))
    flo = F.relu(_conv(flo, sd, e + ".convf2", 1))
    out = F.relu(_conv(torch.cat([cor, flo], 1), sd, e + ".conv", 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, h, x):
    g = "update_block.gru"
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(hx, sd, g + ".convz" + sfx, pad))
        r = torch.sigmoid(_conv(hx, sd, g + ".convr" + sfx, pad))
        q = torch.tanh(_conv(torch.cat([r * h, x], 1), sd, g + ".convq" + sfx, pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, net, inp, corr, flow):
    """-> net, up_mask, delta_flow, motion_features (core/update.py:229-238)."""
    motion = motion_encoder(sd, flow, corr)
    net = sep_conv_gru(sd, net, torch.cat([inp, motion], 1))
    u = "update_block"
    delta = _conv(F.relu(_conv(net, sd, u + ".flow_head.conv1", 1)), sd, u + ".flow_head.conv2", 1)
    mask = 0.25 * _conv(F.relu(_conv(net, sd, u + ".mask.0", 1)), sd, u + ".mask.2", 0)
    return net, mask, delta, motion


# ---------------------------------------------------------------------------
# a11: occlusion + uncertainty heads (core/update.py:17-75,196-214)
# ---------------------------------------------------------------------------

def ou_block(sd, net, inp, corr, flow, delta_flow, motion):
    x = torch.cat([net, inp, corr, flow, delta_flow, motion], 1)   # 712 channels
    o = "occlusion_block"
    occl = _conv(F.relu(_conv(x, sd, o + ".occl_head.conv1", 1)), sd, o + ".occl_head.conv2", 1)
    unc = _conv(F.relu(_conv(x, sd, o + ".uncertainty_head.conv1", 1)), sd,
                o + ".uncertainty_head.conv2", 1)
    return occl, unc


# ---------------------------------------------------------------------------
# a10: convex upsampling (core/raft.py:83-94)
# ---------------------------------------------------------------------------

def convex_upsample(x, mask, mult):
    """x [1,C,h,w], mask [1,576,h,w] (channel k*64+sy*8+sx) -> [1,C,8h,8w]."""
    _, C, h, w = x.shape
    m = torch.softmax(mask.reshape(9, 8, 8, h, w), dim=0)
    xp = F.pad(mult * x[0], [1, 1, 1, 1])                      # zero pad
    out = torch.zeros(C, 8, 8, h, w)
    for k in range(9):
        ky, kx = divmod(k, 3)
        nb = xp[:, ky:ky + h, kx:kx + w]                       # x[c, y+ky-1, x+kx-1]
        out = out + m[k][None] * nb[:, None, None]
    return out.permute(0, 3, 1, 4, 2).reshape(1, C, 8 * h, 8 * w)


# ---------------------------------------------------------------------------
# a2: RAFT forward in test mode (core/raft.py:97-259)
# ---------------------------------------------------------------------------

class _Stage:
    """`with _Stage(timers, key):` adds the wall time of the block to timers[key] (bench.py's
    per-stage split of the CPU baseline); a no-op when timers is None."""

    def __init__(self, timers, key):
        self.timers, self.key = timers, key

    def __enter__(self):
        if self.timers is not None:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.timers is not None:
            import time
            self.timers[self.key] = self.timers.get(self.key, 0.0) + time.perf_counter() - self.t0


def raft_refine(sd, fmap1, fmap2, net, inp, iters, flow_init=None, trace=None, timers=None):
    """The iterative part of RAFT.forward, from feature maps onwards."""
    _, _, h, w = fmap1.shape
    with _Stage(timers, "corr_volume_pyramid"):
        pyr = corr_pyramid(corr_volume(fmap1, fmap2))
    coords0 = pixel_grid(h, w)[None]
    coords1 = coords0.clone()
    if flow_init is not None:
        coords1 = coords1 + flow_init
    for itr in range(iters):
        with _Stage(timers, "corr_lookup"):
            corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        with _Stage(timers, "update_block"):
            net, mask, delta, motion = update_block(sd, net, inp, corr, flow)
        coords1 = coords1 + delta
        if trace is not None:
            trace.append(dict(corr=corr, net=net, delta=delta, coords1=coords1))
    flow_lr = coords1 - coords0
    with _Stage(timers, "ou_block"):
        occl, unc = ou_block(sd, net, inp, corr, flow_lr, delta, motion)
    with _Stage(timers, "convex_upsample"):
        return dict(flow=convex_upsample(flow_lr, mask, 8.0),
                    occlusion=convex_upsample(occl, mask, 1.0),
                    uncertainty=convex_upsample(unc, mask, 1.0),
                    coords=flow_lr)


def raft_forward(sd, image1, image2, iters=12, flow_init=None, timers=None):
    with _Stage(timers, "encoders"):
        fmap1 = features(sd, image1)
        fmap2 = features(sd, image2)
        net, inp = context(sd, image1)
    return raft_refine(sd, fmap1, fmap2, net, inp, iters, flow_init, timers=timers)


def postprocess(pred, H0, W0):
    """unpad; occl = softmax(logits)[1]; sigma = sqrt(exp(u)) (MFT/raft.py:57-62)."""
    flow = unpad(pred["flow"], H0, W0)[0]
    occl = unpad(torch.softmax(pred["occlusion"], dim=1)[:, 1:2], H0, W0)[0]
    sigma = torch.sqrt(torch.exp(unpad(pred["uncertainty"], H0, W0)[0]))
    return flow, occl, sigma


def downsample_flow_8(flow):
    """(B, xy, H, W) -> (B, xy, H/8, W/8), bilinear with align_corners=True, values / 8
    (MFT/raft.py:98-101)."""
    size = (flow.shape[2] // 8, flow.shape[3] // 8)
    return F.interpolate(flow, size=size, mode="bilinear", align_corners=True) / 8


def compute_flow(sd, src_img, dst_img, iters=12, timers=None, init_flow=None):
    """RAFTWrapper.compute_flow(mode='flow') (MFT/raft.py:30-73) ->
    flow[2,H,W], occlusion[1,H,W], sigma[1,H,W].  init_flow: optional (2, H, W) initial flow, padded like
    the images and brought to 1/8 resolution (MFT/raft.py:49-52), added to coords1 (core/raft.py:153-154)."""
    H0, W0 = src_img.shape[:2]
    flow_init = None
    if init_flow is not None:
        l, r, t, b = pad_amounts(H0, W0)
        flow_init = downsample_flow_8(F.pad(torch.as_tensor(init_flow).to(F32)[None], [l, r, t, b], mode="replicate"))
    pred = raft_forward(sd, preprocess(src_img), preprocess(dst_img), iters, flow_init=flow_init, timers=timers)
    return postprocess(pred, H0, W0)


# ---------------------------------------------------------------------------
# a14: chaining (MFT/MFT.py:233-239, MFT/results.py:87-136)
# ---------------------------------------------------------------------------

def chain(L, R):
    """L = (flow, occl, sigma) template->left, R = left->right; -> template->right."""
    flowL, occL, sigL = L
    flowR, occR, sigR = R
    _, H, W = flowL.shape
    G = pixel_grid(H, W)
    p = G + flowL
    ix = _norm_interp(p[0], W)
    iy = _norm_interp(p[1], H)
    flow = p + bilinear_zeros(flowR, ix, iy) - G          # (coords_B + sampled) - coords_A
    occ = torch.maximum(occL, bilinear_zeros(occR, ix, iy))
    sig = torch.sqrt(torch.square(sigL) + torch.square(bilinear_zeros(sigR, ix, iy)))
    return flow, occ, sig


# ---------------------------------------------------------------------------
# a15: per-pixel best-chain selection (MFT/MFT.py:112-143, results.py:250-265)
# ---------------------------------------------------------------------------

def invalid_mask(flow):
    _, H, W = flow.shape
    q = pixel_grid(H, W) + flow
    return (q[0] < 0) | (q[1] < 0) | (q[0] >= W) | (q[1] >= H)


def select(cands, thr):
    """cands: list of (flow, occl, sigma) already ordered [inf, 1, 2, ...].
    -> flow, occl, sigma, chosen index map (first arg-max of -sigma, occluded
    candidates scored -inf; all -inf picks index 0)."""
    flows = torch.stack([c[0] for c in cands])
    occs = torch.stack([c[1] for c in cands])
    sigs = torch.stack([c[2] for c in cands])
    scores = -sigs
    scores = torch.where(occs > thr, torch.full_like(scores, -float("inf")), scores)
    K = len(cands)
    # first maximal index along dim 0
    best = scores.max(dim=0, keepdim=True).values
    ks = torch.arange(K).reshape(K, 1, 1, 1).expand_as(scores)
    idx = torch.where(scores == best, ks, torch.full_like(ks, K)).min(dim=0, keepdim=True).values
    flow = flows.gather(0, idx.expand(1, 2, *idx.shape[2:]))[0]
    occ = occs.gather(0, idx)[0].clone()
    sig = sigs.gather(0, idx)[0]
    occ[0][invalid_mask(flow)] = 1
    return flow, occ, sig, idx[0, 0]


# ---------------------------------------------------------------------------
# a16: tracker bookkeeping (MFT/MFT.py:22-185)
# ---------------------------------------------------------------------------

class Tracker:
    """MFT.init()/track() restated; flows come from ``flow_fn(left_id, right_id,
    left_img, right_img) -> (flow, occl, sigma)`` so tests can plug in either
    the oracle RAFT or recorded flows."""

    def __init__(self, flow_fn, deltas=(np.inf, 1, 2, 4, 8, 16, 32), occlusion_threshold=0.02, timers=None):
        self.flow_fn = flow_fn
        self.deltas = list(deltas)
        self.thr = occlusion_threshold
        self.timers = timers            # optional {stage: seconds} accumulator (bench.py)

    def init(self, img, start_frame_i=0, time_direction=1):
        H, W = img.shape[:2]
        self.start, self.cur, self.dir = start_frame_i, start_frame_i, time_direction
        ident = (torch.zeros(2, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W))
        self.memory = {self.start: dict(img=img, result=ident)}
        return SimpleNamespace(result=ident)

    def _before_start(self, i):
        return (self.dir > 0 and i < self.start) or (self.dir < 0 and i > self.start)

    def track(self, img):
        self.cur += self.dir
        cands, used, pairs = {}, [], []
        for d in self.deltas:
            left = self.cur - d * self.dir
            if self._before_start(left):
                if np.isinf(d):
                    left = self.start
                else:
                    continue
            left = int(left)
            if left in used:
                continue
            R = self.flow_fn(left, self.cur, self.memory[left]["img"], img)
            with _Stage(self.timers, "chain"):
                cands[d] = chain(self.memory[left]["result"], R)
            used.append(left)
            pairs.append((left, self.cur))
        order = sorted(cands.keys(), key=lambda d: 0 if np.isinf(d) else d)
        with _Stage(self.timers, "select"):
            flow, occ, sig, idx = select([cands[d] for d in order], self.thr)
        self.memory[self.cur] = dict(img=img, result=(flow, occ, sig))
        self._cleanup()
        return SimpleNamespace(result=(flow, occ, sig), chosen=idx, pairs=pairs,
                               order=order, memory_keys=sorted(self.memory.keys()))

    def _cleanup(self):
        finite = [d for d in self.deltas if np.isfinite(d)]
        max_delta = max(finite) if finite else 0
        has_inf = any(np.isinf(d) for d in self.deltas)
        for m in list(self.memory.keys()):
            if m == self.start and has_inf:
                continue
            if self.dir > 0 and m + max_delta > self.cur:
                continue
            if self.dir < 0 and m - max_delta < self.cur:
                continue
            del self.memory[m]


# ---------------------------------------------------------------------------
# flow-cache codec -- MFT/utils/io.py:495-512 and :535-551
# "parity unpinned" for the PNG container: cv2 is absent here, so the reference's writer cannot
# be run; the arithmetic below restates compress_channel / u16_to_3u8 / data_3u8_to_u16 /
# decompress_channel line by line in numpy, like the reference evaluates them.
# ---------------------------------------------------------------------------
def quantize_u16(xs):
    """compress_channel (io.py:496-506): -> (uint16 array, lb float32, ub float32)."""
    f_xs = np.float32(xs)
    lb = np.amin(f_xs)
    ub = np.amax(f_xs)
    if np.abs(ub - lb) < 1e-8:
        xs_01 = np.zeros_like(f_xs)
    else:
        xs_01 = (f_xs - lb) / (ub - lb)
    return np.uint16(np.round(xs_01 * (2 ** 16 - 1))), lb, ub


def dequantize_u16(compressed_xs, lb, ub):
    """decompress_channel (io.py:548-551)."""
    xs_01 = np.float32(compressed_xs) / (2 ** 16 - 1)
    return (xs_01 * (ub - lb)) + lb


def u16_to_bgr(xs):
    """u16_to_3u8 (io.py:508-513): the (B, G, R) planes cv2.imencode is handed."""
    return np.dstack((np.zeros_like(xs, np.uint8), np.uint8((xs & 0xFF00) >> 8), np.uint8(xs & 0x00FF)))


def bgr_to_u16(xs):
    """data_3u8_to_u16 (io.py:541-546)."""
    b3, b2, b1 = np.dsplit(np.uint16(xs), 3)
    return ((b2 << 8) | b1)[..., 0]
