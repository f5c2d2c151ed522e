"""Flow-estimator plugin -- drop-in for ``MFT/raft.py:16-73`` (``RAFTWrapper``).

``config.flow_config.of_class(config.flow_config)`` must expose
``compute_flow(src_img, dst_img, mode='flow', init_flow=None) -> (flow,
{'occlusion', 'sigma', 'debug'})`` (``MFT/MFT.py:223-225``); this class does,
and adds ``compute_flow_many`` which the tracker uses to run all the flow
deltas of a frame as ONE batched pass through the native engine.

Split of work:
  * feature / context encoders (a3, ``core/extractor.py``): PyTorch-ROCm convs
    -- on the path but not a HIP target in this tier; each frame is encoded
    once and cached (fnet is per-sample instance-normalised and cnet uses
    eval-mode batch norm, so per-frame encoding is exact);
  * everything after the encoders (a4-a12): ``libmftx`` via ``ops.RaftEngine``.
"""
from __future__ import annotations

import logging
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .weights import make_weights, strip_module_prefix

logger = logging.getLogger(__name__)


# ---------------------------------------------------------------------------
# a3: BasicEncoder forward from a flat state_dict (core/extractor.py:118-195)
# ---------------------------------------------------------------------------

class Encoder:
    """7x7/2 stem, three 2-block residual stages (64, 96 /2, 128 /2), 1x1 head.
    norm='instance' (fnet: no affine, eps 1e-5) or 'batch' (cnet: eval mode;
    folded into a per-channel scale/shift once at load)."""

    def __init__(self, sd, prefix, norm):
        self.norm = norm
        self.p = prefix
        self.sd = sd
        self.bn = {}
        if norm == "batch":
            for k in sd:
                if k.startswith(prefix + ".") and k.endswith(".running_mean"):
                    name = k[: -len(".running_mean")]
                    scale = sd[name + ".weight"] / torch.sqrt(sd[name + ".running_var"] + 1e-5)
                    shift = sd[name + ".bias"] - sd[name + ".running_mean"] * scale
                    self.bn[name] = (scale.reshape(1, -1, 1, 1), shift.reshape(1, -1, 1, 1))

    def _n(self, x, name):
        if self.norm == "instance":
            return F.instance_norm(x, eps=1e-5)
        scale, shift = self.bn[name]
        return x * scale + shift

    def _conv(self, x, name, stride=1, padding=0):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding)

    def _block(self, x, name, stride):
        y = F.relu(self._n(self._conv(x, name + ".conv1", stride, 1), name + ".norm1"))
        y = F.relu(self._n(self._conv(y, name + ".conv2", 1, 1), name + ".norm2"))
        if stride != 1:
            x = self._n(self._conv(x, name + ".downsample.0", stride, 0),
                        name + (".downsample.1" if self.norm == "batch" else ".norm3"))
        return F.relu(x + y)

    def __call__(self, x):
        p = self.p
        x = F.relu(self._n(self._conv(x, p + ".conv1", 2, 3), p + ".norm1"))
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            x = self._block(x, f"{p}.layer{li}.0", stride)
            x = self._block(x, f"{p}.layer{li}.1", 1)
        return self._conv(x, p + ".conv2")


def pad_amounts(H0, W0):
    """InputPadder('sintel') (core/utils/utils.py:9-19): (left, right, top, bottom)."""
    ph = (((H0 // 8) + 1) * 8 - H0) % 8
    pw = (((W0 // 8) + 1) * 8 - W0) % 8
    return pw // 2, pw - pw // 2, ph // 2, ph - ph // 2


def _pixel_major(x):
    """[B,C,h,w] -> contiguous [B, h*w, C]."""
    B, C, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, h * w, C).contiguous()


class FrameFeatures:
    __slots__ = ("fmap", "net", "inp", "h", "w", "pads", "shape")

    def __init__(self, fmap, net, inp, h, w, pads, shape):
        self.fmap, self.net, self.inp, self.h, self.w, self.pads, self.shape = fmap, net, inp, h, w, pads, shape


class RAFTWrapper:
    def __init__(self, config, device="cuda", state_dict=None):
        self.C = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("mft_amd.RAFTWrapper runs on an MI355X (HIP) device only; there is no CPU path")
        if state_dict is None:
            state_dict = self._load_weights(config)
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v)))
              for k, v in strip_module_prefix(state_dict).items()}
        self.sd = {k: v.to(self.device) for k, v in sd.items()}
        # encoders: native HIP (default) or the PyTorch-ROCm/MIOpen implementation (C.torch_encoders)
        self.native_encoders = not getattr(config, "torch_encoders", False)
        if self.native_encoders:
            self.fnet_engine = ops.EncoderEngine(self.sd, "fnet", True, self.device)
            self.cnet_engine = ops.EncoderEngine(self.sd, "cnet", False, self.device)
        else:
            self.fnet = Encoder(self.sd, "fnet", "instance")
            self.cnet = Encoder(self.sd, "cnet", "batch")
        self.engine = ops.RaftEngine(self.sd, self.device)
        self._frames = {}
        # Optional (C.async_encode): encode new frames on a side stream.  The encoders of frame t
        # only need the image, so with results kept on the device (no per-frame host sync) their
        # small MIOpen kernels overlap the tail of frame t-1's GEMM-bound refinement instead of
        # sitting on the critical path.  Device-tensor frames must be complete when passed in.
        self._enc_stream = torch.cuda.Stream(device=self.device) if getattr(config, "async_encode", False) else None

    @staticmethod
    def _load_weights(config):
        model = getattr(config, "model", None)
        if model and Path(str(model)).exists():
            logger.info("loading checkpoint %s", model)
            return torch.load(model, map_location="cpu")
        seed = config.synthetic_weights_seed if isinstance(getattr(config, "synthetic_weights_seed", None), int) else 0
        logger.warning("checkpoint %s not found: using seeded synthetic weights (seed %d)", model, seed)
        return make_weights(seed)

    # ---- per-frame encoding + cache ---------------------------------------
    @torch.no_grad()
    def encode(self, img_bgr, want_context=True) -> FrameFeatures:
        """uint8 BGR (H,W,3) -> cached pixel-major features (MFT/raft.py:41-48,
        core/raft.py:122-149)."""
        H0, W0 = img_bgr.shape[:2]
        if self.native_encoders:
            img = img_bgr if isinstance(img_bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_bgr))
            img = img.to(self.device, non_blocking=True).contiguous()
            pads = pad_amounts(H0, W0)
            h, w = (H0 + pads[2] + pads[3]) // 8, (W0 + pads[0] + pads[1]) // 8
            fmap, _ = self.fnet_engine.forward(img)
            net = inp = None
            if want_context:
                net, inp = self.cnet_engine.forward(img)
            return FrameFeatures(fmap, net, inp, h, w, pads, (H0, W0))
        if isinstance(img_bgr, torch.Tensor):       # frame already resident in HBM (uint8 H,W,3 BGR)
            rgb = img_bgr.to(self.device).flip(-1)
        else:
            rgb = torch.from_numpy(np.ascontiguousarray(img_bgr[:, :, ::-1])).to(self.device, non_blocking=True)
        x = rgb.permute(2, 0, 1)[None].float()
        pads = pad_amounts(H0, W0)
        x = F.pad(x, list(pads), mode="replicate")
        x = (2 * (x / 255.0) - 1.0).contiguous()
        fmap = self.fnet(x).float()
        h, w = fmap.shape[-2:]
        net = inp = None
        if want_context:
            c = self.cnet(x)
            net = _pixel_major(torch.tanh(c[:, :128]))[0]
            inp = _pixel_major(torch.relu(c[:, 128:]))[0]
        return FrameFeatures(_pixel_major(fmap)[0], net, inp, h, w, pads, (H0, W0))

    def reset_cache(self):
        self._frames = {}

    def retain(self, frame_ids):
        keep = set(frame_ids)
        for k in [k for k in self._frames if k not in keep]:
            del self._frames[k]

    def _encode(self, img):
        if self._enc_stream is None:
            return self.encode(img)
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._enc_stream):
            f = self.encode(img)
        main.wait_stream(self._enc_stream)
        for t in (f.fmap, f.net, f.inp):          # allocated on the side stream, consumed on `main`
            if t is not None:
                t.record_stream(main)
        return f

    def _features(self, key, img):
        if key is None:
            return self._encode(img)
        f = self._frames.get(key)
        if f is None or f.shape != img.shape[:2]:
            f = self._encode(img)
            self._frames[key] = f
        return f

    # ---- batched entry point used by the tracker -------------------------
    @torch.no_grad()
    def compute_flow_many(self, lefts, right, iters=None):
        """lefts: [(frame_id | None, img)], right: (frame_id | None, img).
        Returns [(flow[2,H,W], occl[1,H,W], sigma[1,H,W])] in the order of
        ``lefts`` -- left_i -> right for every i, one engine call."""
        iters = int(iters if iters is not None else self.C.flow_iters)
        fr = self._features(right[0], right[1])
        fls = [self._features(k, im) for k, im in lefts]
        for f in fls:
            if f.shape != fr.shape:
                raise ValueError("all frames of a batch must have the same size")
        fmap1 = torch.stack([f.fmap for f in fls])
        fmap2 = fr.fmap[None].expand(len(fls), -1, -1).contiguous()
        net = torch.stack([f.net for f in fls])
        inp = torch.stack([f.inp for f in fls])
        flow, occl, sigma = self.engine.refine(fmap1, fmap2, net, inp, fr.h, fr.w, iters, pads=fr.pads)
        return [(flow[i], occl[i], sigma[i]) for i in range(len(fls))]

    # ---- reference plugin API ---------------------------------------------
    @torch.no_grad()
    def compute_flow(self, src_img, dst_img, mode="TC", vis=False, src_img_identifier=None, numpy_out=False,
                     init_flow=None, vis_debug=False):
        """(H,W,3) uint8 BGR images -> flow (2,H,W) + {'occlusion','sigma','debug'}
        (mode='flow'), or (src_coords, dst_coords, extra) (mode='TC')."""
        if init_flow is not None:
            raise NotImplementedError("init_flow is not supported by the native engine (MFT never passes it, "
                                      "MFT/MFT.py:98)")
        H, W = src_img.shape[:2]
        (flow, occl, sigma), = self.compute_flow_many([(None, src_img)], (None, dst_img))
        assert flow.shape == (2, H, W)
        conv = (lambda t: t.detach().cpu().numpy()) if numpy_out else (lambda t: t)
        if mode == "flow":
            return conv(flow), {"occlusion": conv(occl), "sigma": conv(sigma), "debug": None}
        if mode == "TC":
            idx = torch.arange(H * W, device=flow.device)
            src = torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")]).to(torch.float32)
            dst = src + flow.reshape(2, H * W)
            return conv(src), conv(dst), {"occlusion": conv(occl.reshape(-1)) if numpy_out else occl,
                                          "sigma": conv(sigma.reshape(-1)) if numpy_out else sigma,
                                          "debug": None}
        raise ValueError(f"unknown mode {mode!r}")
