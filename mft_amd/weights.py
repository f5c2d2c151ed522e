"""Checkpoint schema + seeded synthetic weights for the RAFT-OU flow network.

The reference loads a ``state_dict`` whose keys carry a ``module.`` prefix
(``MFT/raft.py:20-23`` wraps the net in ``nn.DataParallel`` before
``load_state_dict``).  The trained checkpoint is not available in this
environment (``/root/reference/.MISSING_LARGE_BLOBS``), so parity and benchmarks
use weights drawn from a seeded numpy PCG64 stream keyed by the reference's
tensor names and shapes (187 tensors, probed from
``MFT/RAFT/core/raft.py:42-67`` / ``core/extractor.py:118-166`` /
``core/update.py:142-238``).  A real checkpoint with the same keys loads through
:func:`strip_module_prefix` unchanged.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# schema
# ---------------------------------------------------------------------------

_ENCODER_STAGES = ((64, 1), (96, 2), (128, 2))  # (planes, stride of first block)


def _encoder_schema(prefix: str, out_dim: int, batch_norm: bool):
    """BasicEncoder tensors in state_dict order (core/extractor.py:118-166)."""
    items = []

    def bn(name, ch):
        if batch_norm:
            items.append((f"{name}.weight", (ch,), "bn_w"))
            items.append((f"{name}.bias", (ch,), "bn_b"))
            items.append((f"{name}.running_mean", (ch,), "bn_mean"))
            items.append((f"{name}.running_var", (ch,), "bn_var"))
            items.append((f"{name}.num_batches_tracked", (), "bn_count"))

    def conv(name, cout, cin, kh, kw):
        items.append((f"{name}.weight", (cout, cin, kh, kw), "conv_w"))
        items.append((f"{name}.bias", (cout,), "conv_b"))

    bn(f"{prefix}.norm1", 64)
    conv(f"{prefix}.conv1", 64, 3, 7, 7)
    in_planes = 64
    for li, (planes, stride) in enumerate(_ENCODER_STAGES, start=1):
        for bi in range(2):
            blk = f"{prefix}.layer{li}.{bi}"
            cin = in_planes if bi == 0 else planes
            conv(f"{blk}.conv1", planes, cin, 3, 3)
            conv(f"{blk}.conv2", planes, planes, 3, 3)
            bn(f"{blk}.norm1", planes)
            bn(f"{blk}.norm2", planes)
            if bi == 0 and stride != 1:
                bn(f"{blk}.norm3", planes)
                conv(f"{blk}.downsample.0", planes, cin, 1, 1)
                bn(f"{blk}.downsample.1", planes)
        in_planes = planes
    conv(f"{prefix}.conv2", out_dim, 128, 1, 1)
    return items


def _update_schema():
    """BasicUpdateBlock + OcclusionAndUncertaintyBlock (core/update.py)."""
    items = []

    def conv(name, cout, cin, kh, kw):
        items.append((f"{name}.weight", (cout, cin, kh, kw), "conv_w"))
        items.append((f"{name}.bias", (cout,), "conv_b"))

    e = "update_block.encoder"
    conv(f"{e}.convc1", 256, 324, 1, 1)
    conv(f"{e}.convc2", 192, 256, 3, 3)
    conv(f"{e}.convf1", 128, 2, 7, 7)
    conv(f"{e}.convf2", 64, 128, 3, 3)
    conv(f"{e}.conv", 126, 256, 3, 3)
    g = "update_block.gru"
    for n in ("convz1", "convr1", "convq1"):
        conv(f"{g}.{n}", 128, 384, 1, 5)
    for n in ("convz2", "convr2", "convq2"):
        conv(f"{g}.{n}", 128, 384, 5, 1)
    conv("update_block.flow_head.conv1", 256, 128, 3, 3)
    conv("update_block.flow_head.conv2", 2, 256, 3, 3)
    conv("update_block.mask.0", 256, 128, 3, 3)
    conv("update_block.mask.2", 576, 256, 1, 1)
    conv("occlusion_block.occl_head.conv1", 128, 712, 3, 3)
    conv("occlusion_block.occl_head.conv2", 2, 128, 3, 3)
    conv("occlusion_block.uncertainty_head.conv1", 128, 712, 3, 3)
    conv("occlusion_block.uncertainty_head.conv2", 1, 128, 3, 3)
    return items


def schema():
    """[(name, shape, kind)] for every tensor of the reference checkpoint."""
    return (_encoder_schema("fnet", 256, batch_norm=False)
            + _encoder_schema("cnet", 256, batch_norm=True)
            + _update_schema())


# ---------------------------------------------------------------------------
# seeded generator
# ---------------------------------------------------------------------------

# Per-tensor gain overrides.  With no trained checkpoint the net must still be a
# well-conditioned map (a trained RAFT is contractive in its refinement loop):
# the flow head is damped so every iteration moves coordinates by a fraction of
# a pixel, and the OU head biases are set so that roughly a third of the pixels
# exceed MFT's occlusion threshold (0.02, configs/MFT_cfg.py:16) and sigma stays
# O(1).
_GAIN = {
    "fnet.conv2.weight": 0.3,
    "cnet.conv2.weight": 0.3,
    "update_block.encoder.convc1.weight": 0.5,
    "update_block.gru.convz1.weight": 0.5,
    "update_block.gru.convr1.weight": 0.5,
    "update_block.gru.convq1.weight": 0.5,
    "update_block.gru.convz2.weight": 0.5,
    "update_block.gru.convr2.weight": 0.5,
    "update_block.gru.convq2.weight": 0.5,
    "update_block.flow_head.conv2.weight": 0.05,
    "update_block.mask.2.weight": 2.0,
    "occlusion_block.occl_head.conv1.weight": 0.3,
    "occlusion_block.uncertainty_head.conv1.weight": 0.3,
    "occlusion_block.occl_head.conv2.weight": 4.0,
    "occlusion_block.uncertainty_head.conv2.weight": 3.0,
}
_BIAS = {
    "occlusion_block.occl_head.conv2.bias": np.array([3.7, -3.7], np.float32),
    "occlusion_block.uncertainty_head.conv2.bias": np.array([-2.5], np.float32),
}


def make_weights(seed: int = 0, module_prefix: bool = False) -> dict:
    """Deterministic fp32 weights keyed like the reference ``state_dict``.

    Conv weights ~ N(0, gain * sqrt(2 / fan_in)), conv biases ~ U(-0.05, 0.05),
    batch-norm affine/statistics mildly perturbed around identity.  Each tensor
    gets its own PCG64 stream derived from ``(seed, index)`` so the values do not
    depend on generation order.
    """
    out = {}
    for idx, (name, shape, kind) in enumerate(schema()):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        if kind == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            std = np.sqrt(2.0 / fan_in) * _GAIN.get(name, 1.0)
            t = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif kind == "conv_b":
            t = _BIAS[name].copy() if name in _BIAS else \
                rng.uniform(-0.05, 0.05, shape).astype(np.float32)
        elif kind == "bn_w":
            t = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif kind == "bn_b":
            t = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif kind == "bn_mean":
            t = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif kind == "bn_var":
            t = rng.uniform(0.7, 1.3, shape).astype(np.float32)
        elif kind == "bn_count":
            t = np.array(1, dtype=np.int64)
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[("module." + name) if module_prefix else name] = t
    return out


def strip_module_prefix(state_dict: dict) -> dict:
    """Accept both bare and ``module.``-prefixed (DataParallel) checkpoints."""
    return {(k[len("module."):] if k.startswith("module.") else k): v
            for k, v in state_dict.items()}


def to_torch(state_dict: dict, device=None) -> dict:
    import torch
    return {k: torch.as_tensor(np.asarray(v)).to(device) if device is not None
            else torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
