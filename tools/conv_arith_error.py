#!/usr/bin/env python3
"""Error of mftx_conv2d against an fp64 convolution, per arithmetic (argument 0: exact fp32 MFMA,
1: split fp16): max and rms of |y - y64| relative to rms(y64), on update-block shapes with operand
magnitudes spread over several decades (the split must not care)."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

ARITH = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(0)
dev = "cuda"
P, h, w = 2, 64, 64
for name, cin, cout, kh, kw, scale in (("3x3 256->192", 256, 192, 3, 3, 1.0), ("1x5 256->256", 256, 256, 1, 5, 1.0),
                                       ("1x1 324->256", 324, 256, 1, 1, 30.0), ("3x3 128->256 tiny", 128, 256, 3, 3, 1e-4),
                                       ("3x3 712->256", 712, 256, 3, 3, 1.0)):
    x = torch.randn(P, cin, h, w, device=dev) * scale * torch.exp(torch.randn(P, cin, h, w, device=dev))   # log-normal magnitudes
    wt = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y64 = F.conv2d(x.double(), wt.double(), b.double(), padding=(kh // 2, kw // 2))
    y32 = F.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    xm = x.permute(0, 2, 3, 1).reshape(P * h * w, cin).contiguous()
    wp = ops.pack_conv_weight(wt)
    y = ops.conv2d(xm, ops.split_weights(wp) if ARITH else wp, b, P, h, w, cout, kh, kw, arith=ARITH)
    y = y.reshape(P, h, w, cout).permute(0, 3, 1, 2)
    ref = y64.pow(2).mean().sqrt()
    e = (y.double() - y64).abs()
    e32 = (y32.double() - y64).abs()
    print(f"{name:20s} mftx: max {e.max() / ref:.3e} rms {e.pow(2).mean().sqrt() / ref:.3e}   "
          f"vendor fp32 conv: max {e32.max() / ref:.3e} rms {e32.pow(2).mean().sqrt() / ref:.3e}")
