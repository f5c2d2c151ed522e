"""Worker of test_conv_tile_shapes_bitwise: a few conv GEMMs with the tile shape forced by
MFTX_CONV_TILE (read once per process, hence the subprocess), outputs saved to argv[1]."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

import os  # noqa: E402

ARITH = int(os.environ.get("MFTX_TILE_WORKER_ARITH", "0"))        # 1: split-fp16 arithmetic
g = torch.Generator().manual_seed(5)
outs = []
for (cin, cout, kh, kw, P, h, w, act) in [(256, 128, 1, 5, 1, 64, 64, "tanh"), (256, 126, 3, 3, 2, 33, 47, "relu"),
                                          (128, 64, 3, 3, 1, 64, 64, "relu"), (324, 256, 1, 1, 1, 17, 23, None)]:
    x = torch.randn(P * h * w, cin, generator=g).cuda()
    wt = ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) * 0.05).cuda())
    b = torch.randn(cout, generator=g).cuda()
    if ARITH:
        wt = ops.split_weights(wt)
    outs.append(ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act=act, arith=ARITH).cpu().numpy().ravel())
np.save(sys.argv[1], np.concatenate(outs))
