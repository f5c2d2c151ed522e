"""Worker of test_conv_tile_shapes_bitwise: a few conv GEMMs with the tile shape forced by
MFTX_CONV_TILE (read once per process, hence the subprocess), outputs saved to argv[1]."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

import os  # noqa: E402

ARITH = int(os.environ.get("MFTX_TILE_WORKER_ARITH", "0"))        # 1: split-fp16 arithmetic, 2: with A stored in split form
g = torch.Generator().manual_seed(5)
outs = []
LAYERS = [(256, 128, 1, 5, 1, 64, 64, "tanh"), (256, 126, 3, 3, 2, 33, 47, "relu"),
          (128, 64, 3, 3, 1, 64, 64, "relu"), (324, 256, 1, 1, 1, 17, 23, None)]
if ARITH == 2:      # channel counts in whole 8-channel groups; N = 256 wide, two input segments, M not a multiple of any tile
    LAYERS = [(256, 128, 1, 5, 1, 64, 64, "tanh"), (256, 126, 3, 3, 2, 33, 47, "relu"),
              (384, 256, 5, 1, 2, 40, 56, "relu"), (328, 256, 1, 1, 1, 17, 23, None), (192, 384, 3, 3, 1, 24, 31, "relu")]
for (cin, cout, kh, kw, P, h, w, act) in LAYERS:
    x = torch.randn(P * h * w, cin, generator=g).cuda()
    wt = ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) * 0.05).cuda())
    b = torch.randn(cout, generator=g).cuda()
    if ARITH:
        wt = ops.split_weights(wt)
    if ARITH == 2:
        x2 = None
        if cin == 384:                                   # as the GRU gates: [h | x] from two tensors
            x, x2 = x[:, :128].contiguous(), x[:, 128:].contiguous()
        y = ops.conv2d(ops.split_activations(x), wt, b, P, h, w, cout, kh, kw, act=act, arith=1, a_split=True,
                       x2=None if x2 is None else ops.split_activations(x2))
        if os.environ.get("MFTX_TILE_WORKER_REF"):       # the same layer with A split in registers: same bits
            ref = ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act=act, arith=1, x2=x2)
            assert torch.equal(y, ref), ((cin, cout, kh, kw), float((y - ref).abs().max()))
        outs.append(y.cpu().numpy().ravel())
        continue
    outs.append(ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act=act, arith=ARITH).cpu().numpy().ravel())
np.save(sys.argv[1], np.concatenate(outs))
