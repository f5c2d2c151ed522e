"""Worker of test_engine_register_split_bitwise: one refinement of the RAFT engine on seeded random features, outputs saved
to argv[1].  The engine's tuning switches are read once per process (MFTX_RAFT_NOPRESPLIT, MFTX_RAFT_NOFUSE, ...), hence
the subprocess."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402

sd = {k: torch.from_numpy(v).cuda() for k, v in make_weights(7).items()}
eng = ops.RaftEngine(sd, "cuda")
g = torch.Generator().manual_seed(11)
P, h, w = 3, 24, 40
f1 = torch.randn(P, h * w, 256, generator=g).cuda()
f2 = (f1.cpu() + 0.3 * torch.randn(P, h * w, 256, generator=g)).cuda()
net = torch.tanh(torch.randn(P, h * w, 128, generator=g)).cuda()
inp = torch.relu(torch.randn(P, h * w, 128, generator=g)).cuda()
flow, occl, sigma = eng.refine(f1, f2, net, inp, h, w, 4)[:3]
np.save(sys.argv[1], np.concatenate([t.cpu().numpy().ravel() for t in (flow, occl, sigma)]))
