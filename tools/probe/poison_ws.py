"""Probe (ONE process, GPU to itself): does the result depend on what the workspaces held before the call?  A correct engine never
reads a workspace region it has not written in the same call."""
import hashlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mft_amd.config import AttrDict, Config
from mft_amd.raft import RAFTWrapper
from mft_amd.synth import SyntheticVideo
from mft_amd.weights import make_weights
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opts = dict(kv.split("=") for kv in sys.argv[2:])
opts = {k: int(v) for k, v in opts.items()}
c = Config(); c.flow_iters = 12; c.raft_params = AttrDict(engine_options=opts)
fl = RAFTWrapper(c, state_dict=make_weights(7))
vid = SyntheticVideo(512, 512, n_frames=9, seed=9)
pairs = [(i, vid[i], 8, vid[8]) for i in range(P)]
fl.compute_pairs(pairs, packed_out=True, planar=False)
torch.cuda.synchronize()
seen = {}
for rep, fill in enumerate([None, 0.0, float("nan"), 1e30, -3.0, "rand", None, float("nan")]):
    for eng in [fl.engine]:
        ws = eng._ws
        if fill == "rand":
            ws.view(torch.float32).normal_(0, 100.0)
        elif fill is not None:
            ws.view(torch.float32).fill_(fill)
    for enc in (fl.fnet_engine, fl.cnet_engine):
        if fill is not None and enc._ws is not None:
            enc._ws.view(torch.float32).fill_(0.0 if fill == "rand" else fill)
    fl.reset_cache()
    out = fl.compute_pairs(pairs, packed_out=True, planar=False)
    torch.cuda.synchronize()
    t = torch.stack([o[3] for o in out])
    h = hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"P={P} opts={opts} fill={fill}: {h} finite={bool(torch.isfinite(t).all())}", flush=True)
