import sys, time, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, numpy as np
from mft_amd.config import load_config
from mft_amd.synth import SyntheticVideo
from mft_amd.video import FrameRing, ResultDrain
from mft_amd.MFT import MFT as MFTClass
conf = load_config(str(__import__("pathlib").Path(__file__).resolve().parents[1] / "configs" / "MFT_cfg.py"))
conf.flow_config.model = None; conf.flow_config.synthetic_weights_seed = 0; conf.flow_config.async_encode = True
conf.keep_result_on_device = True
if os.environ.get("NF_OFF"): conf.nonfinite_check_every = 0
tr = conf.tracker_class(conf)
vid = SyntheticVideo(512, 512, n_frames=64, seed=0)
frames = [vid[i % 64] for i in range(300)]
tr.init(torch.from_numpy(frames[0]).cuda())
for i in range(1, 40): tr.track(torch.from_numpy(frames[i]).cuda())
torch.cuda.synchronize()
orig = MFTClass._check_nonfinite
tcheck = []
def timed(self, synced):
    t = time.perf_counter(); r = orig(self, synced); tcheck.append(time.perf_counter() - t); return r
MFTClass._check_nonfinite = timed
enc = tr.flower._enc_stream
ring = FrameRing((frames[i] for i in range(40, 300)), keep=40, streams=[enc]).prepare(frames[0].shape)
drain = ResultDrain(depth=4, nonfinite_from=tr).prepare(tr.memory[tr.current_frame_i]['result'])
torch.set_num_threads(1)
ts = []
t0 = time.perf_counter()
for n, frame in enumerate(ring):
    a = time.perf_counter()
    m = tr.track(frame)
    b = time.perf_counter()
    drain.submit(m.result)
    c = time.perf_counter()
    if len(drain) > 2: drain.collect()
    d = time.perf_counter()
    ts.append((b - a, c - b, d - c))
torch.cuda.synchronize()
print("fps", len(ts) / (time.perf_counter() - t0))
arr = np.array(ts) * 1e3
print("track ms: mean %.2f max %.2f | submit mean %.2f max %.2f | collect mean %.2f max %.2f" % (arr[:,0].mean(), arr[:,0].max(), arr[:,1].mean(), arr[:,1].max(), arr[:,2].mean(), arr[:,2].max()))
tc = np.array(tcheck) * 1e3
print("check ms: mean %.3f max %.3f; calls > 1 ms:" % (tc.mean(), tc.max()), [(i, round(x, 2)) for i, x in enumerate(tc) if x > 1.0][:10])
print("track > 8 ms at", [(i, round(x, 1)) for i, x in enumerate(arr[:,0]) if x > 8][:10], "collect > 8 ms at", [(i, round(x,1)) for i, x in enumerate(arr[:,2]) if x > 8][:10])
