import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from types import SimpleNamespace
from mft_amd.synth import SyntheticVideo
from mft_amd.video import FrameRing, ResultDrain
args = SimpleNamespace(iters=12, sync_encode=False)
vid = SyntheticVideo(512, 512, n_frames=120, seed=0)
host = [vid[i] for i in range(120)]
dev = [torch.from_numpy(f).cuda() for f in host]
tr, conf = bench.build_tracker(args, False)
tr.init(dev[0])
for i in range(1, 40): tr.track(dev[i])
torch.cuda.synchronize()
def timeit(name, fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{name}: {n / (time.perf_counter() - t0):.1f} fps")
st = [40]
def dev_loop():
    for i in range(st[0], st[0] + 20): tr.track(dev[i])
    st[0] += 20
timeit("device frames, device results", dev_loop)
def ring_loop():
    for f in FrameRing(host[st[0]: st[0] + 20], depth=4): tr.track(f)
    st[0] += 20
timeit("ring in, device results", ring_loop)
def drain_loop():
    d = ResultDrain()
    for i in range(st[0], st[0] + 20):
        d.submit(tr.track(dev[i]).result)
        while len(d) > 2: d.collect()
    while len(d): d.collect()
    st[0] += 20
timeit("device frames, drain out", drain_loop)
def simple_loop():
    conf.keep_result_on_device = False
    for i in range(st[0], st[0] + 20): tr.track(host[i])
    conf.keep_result_on_device = True
    st[0] += 20
timeit("numpy in, .cpu() out (round 1 path)", simple_loop)
