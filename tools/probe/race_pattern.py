"""Probe: under load, WHERE does a bad output of the 32-cell tile_conv kernel differ from the good one?"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mft_amd import ops
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(3)
P, h, w = 1, 64, 64
M = P * h * w
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
x128 = ops.split_activations(rnd(M, 128))
kh, kw, N, cin = 5, 1, 128, 128
wt = ops.pack_tile_conv_weights(ops.pack_conv_weight(rnd(N, cin, kh, kw, sc=0.05)), N, cin)
bias = rnd(N, sc=0.1)
fn = lambda: ops.tile_conv2d(x128, wt, bias, P, h, w, N, kh, kw, act="relu")
ref = fn().clone()
torch.cuda.synchronize()
bad = []
masks = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    out = fn()
    ne = out != ref
    masks.append(torch.stack([ne.any(), ]))
    if len(bad) < 400:
        bad.append((ne.any(1).clone(), ne.any(0).clone(), (out - ref).abs().max().clone(), ne.sum().clone()))
torch.cuda.synchronize()
n = 0
for rows, cols, mx, cnt in bad:
    if int(cnt) == 0: continue
    n += 1
    r = rows.nonzero().flatten().tolist(); c = cols.nonzero().flatten().tolist()
    ys = sorted(set(i // w for i in r)); xs = sorted(set(i % w for i in r))
    print(f"bad call: {int(cnt)} values differ, max abs {float(mx):.3e}; cells {len(r)} rows y in [{ys[0]}..{ys[-1]}] ({len(ys)}), x in {xs[:8]}{'...' if len(xs) > 8 else ''} ({len(xs)}); channels {c[0]}..{c[-1]} ({len(c)}): {c[:40]}", flush=True)
    if n >= 12: break
print("bad calls among first 400:", sum(int(b[3]) > 0 for b in bad))
