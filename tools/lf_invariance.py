import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops
P, h, w = 7, 64, 64
N = h * w
g = torch.Generator().manual_seed(3)
f1 = (torch.randn(P, N, 256, generator=g) * 0.5).cuda(); f2 = (torch.randn(P, N, 256, generator=g) * 0.5).cuda()
lv = ops.corr_pyramid(f1, f2, h, w, arith=ops.ARITH_SPLIT)
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, N, 2) + torch.randn(P, N, 2, generator=g) * 5).cuda().contiguous()
wpk = ops.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).cuda())
bias = torch.randn(256, generator=g).cuda()
wf = ops.pack_lookup_convc1_weights(wpk)
full = ops.corr_lookup_convc1(lv, coords, h, w, wf, bias)
for k in range(P):
    one = ops.corr_lookup_convc1([t[k:k + 1].contiguous() for t in lv], coords[k:k + 1].contiguous(), h, w, wf, bias)
    d = (one.reshape(N, -1) - full.reshape(P, N, -1)[k]).abs()
    bad = (d.max(1).values > 0).nonzero().flatten()
    print("pair", k, "equal" if bad.numel() == 0 else f"{bad.numel()} cells differ, first {bad[:8].tolist()}, max {float(d.max()):.3e}")
again = ops.corr_lookup_convc1(lv, coords, h, w, wf, bias)
print("repeat identical:", torch.equal(full, again))
