#!/usr/bin/env python3
"""Reference point for the conv GEMM's rates: the vendor library's fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) on plain
matrices with the (M, N, K) of the update-block layers at 7 pairs -- i.e. WITHOUT the implicit-conv gather, the bias /
activation / GRU epilogues and the 3.5-tiles-per-CU geometry the convolutions have to live with.  Run on the GPU box."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
M = 7 * 4096
LAYERS = [("convc1 1x1 324->256", 256, 324), ("convc2 3x3 256->192", 192, 2304), ("conv 3x3 256->126", 126, 2304),
          ("gru zr 1x5 256->256", 256, 1280), ("gru q 1x5 256->128", 128, 1280), ("fh1 3x3 128->256", 256, 1152),
          ("ou1 3x3 712->256", 256, 6408)]
for name, N, K in LAYERS:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda")
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{name:24s} M={M} N={N:4d} K={K:5d}: {t * 1e6:7.1f} us  {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s")
# the correlation volume as a batched GEMM
a = torch.randn(7, 4096, 256, device="cuda"); b = torch.randn(7, 256, 4096, device="cuda")
for _ in range(3):
    torch.bmm(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    torch.bmm(a, b)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f"{'corr volume (bmm)':24s} 7 x 4096 x 4096 x 256: {t * 1e6:7.1f} us  {2.0 * 7 * 4096 * 4096 * 256 / t / 1e12:6.1f} TFLOP/s (no pooling, row-major output)")
