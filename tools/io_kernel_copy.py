import sys, time, torch
sys.path.insert(0, '/root/repo')
from mft_amd import ops
dev='cuda'
for mb in (1, 4):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev); h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, fn in (("kernel D2H", lambda: ops.copy_bytes(d, h)), ("kernel H2D", lambda: ops.copy_bytes(h, d)),
                     ("sdma D2H", lambda: h.copy_(d, non_blocking=True)), ("sdma H2D", lambda: d.copy_(h, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{mb} MB {name}: {dt*1e6:.0f} us = {n/dt/1e9:.2f} GB/s")
