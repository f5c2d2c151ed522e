"""Probe: can several ranks share cuda:0?  (a) RCCL, (b) gloo with device tensors, (c) gloo with host staging."""
import os, sys, time
import torch, torch.distributed as dist
backend = sys.argv[1]
torch.cuda.set_device(0)
kw = {}
dist.init_process_group(backend, **kw)
r, w = dist.get_rank(), dist.get_world_size()
send = torch.full((4, 1000), float(r), device="cuda")
recv = torch.empty(w * 4, 1000, device="cuda")
try:
    t0 = time.time()
    work = dist.all_gather_into_tensor(recv, send, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    ok = all(float(recv[4 * i, 0]) == i for i in range(w))
    print(f"[{backend}] rank {r}/{w}: device all_gather_into_tensor ok={ok} {time.time()-t0:.2f}s", flush=True)
except Exception as e:
    print(f"[{backend}] rank {r}: device all_gather failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
dist.barrier()
dist.destroy_process_group()
