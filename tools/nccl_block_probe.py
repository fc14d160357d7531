import os, sys, time, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29652")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.zeros(50_000_000, device="cuda")
y = torch.randn(8192, 8192, device="cuda")
def busy():
    for _ in range(20):
        torch.mm(y, y)
for mode in ("async", "sync_op", "none"):
    for _ in range(2):
        busy()
        if mode == "async":
            h = dist.all_reduce(x, async_op=True); h.wait()
        elif mode == "sync_op":
            dist.all_reduce(x, async_op=False)
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    if mode == "async":
        h = dist.all_reduce(x, async_op=True)
        t1 = time.perf_counter()
        h.wait()
    elif mode == "sync_op":
        dist.all_reduce(x, async_op=False)
        t1 = time.perf_counter()
    else:
        t1 = time.perf_counter()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-8s issue %.2f ms, wait %.2f ms, then device sync %.2f ms (GPU had ~%s ms of work queued)" % (mode, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), "20 mm"), flush=True)
dist.destroy_process_group()
