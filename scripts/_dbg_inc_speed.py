import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16").freeze_packed()
ctx1 = torch.randn(1, 288, device=dev)
def T(fn, n):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    for nd in (4096, 32768, 131072):
        z = torch.randn(nd, 15, device=dev)
        for inc in (False, None):
            flow.incremental_inverse = inc
            dt = T(lambda: flow.inverse(z, ctx1), 5)
            print(f"draws={nd} incremental={inc is None}: {dt*1e3:.3f} ms  {nd/dt/1e6:.2f} M draws/s", flush=True)
