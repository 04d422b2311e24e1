"""The in-kernel (sum nll, rows) reduction: launches of pf_flow_forward vs pf_flow_forward_reduce back to back (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16").freeze_packed()
x, ctx = bench.make_inputs(4096, 1, dev)
nll = torch.empty(4096, device=dev)
red = [torch.zeros(16, 2, device=dev) for _ in range(3)]
plain = flow.bind_nll(x, ctx, nll)
reduce_ = flow.bind_nll(x, ctx, nll, sum_count=red)
def t(fn, n=300):
    for i in range(50): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(3):
    print(f"plain {t(plain):.2f} us   reduce {t(reduce_):.2f} us", flush=True)
