"""Incremental-inverse variants: draws/s of pf_flow_inverse_inc at several batch sizes (config 3 flow, D = 15, L = 8,
one context row).  $PF_INC_THREADS=512 selects the 8-wave variant."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16").freeze_packed()
ctx1 = torch.randn(1, 288, device=dev)
out = []
for n in [int(a) for a in sys.argv[1:]] or [4096, 12288, 32768, 131072]:
    z = torch.randn(n, 15, device=dev)
    with torch.no_grad():
        for _ in range(2): flow.inverse(z, ctx1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): flow.inverse(z, ctx1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    out.append(f"{n}: {dt*1e3:.3f} ms {n/dt/1e6:.2f} M/s")
print("threads", os.environ.get("PF_INC_THREADS", "256"), "|", " | ".join(out))
