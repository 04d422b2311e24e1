"""fp32 sampling: incremental inverse vs the D-pass kernel (config 3 flow, one context row), and their agreement."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "fp32")
ctx1 = torch.randn(1, 288, device=dev)
for n in [4096, 32768, 131072]:
    z = torch.randn(n, 15, device=dev)
    res = {}
    for name, inc in (("inc", None), ("dpass", False)):
        flow.incremental_inverse = inc
        with torch.no_grad():
            x, ld = flow.inverse(z, ctx1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): flow.inverse(z, ctx1)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        res[name] = (x, ld, dt)
    dx = (res["inc"][0] - res["dpass"][0]).abs()
    dl = (res["inc"][1] - res["dpass"][1]).abs()
    print(f"{n}: inc {res['inc'][2]*1e3:.2f} ms {n/res['inc'][2]/1e6:.2f} M/s | dpass {res['dpass'][2]*1e3:.2f} ms {n/res['dpass'][2]/1e6:.2f} M/s | "
          f"|dx| median {dx.median():.2e} max {dx.max():.2e} | |dlogdet| median {dl.median():.2e} max {dl.max():.2e}", flush=True)
