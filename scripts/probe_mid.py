"""Mid-batch kernel (PF_FLOW_MID=1 over the PF_FLAG_WIDE layout) against the 16-row kernel: values and time.
usage: PF_FLOW_WIDE=1 PF_FLOW_MID=1 python scripts/probe_mid.py [rows]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16")
x, ctx = bench.make_inputs(rows, 3, dev)
out = torch.empty(rows, device=dev)
print("kernel:", flow.forward_kernel_name(rows), flush=True)
with torch.no_grad():
    got = flow.compute_psd_aware_nll(x, ctx, None).clone()
    z, ld = flow(x, ctx)
    torch.cuda.synchronize()
    os.environ["PF_FLOW_WIDE"] = "0"
    want = flow.compute_psd_aware_nll(x, ctx, None).clone()
    z0, ld0 = flow(x, ctx)
    os.environ["PF_FLOW_WIDE"] = "1"
    torch.cuda.synchronize()
e = (got - want).abs()
print(f"finite {bool(torch.isfinite(got).all())}  |nll - 16-row kernel|: median {e.median():.3e} p99 {e.quantile(0.99):.3e} max {e.max():.3e}; "
      f"|z| median {(z - z0).abs().median():.3e} max {(z - z0).abs().max():.3e}; |ld| median {(ld - ld0).abs().median():.3e}; mean nll {got.mean():.4f} vs {want.mean():.4f}", flush=True)
launch = flow.bind_nll(x, ctx, out)
for _ in range(20): launch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): launch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"{rows} rows: {dt*1e6:.1f} us = {rows/dt/1e6:.1f} M samples/s = {rows*bench.flops_per_sample()/dt/1e12/2500:.3f} of the bf16 peak")
