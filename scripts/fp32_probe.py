import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "fp32").freeze_packed()
x, ctx = bench.make_inputs(4096, 1, dev)
nll = torch.empty(4096, device=dev)
ms = bench.kernel_time_ms(flow, x, ctx, nll, torch.cuda.Stream(dev), 20)
n64, sens = bench.fp64_reference(flow, x.cpu(), ctx.cpu())
want = bench.oracle_for(flow).compute_psd_aware_nll(x.cpu(), ctx.cpu(), torch.zeros_like(x.cpu()))
got = flow.nll_into(x, ctx, nll).cpu().double()
den = n64.abs().clamp_min(1.0)
rh, rc = (got - n64).abs() / den, (want.double() - n64).abs() / den
print(f"fp32 kernel {ms*1e3:.1f} us; HIP vs fp64: p50 {rh.median():.3e} p99 {rh.quantile(0.99):.3e} max {rh.max():.3e} over1e-5 {(rh>1e-5).double().mean():.4f} | "
      f"CPU: p50 {rc.median():.3e} p99 {rc.quantile(0.99):.3e} max {rc.max():.3e} over1e-5 {(rc>1e-5).double().mean():.4f}")
z = torch.randn(131072, 15, device=dev); c1 = torch.randn(1, 288, device=dev)
import time
with torch.no_grad():
    flow.inverse(z, c1); torch.cuda.synchronize(); t0 = time.perf_counter(); flow.inverse(z, c1); flow.inverse(z, c1); torch.cuda.synchronize()
print(f"fp32 inverse 131072 draws: {(time.perf_counter()-t0)/2*1e3:.2f} ms")
