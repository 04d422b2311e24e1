"""Host time of the plain (not pre-bound) forward entry points at batch 4096: where the ~250 us per call go."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16")
x, ctx = bench.make_inputs(4096, 1, dev)
flow.eval()
def run(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    print(f"compute_psd_aware_nll: {run(lambda: flow.compute_psd_aware_nll(x, ctx, None)):.1f} us/call")
    print(f"log_prob:              {run(lambda: flow.log_prob(x, ctx)):.1f} us/call")
    print(f"forward:               {run(lambda: flow(x, ctx)):.1f} us/call")
    flow.freeze_packed()
    print(f"compute_psd_aware_nll (frozen weights): {run(lambda: flow.compute_psd_aware_nll(x, ctx, None)):.1f} us/call")
    flow.freeze_packed(False)
    if len(sys.argv) > 1:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(300): flow.compute_psd_aware_nll(x, ctx, None)
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
