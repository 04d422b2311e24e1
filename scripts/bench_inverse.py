"""Sampling throughput: draws/s of NSFPosteriorFlow.inverse for one event (one context row)."""
import sys, time, torch
sys.path.insert(0, ".")
from bench import build_flow, D, C
for n in (4096, 131072):
    for hoist in (True, False):
        flow = build_flow("cuda", "bf16"); flow.hoist_context = hoist; flow.freeze_packed()
        ctx = torch.randn(1, C, device="cuda"); z = torch.randn(n, D, device="cuda")
        with torch.no_grad():
            for _ in range(3): flow.inverse(z, ctx)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            k = 10 if n <= 4096 else 3
            for _ in range(k): flow.inverse(z, ctx)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
        print(f"draws={n} hoist={hoist}: {dt*1e3:.3f} ms  {n/dt/1e6:.2f} M draws/s", flush=True)
