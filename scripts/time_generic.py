"""Throughput of the generic-shape kernel on FlowHead(12, 384, 24) of experiments/frozen_context_heads.py:159-163."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import NSFPosteriorFlow
torch.manual_seed(0)
flow = NSFPosteriorFlow(11, 288, 384, 12, 24, 3.0, temperature_scale=1.0, use_masked_context=False).cuda().eval()
for p in flow.parameters(): p.requires_grad_(False)
B = 4096
x = torch.rand(B, 11, device="cuda") * 2 - 1
ctx = torch.randn(B, 288, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for prec in ("fp32", "bf16"):
    flow.precision = prec
    ms = t(lambda: flow.compute_psd_aware_nll(x, ctx, None))
    flop = 2 * 12 * (11 * 384 + 288 * 384 + 2 * (2 * 384 * 384 + 288 * 384) + 384 * 11 * 71)
    z = torch.randn(B, 11, device="cuda")
    mi = t(lambda: flow.inverse(z, ctx[:1].expand(B, -1)), n=3)
    print(f"{prec}: forward {ms:.3f} ms per {B} rows = {B / ms / 1e3:.2f} M samples/s = {B * flop / ms / 1e9:.1f} TFLOP/s (dense count); "
          f"inverse {mi:.2f} ms per {B} draws = {B / mi / 1e3:.3f} M draws/s")
