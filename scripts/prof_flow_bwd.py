"""Kernel mix of one flow forward + backward at 2048 rows (run under rocprofv3 --kernel-trace)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda"); torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train(); model.flow.precision = prec
flow = model.flow
if os.environ.get("PF_FLAT", "1") != "0":
    flow.flatten_parameters()                  # one flat leaf (PF_FLAT=0: 180 per-tensor Parameters through autograd)
n_rows = 2048
ctx = torch.randn(n_rows, flow.context_features, device=dev, requires_grad=True)
x = (torch.rand(n_rows, flow.features, device=dev) * 2 - 1)
def flow_fb():
    for p_ in flow.parameters(): p_.grad = None        # what optimizer.zero_grad() does (set_to_none)
    ctx.grad = None
    flow.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).mean().backward()
for _ in range(3): flow_fb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): flow_fb()
torch.cuda.synchronize(); print(f"flow fwd+bwd {n_rows} rows {prec}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
if len(sys.argv) > 2 and sys.argv[2] == "cprofile":      # where the host time of one iteration goes
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): flow_fb()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
