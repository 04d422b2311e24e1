"""bf16 error of the 16-row kernel on the bench workload against fp64 and against the bf16-emulating oracle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from helpers import make_pair
from oracle import nflows_restated as nfr
torch.set_num_threads(16)
ref, ref64, flow = make_pair(bench.D, bench.C, bench.H, bench.L, bench.K, bench.TB, scale=bench.FINAL_LAYER_SCALE)
flow.precision = "bf16"
x, ctx = bench.make_inputs(4096, 1, "cpu")
zeros = torch.zeros_like(x)
with torch.no_grad():
    n64 = ref64.compute_psd_aware_nll(x.double(), ctx.double(), zeros.double())
    with nfr.gemm_emulation("bf16"):
        nemu = ref.compute_psd_aware_nll(x, ctx, zeros).double()
    got = flow.compute_psd_aware_nll(x.cuda(), ctx.cuda(), None).cpu().double()
q = lambda t: "p50 %.2e p90 %.2e p99 %.2e max %.2e" % tuple(t.quantile(torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=t.dtype)).tolist())
print("kernel vs emulation:", q((got - nemu).abs()))
print("kernel vs fp64     :", q((got - n64).abs()))
print("emulation vs fp64  :", q((nemu - n64).abs()))
tails = (x.abs() > bench.TB).any(dim=1)
print("rows with a tail entry:", int(tails.sum()), " kernel vs fp64 on rows WITHOUT tails:", q((got - n64).abs()[~tails]), " WITH:", q((got - n64).abs()[tails]))
print("emulation vs fp64 without tails:", q((nemu - n64).abs()[~tails]))
