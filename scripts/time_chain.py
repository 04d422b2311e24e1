"""Kernel time of pf_flow_backward_chain / pf_flow_reevaluate alone (bf16 mode, LeanNPE's flow, 2048 rows), by HIP events
around 20 calls of the backward (side builds with -DPF_CHAIN_ABLATE=<mask> through $PF_LIBPFHIP)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe, _lib, _flow_autograd as fa
dev = torch.device("cuda"); torch.manual_seed(0)
flow = npe.LeanNPE().to(dev).train().flow; flow.precision = "bf16"
B = 2048
ctx = torch.randn(B, flow.context_features, device=dev); x = torch.rand(B, flow.features, device=dev) * 2 - 1
U = torch.empty(flow.num_layers, B, flow.features, device=dev)
with torch.no_grad():
    flow._forward_call(x, ctx, None, layer_inputs=U)
gz, gl = torch.randn(B, flow.features, device=dev), torch.randn(B, device=dev)
# time the chain call alone by patching the library call
lib = _lib.lib(); orig = lib.pf_flow_backward_chain; times = []
def timed(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = orig(*a); e1.record(); times.append((e0, e1)); return rc
class L2:                                    # proxy that times one entry point
    def __getattr__(self, n): return timed if n == "pf_flow_backward_chain" else getattr(lib, n)
_lib._lib = L2()
for _ in range(25): fa._flow_backward_batched(flow, U, ctx, gz, gl)
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in times[5:])
print(f"{os.environ.get('PF_LIBPFHIP', 'default'):55s} chain kernel median {t[len(t) // 2] * 1e3:.1f} us")
