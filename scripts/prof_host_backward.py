"""Host overhead of the flow's forward + backward: at 16 rows the kernels take ~0.3 ms, what remains is the host."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe, _flow_autograd as fa
dev = torch.device("cuda"); torch.manual_seed(0)
flow = npe.LeanNPE().to(dev).train().flow; flow.precision = "bf16"
for B in (16, 2048):
    ctx = torch.randn(B, flow.context_features, device=dev, requires_grad=True)
    x = torch.rand(B, flow.features, device=dev) * 2 - 1
    U = torch.empty(flow.num_layers, B, flow.features, device=dev)
    with torch.no_grad():
        z, _, _ = flow._forward_call(x, ctx.detach(), None, layer_inputs=U)
    g = torch.randn(B, device=dev)
    params = flow._ordered_parameters()
    def fb():
        for q in params: q.grad = None
        ctx.grad = None
        flow.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).mean().backward()
    def bwd_only():
        fa._flow_backward_batched(flow, U, ctx.detach(), None, None, None, (g, z, None))
    for name, fn in (("forward + backward through autograd", fb), ("_flow_backward_batched alone", bwd_only)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"B={B:5d} {name:40s} host {(t1 - t0) / 50 * 1e3:.2f} ms/call, with the GPU drained {(t2 - t0) / 50 * 1e3:.2f}")
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): bwd_only()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
