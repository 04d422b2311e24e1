"""Geometry features of the coherent encoder (pf_geom_features) against the tensor-op (rocFFT) route: ms per batch."""
import sys, time
import torch
sys.path.insert(0, ".")
from posteriflow_amd import npe

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
enc = npe.CoherentEncoder(context_dim=256, psd_bands=16).cuda().eval()
x = torch.randn(B, 3, 16384, device="cuda")


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    hip = timed(lambda: enc._geometry_rel(x))
    enc.__dict__["_geom_plan"] = False
    ops = timed(lambda: enc._geometry_rel(x))
print(f"B={B}: pf_geom_features {hip:.3f} ms ({B / hip * 1e3:.0f} events/s), tensor ops + rocFFT {ops:.3f} ms", flush=True)
