"""LayerNorm forward / backward row kernels (pf_enc_ln_*) alone: microseconds and bytes per second at [rows] rows x 192."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 183
E = 192
dev = "cuda"
x = torch.randn(M, E, device=dev); dy = torch.randn(M, E, device=dev).bfloat16(); dres = torch.randn(M, E, device=dev)
y = torch.empty(M, E, device=dev, dtype=torch.bfloat16); dx = torch.empty(M, E, device=dev); gout = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
g = torch.randn(E, device=dev); b = torch.randn(E, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dg = torch.zeros(E, device=dev); db = torch.zeros(E, device=dev)
a = _lib.PfLnArgs()
a.x, a.gamma, a.beta, a.M, a.y, a.mean, a.rstd = x.data_ptr(), g.data_ptr(), b.data_ptr(), M, y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
a.dy, a.dres, a.dx, a.gout, a.dgamma, a.dbeta = dy.data_ptr(), dres.data_ptr(), dx.data_ptr(), gout.data_ptr(), dg.data_ptr(), db.data_ptr()
a.drop_p, a.seed, a.site = 0.05, 1, 2
s = torch.cuda.current_stream().cuda_stream
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
f = t(lambda: _lib.check(L.pf_enc_ln_forward(1, C.byref(a), s), "f"))
bw = t(lambda: _lib.check(L.pf_enc_ln_backward(1, C.byref(a), s), "b"))
print(f"{M} rows: ln_fwd {f:.1f} us = {M * E * (4 + 2) / f / 1e6:.2f} TB/s; ln_bwd {bw:.1f} us = {M * E * (4 + 2 + 4 + 4 + 2) / bw / 1e6:.2f} TB/s")
