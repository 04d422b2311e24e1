"""Timing of the strip GEMM (pf_dense_nt) per epilogue on the token mixer's shapes, M = 1024 events x 183 tokens."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_enc_blocks_gpu import pack_matrix, PREC, stream, act_dtype
from posteriflow_amd import _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 183
prec = "bf16"
dt = act_dtype(prec)


def run(name, epi, k, n, p=0.0, reps=20, **kw):
    a = torch.randn(M, k, device="cuda").to(dt)
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    fr = pack_matrix(w, prec)
    out = torch.empty(M, n, dtype=torch.float32 if epi == 2 else dt, device="cuda")
    args = _lib.PfDenseArgs()
    args.A, args.M, args.rows_per_seq, args.lda, args.K, args.N, args.KC = a.data_ptr(), M, M, k, k, n, (k if k <= 256 else 192)
    args.wfrags, args.bias, args.out, args.ldo = fr.data_ptr(), b.data_ptr(), out.data_ptr(), n
    keep = []
    if epi == 1:
        d = torch.empty(M, n, dtype=dt, device="cuda"); args.dact = d.data_ptr(); keep.append(d)
    if epi == 2:
        r = torch.randn(M, n, device="cuda"); args.resid = r.data_ptr(); keep.append(r)
    if epi == 3:
        m = torch.randn(M, n, device="cuda").to(dt); args.mul = m.data_ptr(); keep.append(m)
    args.drop_p, args.seed, args.site = p, 1, 2
    for _ in range(3):
        _lib.check(L.pf_dense_nt(PREC[prec], epi, C.byref(args), stream()), name)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        L.pf_dense_nt(PREC[prec], epi, C.byref(args), stream())
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / reps * 1e6
    flop = 2.0 * M * k * n
    print(f"{name:34s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")


run("plain 192->576 (QKV)", 0, 192, 576)
run("plain 192->768", 0, 192, 768)
run("gelu 192->768 p=0", 1, 192, 768)
run("gelu 192->768 p=0.05", 1, 192, 768, p=0.05)
run("mul 192->768", 3, 192, 768)
run("resid 768->192 p=0", 2, 768, 192)
run("resid 768->192 p=0.05", 2, 768, 192, p=0.05)
run("resid 192->192 p=0.05", 2, 192, 192, p=0.05)
run("plain 768->192", 0, 768, 192)
run("plain 576->192", 0, 576, 192)
run("plain 192->192", 0, 192, 192)
