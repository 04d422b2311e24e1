"""Timing of the transposed-convolution GEMMs (pf_dense_nt, multiply epilogue over padded gradient images) at 1024 events."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_enc_blocks_gpu import pack_matrix, PREC, stream
from posteriflow_amd import _lib
L = _lib.lib()
NSEQ = 3072
CONV = {2: (32, 64, 16, 4, 2041, 507), 3: (64, 128, 8, 4, 507, 125), 4: (128, 192, 4, 2, 125, 61)}   # cin cout kw s lin lout
GPAD = {2: 514, 3: 128, 4: 64}
DXR = {2: 511, 3: 127, 4: 63}


def run(l, reps=10, kc=None):
    cin, cout, kw, s, lin, lout = CONV[l]
    K, N = (kw // s) * cout, s * cin
    g = torch.randn(NSEQ, GPAD[l], cout, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    fr = pack_matrix(w, "bf16")
    mul = torch.randn(NSEQ, lin, cin, device="cuda").bfloat16()
    out = torch.zeros(NSEQ, lin, cin, device="cuda").bfloat16()
    a = _lib.PfDenseArgs()
    a.A, a.M, a.rows_per_seq, a.a_seq_stride, a.lda = g.data_ptr(), NSEQ * DXR[l], DXR[l], GPAD[l] * cout, cout
    a.K, a.N, a.KC = K, N, kc or (K if K <= 256 else 192)
    a.wfrags, a.out, a.o_seq_stride, a.ldo = fr.data_ptr(), out.data_ptr(), lin * cin, N
    a.o_valid_per_seq, a.x_seq_stride, a.mul = lin * cin, lin * cin, mul.data_ptr()
    for _ in range(2):
        _lib.check(L.pf_dense_nt(PREC["bf16"], 3, C.byref(a), stream()), "dx")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        L.pf_dense_nt(PREC["bf16"], 3, C.byref(a), stream())
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / reps * 1e6
    flop = 2.0 * NSEQ * DXR[l] * K * N
    byts = g.numel() * 2 + mul.numel() * 2 * 2
    # reference: windows of g times w^T, times mul
    win = torch.cat([g[:, j:j + DXR[l]] for j in range(kw // s)], dim=2).float()          # [NSEQ, rows, K]
    ref = (win[:4] @ w.t().bfloat16().float()).reshape(4, -1)[:, :lin * cin] * mul[:4].float().reshape(4, -1)
    got = out[:4].float().reshape(4, -1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"conv{l} dX  KC={a.KC} K={K} N={N} rows={NSEQ * DXR[l]}: {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s  {byts / us / 1e6:5.2f} TB/s  rel err {err:.1e}", flush=True)


for l in (2, 3, 4):
    run(l)
    run(l, kc=128)
    run(l, kc=64)
