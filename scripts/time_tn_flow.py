"""Timing of the batched transposed GEMM (pf_dense_tn) on the flow's weight-gradient shapes (few rows, many small problems):
python scripts/time_tn_flow.py [rows]; PF_TN_CFG=<n> forces a tile configuration, the split count is swept."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
st = torch.cuda.current_stream().cuda_stream


def run(name, n1, n2, batch, splits, reps=30):
    g = torch.randn(batch, B, n1, device="cuda").bfloat16()
    a = torch.randn(batch, B, n2, device="cuda").bfloat16()
    dw = torch.zeros(batch, n1, n2, device="cuda")
    db = torch.zeros(batch, n1, device="cuda")
    t = _lib.PfDenseTnArgs()
    t.G, t.g_seq_stride, t.ldg, t.A, t.a_seq_stride, t.lda = g.data_ptr(), 0, n1, a.data_ptr(), 0, n2
    t.M, t.rows_per_seq, t.N1, t.N2, t.dW, t.ldw, t.db, t.splits = B, B, n1, n2, dw.data_ptr(), n2, db.data_ptr(), splits
    t.batch, t.g_batch_stride, t.a_batch_stride, t.w_batch_stride, t.b_batch_stride = batch, B * n1, B * n2, n1 * n2, n1
    for _ in range(3):
        _lib.check(L.pf_dense_tn(_lib.PRECISIONS["bf16"], C.byref(t), st), name)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        L.pf_dense_tn(_lib.PRECISIONS["bf16"], C.byref(t), st)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / reps * 1e6
    dw.zero_(); L.pf_dense_tn(_lib.PRECISIONS["bf16"], C.byref(t), st)
    ref = torch.einsum("zmi,zmj->zij", g.float(), a.float())
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:22s} cfg {os.environ.get('PF_TN_CFG', '-'):>2s} splits {splits:3d}: {us:7.1f} us   rel err {err:.1e}", flush=True)


if len(sys.argv) > 3 and sys.argv[3] == "enc":      # the token mixer's weight gradients: 1024 events x 183 tokens
    B = 1024 * 183
    for n1, n2 in ((768, 192), (192, 768), (576, 192), (192, 192), (192, 576)):
        run(f"{n1}x{n2} (mixer)", n1, n2, 1, 0, reps=10)
    sys.exit(0)
for sp in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,4,8,16".split(","))]:
    run("256x256 x10 (W1/W2)", 256, 256, 10, sp)
    run("256x288 x10 (Wc)", 256, 288, 10, sp)
    run("256x16 x10 (W0)", 256, 16, 10, sp)
    run("520x256 x10 (Wf)", 520, 256, 10, sp)
