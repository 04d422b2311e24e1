"""GEMM fuzz: pf_dense_nt (plain epilogue, optional split reduction over slab chunks) and pf_dense_tn (random explicit split
counts, batches) at random ragged shapes against float64 matmuls.  python scripts/fuzz_gemm.py [seed] [n]"""
import ctypes as C, math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_enc_blocks_gpu as T

seed, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
rnd = random.Random(seed)
lib, L = T._L()
fails = 0
for it in range(n):
    prec = rnd.choice(["bf16", "fp32"])
    g = torch.Generator().manual_seed(seed * 1000 + it)
    # ---- NT: out[M, N] = A[M, K] W[N, K]^T (+ bias)
    m, k, nn = rnd.choice([1, 5, 16, 127, 128, 129, rnd.randint(1, 3000)]), 64 * rnd.randint(1, 12), 16 * rnd.randint(1, 48)
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(nn, k, generator=g) / math.sqrt(k)).cuda()
    b = torch.randn(nn, generator=g).cuda()
    kc = k if k <= 256 else rnd.choice([c for c in (64, 128, 192, 256) if k % c == 0])
    try:
        got = T.dense_nt(prec, 0, a.to(T.act_dtype(prec)).contiguous(), w, b, kc=kc, out_f32=True)
        want = T.rnd(a, prec).double() @ T.rnd(w, prec).double().t() + b.double()
        e = (got.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-9)
        ok = bool(torch.isfinite(got).all()) and e < 1e-4
    except NotImplementedError:
        ok, e = True, float("nan")            # PF_ERR_UNSUPPORTED: refused, not wrong
    fails += 0 if ok else 1
    print(("ok   " if ok else "FAIL ") + f"nt {prec} {m}x{k}x{nn} kc {kc}: {e:.1e}", flush=True)
    # ---- NT with a split reduction over slabs (the flow's context gradient): A = [slabs][M][H], out += sum over slabs
    H, slabs = rnd.choice([64, 128, 256]), rnd.randint(1, 36)
    m2, n2 = rnd.choice([1, 100, 517, rnd.randint(1, 2500)]), 16 * rnd.randint(1, 12)
    A3 = torch.randn(slabs, m2, H, generator=g).cuda()
    W3 = (torch.randn(n2, slabs * H, generator=g) / math.sqrt(slabs * H)).cuda()
    Ad = A3.to(T.act_dtype(prec)).contiguous()
    frags = T.pack_matrix(W3, prec)
    splits = rnd.randint(1, slabs)
    out = torch.zeros(m2, n2, device="cuda")
    args = lib.PfDenseArgs()
    args.A, args.M, args.rows_per_seq, args.a_seq_stride, args.lda = Ad.data_ptr(), m2, m2, 0, H
    args.K, args.N, args.KC, args.a_chunk_stride, args.a_slab_chunks = slabs * H, n2, H, m2 * H, 1
    args.wfrags, args.out, args.o_seq_stride, args.ldo, args.out_f32, args.k_splits = frags.data_ptr(), out.data_ptr(), 0, n2, 1, splits
    lib.check(L.pf_dense_nt(T.PREC[prec], 0, C.byref(args), T.stream()), "pf_dense_nt split")
    want = (T.rnd(A3, prec).double().permute(1, 0, 2).reshape(m2, slabs * H) @ T.rnd(W3, prec).double().t())
    e = (out.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-9)
    ok = bool(torch.isfinite(out).all()) and e < 1e-4
    fails += 0 if ok else 1
    print(("ok   " if ok else "FAIL ") + f"nt-split {prec} {slabs} slabs x {m2}x{H} -> {n2}, {splits} splits: {e:.1e}", flush=True)
    # ---- TN: dW[N1, N2] += G[M, N1]^T A[M, N2], db += column sums of G; batches, explicit splits
    m3, n1, n2b, nb = rnd.choice([1, 33, 97, rnd.randint(1, 6000)]), 8 * rnd.randint(1, 96), 8 * rnd.randint(1, 96), rnd.choice([1, 1, 3])
    G = torch.randn(nb, m3, n1, generator=g).cuda()
    A = torch.randn(nb, m3, n2b, generator=g).cuda()
    Gd, Ad = G.to(T.act_dtype(prec)).contiguous(), A.to(T.act_dtype(prec)).contiguous()
    dW, db = torch.zeros(nb, n1, n2b, device="cuda"), torch.zeros(nb, n1, device="cuda")
    t = lib.PfDenseTnArgs()
    t.G, t.g_seq_stride, t.ldg, t.A, t.a_seq_stride, t.lda = Gd.data_ptr(), 0, n1, Ad.data_ptr(), 0, n2b
    t.M, t.rows_per_seq, t.N1, t.N2, t.dW, t.ldw, t.db = m3, m3, n1, n2b, dW.data_ptr(), n2b, db.data_ptr()
    t.splits = rnd.choice([0, 0, 1, rnd.randint(1, 80)])
    t.batch, t.g_batch_stride, t.a_batch_stride, t.w_batch_stride, t.b_batch_stride = nb, m3 * n1, m3 * n2b, n1 * n2b, n1
    lib.check(L.pf_dense_tn(T.PREC[prec], C.byref(t), T.stream()), "pf_dense_tn")
    want = T.rnd(G, prec).double().transpose(1, 2) @ T.rnd(A, prec).double()
    e1 = (dW.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-9)
    e2 = (db.double() - T.rnd(G, prec).double().sum(1)).abs().max().item()
    ok = e1 < 2e-5 and e2 < 1e-3 * math.sqrt(m3)
    fails += 0 if ok else 1
    print(("ok   " if ok else "FAIL ") + f"tn {prec} batch {nb} {m3}x{n1}x{n2b} splits {t.splits}: {e1:.1e} bias {e2:.1e}", flush=True)
print(f"{fails} failures")
sys.exit(1 if fails else 0)
