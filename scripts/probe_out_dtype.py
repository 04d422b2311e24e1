"""does torch's library GEMM take bf16 operands with an fp32 result on this ROCm build? (bmm / baddbmm / mm / addmm out_dtype)"""
import time, torch
dev = torch.device("cuda")
L, B, H = 10, 2048, 256
a = torch.randn(L, B, H, device=dev); w = torch.randn(L, H, H, device=dev); b = torch.randn(L, H, device=dev)
ab, wb = a.bfloat16(), w.bfloat16()
ref = torch.baddbmm(b[:, None, :], a, w.transpose(1, 2))
def t(name, f):
    try:
        out = f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): out = f()
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item() if out.shape == ref.shape else float("nan")
        print(f"{name}: ok dtype {out.dtype} {(time.perf_counter() - t0) / 20 * 1e6:.1f} us rel err {err:.2e}")
    except Exception as e:
        print(f"{name}: {type(e).__name__} {str(e)[:160]}")
t("fp32 baddbmm", lambda: torch.baddbmm(b[:, None, :], a, w.transpose(1, 2)))
t("bf16 bmm -> bf16", lambda: torch.bmm(ab, wb.transpose(1, 2)))
t("bf16 bmm out_dtype fp32", lambda: torch.bmm(ab, wb.transpose(1, 2), out_dtype=torch.float32))
t("bf16 baddbmm out_dtype fp32 (fp32 bias)", lambda: torch.baddbmm(b[:, None, :], ab, wb.transpose(1, 2), out_dtype=torch.float32))
t("bf16 baddbmm out_dtype fp32 (fp32 bias expanded)", lambda: torch.baddbmm(b[:, None, :].expand(L, B, H), ab, wb.transpose(1, 2), out_dtype=torch.float32))
g = torch.randn(L, B, H, device=dev).bfloat16()
t("wgrad bf16 bmm G^T A out fp32", lambda: torch.bmm(g.transpose(1, 2), ab, out_dtype=torch.float32))
t("wgrad fp32", lambda: torch.bmm(g.float().transpose(1, 2), a))
c = torch.randn(B, 288, device=dev); cb = c.bfloat16(); W = torch.randn(7680, 288, device=dev); Wb = W.bfloat16(); bb = torch.randn(7680, device=dev)
t("addmm fp32", lambda: torch.addmm(bb, c, W.t()))
t("addmm bf16 out fp32", lambda: torch.addmm(bb, cb, Wb.t(), out_dtype=torch.float32))
t("mm bf16 out fp32", lambda: torch.mm(cb, Wb.t(), out_dtype=torch.float32))
t("cast [L,B,H] fp32->bf16", lambda: a.bfloat16())
t("relu+cast", lambda: torch.relu(a).bfloat16())
