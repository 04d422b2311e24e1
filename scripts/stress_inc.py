"""Random shapes: incremental inverse against the D-pass kernel in both precisions (no oracle: kernel vs kernel)."""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posteriflow_amd import NSFPosteriorFlow
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    H = rng.choice([64, 128, 192, 256]); D = rng.randint(2, min(16, H // 16)); C = rng.choice([0, 1, 7, 33, 288])
    K = rng.choice([2, 5, 8, 13, 16]); L = rng.randint(1, 4); tb = rng.choice([1.0, 3.0, 5.0])
    groups = rng.choice([1, 1, 2, 3, 5]); per = rng.choice([1, 7, 16, 33, 100]); B = groups * per
    torch.manual_seed(it)
    flow = NSFPosteriorFlow(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=False).cuda()
    with torch.no_grad():
        for p in flow.parameters():
            if p.dim() == 2 and p.shape[0] == D * (3 * K - 1): p.mul_(1.5)
    z = torch.randn(B, D, device="cuda") * 1.3
    ctx = torch.randn(groups if rng.random() < 0.5 else B, C, device="cuda") if C else None
    for prec, tol in (("fp32", 2e-3), ("bf16", 0.5)):
        flow.precision = prec
        flow.incremental_inverse = None
        with torch.no_grad():
            x1, l1 = flow.inverse(z, ctx)
            used = flow.incremental_inverse is None
            flow.incremental_inverse = False
            x0, l0 = flow.inverse(z, ctx)
        ok = torch.isfinite(x1).all() and torch.isfinite(l1).all()
        dx = (x1 - x0).abs()
        q = dx.flatten().quantile(0.9).item()
        if not ok or q > tol:
            bad += 1
            print(f"BAD it={it} D={D} C={C} H={H} K={K} L={L} B={B} groups={groups} {prec}: finite={bool(ok)} q90={q:.3e} max={dx.max():.3e}", flush=True)
    if it % 10 == 9: print(f"{it + 1} shapes done, bad={bad}, last used_inc={used}", flush=True)
print("bad", bad)
