"""fp32 incremental inverse vs fp32 D-pass inverse, each against the fp64 oracle (config 3 flow, 256 draws) and round trip."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_pair
D, C, H, L, K, tb, B = 15, 288, 256, 8, 16, 5.0, 256
ref, ref64, flow = make_pair(D, C, H, L, K, tb)
g = torch.Generator().manual_seed(5)
z = torch.randn(B, D, generator=g) * 1.2
ctx = torch.randn(B, C, generator=g)
with torch.no_grad():
    x64, ld64 = ref64.inverse_raw(z.double(), ctx.double())
    x32, ld32 = ref.inverse_raw(z, ctx)
    print(f"cpu fp32 oracle: |x-x64| med {(x32.double()-x64).abs().median():.2e} q99 {(x32.double()-x64).abs().flatten().quantile(0.99):.2e} max {(x32.double()-x64).abs().max():.2e} | "
          f"|ld-ld64| med {(ld32.double()-ld64).abs().median():.2e} max {(ld32.double()-ld64).abs().max():.2e}")
    for name, inc in (("inc", None), ("dpass", False)):
        flow.incremental_inverse = inc
        x, ld, flags = flow._inverse_call(z.cuda(), ctx.cuda(), B)
        z2, ldf = flow(x, ctx.cuda())
        ex = (x.cpu().double() - x64).abs(); el = (ld.cpu().double() - ld64).abs()
        rt = (z2.cpu() - z).abs().max(dim=1).values
        print(f"{name:6s}: |x-x64| med {ex.median():.2e} q99 {ex.flatten().quantile(0.99):.2e} max {ex.max():.2e} | |ld-ld64| med {el.median():.2e} max {el.max():.2e} | "
              f"round trip med {rt.median():.2e} q99 {rt.quantile(0.99):.2e} | |ld_f+ld_i| q99 {(ldf+ld).abs().quantile(0.99):.2e}")
