import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import flow_inputs, make_pair
for (D, C, H, L, K, tb, B) in [(11, 288, 256, 3, 16, 5.0, 100), (4, 0, 64, 2, 8, 3.0, 50), (15, 288, 256, 12, 16, 5.0, 64), (2, 7, 128, 2, 5, 3.0, 33)]:
    ref, ref64, flow = make_pair(D, C, H, L, K, tb)
    flow.precision = "bf16"
    torch.manual_seed(1)
    z = torch.randn(B, D)
    ctx = torch.randn(B, C) if C else None
    with torch.no_grad():
        want, ldw = ref64.inverse_raw(z.double(), None if ctx is None else ctx.double())
        flow.incremental_inverse = False
        x0, ld0, f0 = flow._inverse_call(z.cuda(), None if ctx is None else ctx.cuda(), B)
        flow.incremental_inverse = None
        x1, ld1, f1 = flow._inverse_call(z.cuda(), None if ctx is None else ctx.cuda(), B)
    e = lambda a, b: (a.cpu().double() - b).abs().max().item()
    print(f"D={D} C={C} H={H} L={L}: dpass-vs-oracle {e(x0, want):.2e} inc-vs-oracle {e(x1, want):.2e} inc-vs-dpass {e(x1, x0.cpu().double()):.2e} "
          f"| logdet {e(ld0, ldw):.2e} {e(ld1, ldw):.2e} flags {int(f0.sum())} {int(f1.sum())}", flush=True)
