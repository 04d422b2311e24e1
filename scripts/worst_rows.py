import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_pair, flow_inputs
D, C, H, L, K, tb, B = 15, 288, 256, 8, 16, 5.0, 1024
ref, ref64, flow = make_pair(D, C, H, L, K, tb)
x, ctx = flow_inputs(B, D, C, tb)
with torch.no_grad():
    nll64 = ref64.compute_psd_aware_nll(x.double(), ctx.double(), torch.zeros_like(x).double())
    nll32 = ref.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).double()
    nll = flow.compute_psd_aware_nll(x.cuda(), ctx.cuda(), torch.zeros_like(x).cuda()).cpu().double()
    z64, ld64 = ref64(x.double(), ctx.double())
e_gpu = (nll - nll64).abs() / nll64.abs().clamp_min(1)
e_cpu = (nll32 - nll64).abs() / nll64.abs().clamp_min(1)
for name, e in (("gpu", e_gpu), ("cpu32", e_cpu)):
    top = e.topk(5)
    print(name, [f"{i}:{v:.1e}" for v, i in zip(top.values.tolist(), top.indices.tolist())], "median %.1e" % e.median())
i = e_gpu.argmax().item()
print("worst gpu row", i, "x", x[i].tolist(), "nll64", nll64[i].item(), "ld64", ld64[i].item(), "|z|max", z64[i].abs().max().item())
print("corr of log errors", torch.corrcoef(torch.stack([e_gpu.log(), e_cpu.log()]))[0,1].item())
