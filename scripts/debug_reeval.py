import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.nn.functional as F
from helpers import flow_inputs, make_pair
from posteriflow_amd import _flow_autograd as fa
D, C, H, L, K = 4, 0, 64, 1, 8
_, _, flow = make_pair(D, C, H, L, K, 5.0, scale=2.0)
flow.precision = "bf16"
B = 16
x, ctx = flow_inputs(B, D, C, 5.0)
xg = x.cuda()
U = torch.empty(L, B, D, device="cuda")
rb = lambda t: t.bfloat16().float()
with torch.no_grad():
    flow._forward_call(xg, None, None, layer_inputs=U)
    HS, T1, T2, G, PC, H2, params = fa._reevaluate_hip(flow, U, None)
    net = flow._ar_transforms[0].autoregressive_net
    W = rb(net.final_layer.weight * net.final_layer.mask)
    want = F.linear(rb(H2[0]), W, net.final_layer.bias)
    err = (params[0] - want).abs()
    M = 3 * K - 1
    print("err by output index p (max over rows):")
    e = err.max(0).values.cpu()
    for t in range((D * M + 15) // 16):
        print(t, " ".join(f"{v:.1e}" for v in e[16 * t:16 * t + 16].tolist()))
    # which k-steps are missing?  partial sums
    for ks_drop in range(H // 32):
        Wd = W.clone(); Wd[:, 32 * ks_drop:32 * ks_drop + 32] = 0
        w2 = F.linear(rb(H2[0]), Wd, net.final_layer.bias)
        print("dropping k-step", ks_drop, "max err", (params[0] - w2).abs().max().item())
    w3 = F.linear(rb(F.relu(H2[0])), W, net.final_layer.bias)
    print("with relu(h):", (params[0] - w3).abs().max().item())
    w4 = F.linear(rb(HS[1, 0]), W, net.final_layer.bias)
    print("with h1 (before the last block):", (params[0] - w4).abs().max().item())
    w5 = F.linear(rb(F.relu(T1[1, 0])), W, net.final_layer.bias)
    print("with relu(t1_1):", (params[0] - w5).abs().max().item())
    torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
    d = (params[0] - want).cpu()
    print("diff rows x cols 24..40:\n", d[:, 24:40])
    print("bias 24..40", net.final_layer.bias[24:40].cpu())
    nob = F.linear(rb(H2[0]), W)
    print("params - (W h) cols 24..40 row 0..2:\n", (params[0] - nob).cpu()[:3, 24:40])
