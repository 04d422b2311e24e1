"""bf16-mode backward against fp32-mode backward: layer inputs kept by the two forward kernels, then the gradients."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import flow_inputs, make_pair
from posteriflow_amd import _flow_autograd as fa
for (D, C, H, L, K, B) in ((7, 40, 128, 2, 16, 200), (4, 0, 64, 3, 8, 200), (11, 288, 256, 3, 16, 200)):
    ref, _, flow = make_pair(D, C, H, L, K, 5.0)
    x, ctx = flow_inputs(B, D, C, 5.0)
    xg = x.cuda(); cg = ctx.cuda() if C else None
    Us, zs = {}, {}
    for prec in ("fp32", "bf16"):
        flow.precision = prec
        U = torch.empty(L, B, D, device="cuda")
        with torch.no_grad():
            z, ld, nll = flow._forward_call(xg, cg, None, layer_inputs=U)
        Us[prec], zs[prec] = U, z
    dU = (Us["bf16"] - Us["fp32"]).abs()
    print(f"[D{D} C{C} H{H} L{L}] |U_bf16 - U_fp32| per layer max", " ".join(f"{dU[l].max():.1e}" for l in range(L)),
          " rows with |dU| > 0.1:", int((dU.amax(dim=(0, 2)) > 0.1).sum()), f" |dz| max {(zs['bf16'] - zs['fp32']).abs().max():.1e}")
    gz = torch.randn(B, D, device="cuda"); gl = torch.randn(B, device="cuda")
    # same U (the fp32 one), both chains
    out = {}
    for prec in ("fp32", "bf16"):
        flow.precision = prec
        out[prec] = fa._flow_backward_batched(flow, Us["fp32"], cg, gz, gl)
    # bf16 U, fp32 chain: what the inputs kept by the bf16 forward do to the gradients
    flow.precision = "fp32"
    out["bf16U"] = fa._flow_backward_batched(flow, Us["bf16"], cg, gz, gl)
    for k in ("g_x", "W0", "Wf"):
        for tag in ("bf16", "bf16U"):
            a, b = out["fp32"][k], out[tag][k]
            rel = ((a - b).abs().max() / a.abs().max()).item()
            cos = torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
            print(f"    {k} {tag}: rel {rel:.2e} cos {cos:.6f}")
