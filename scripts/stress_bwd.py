"""Random shapes through the bf16 backward kernels: pf_flow_reevaluate against same-rounding tensor ops (tight) and the
bf16 chain against the fp32 chain on identical activations.  usage: stress_bwd.py [seed] [n]"""
import os, sys, random, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import NSFPosteriorFlow, _flow_autograd as fa
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
rb = lambda t: t.bfloat16().float()
worst_re, worst_ch, fails = 0.0, 0.0, 0
for it in range(n):
    H = rng.choice([64, 128, 192, 256])
    D = rng.randint(1, min(16, H // 16))
    C = rng.choice([0, 0, rng.randint(1, 40), rng.randint(41, 300), 288])
    if C > 0 and C % D == 0:
        C += 1                                         # keep the plain (GLU) conditioner
    K, L, B = rng.randint(2, 16), rng.randint(1, 3), rng.choice([1, 7, 16, 33, 200, 257])
    torch.manual_seed(seed * 1000 + it)
    flow = NSFPosteriorFlow(D, C, H, L, K, 4.0, use_masked_context=False).cuda()
    with torch.no_grad():
        for layer in flow._ar_transforms:
            net = layer.autoregressive_net
            net.final_layer.weight.mul_(3.0)
            for blk in net.blocks:
                blk.linear_layers[1].weight.mul_(50.0)
    x = (torch.rand(B, D, device="cuda") * 2 - 1) * 4.4
    ctx = torch.randn(B, C, device="cuda") if C else None
    tag = f"D{D} C{C} H{H} K{K} L{L} B{B}"
    try:
        with torch.no_grad():
            flow.precision = "bf16"
            U = torch.empty(L, B, D, device="cuda")
            flow._forward_call(x, ctx, None, layer_inputs=U)
            HS, T1, T2, G, PC, H2, params = fa._reevaluate_hip(flow, U, ctx)
            lin = lambda m, v, rnd=True: F.linear(rb(v) if rnd else v, rb(m.weight * m.mask), m.bias)
            clin = lambda m, v: F.linear(rb(v), rb(m.weight), m.bias)
            err = 0.0
            def chk(got, want):
                global err
                err = max(err, ((got - want).abs().max() / want.abs().max().clamp_min(1e-6)).item())
            for l, layer in enumerate(flow._ar_transforms):
                net = layer.autoregressive_net
                h = lin(net.initial_layer, U[l], False)
                if C:
                    chk(PC[l], clin(net.context_layer, ctx)); h = h + F.relu(PC[l])
                chk(HS[0, l], h)
                for j, blk in enumerate(net.blocks):
                    h = HS[j, l]
                    chk(T1[j, l], lin(blk.linear_layers[0], F.relu(h)))
                    t2 = lin(blk.linear_layers[1], F.relu(T1[j, l]))
                    if C:
                        chk(T2[j, l], t2); chk(G[j, l], torch.sigmoid(clin(blk.context_layer, ctx)))
                        nxt = h + T2[j, l] * G[j, l]
                    else:
                        nxt = h + t2
                    chk(HS[j + 1, l] if j == 0 else H2[l], nxt)
                chk(params[l], lin(net.final_layer, H2[l]))
            gz, gl = torch.randn(B, D, device="cuda"), torch.randn(B, device="cuda")
            fa.REEVAL_HIP = False
            out = {}
            for prec in ("fp32", "bf16"):
                flow.precision = prec
                out[prec] = fa._flow_backward_batched(flow, U, ctx, gz, gl)
            fa.REEVAL_HIP = True
            cerr = 0.0
            for k, v in out["fp32"].items():
                if v is None: continue
                for a, b in zip(v if isinstance(v, list) else [v], out["bf16"][k] if isinstance(v, list) else [out["bf16"][k]]):
                    assert torch.isfinite(b).all(), k
                    cerr = max(cerr, ((a - b).abs().max() / a.abs().max().clamp_min(1e-12)).item())
        ok = err < 2e-3 and cerr < 3e-2
        worst_re, worst_ch = max(worst_re, err), max(worst_ch, cerr)
        if not ok:
            fails += 1
        print(f"{'ok  ' if ok else 'FAIL'} {tag:32s} reeval {err:.1e}  chain bf16 vs fp32 {cerr:.1e}", flush=True)
    except Exception as e:
        fails += 1
        print(f"EXC  {tag}: {type(e).__name__} {str(e)[:150]}", flush=True)
print(f"{n} shapes, {fails} failures; worst reeval {worst_re:.1e}, worst chain {worst_ch:.1e}")
sys.exit(1 if fails else 0)
