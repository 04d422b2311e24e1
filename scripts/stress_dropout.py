"""Random shapes through the training forward with dropout (flow_train_kernel, fp32 mode): its z / log|det| against the
tensor-op evaluation with the factors pf_flow_dropout_mask returns for the same seed.  usage: stress_dropout.py [seed] [n]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import NSFPosteriorFlow, _flow_autograd as fa
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
fails, worst = 0, 0.0
for it in range(n):
    H = rng.choice([64, 128, 192, 256])
    D = rng.randint(1, min(16, H // 16))
    C = rng.choice([0, rng.randint(1, 40), rng.randint(41, 288), 288])
    masked = C > 0 and C % D == 0 and rng.random() < 0.5
    if C > 0 and C % D == 0 and not masked:
        C += 1
    if C > 288 and H < 256:
        C = 288 - (288 % D == 0)
    K, L, B = rng.randint(2, 16), rng.randint(1, 3), rng.choice([1, 7, 16, 33, 200, 257])
    p = rng.choice([0.05, 0.15, 0.5])
    torch.manual_seed(seed * 1000 + it)
    flow = NSFPosteriorFlow(D, C, H, L, K, 4.0, dropout=p, use_masked_context=masked).cuda().train()
    with torch.no_grad():
        for layer in flow._ar_transforms:
            for blk in layer.autoregressive_net.blocks:
                blk.linear_layers[1].weight.mul_(50.0)
    x = (torch.rand(B, D, device="cuda") * 2 - 1) * 4.4
    ctx = torch.randn(B, C, device="cuda") if C else None
    tag = f"D{D} C{C} H{H} K{K} L{L} B{B} p{p} {'masked-ctx' if masked else ''}"
    try:
        with torch.no_grad():
            sd = rng.getrandbits(62)
            z, ld, _ = flow._forward_call(x, ctx, None, dropout_seed=sd)
            drop = fa.dropout_mask(flow, B, sd, x.device)
            cx = flow._permute_context_blocks(ctx) if (ctx is not None and masked) else ctx
            zr, ldr = fa.flow_forward(flow, x, cx, drop)
            flow.eval(); ze, _, _ = flow._forward_call(x, ctx, None); flow.train()
        ez, el = (z - zr).abs().max().item(), (ld - ldr).abs().max().item()
        moved = (z - ze).abs().max().item()
        ok = ez < 5e-4 and el < 5e-3 and (moved > 10 * ez or L * H < 128)
        worst = max(worst, ez)
        fails += not ok
        print(f"{'ok  ' if ok else 'FAIL'} {tag:44s} |dz| {ez:.1e} |dld| {el:.1e} dropout moves z by {moved:.1e}", flush=True)
    except Exception as e:
        fails += 1
        print(f"EXC  {tag}: {type(e).__name__} {str(e)[:160]}", flush=True)
print(f"{n} shapes, {fails} failures; worst |dz| {worst:.1e}")
sys.exit(1 if fails else 0)
