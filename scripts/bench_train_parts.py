"""Where a training step's time goes: encoder fwd/bwd (stem vs the rest) and flow fwd/bwd, B examples."""
import sys, time, torch
sys.path.insert(0, ".")
from posteriflow_amd import npe
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda"); torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train(); model.encoder.precision = prec; model.flow.precision = prec
enc, flow = model.encoder, model.flow
strain = torch.randn(B, 3, 16384, device=dev)

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def stem_fb():
    clean = enc._sanitize(strain)
    with enc._autocast(dev):
        tok = enc._stem(clean)
    tok.float().square().mean().backward()
def stem_f():
    with torch.no_grad(), enc._autocast(dev): enc._stem(enc._sanitize(strain))
def enc_fb():
    enc(strain).square().mean().backward()
def enc_f():
    with torch.no_grad(): enc(strain)
n_rows = 2 * B
ctx = torch.randn(n_rows, flow.context_features, device=dev, requires_grad=True)
x = (torch.rand(n_rows, flow.features, device=dev) * 2 - 1)
def flow_fb():
    for q in flow.parameters(): q.grad = None          # as optimizer.zero_grad(set_to_none=True): no accumulation kernels
    ctx.grad = None
    flow.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).mean().backward()
def flow_f():
    with torch.no_grad(): flow.compute_psd_aware_nll(x, ctx, torch.zeros_like(x))
print(f"B={B} {prec}: stem(tensor-op) fwd {timed(stem_f):.2f} fwd+bwd {timed(stem_fb):.2f} | encoder fwd(HIP stem) {timed(enc_f):.2f} "
      f"fwd+bwd {timed(enc_fb):.2f} | flow ({n_rows} rows) fwd {timed(flow_f):.2f} fwd+bwd {timed(flow_fb):.2f} ms", flush=True)
