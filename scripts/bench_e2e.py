"""BASELINE config 3 end to end on one GPU: 3-detector strain [B, 3, 16384] -> LeanStrainEncoder ->
flow log-density of the 11 parameters (LeanNPE.nll), batch 4096, bf16 mode; and the single-event
posterior (sample_event: encode once, 1e5 draws + their log q), the reference's inference use."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import recipe
from posteriflow_amd import npe
from posteriflow_amd.inference import sample_event

torch.manual_seed(0)
dev = torch.device("cuda")
for prec in ("bf16", "fp32"):
    model = npe.LeanNPE().to(dev).eval()
    model.encoder.precision = prec; model.flow.precision = prec
    B = 4096
    strain = torch.randn(B, 3, 16384, device=dev)
    theta = torch.stack([recipe.physical_params(64, seed=3)[8:40] for _ in range(B // 32)]).reshape(B, 11).to(dev)
    rank = torch.zeros(B, dtype=torch.long, device=dev)
    def step():
        with torch.no_grad():
            ck = B if prec == "bf16" else 1024
            return torch.cat([model.nll(strain[i:i + ck], theta[i:i + ck], rank[i:i + ck]) for i in range(0, B, ck)])
    for _ in range(2): out = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    assert torch.isfinite(out).all()
    with torch.no_grad():
        sample_event(model, strain[:1], num_samples=4096, seed=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = sample_event(model, strain[:1], num_samples=100000, seed=0)
        torch.cuda.synchronize(); ts = time.perf_counter() - t0
    print(f"{prec}: LeanNPE.nll batch {B}: {dt*1e3:.1f} ms = {B/dt:.0f} events/s | single event, 1e5 draws + log q: "
          f"{ts*1e3:.0f} ms = {1e5/ts:.0f} draws/s", flush=True)
