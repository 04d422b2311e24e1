"""BASELINE config 3 end to end, bf16 only, for a rocprofv3 kernel trace: 8 calls of LeanNPE.nll on 4096 three-detector events."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import recipe
from posteriflow_amd import npe

torch.manual_seed(0)
dev = torch.device("cuda")
model = npe.LeanNPE().to(dev).eval().set_precision("bf16")
B = 4096
strain = torch.randn(B, 3, 16384, device=dev)
theta = torch.stack([recipe.physical_params(64, seed=3)[8:40] for _ in range(B // 32)]).reshape(B, 11).to(dev)
rank = torch.zeros(B, dtype=torch.long, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = model.nll(strain, theta, rank)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        out = model.nll(strain, theta, rank)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
assert torch.isfinite(out).all()
print(f"bf16: LeanNPE.nll batch {B}: {dt*1e3:.2f} ms = {B/dt:.0f} events/s", flush=True)
