"""HIP stem alone (4096 events x 3 detectors): ms per call in bf16 and fp32 modes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
dev = torch.device("cuda"); torch.manual_seed(0)
enc = npe.LeanStrainEncoder().to(dev).eval()
strain = torch.randn(4096, 3, 16384, device=dev)
for prec in ("bf16", "fp32"):
    enc.precision = prec
    with torch.no_grad():
        for _ in range(3): enc._stem_hip(strain)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): enc._stem_hip(strain)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{prec}: stem {dt*1e3:.3f} ms per 4096 events")
