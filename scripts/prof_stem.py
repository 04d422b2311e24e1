"""HIP stem alone, 4096 events x 3 detectors, bf16: for a rocprofv3 kernel trace (per-layer times)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
enc = npe.LeanStrainEncoder().cuda().eval(); enc.precision = "bf16"
strain = torch.randn(4096, 3, 16384, device="cuda")
with torch.no_grad():
    for _ in range(5): enc._stem_hip(strain)
torch.cuda.synchronize()
