"""One LeanStrainEncoder forward over 2048 events (chunks of 512) for a rocprofv3 kernel trace."""
import sys, torch
sys.path.insert(0, ".")
from posteriflow_amd import npe
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
enc = npe.LeanStrainEncoder().cuda().eval(); enc.precision = prec
strain = torch.randn(2048, 3, 16384, device="cuda")
with torch.no_grad():
    for _ in range(3):
        [enc(strain[i:i + 512]) for i in range(0, 2048, 512)]
torch.cuda.synchronize()
