"""Kernel mix of one encoder forward + backward at B events (run under rocprofv3 --kernel-trace --stats)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda"); torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train(); model.encoder.precision = "bf16"
enc = model.encoder
strain = torch.randn(B, 3, 16384, device=dev)
def enc_fb():
    for q in enc.parameters(): q.grad = None
    enc(strain).square().mean().backward()
for _ in range(2): enc_fb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): enc_fb()
torch.cuda.synchronize(); print(f"encoder fwd+bwd B={B}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
