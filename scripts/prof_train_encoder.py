"""Kernel mix of the encoder's training step (tensor-op transformer, bf16 autocast): 4 x forward + backward, B = 1024."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda"); torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train(); model.encoder.precision = "bf16"
enc = model.encoder
strain = torch.randn(B, 3, 16384, device=dev)
for _ in range(4):
    enc(strain).square().mean().backward()
torch.cuda.synchronize()
