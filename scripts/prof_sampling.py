"""Sampling through the incremental inverse for a rocprofv3 kernel trace: 131 072 draws x 10, one context row,
BASELINE config 3 flow (D = 15, L = 8), bf16; then 3 launches of the D-pass inverse for comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16").freeze_packed()
ctx1 = torch.randn(1, 288, device=dev)
z = torch.randn(131072, 15, device=dev)
with torch.no_grad():
    for _ in range(10):
        flow.inverse(z, ctx1)
    flow.incremental_inverse = False
    for _ in range(3):
        flow.inverse(z, ctx1)
torch.cuda.synchronize()
