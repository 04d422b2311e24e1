"""PF_MID_TRACE build: a few launches at [rows]; the library prints the spans of one wave of workgroup 0"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16")
x, ctx = bench.make_inputs(rows, 3, dev)
with torch.no_grad():
    for _ in range(4):
        flow.compute_psd_aware_nll(x, ctx, None)
torch.cuda.synchronize()
