"""Stage timestamps of the incremental inverse (PF_INC_TRACE=1): one launch of n draws, trace printed by the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16").freeze_packed()
ctx1 = torch.randn(1, 288, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
z = torch.randn(n, 15, device=dev)
with torch.no_grad():
    os.environ.pop("PF_INC_TRACE", None)
    flow.inverse(z, ctx1); torch.cuda.synchronize()
    os.environ["PF_INC_TRACE"] = "1"
    flow.inverse(z, ctx1); torch.cuda.synchronize()
