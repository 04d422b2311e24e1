"""PMC driver: a few launches of the forward kernels of the bench workload -- the 16-row kernel at B = 4096 (the headline
batch), the mid-batch kernel at B = 16384 (round 4) and the large-batch kernel at B = 65536 -- for rocprofv3 --pmc /
--kernel-trace passes.  PMC_ROWS=4096,16384 restricts the sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
flow = bench.build_flow(dev, "bf16").freeze_packed()
for B in [int(v) for v in os.environ.get("PMC_ROWS", "4096,16384,65536").split(",")]:
    x, ctx = bench.make_inputs(B, 1, dev)
    nll = torch.empty(B, device=dev)
    for _ in range(12):
        flow.nll_into(x, ctx, nll)
    torch.cuda.synchronize()
    print(B, flow.forward_kernel_name(B), nll.double().mean().item(), flush=True)
