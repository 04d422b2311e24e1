"""Round-2 PMC driver: a few launches of the two forward kernels of the bench workload -- the 16-row kernel at
B = 4096 (the headline batch) and the wide kernel at B = 65536 -- for rocprofv3 --pmc / --kernel-trace passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
flow = bench.build_flow(dev, "bf16").freeze_packed()
for B in (4096, 65536):
    x, ctx = bench.make_inputs(B, 1, dev)
    nll = torch.empty(B, device=dev)
    for _ in range(12):
        flow.nll_into(x, ctx, nll)
    torch.cuda.synchronize()
    print(B, flow.forward_kernel_name(B), nll.double().mean().item(), flush=True)
