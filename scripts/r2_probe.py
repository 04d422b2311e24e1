"""Round-2 probe: per-kernel times of the forward variants on the bench workload (run under
rocprofv3 --kernel-trace --stats to split the hoisted path into projection + chain)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    batches = [int(b) for b in os.environ.get("PROBE_BATCHES", "4096").split(",")]
    for prec in os.environ.get("PROBE_PRECS", "bf16,fp32").split(","):
        for hoist in (False, True):
            flow = bench.build_flow(dev, prec)
            flow.hoist_context = hoist
            flow.freeze_packed()
            for B in batches:
                x, ctx = bench.make_inputs(B, 1, dev)
                nll = torch.empty(B, device=dev)
                for _ in range(5):
                    flow.nll_into(x, ctx, nll)
                torch.cuda.synchronize()
                n = 50
                t0 = time.perf_counter()
                for _ in range(n):
                    flow.nll_into(x, ctx, nll)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"{prec} hoist={hoist} B={B}: {dt * 1e6:.1f} us/call  {B / dt / 1e6:.2f} M samples/s  "
                      f"{B * bench.flops_per_sample() / dt / 1e12:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
