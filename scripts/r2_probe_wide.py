"""Round-2 probe: the large-batch (wide) forward kernel against the 16-row kernel over batch sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    flow = bench.build_flow(dev, "bf16")
    batches = [int(b) for b in os.environ.get("PROBE_BATCHES", "4096,8192,16384,32768,65536,131072").split(",")]
    for B in batches:
        x, ctx = bench.make_inputs(B, 1, dev)
        nll = torch.empty(B, device=dev)
        res = {}
        for name, mb in (("16-row", 1 << 40), ("wide", 1)):
            flow.wide_min_batch = mb
            for _ in range(3):
                flow.nll_into(x, ctx, nll)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                flow.nll_into(x, ctx, nll)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res[name] = nll.double().mean().item()
            print(f"B={B:7d} {name:7s}: {dt * 1e6:9.1f} us  {B / dt / 1e6:7.2f} M samples/s  "
                  f"{B * bench.flops_per_sample() / dt / 1e12:7.1f} TFLOP/s = {B * bench.flops_per_sample() / dt / 2.5e15:.3f} of peak  "
                  f"mean nll {res[name]:.4f}", flush=True)


if __name__ == "__main__":
    main()
