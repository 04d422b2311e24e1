"""time only: the kernel pf_flow_forward picks at [rows] (library from $PF_LIBPFHIP)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda")
flow = bench.build_flow(dev, "bf16")
x, ctx = bench.make_inputs(rows, 3, dev)
out = torch.empty(rows, device=dev)
launch = flow.bind_nll(x, ctx, out)
for _ in range(20): launch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): launch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"{os.environ.get('PF_LIBPFHIP', 'default').split('/')[-1]} {flow.forward_kernel_name(rows)} {rows} rows: {dt*1e6:.1f} us")
