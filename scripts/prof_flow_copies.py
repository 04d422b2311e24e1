"""Which host lines issue the small device-to-device copies of one flow forward + backward (torch profiler, stacks)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda"); torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train(); flow = model.flow
flow.precision = 'bf16'
flow.flatten_parameters()
ctx = torch.randn(2048, flow.context_features, device=dev, requires_grad=True)
x = torch.rand(2048, flow.features, device=dev) * 2 - 1
def fb():
    for p_ in flow.parameters(): p_.grad = None
    ctx.grad = None
    flow.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).mean().backward()
for _ in range(3): fb()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fb(); torch.cuda.synchronize()
ev = prof.events()
names = {}
for e in ev:
    if e.device_type.name == "CUDA" or "emcpy" in e.name or "copy_" in e.name or e.name in ("aten::fill_", "aten::zero_", "aten::mul_"):
        key = e.name[:60]
        names[key] = names.get(key, 0) + 1
for k, v in sorted(names.items(), key=lambda kv: -kv[1]): print(v, k)
print("---- by stack")
for row in prof.key_averages(group_by_stack_n=8):
    if row.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::mul_", "aten::cat", "aten::index"):
        fr = [f for f in row.stack if "posteriflow_amd" in f or "scripts/" in f][:3]
        print(row.count, row.key, fr)
