"""Host (enqueue) time of one training step by phase, without synchronising inside the step: if the sum is close to the
measured step time the step is host-bound, whatever the kernels take.  usage: prof_host_step.py [events] [precision]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe, train
from posteriflow_amd.remix import synthetic_dataset
events = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda")
ds = synthetic_dataset(dev, n_noise=512, n_events=512, seed=0)
torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train().set_precision(prec).flatten_parameters()
opt = train.make_optimizer(model); sched = train.make_scheduler(opt, 10000)
g = torch.Generator(device=dev).manual_seed(0)
names = ["remix", "batch_nll", "zero+backward", "clip", "adamw+sched"]
acc = [0.0] * len(names)
def step(record):
    t = [time.perf_counter()]
    idx = torch.randint(0, ds.n_events, (events,), device=dev, generator=g)
    strain, labels, nsig, _ = ds.batch(idx, generator=g); t.append(time.perf_counter())
    total, count = npe.batch_nll(model, strain, labels, nsig, row_cap="exact", reduction="sum"); loss = total / count.clamp_min(1.0); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); loss.backward(); t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(model.parameters(), train.GRAD_CLIP); t.append(time.perf_counter())
    opt.step(); sched.step(); t.append(time.perf_counter())
    if record:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"events {events} {prec}: host enqueue {host / N * 1e3:.2f} ms/step, wall {wall / N * 1e3:.2f} ms/step")
print("  " + ", ".join(f"{n} {a / N * 1e3:.2f}" for n, a in zip(names, acc)))
