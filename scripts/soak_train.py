"""Soak: N training steps on the synthetic dataset; the loss must stay finite and go down, nothing may fault."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe, train
from posteriflow_amd.remix import synthetic_dataset
steps, events = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda")
ds = synthetic_dataset(dev, n_noise=256, n_events=512, seed=0)
torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train().set_precision(os.environ.get("PF_SOAK_PREC", "bf16")).flatten_parameters()
opt = train.make_optimizer(model); sched = train.make_scheduler(opt, 2000, warmup_steps=20) if "warmup_steps" in train.make_scheduler.__code__.co_varnames else train.make_scheduler(opt, 2000)
g = torch.Generator(device=dev).manual_seed(0)
losses = []
t0 = time.perf_counter()
RAGGED = [1, 7, 33, 100, 183, 256, 333, 500] if os.environ.get("PF_SOAK_RAGGED") else None      # event counts cycled step by step
for it in range(steps):
    if RAGGED:
        events = RAGGED[it % len(RAGGED)]
    idx = torch.randint(0, ds.n_events, (events,), device=dev, generator=g)
    strain, labels, nsig, _ = ds.batch(idx, generator=g)
    out = train.train_step(model, opt, sched, strain, labels, nsig, sync=(it % 20 == 19))
    if it % 20 == 19:
        losses.append(out["loss"])
        print(f"step {it + 1}: loss {out['loss']:.3f} grad norm {out['grad_norm']:.2f}  ({(time.perf_counter() - t0) / (it + 1) * 1e3:.1f} ms/step)", flush=True)
assert all(l == l and abs(l) < 1e6 for l in losses), losses
assert losses[-1] < losses[0], losses
print("soak ok", losses[0], "->", losses[-1])
