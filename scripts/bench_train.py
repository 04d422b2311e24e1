"""Config-4 rehearsal on one GPU: on-GPU remix -> LeanNPE batch_nll -> backward -> AdamW, 1024 examples
per step (the per-GPU share of batch 8192 on 8 GPUs).  Prints ms per phase and examples/s."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, ".")
from posteriflow_amd import npe, train
from posteriflow_amd.remix import RemixDataset, T_LEN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
n_noise, n_ev = 1024, 1024
counts = rng.integers(1, 4, n_ev); m = int(counts.sum())
np.save(f"{tmp}/noise.npy", rng.standard_normal((n_noise, 3, T_LEN), dtype=np.float32).astype(np.float16))
np.save(f"{tmp}/signals.npy", (0.1 * rng.standard_normal((m, 3, T_LEN), dtype=np.float32)).astype(np.float16))
lo = np.array([5, 5, 100, 0, -1.5, 0, 0, 0, -1.2, 0, 0], np.float32)
hi = np.array([80, 60, 1500, 6.2, 1.5, 3.1, 3.1, 6.2, 1.2, 1, 1], np.float32)
np.save(f"{tmp}/params.npy", (lo + (hi - lo) * rng.random((m, 11), dtype=np.float32)))
starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
json.dump({"n_noise": n_noise, "n_signals": m, "events": [[int(a), int(b)] for a, b in zip(starts, counts)]},
          open(f"{tmp}/events.json", "w"))
ds = RemixDataset(tmp, seed=0)
torch.manual_seed(0)
model = npe.LeanNPE().to(dev).train()
model.encoder.precision = prec; model.flow.precision = prec
model.flatten_parameters()
opt = train.make_optimizer(model); sched = train.make_scheduler(opt, 10000)
g = torch.Generator(device="cuda").manual_seed(0)

ROW_CAP = {"exact": "exact", "none": None}.get(os.environ.get("PF_ROW_CAP", "exact"), "exact")     # PF_ROW_CAP=none: 5 rows per event


def phase_times(n):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    for it in range(n):
        e = ev[it]
        idx = torch.randint(0, n_ev, (B,), device=dev, generator=g)
        e[0].record()
        strain, labels, nsig, snr = ds.batch(idx, generator=g)
        e[1].record()
        loss = npe.batch_nll(model, strain, labels, nsig, row_cap=ROW_CAP)
        e[2].record()
        opt.zero_grad(set_to_none=True); loss.backward()
        e[3].record()
        torch.nn.utils.clip_grad_norm_(model.parameters(), train.GRAD_CLIP); opt.step(); sched.step()
        e[4].record()
    torch.cuda.synchronize()
    return np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in ev]), float(loss)

phase_times(3)
t0 = time.perf_counter(); t, loss = phase_times(10); wall = (time.perf_counter() - t0) / 10
med = np.median(t, 0)
print(f"B={B} {prec}: remix {med[0]:.2f} ms, forward {med[1]:.2f} ms, backward {med[2]:.2f} ms, clip+AdamW {med[3]:.2f} ms; "
      f"wall {wall*1e3:.1f} ms/step = {B/wall:.0f} examples/s (loss {loss:.3f})", flush=True)
