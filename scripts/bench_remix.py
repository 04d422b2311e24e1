"""Remix assembly throughput (pf_remix_forward): examples/s and HBM GB/s against the ~8 TB/s peak.
Synthetic pools (2048 noise rows, 4096 signal rows = 1.2 GB fp16), 1024 and 8192 examples per call,
2 signals per example on average (algorithmic bytes: 98 KB fp16 per pool row read + 196 KB fp32 out)."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd.remix import RemixDataset, T_LEN

dev = torch.device("cuda")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
n_noise, n_ev = 2048, 2048
counts = rng.integers(1, 4, n_ev)                   # 1..3 signals, mean 2
m = int(counts.sum())
np.save(f"{tmp}/noise.npy", rng.standard_normal((n_noise, 3, T_LEN), dtype=np.float32).astype(np.float16))
np.save(f"{tmp}/signals.npy", (0.1 * rng.standard_normal((m, 3, T_LEN), dtype=np.float32)).astype(np.float16))
par = np.zeros((m, 11), np.float32); par[:, 0] = 30; par[:, 1] = 20; par[:, 2] = rng.uniform(100, 1500, m)
np.save(f"{tmp}/params.npy", par)
starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
json.dump({"n_noise": n_noise, "n_signals": m, "events": [[int(a), int(b)] for a, b in zip(starts, counts)]},
          open(f"{tmp}/events.json", "w"))
ds = RemixDataset(tmp, seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
for B in (1024, 8192):
    idx = torch.randint(0, n_ev, (B,), device=dev, generator=g)
    plan = ds.device_plan(idx, generator=g)
    for _ in range(3): ds.assemble(plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ds.assemble(plan)
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20): ds.assemble(ds.device_plan(idx, generator=g))
    torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 20
    nsig = float(plan.nsig.float().mean())
    byt = B * (3 * T_LEN * 2 * (1 + nsig) + 3 * T_LEN * 4)
    print(f"B={B}: assemble {ta*1e3:.3f} ms ({B/ta:.0f} examples/s, {byt/ta/1e9:.0f} GB/s algorithmic, "
          f"{byt/ta/8e12*100:.1f}% of 8 TB/s); plan+assemble {tp*1e3:.3f} ms ({B/tp:.0f} examples/s)", flush=True)
