"""BASELINE config 5: N draws from the posterior of ONE event with the 12-layer flow (D = 15), sharded over the
ranks of `python -m torch.distributed.run --nproc-per-node G scripts/bench_sampling.py` (or one process):
every rank draws its shard with its own generator (seed + rank) through pf_flow_inverse with the single context
row passed un-expanded; no collective on the data path, rank 0 gathers per-dimension moments.  Prints draws/s."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd.dist import rank_generator, shard_bounds
from posteriflow_amd.flows import NSFPosteriorFlow

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl" if torch.cuda.device_count() >= world else "gloo", rank=rank, world_size=world)
torch.manual_seed(0)
flow = NSFPosteriorFlow(features=15, context_features=288, hidden_features=256, num_layers=12, num_bins=16,
                        tail_bound=5.0).to(dev).eval()
flow.precision = "bf16"
flow.freeze_packed()
ctx = torch.randn(1, 288, generator=torch.Generator().manual_seed(1)).to(dev)
lo, hi = shard_bounds(n_total, rank, world)
gen = rank_generator(1234, rank, dev)
chunk = 131072
def draw():
    out = []
    for i in range(lo, hi, chunk):
        z = torch.randn(min(chunk, hi - i), 15, device=dev, generator=gen)
        out.append(flow.inverse(z, ctx)[0])
    return torch.cat(out)
with torch.no_grad():
    draw(); torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter(); x = draw(); torch.cuda.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
mom = torch.stack([x.double().sum(0), x.double().square().sum(0)])
if dist:
    mom = mom.cpu() if dist.get_backend() == "gloo" else mom
    dist.all_reduce(mom)
if rank == 0:
    mean = mom[0] / n_total
    print(json.dumps({"config": "BASELINE config 5: 12-layer flow, D=15, one context row", "draws": n_total, "n_gpus": world,
                      "seconds": dt, "draws_per_s": n_total / dt,
                      "mean": [round(v, 4) for v in mean.tolist()],
                      "std": [round(v, 4) for v in (mom[1] / n_total - mean ** 2).clamp_min(0).sqrt().tolist()]}))
if dist:
    dist.destroy_process_group()
