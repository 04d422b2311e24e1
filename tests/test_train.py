"""The training step: the trainer recipe, and the data-parallel gradient all-reduce (gloo, 2 ranks)."""
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lr_schedule_matches_reference_formula():
    from posteriflow_amd.train import lr_factor
    assert lr_factor(0, 1000) == 0.0 and abs(lr_factor(250, 1000) - 0.5) < 1e-12
    assert abs(lr_factor(500, 1000) - 1.0) < 1e-12
    assert abs(lr_factor(750, 1000) - (0.01 + 0.99 * 0.5)) < 1e-12
    assert abs(lr_factor(5000, 1000) - 0.01) < 1e-12


GRAD_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from posteriflow_amd.train import OverlappedGradReducer, allreduce_gradients
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7))
x = torch.randn(64, 40); y = torch.randn(64, 7)
full = torch.nn.functional.mse_loss(net(x), y)
want = torch.autograd.grad(full, list(net.parameters()))
lo, hi = rank * 32, rank * 32 + 32
torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
allreduce_gradients(net.parameters(), bucket_bytes=20000)          # several buckets
for p, w in zip(net.parameters(), want):
    assert torch.allclose(p.grad, w, atol=1e-6), (p.grad - w).abs().max()
# the overlapped reducer: gradients are views into 3 flat buckets, all-reduced from hooks during backward
red = OverlappedGradReducer(net.parameters(), n_buckets=3)
assert 1 <= len(red.buckets) <= 3 and sum(b["flat"].numel() for b in red.buckets) == sum(p.numel() for p in net.parameters())
for step in range(2):                                               # twice: the buckets re-arm
    red.zero()
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    assert all(b["launched"] for b in red.buckets)                  # every bucket went out before backward() returned
    red.finish()
    for p, w in zip(net.parameters(), want):
        assert torch.allclose(p.grad, w, atol=1e-6), (step, (p.grad - w).abs().max())
        assert any(p.grad.data_ptr() >= b["flat"].data_ptr() and p.grad.data_ptr() < b["flat"].data_ptr() + 4 * b["flat"].numel() for b in red.buckets)
red.close()
# a parameter without a gradient this step: finish() still reduces its bucket
extra = torch.nn.Parameter(torch.zeros(5))
red2 = OverlappedGradReducer(list(net.parameters()) + [extra], n_buckets=2)
red2.zero()
torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
red2.finish()
for p, w in zip(net.parameters(), want):
    assert torch.allclose(p.grad, w, atol=1e-6)
assert torch.equal(extra.grad, torch.zeros(5))
red2.close()
# the one-backward-per-zero() contract is enforced: a second backward would add to a bucket already being reduced
red3 = OverlappedGradReducer(net.parameters(), n_buckets=2)
red3.zero()
torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
try:
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    raise SystemExit("second backward() before finish() was accepted")
except RuntimeError as e:
    assert "One backward per zero()" in str(e), e
red3.finish()
red3.close()
# graphs that differ between ranks (rank 1 does not use the first layer's bias... here: the whole first Linear is
# detached on rank 1): every rank still issues the all-reduces of buckets 0, 1, 2 in that order -- no hang, and the mean
# counts the missing gradient as zero
red4 = OverlappedGradReducer(net.parameters(), n_buckets=3)
red4.zero()
h = net[0](x[lo:hi])
if rank == 1:
    h = h.detach()
torch.nn.functional.mse_loss(net[2](torch.relu(h)), y[lo:hi]).backward()
red4.finish()
g0 = torch.autograd.grad(torch.nn.functional.mse_loss(net(x[:32]), y[:32]), list(net[0].parameters()))
for p, w in zip(net[0].parameters(), g0):
    assert torch.allclose(p.grad, 0.5 * w, atol=1e-6), (p.grad - 0.5 * w).abs().max()
for p, w in zip(net[2].parameters(), want[2:]):
    assert torch.allclose(p.grad, w, atol=1e-6)
red4.close()
dist.destroy_process_group()
print("ok", rank)
"""


def test_gradient_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "g.py"
    script.write_text(GRAD_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29537", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


DP_WORKER = r"""
# Data-parallel train_step against the reference's single-process loss sum(nll) / sum(nsig) over the CONCATENATED batch
# (experiments/train_lean_npe.py:108-127), with ranks that hold DIFFERENT numbers of (event, rank) pairs.
import os, sys, copy, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
backend = sys.argv[2]
from posteriflow_amd.train import OverlappedGradReducer, train_step
from posteriflow_amd.npe import batch_nll
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if backend == "nccl":
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dev = torch.device("cuda" if backend == "nccl" else "cpu")
dist.init_process_group(backend, rank=rank, world_size=world)
torch.set_default_dtype(torch.float64)

class Toy(torch.nn.Module):
    # the (encode, nll) interface batch_nll drives; smooth, every parameter reached by every row
    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Linear(24, 8)
        self.rank_embed = torch.nn.Embedding(5, 3)
        self.head = torch.nn.Linear(8 + 3 + 4, 1)
    def encode(self, strain, asd_bands=None):
        return torch.tanh(self.enc(strain))
    def nll(self, strain, params, rank, context=None, asd_bands=None):
        h = torch.cat([context, self.rank_embed(rank), params], dim=1)
        return torch.nn.functional.softplus(self.head(h)).squeeze(1) + (params ** 2).sum(1)

torch.manual_seed(0)
model0 = Toy().to(dev)
B = 12                                           # per rank
strain = torch.randn(world * B, 24, device=dev)
params = torch.randn(world * B, 5, 4, device=dev)
nsig = torch.tensor([1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1, 1] + [5, 4, 5, 3, 5, 5, 4, 5, 5, 2, 5, 5], device=dev)   # 14 vs 53 pairs
assert int(nsig[:B].sum()) != int(nsig[B:].sum())
lo, hi = rank * B, rank * B + B

def reference_gradient(model):
    # what one process does on the whole batch: the reference's per-rank loop
    ctx = model.encode(strain)
    tot, cnt = 0.0, 0
    for r in range(5):
        m = nsig > r
        if m.any():
            rr = torch.full((int(m.sum()),), r, dtype=torch.long, device=dev)
            tot = tot + model.nll(None, params[m, r], rr, context=ctx[m]).sum()
            cnt += int(m.sum())
    loss = tot / cnt
    return loss.detach(), torch.autograd.grad(loss, list(model.parameters()))

class NoStep:                                    # the gradients are what is compared: no parameter update
    def zero_grad(self, set_to_none=True):
        for p in self.params: p.grad = None
    def step(self): pass

for row_cap in ("exact", None, 56):                # 56: a static cap above both ranks' pair counts (14, 53)
    for use_reducer in (False, True):
        model = copy.deepcopy(model0)
        want_loss, want = reference_gradient(model)
        opt = NoStep(); opt.params = list(model.parameters())
        red = OverlappedGradReducer(model.parameters(), n_buckets=2) if use_reducer else None
        out = train_step(model, opt, None, strain[lo:hi], params[lo:hi], nsig[lo:hi], reducer=red, row_cap=row_cap)
        # clip_grad_norm_(5.0) ran inside the step: undo nothing unless it clipped
        gn = torch.sqrt(sum((w ** 2).sum() for w in want))
        scale = min(1.0, 5.0 / (float(gn) + 1e-6))
        assert abs(out["loss"] - float(want_loss)) < 1e-9, (row_cap, use_reducer, out["loss"], float(want_loss))
        assert abs(out["grad_norm"] - float(gn)) < 1e-6 * float(gn)
        for (name, p), w in zip(model.named_parameters(), want):
            err = (p.grad - scale * w).abs().max().item()
            assert err < 1e-6 * max(1.0, w.abs().max().item()), (row_cap, use_reducer, name, err)
        # the mean of the ranks' LOCAL means is a different gradient on this data: the test can see the difference
        local = batch_nll(copy.deepcopy(model0), strain[lo:hi], params[lo:hi], nsig[lo:hi], row_cap=row_cap)
        both = torch.stack([local.detach()]); dist.all_reduce(both)
        assert abs(float(both[0]) / world - float(want_loss)) > 1e-3
        if red is not None: red.close()
dist.destroy_process_group()
print("ok", rank)
"""


def test_data_parallel_step_is_the_global_batch_gradient_gloo_world2(tmp_path):
    """ranks with different signal counts; every reduced gradient = the single-process gradient of
    sum nll / sum nsig over the concatenated batch (1e-6), through both reducers and every row_cap mode."""
    script = tmp_path / "dp.py"
    script.write_text(DP_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT, "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


@pytest.mark.gpu
def test_data_parallel_step_is_the_global_batch_gradient_nccl_world2(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "dp.py"
    script.write_text(DP_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29543", str(script), ROOT, "nccl"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


NCCL_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from posteriflow_amd.train import OverlappedGradReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", rank=rank, world_size=world)
dev = torch.device("cuda")
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7)).to(dev)
x = torch.randn(64, 40, device=dev); y = torch.randn(64, 7, device=dev)
want = torch.autograd.grad(torch.nn.functional.mse_loss(net(x), y), list(net.parameters()))
lo, hi = rank * 32, rank * 32 + 32
red = OverlappedGradReducer(net.parameters(), n_buckets=3)
for step in range(2):
    red.zero()
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    red.finish()
    for p, w in zip(net.parameters(), want):
        assert torch.allclose(p.grad, w, atol=1e-5), (step, (p.grad - w).abs().max())
red.close()
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.gpu
def test_gradient_allreduce_nccl_world2(tmp_path):
    """OverlappedGradReducer over RCCL, one GPU per rank: runs where two GPUs are visible (skips on the pool's one-GPU boxes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "n.py"
    script.write_text(NCCL_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29539", str(script), ROOT]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_step_reduces_loss_gpu(precision):
    """eight optimiser steps on one small batch; bf16: the throughput mode (bf16 training forward, HIP re-evaluation,
    bf16 chain, weights re-packed into all three streams after every optimiser step)"""
    import recipe
    from posteriflow_amd import LeanNPE
    from posteriflow_amd.train import checkpoint_dict, make_optimizer, make_scheduler, train_step
    torch.manual_seed(0)
    model = LeanNPE(flow_layers=2).cuda()
    model.flow.precision = model.encoder.precision = precision
    opt = make_optimizer(model, lr=1e-3)
    sched = make_scheduler(opt, total_steps=100, warmup_steps=2)
    strain = recipe.strain_batch(8, 3, seed=1).cuda()
    params = torch.stack([recipe.physical_params(8, seed=30 + r) for r in range(5)], dim=1).cuda()
    nsig = torch.tensor([1, 2, 1, 3, 1, 1, 2, 1]).cuda()
    losses = [train_step(model, opt, sched, strain, params, nsig)["loss"] for _ in range(8)]
    # (the trajectory of 8 events at lr 1e-3 is spiky and, with float atomics in the gradients, not bit-reproducible: 14.1 ->
    # 9 .. 12 over the last steps in every run observed)
    assert all(math.isfinite(v) for v in losses) and min(losses[-3:]) < losses[0] - 1.0, losses
    ck = checkpoint_dict(model, 0, losses[-1])
    assert set(ck) == {"model_state_dict", "epoch", "val_nll", "diagnostics", "args"}
    model.eval()
    with torch.no_grad():                       # inference path (HIP stem + flow) still works after updates
        assert torch.isfinite(model.sample_posterior(strain[:2], n_samples=16)).all()


@pytest.mark.gpu
def test_config4_training_step_at_per_gpu_size():
    """BASELINE config 4 at its per-GPU size (8192 / 8 = 1024 events): on-GPU remix of resident pools -> batch_nll
    (encoder + flow, one static-shape flow call) -> backward -> clip + AdamW, in the throughput mode; three steps, finite
    loss and gradient norm, parameters really move (experiments/train_lean_npe.py:108-127, 357-368)."""
    from posteriflow_amd import LeanNPE
    from posteriflow_amd.remix import synthetic_dataset
    from posteriflow_amd.train import make_optimizer, make_scheduler, train_step
    dev = torch.device("cuda")
    ds = synthetic_dataset(dev, n_noise=256, n_events=256, seed=0)
    torch.manual_seed(0)
    model = LeanNPE().to(dev).train().set_precision("bf16")
    model.flatten_parameters()                      # flow and encoder as one leaf each (what bench.py's config-4 step uses)
    opt = make_optimizer(model)
    sched = make_scheduler(opt, total_steps=1000, warmup_steps=2)
    g = torch.Generator(device=dev).manual_seed(0)
    before = [p.detach().clone() for p in list(model.parameters())[:4]]
    out = []
    for _ in range(3):
        idx = torch.randint(0, ds.n_events, (1024,), device=dev, generator=g)
        strain, labels, nsig, _ = ds.batch(idx, generator=g)
        assert strain.shape == (1024, 3, 16384)
        out.append(train_step(model, opt, sched, strain, labels, nsig, row_cap=2048))
    assert all(math.isfinite(o["loss"]) and math.isfinite(o["grad_norm"]) and o["grad_norm"] > 0 for o in out), out
    assert any(not torch.equal(b, p.detach()) for b, p in zip(before, list(model.parameters())[:4]))


@pytest.mark.gpu
def test_fused_optimizer_updates_reach_the_kernels():
    """torch's fused AdamW updates parameters without bumping their version counters (leaf and views: 0 -> 0); the packed
    weights are keyed on an optimiser-step epoch as well, so the step after a fused update evaluates the NEW weights: the
    trajectory equals the unfused optimiser's, and an eval-mode call right after a step sees the update."""
    from posteriflow_amd import npe, train
    from posteriflow_amd.remix import synthetic_dataset
    dev = torch.device("cuda")
    ds = synthetic_dataset(dev, n_noise=64, n_events=64, seed=0)
    traj = {}
    for fused in (False, True):
        torch.manual_seed(0)
        model = npe.LeanNPE().to(dev).train().set_precision("fp32").flatten_parameters()
        for l in model.encoder.fusion.layers:                          # (no dropout: the two runs must be comparable)
            l.dropout.p = l.dropout1.p = l.dropout2.p = 0.0
            l.self_attn.dropout = 0.0
        opt = train.make_optimizer(model, lr=1e-4, fused=fused)
        g = torch.Generator(device=dev).manual_seed(0)
        losses = []
        for it in range(3):
            idx = torch.randint(0, ds.n_events, (8,), device=dev, generator=g)
            strain, labels, nsig, _ = ds.batch(idx, generator=g)
            losses.append(train.train_step(model, opt, None, strain, labels, nsig)["loss"])
        rank0 = torch.zeros(8, dtype=torch.long, device=dev)
        model.eval()
        with torch.no_grad():
            before = model.nll(strain, labels[:, 0], rank0).sum().item()
        model.train()
        train.train_step(model, opt, None, strain, labels, nsig)
        model.eval()
        with torch.no_grad():
            after = model.nll(strain, labels[:, 0], rank0).sum().item()
        assert abs(after - before) > 1e-4 * abs(before), (fused, before, after)      # the eval call saw the step
        traj[fused] = losses
    assert traj[False][0] == traj[True][0]
    for a, b in zip(traj[False], traj[True]):          # same trajectory: the step after a fused update used the new weights
        assert abs(a - b) < 2e-3 * abs(a), traj
