"""bench.py end to end on the GPU: the default single-GPU step and the N > 1 step (in-kernel loss reduction +
asynchronous all-reduce over a real NCCL/RCCL communicator, rehearsed with one rank), checked for the JSON
contract and for the loss the collective path reports."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                          "--no-extras", "--no-cpu-baseline", *flags],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_default_step_contract():
    d = _run()
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("BASELINE config 3") and "model" not in d["config"]
    r = d["roofline"]
    assert r["kernel"].startswith("pf::") and r["flop_per_sample"] == 10682368 and r["flop_per_sample_mask_aware"] == 7275264
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "TFLOP/s"
    assert d["value"] > 1e7 and abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_gpus_flag_spawns_the_ranks_itself():
    """`bench.py --gpus 2` with no launcher around it starts two ranks as child processes before touching the GPU
    (gloo rehearsal: both ranks share the one GPU of the box) and reports n_gpus = 2; a --gpus that contradicts
    WORLD_SIZE is an error, not a silent single-rank run (VERDICT r1 / ADVICE r1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "12",
                          "--warmup", "2", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "dp2"
    assert "every 1 steps" in d["config"]["launch"] and d["extras"]["value_allreduce_every_16"] > 0
    assert 50.0 < d["config"]["global_mean_nll"] < 500.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"],
                         capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


@pytest.mark.parametrize("every", [16, 1, 7])
def test_collective_step_reports_the_same_loss(every):
    # windows of `every` steps: 20 timed steps end in a partly filled window for 16 and 7 (flushed inside the timed region)
    d = _run("--collective", "--allreduce-every", str(every))
    assert d["config"]["launch"] == f"pre-bound launch, in-kernel (sum nll, rows) + async all-reduce every {every} steps"
    loss = d["config"]["global_mean_nll"]
    assert loss is not None and 50.0 < loss < 500.0          # mean NLL of the synthetic batch (134.9 for seed 1)
    if every == 16:
        test_collective_step_reports_the_same_loss.ref = loss
    elif hasattr(test_collective_step_reports_the_same_loss, "ref"):
        assert abs(loss - test_collective_step_reports_the_same_loss.ref) < 1e-3 * abs(loss)
    assert d["value"] > 5e6


def test_gpus_flag_nccl_world2():
    """The same two-rank run over RCCL (backend nccl, one GPU per rank): runs wherever two GPUs are visible, skips on the
    one-GPU boxes of the pool.  Weak scaling: the global batch doubles, the mean NLL is the one-rank value of the same
    seeds' union, and both ranks' per-step all-reduce completes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_PORT="29549", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and 50.0 < d["config"]["global_mean_nll"] < 500.0
    assert d["value"] > 2e7                                   # two GPUs: more than one GPU's 4e7 / 2


@pytest.mark.gpu
def test_size_fuzz_script():
    """scripts/fuzz_sizes.py: forward / inverse / backward of four flow shapes and the encoder at ragged batch sizes (1 .. 33 333):
    no fault, every row independent of the batch it travels in (a ragged ~500-row batch once sent a split GEMM out of bounds)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_sizes.py"), "1", "16"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
