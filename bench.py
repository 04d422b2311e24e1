#!/usr/bin/env python3
"""bench.py -- flow.log_prob samples/sec at batch 4096 on N MI355X (BASELINE.json metric).

One step = one pass of the hot path (NSFPosteriorFlow.compute_psd_aware_nll ==
pf_flow_forward with the fused N(0,I) base log-density) over one batch of
synthetic inputs already resident in HBM, followed -- when N > 1 -- by the RCCL
all-reduce of (sum nll, count), the only exchange the path has.
Workload: BASELINE config 3's flow (8-layer MAF-RQS, D=15, C=288, H=256, K=16,
tail_bound 5) at batch 4096 PER GPU (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     achieved = batch * dense-GEMM FLOP/sample / avg kernel time (HIP events)
  cpu_baseline the CPU oracle (reference algorithm, fp32, all host cores) on the same batch
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, C, H, K, L, TB = 15, 288, 256, 16, 8, 5.0
FINAL_LAYER_SCALE = 2.0       # DESIGN.md "Measurement": exercises the bin search, stays well-conditioned
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cpu_info():
    """What the box grants the CPU leg: the scheduler affinity, the cgroup CPU quota and os.cpu_count(), plus lscpu's model
    name.  `granted` = min(affinity, quota) -- a GPU box shows the whole host in os.cpu_count() but grants a share."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity"] = os.cpu_count() or 1
    quota = None
    try:   # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(p)))
    except Exception:
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // p)
        except Exception:
            pass
    info["cgroup_quota"] = quota
    info["granted"] = max(1, min(info["affinity"], quota) if quota else info["affinity"])
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        info["model"] = next((l.split(":", 1)[1].strip() for l in out.splitlines() if l.startswith("Model name")), None)
    except Exception:
        info["model"] = None
    return info


def flops_per_sample():
    return 2 * L * (D * H + C * H + 2 * (2 * H * H + C * H) + H * D * (3 * K - 1))


def masked_flops_per_sample():
    """The same count with the autoregressive masks discounted (7.275 MFLOP for this flow;
    reported alongside, never instead)."""
    deg_h = [u % max(1, D - 1) + min(1, D - 1) for u in range(H)]
    n_in = sum(1 for u in range(H) for d in range(D) if deg_h[u] >= d + 1)
    n_hh = sum(1 for u in range(H) for v in range(H) if deg_h[u] >= deg_h[v])
    n_out = (3 * K - 1) * sum(1 for f in range(D) for v in range(H) if f + 1 > deg_h[v])
    return 2 * L * (n_in + C * H + 2 * (2 * n_hh + C * H) + n_out)


def make_inputs(batch, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, D, generator=g) * 2 - 1
    m = torch.rand(batch, D, generator=g) < 0.02
    x = torch.where(m, (torch.rand(batch, D, generator=g) * 2 - 1) * 6.0, x)
    ctx = torch.randn(batch, C, generator=g)
    return x.to(device), ctx.to(device)


def build_flow(device, precision):
    from posteriflow_amd import NSFPosteriorFlow
    torch.manual_seed(0)
    flow = NSFPosteriorFlow(D, C, H, L, K, TB, temperature_scale=1.0, use_masked_context=False)
    with torch.no_grad():
        for layer in flow._ar_transforms:
            layer.autoregressive_net.final_layer.weight.mul_(FINAL_LAYER_SCALE)
            layer.autoregressive_net.final_layer.bias.mul_(FINAL_LAYER_SCALE)
    flow = flow.to(device)
    flow.precision = precision
    return flow


def oracle_for(flow):
    """The CPU oracle with the bench flow's weights (checker / CPU baseline only)."""
    from oracle.flow_ref import NSFPosteriorFlowRef
    ref = NSFPosteriorFlowRef(D, C, H, L, K, TB, temperature_scale=1.0)
    sd = {k: v.cpu() for k, v in flow.state_dict().items() if not k.startswith("flow.")}
    ref.load_state_dict(sd)
    return ref


def cpu_baseline(flow, batch, budget_s=15.0):
    """The oracle (CPU restatement of the reference's algorithm, fp32) on the host cores the box grants.
    Thread count: every core granted (affinity and cgroup quota; $PF_BENCH_CPU_THREADS overrides) -- and, because an
    intra-op pool wider than the GEMMs' parallelism can be SLOWER, a short probe of 16 / 32 / 64 threads beside it; the
    sample is timed at the fastest and the probe is reported.  Returns (baseline dict, oracle nll of the seed-1 batch)."""
    ref = oracle_for(flow)
    info = host_cpu_info()
    forced = os.environ.get("PF_BENCH_CPU_THREADS")
    x, ctx = make_inputs(batch, 1, "cpu")
    ls = torch.zeros_like(x)
    probe = {}
    with torch.no_grad():
        cands = [int(forced)] if forced else sorted({info["granted"]} | {c for c in (16, 32, 64) if c < info["granted"]})
        for c in cands:
            torch.set_num_threads(c)
            ref.compute_psd_aware_nll(x, ctx, ls)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ref.compute_psd_aware_nll(x, ctx, ls)
                ts.append(time.perf_counter() - t0)
            probe[c] = min(ts) * 1e3
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
    log(f"cpu baseline: {info}; probe ms per batch by thread count {probe}; timing with {cores} threads")
    times = []
    with torch.no_grad():
        for _ in range(3):
            want = ref.compute_psd_aware_nll(x, ctx, ls)
        t_end = time.time() + budget_s                    # a bounded sample: ~15 s of host work
        while len(times) < 200 and (time.time() < t_end or len(times) < 20):
            t0 = time.perf_counter()
            ref.compute_psd_aware_nll(x, ctx, ls)
            times.append(time.perf_counter() - t0)
            if len(times) % 10 == 1:
                log(f"cpu baseline iter {len(times)}: {times[-1] * 1e3:.1f} ms")
        # the sampling direction as the reference executes it (D conditioner passes per layer), one chunk of 4096 draws
        # for one context row (pipeline.py:169-173): ~2-3 s of host work, reported beside the GPU's draws/s
        z = torch.randn(4096, D, generator=torch.Generator().manual_seed(3))
        t0 = time.perf_counter()
        y, _ = ref.inverse(z, ctx[:1].expand(4096, -1))
        inv_s = time.perf_counter() - t0
        # ... and what BASELINE.md's published figures time (pipeline.py:169-181): draws + log q of the same chunk
        t0 = time.perf_counter()
        ref.compute_psd_aware_nll(y, ctx[:1].expand(4096, -1), torch.zeros_like(y))
        dens_s = time.perf_counter() - t0
    log(f"cpu baseline inverse: 4096 draws in {inv_s:.2f} s (+ density {dens_s:.2f} s)")
    ts = sorted(times)
    med = statistics.median(ts)
    p10, p90 = ts[int(0.1 * (len(ts) - 1))], ts[int(round(0.9 * (len(ts) - 1)))]
    return {"value": batch / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"batch {batch}, fp32, {len(times)} iterations (~{budget_s:.0f} s of host work) after 3 warm-ups, median",
            "ms_per_batch": med * 1e3, "ms_per_batch_p10": p10 * 1e3, "ms_per_batch_p90": p90 * 1e3,
            "inverse_draws_per_s": 4096 / inv_s, "draws_plus_logq_per_s": 4096 / (inv_s + dens_s),
            "host": info, "threads_probe_ms_per_batch": {str(k): v for k, v in probe.items()}}, want


def fp64_reference(flow, x, ctx):
    """(nll in float64 [B], first-order sensitivity [B]) from the oracle evaluated in float64 on the CPU.
    sensitivity[row] = sum over layers and raw spline parameters p of |d nll / d p| * |p| * 2^-24: the change of the row's
    nll, to first order, when every raw spline parameter (the conditioner's final-layer output, which any fp32 evaluation
    holds as an fp32 number) is perturbed by half an fp32 ulp.  It is a property of the map in exact arithmetic -- a row
    where it is large cannot be evaluated to 1e-5 by ANY fp32 implementation."""
    from oracle import nflows_restated as nfr
    from oracle.flow_ref import NSFPosteriorFlowRef
    ref64 = NSFPosteriorFlowRef(D, C, H, L, K, TB, temperature_scale=1.0).double()
    ref64.load_state_dict({k: v.cpu() for k, v in flow.state_dict().items() if not k.startswith("flow.")})
    kept, hooks = [], []

    def keep(_m, _i, out):
        out.retain_grad()
        kept.append(out)

    for t in ref64.transform._transforms:
        if isinstance(t, nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform):
            hooks.append(t.autoregressive_net.final_layer.register_forward_hook(keep))
    xd, cd = x.double(), ctx.double()
    nll = ref64.compute_psd_aware_nll(xd, cd, torch.zeros_like(xd))
    nll.sum().backward()                                  # rows are independent: the gradient of the sum is per row
    sens = sum((p.grad.abs() * p.detach().abs()).sum(dim=1) for p in kept) * 2.0 ** -24
    for h in hooks:
        h.remove()
    return nll.detach(), sens.detach()


def parity_check(flow, dev, batch, want):
    """GPU nll of the timed workload (seed-1 batch) against the oracle's, both precisions, on EVERY row.
    fp32 parity mode, against a float64 evaluation of the oracle: the HIP path's and the CPU fp32 oracle's relative
    distances side by side (north_star's tolerance is 1e-5 relative: `frac_rows_over_1e-5` of each), and -- for the rows
    either of them misses -- the float64 first-order sensitivity of the row's nll to half-ulp perturbations of the raw
    spline parameters (`fp64_reference`): evidence, not assertion, that those rows are the ill-conditioned ones of this
    random-weight map.  Hard asserts: the HIP path within 2x / 4x the CPU fp32 path's own distance from float64 at
    p99 / worst row, and the two fp32 evaluations agreeing to 5e-5 at p99.
    bf16 throughput mode: checked against the oracle evaluated with the SAME operand rounding (bf16 GEMM operands, fp32
    accumulate: oracle.nflows_restated.gemm_emulation) -- median 2e-2 nats, p99 3; what bf16 operands cost against the
    fp32 oracle on this random-weight workload (median ~0.6 nats of ~135, p99 ~8) is recorded, not asserted: it is the
    arithmetic's error, the CPU emulation shows the same."""
    from oracle import nflows_restated as nfr
    x, ctx = make_inputs(batch, 1, dev)
    out = {}
    prec = flow.precision
    was_frozen = flow._frozen
    nll = torch.empty(batch, device=dev)
    stats = lambda e: {"p50": e.median().item(), "p99": e.quantile(0.99).item(), "max": e.max().item()}
    n64, sens = fp64_reference(flow, x.cpu(), ctx.cpu())
    den = n64.abs().clamp_min(1.0)
    with torch.no_grad():
        for name in ("fp32", "bf16"):
            flow.precision = name
            if was_frozen:
                flow.freeze_packed()
            got = flow.nll_into(x, ctx, nll).cpu().double()
            err = (got - want.double()).abs()
            rel = err / want.double().abs().clamp_min(1.0)
            out[name] = {"max_abs": err.max().item(), "p50_abs": err.median().item(), "p99_rel": rel.quantile(0.99).item(),
                         "max_rel": rel.max().item(), "rows": batch}
            if name == "fp32":
                r_hip, r_cpu = (got - n64).abs() / den, (want.double() - n64).abs() / den
                rs = sens / den                                       # relative first-order sensitivity
                over_hip, over_cpu = r_hip > 1e-5, r_cpu > 1e-5
                either = over_hip | over_cpu
                q = lambda t, p_: t.quantile(p_).item() if t.numel() else None
                out[name].update({
                    "reference": "float64 evaluation of the oracle (CPU)",
                    "hip_vs_fp64": {"p50_rel": r_hip.median().item(), "p99_rel": r_hip.quantile(0.99).item(),
                                    "max_rel": r_hip.max().item(), "frac_rows_over_1e-5": over_hip.double().mean().item()},
                    "cpu_fp32_vs_fp64": {"p50_rel": r_cpu.median().item(), "p99_rel": r_cpu.quantile(0.99).item(),
                                         "max_rel": r_cpu.max().item(), "frac_rows_over_1e-5": over_cpu.double().mean().item()},
                    "rows_over_1e-5_in_both": int((over_hip & over_cpu).sum()), "rows_over_1e-5_hip_only": int((over_hip & ~over_cpu).sum()),
                    "rows_over_1e-5_cpu_only": int((over_cpu & ~over_hip).sum()),
                    # half-ulp sensitivity of the row's nll (relative), rows over 1e-5 (in either fp32 evaluation) vs the rest
                    "sensitivity_rel": {"definition": "sum |d nll/d p| |p| 2^-24 / max(|nll|, 1) over the raw spline parameters p of all layers, float64",
                                        "all_rows_p50": rs.median().item(), "all_rows_p99": rs.quantile(0.99).item(),
                                        "rows_over_1e-5_p10": q(rs[either], 0.1), "rows_over_1e-5_p50": q(rs[either], 0.5),
                                        "rows_within_1e-5_p50": q(rs[~either], 0.5), "rows_within_1e-5_p99": q(rs[~either], 0.99),
                                        # how well the sensitivity ranks the rows: share of the over-1e-5 rows among the
                                        # n most sensitive rows, n = number of over-1e-5 rows
                                        "top_n_overlap": (float(either[torch.argsort(rs, descending=True)[: int(either.sum())]].double().mean())
                                                          if int(either.sum()) else None),
                                        "err_over_sensitivity_p50_hip": (r_hip / rs.clamp_min(1e-30)).median().item(),
                                        "err_over_sensitivity_p50_cpu": (r_cpu / rs.clamp_min(1e-30)).median().item()},
                })
                assert r_hip.median() < 1e-5, ("fp32 nll off float64 at the median", out)
                assert r_hip.quantile(0.99) < max(1e-5, 2 * r_cpu.quantile(0.99).item()), ("fp32 p99 beyond 2x the CPU fp32 path's", out)
                assert r_hip.max() < max(1e-5, 4 * r_cpu.max().item()), ("fp32 worst row beyond 4x the CPU fp32 path's", out)
            if name == "bf16":
                xc, cc = x.cpu(), ctx.cpu()
                with nfr.gemm_emulation("bf16"):
                    emu = oracle_for(flow).compute_psd_aware_nll(xc, cc, torch.zeros_like(xc)).double()
                out[name]["abs_vs_same_rounding_oracle"] = stats((got - emu).abs())
                out[name]["same_rounding_oracle_vs_fp32_oracle"] = stats((emu - want.double()).abs())
    flow.precision = prec
    if was_frozen:
        flow.freeze_packed()
    log("parity vs oracle: " + json.dumps(out))
    assert out["fp32"]["p99_rel"] < 5e-5 and out["fp32"]["max_rel"] < 5e-3, ("fp32 nll off the oracle", out)
    e = out["bf16"]["abs_vs_same_rounding_oracle"]
    assert e["p50"] < 2e-2 and e["p99"] < 3.0, ("bf16 nll off the same-rounding oracle", out)
    return out


def ingest_floor_us(flow, batch, dev, stream):
    """The compute-free read of the SAME packed weight stream by one workgroup per row group (pf_diag_stream_ingest): the
    floor under the 16-row kernel, whose every workgroup streams all of it.  Best of 4 / 8 KiB in flight per wave."""
    import ctypes
    lib = __import__("posteriflow_amd")._lib.lib()
    packed = flow.packed_weights(wide=flow._use_wide(batch))
    rows_wg = int(lib.pf_flow_rows_per_workgroup(flow._desc(wide=flow._use_wide(batch)), batch))
    wgs = (batch + rows_wg - 1) // rows_wg
    sink = torch.zeros(max(wgs, 1), dtype=torch.int32, device=dev)
    nbytes = packed.numel() * packed.element_size()
    best = None
    with torch.cuda.stream(stream):
        for infl in (4, 8):
            call = lambda: lib.pf_diag_stream_ingest(ctypes.c_void_p(packed.data_ptr()), nbytes, wgs, infl,
                                                     ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
            for _ in range(10):
                assert call() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(50):
                call()
            e1.record(stream)
            stream.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            best = us if best is None else min(best, us)
    return best, nbytes, wgs


def kernel_time_ms(flow, x, ctx, nll, stream, n_k):
    """kernel-only duration: HIP events (on the launch stream) bracketing a captured chain of back-to-back launches"""
    dev = x.device
    with torch.cuda.stream(stream):
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            flow.nll_into(x, ctx, nll)
        kg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(kg, stream=stream):
            for _ in range(n_k):
                flow.nll_into(x, ctx, nll)
        kg.replay()
        stream.synchronize()
        k0.record(stream)
        kg.replay()
        k1.record(stream)
        stream.synchronize()
        return k0.elapsed_time(k1) / n_k


def extras(flow, dev, batch):
    """Side measurements on the same GPU (not part of `value`): the fp32 parity mode's log_prob throughput,
    the forward at a large batch, sampling through the inverse (one context row, as inference/pipeline.py:169-173)
    and the strain embedding of BASELINE config 3 (3 detectors, batch rows)."""
    from posteriflow_amd import npe
    out = {}

    def timed(fn, reps):
        for _ in range(3):                      # (one warm-up call once read 2.3x slow on a fresh box)
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    with torch.no_grad():
        if flow.precision == "bf16":           # the fp32 parity mode (v_mfma_f32_16x16x4_f32) on the bench workload
            x, ctx = make_inputs(batch, 1, dev)
            nll = torch.empty(batch, device=dev)
            flow.precision = "fp32"
            flow.freeze_packed()
            ms = kernel_time_ms(flow, x, ctx, nll, torch.cuda.Stream(dev), 20)
            flow.precision = "bf16"
            flow.freeze_packed()
            out[f"forward_samples_per_s_fp32_{batch}"] = batch / (ms * 1e-3)
            out[f"forward_kernel_us_fp32_{batch}"] = ms * 1e3
            out[f"forward_roofline_frac_fp32_{batch}"] = batch * flops_per_sample() / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["fp32"]
            out[f"forward_roofline_frac_mask_aware_fp32_{batch}"] = (batch * masked_flops_per_sample() / (ms * 1e-3) / 1e12
                                                                      / PEAK_TFLOPS["fp32"])
        # the forward on a batch large enough to amortise the per-CU weight ingest
        nb = 65536
        xb, cb = make_inputs(nb, 3, dev)
        ob = torch.empty(nb, device=dev)
        launch = flow.bind_nll(xb, cb, ob)
        dt = timed(launch, 20)
        out["forward_samples_per_s_65536"] = nb / dt
        out["forward_kernel_65536"] = flow.forward_kernel_name(nb)
        out["forward_tflops_65536"] = nb * flops_per_sample() / dt / 1e12
        out["forward_roofline_frac_65536"] = out["forward_tflops_65536"] / PEAK_TFLOPS[flow.precision]
        out["forward_roofline_frac_mask_aware_65536"] = (nb * masked_flops_per_sample() / dt / 1e12
                                                         / PEAK_TFLOPS[flow.precision])
        del xb, cb, ob
        nm = 16384                               # one round of the mid-batch kernel (64 rows per workgroup, two waves per SIMD)
        xb, cb = make_inputs(nm, 4, dev)
        ob = torch.empty(nm, device=dev)
        dt = timed(flow.bind_nll(xb, cb, ob), 20)
        out["forward_samples_per_s_16384"] = nm / dt
        out["forward_kernel_16384"] = flow.forward_kernel_name(nm)
        out["forward_kernel_us_16384"] = dt * 1e6
        out["forward_roofline_frac_16384"] = nm * flops_per_sample() / dt / 1e12 / PEAK_TFLOPS[flow.precision]
        del xb, cb, ob
        ctx1 = torch.randn(1, C, device=dev)
        for n in (4096, 131072):
            z = torch.randn(n, D, device=dev)
            dt = timed(lambda: flow.inverse(z, ctx1), 5 if n <= 4096 else 2)
            out[f"inverse_draws_per_s_{n}"] = n / dt
        if flow.precision == "bf16":           # the same draws in the fp32 parity mode (fp32 incremental inverse)
            prec, flow.precision = flow.precision, "fp32"
            dt = timed(lambda: flow.inverse(z, ctx1), 2)
            out["inverse_draws_per_s_fp32_131072"] = z.shape[0] / dt
            flow.precision = prec
        enc = npe.LeanStrainEncoder().to(dev).eval()
        enc.precision = flow.precision
        strain = torch.randn(batch, 3, 16384, device=dev)
        dt = timed(lambda: enc._stem_hip(strain), 3)
        out["stem_events_per_s"] = batch / dt
        out["stem_tflops"] = batch * 3 * 34.98e6 * 2 / dt / 1e12          # 34.98 M MAC / detector
        out["stem_strain_read_GBps"] = batch * 3 * 16384 * 4 / dt / 1e9
        # fp32 parity mode evaluates the Transformer with tensor ops ([chunk, 183, 768] hidden): keep chunks small there
        chunk = batch if flow.precision == "bf16" else 512
        dt = timed(lambda: [enc(strain[i:i + chunk]) for i in range(0, batch, chunk)], 2)
        out["encoder_events_per_s"] = batch / dt
        nll = torch.empty(batch, device=dev)
        x, ctx = make_inputs(batch, 1, dev)
        out["config3_end_to_end_events_per_s"] = batch / (dt + timed(lambda: flow.nll_into(x, ctx, nll), 10))
    # the training side of the same flow: forward + backward of the NLL at 2048 rows (the per-GPU share of a 1024-event
    # batch with ~2 signals per event), LeanNPE's flow shape, in the mode being benchmarked
    tf = npe.LeanNPE().to(dev).train().flow.flatten_parameters()     # one flat leaf: what a trainer should use (flows.py)
    tf.precision = flow.precision
    xt = torch.rand(2048, tf.features, device=dev) * 2 - 1
    ct = torch.randn(2048, tf.context_features, device=dev, requires_grad=True)
    params = list(tf.parameters())

    def fwd_bwd():
        for q in params:
            q.grad = None
        ct.grad = None
        tf.compute_psd_aware_nll(xt, ct, torch.zeros_like(xt)).mean().backward()

    for _ in range(3):
        fwd_bwd()
    out[f"flow_fwd_bwd_ms_2048_{flow.precision}"] = timed(fwd_bwd, 10) * 1e3
    out.update(config5_sampling(dev, flow.precision))
    out.update(sample_event_rate(dev, flow.precision))
    out.update(config4_train_step(dev, flow.precision))
    out.update(generic_head(dev, flow.precision))
    out.update(coherent_geometry(dev))
    log("extras: " + ", ".join(f"{k}={v:.3g}" for k, v in out.items() if isinstance(v, (int, float))))
    return out


def config5_sampling(dev, precision, n_rank=125_000, reps=3):
    """BASELINE config 5 on this GPU: the 12-layer D = 15 flow, one context row, this rank's share (125 000 draws = 1e6 / 8)
    of the posterior draws through flow.inverse in chunks of 131 072 (inference/pipeline.py:169-173 uses 4096)."""
    from posteriflow_amd import NSFPosteriorFlow
    torch.manual_seed(0)
    f5 = NSFPosteriorFlow(features=15, context_features=288, hidden_features=256, num_layers=12, num_bins=16,
                          tail_bound=5.0).to(dev).eval()
    f5.precision = precision
    ctx = torch.randn(1, 288, generator=torch.Generator().manual_seed(1)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)

    def draw():
        with torch.no_grad():
            z = torch.randn(n_rank, 15, device=dev, generator=gen)
            return f5.inverse(z, ctx)[0]

    draw()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        draw()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    return {"config5_draws_per_s": n_rank / dt, "config5_ms_per_125000_draws": dt * 1e3}


def sample_event_rate(dev, precision, n=10_000):
    """What BASELINE.md's published figures time (src/ahsd/inference/pipeline.py:161-186, `infer()`): for ONE event, encode
    the strain, draw n = 10 000 posterior samples through flow.inverse and evaluate log q of every draw -- the reference's
    result.json files record ~1 120 draws/s on its CPU and ~1 850 on mps (LeanNPE: 10-layer D = 11 flow, 3 detectors).
    Here: posteriflow_amd.inference.sample_event, encode included, median of single calls."""
    from posteriflow_amd import npe
    from posteriflow_amd.inference import sample_event
    torch.manual_seed(0)
    model = npe.LeanNPE().to(dev).eval()
    strain = torch.randn(1, 3, 16384, device=dev)
    out = {}
    for prec in dict.fromkeys((precision, "fp32")):
        model.set_precision(prec)
        sample_event(model, strain, n, seed=1)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = sample_event(model, strain, n, seed=1)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(r["logq"]).all())
        key = "sample_event_draws_plus_logq_per_s" + ("" if prec == precision else f"_{prec}")
        out[key] = n / sorted(ts)[len(ts) // 2]
    out["sample_event_reference_published_draws_per_s"] = {"cpu": 1120, "mps": 1850}     # BASELINE.md section 1 (other hardware)
    return out


def generic_head(dev, precision, batch=4096):
    """A flow size outside the scheduled kernels' set -- FlowHead(12, 384, 24) of experiments/frozen_context_heads.py:159-163 --
    through the generic kernel: log-density of 4096 rows."""
    from posteriflow_amd import NSFPosteriorFlow
    torch.manual_seed(0)
    fh = NSFPosteriorFlow(11, 288, 384, 12, 24, 3.0, temperature_scale=1.0, use_masked_context=False).to(dev).eval()
    for q in fh.parameters():
        q.requires_grad_(False)
    fh.precision = precision
    x = torch.rand(batch, 11, device=dev) * 2 - 1
    c = torch.randn(batch, 288, device=dev)
    fh.compute_psd_aware_nll(x, c, None)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fh.compute_psd_aware_nll(x, c, None)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    return {"generic_head_12x384x24_samples_per_s": batch / sorted(ts)[len(ts) // 2]}


def coherent_geometry(dev, events=1024):
    """CoherentEncoder's frequency-domain geometry features (coherent_encoder.py:79-116: 3 rfft + 3 irfft of 16 384 samples
    per event + band reductions) by pf_geom_features: events per second."""
    from posteriflow_amd import npe
    enc = npe.CoherentEncoder(context_dim=256, psd_bands=16).to(dev).eval()
    x = torch.randn(events, 3, 16384, device=dev)
    ts = []
    with torch.no_grad():
        enc._geometry_rel(x)
        for _ in range(7):                  # median of single calls: one stall of the box does not decide the figure
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            enc._geometry_rel(x)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
    return {"coherent_geometry_events_per_s": events / sorted(ts)[len(ts) // 2]}


def config4_train_step(dev, precision, events=1024, reps=5):
    """BASELINE config 4's per-GPU share: one training step at 1024 events -- on-GPU remix of resident pools ->
    LeanNPE batch_nll (encoder + flow) -> backward -> clip + AdamW (experiments/train_lean_npe.py:357-368)."""
    from posteriflow_amd import npe, train
    from posteriflow_amd.remix import synthetic_dataset
    ds = synthetic_dataset(dev, n_noise=512, n_events=512, seed=0)
    torch.manual_seed(0)
    model = npe.LeanNPE().to(dev).train().set_precision(precision)
    model.flatten_parameters()                       # one flat leaf each for the flow and the encoder
    opt = train.make_optimizer(model)
    sched = train.make_scheduler(opt, 10000)
    g = torch.Generator(device=dev).manual_seed(0)

    def step():
        idx = torch.randint(0, ds.n_events, (events,), device=dev, generator=g)
        strain, labels, nsig, _ = ds.batch(idx, generator=g)
        # row_cap="exact": the existing pairs; sync=False: the loss stays on the device (read once, after the timed steps)
        return train.train_step(model, opt, sched, strain, labels, nsig, sync=False)["loss"]

    for _ in range(2):
        loss = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    assert bool(torch.isfinite(torch.as_tensor(loss)).all()), "training step produced a non-finite loss"
    return {f"train_step_ms_{events}": dt * 1e3, f"train_events_per_s_{events}": events / dt}


def pmc_traffic(args):
    """bytes per launch at the L2's memory side from the committed PMC passes of the build being timed (rocprofv3
    cannot run inside this process): only for the configuration they were collected on, else null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if args.batch != 4096 or args.precision != "bf16" or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)["bytes_per_launch"]


def spawn_ranks(args):
    """`bench.py --gpus N` outside a launcher: start N ranks (one process per GPU, torch.distributed.run) as CHILD
    processes before this process touches the GPU, pass their output through and exit with their code."""
    n_dev = torch.cuda.device_count()                    # does not initialise the GPU runtime
    if args.backend == "nccl" and n_dev < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible (use --backend gloo to rehearse on fewer)")
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 400))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="N = 1: replay one HIP graph per step instead of one pre-bound launch")
    ap.add_argument("--no-graph", action="store_true", help="(default now; kept for older command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sampling / embedding side measurements")
    ap.add_argument("--collective", action="store_true", help="run the N > 1 step (with its all-reduce) on a single rank")
    ap.add_argument("--allreduce-every", type=int, default=1,
                    help="N > 1: steps per all-reduce of the in-kernel (sum nll, rows) accumulator (1 = every step, the "
                         "contract's step; the windowed figure is reported under extras)")
    ap.add_argument("--settle", type=int, default=300,
                    help="untimed launches BEFORE the W warm-up steps (part of set-up, like packing the weights): an idle GPU "
                         "takes tens of milliseconds to reach the clock it holds under load, and K = 20 steps are 2 ms")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path on fewer GPUs than ranks (collective through the host)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                 # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks or drop the flag")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()         # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    collective = world > 1 or args.collective            # --collective: rehearse the N > 1 step on one rank
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)}")
    flow = build_flow(dev, args.precision).freeze_packed()
    log("weights packed")
    x, ctx = make_inputs(args.batch, 1 + rank, dev)
    nll = torch.empty(args.batch, device=dev)
    # N > 1: (sum nll, count) all-reduced over RCCL -- the path's only exchange.  The flow kernel reduces the pair
    # itself (wave shuffle + one pair of atomics per workgroup) and zeroes the accumulator of the NEXT window, so a
    # step is exactly one kernel launch (a pre-bound C call: the Python wrapper's per-call work would make the loop
    # host-bound at ~250 us) and one asynchronous 128-byte all-reduce that overlaps the next steps' kernels (three
    # rotating accumulators).  --allreduce-every M > 1 reduces a window of M steps at once.
    NBUF = 3
    red = [torch.zeros(16, 2, device=dev, dtype=torch.float32) for _ in range(NBUF)]    # PF_REDUCE_SLOTS pairs each
    works = [None] * NBUF
    state = {"k": 0, "last": 0, "M": max(1, args.allreduce_every)}

    use_graph = args.graph and not collective
    graph = None
    # N > 1: the flow's stream gets the HIGH priority, so that when a step's kernel and the previous step's RCCL kernel
    # become ready together the flow's 256 workgroups are placed first and the collective fills in behind them
    stream = torch.cuda.Stream(dev, priority=-1 if (collective and not os.environ.get("PF_BENCH_FLAT_PRIORITY")) else 0)
    with torch.cuda.stream(stream):
        plain = flow.bind_nll(x, ctx, nll, stream=stream)
        reduce_launch = flow.bind_nll(x, ctx, nll, sum_count=red, stream=stream) if collective else None

        def flush(a):
            """all-reduce the window accumulated in red[a]"""
            if os.environ.get("PF_BENCH_SKIP_ALLREDUCE"):           # timing experiment only
                pass
            elif args.backend == "nccl":
                works[a] = dist.all_reduce(red[a], async_op=True)
            else:                                         # gloo rehearsal on CPU tensors
                host = red[a].cpu()
                dist.all_reduce(host)
                red[a].copy_(host)
            state["last"] = a

        def drain():
            M = state["M"]
            if collective and state["k"] % M:             # a partly filled window: reduce it now
                w = state["k"] // M
                flush(w % NBUF)
                state["k"] = (w + 1) * M
            for i, w in enumerate(works):
                if w is not None:
                    w.wait()
                    works[i] = None

        def step():
            if not collective:
                plain()
                return
            # every launch adds its (sum nll, rows) to the window's accumulator red[a] inside the kernel and zeroes
            # the next window's; the window (M steps) is all-reduced once, asynchronously, over three rotating pairs
            M = state["M"]
            k = state["k"]
            w, pos = divmod(k, M)
            a = w % NBUF
            if pos == 0:
                nxt = (a + 1) % NBUF
                if works[nxt] is not None:
                    works[nxt].wait()                     # stream-side: its all-reduce is done before a launch zeroes it
                    works[nxt] = None
            reduce_launch(a)
            if pos == M - 1:
                flush(a)
            state["k"] = k + 1

        def timed_loop(run, n_steps):
            """barrier + synchronize, n_steps steps, synchronize + barrier; max over ranks"""
            drain()
            stream.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(n_steps):
                run()
            drain()
            e1.record(stream)
            stream.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            return el, e0.elapsed_time(e1)

        step()
        drain()
        stream.synchronize()
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                step()
        log("step captured" if use_graph else "one pre-bound launch per step" + (" + async all-reduce" if collective else ""))
        run = graph.replay if graph is not None else step
        for _ in range(max(0, args.settle)):              # clock settling (set-up), then the contract's W warm-up steps
            plain()
        stream.synchronize()
        for _ in range(args.warmup):
            run()
        elapsed, dev_ms = timed_loop(run, args.steps)
        log(f"timed {args.steps} steps: {elapsed * 1e3 / args.steps:.4f} ms/step (device {dev_ms / args.steps:.4f})")
        mean_nll = None
        if collective:
            last = red[state["last"]].double().sum(0).cpu()   # (sum nll, count) of the last window over all ranks and slots
            mean_nll = (last[0] / last[1]).item()
        windowed = None
        if collective and state["M"] == 1:               # beside the contract number: one all-reduce per 16 steps
            state["M"], state["k"] = 16, 0
            for i in range(NBUF):
                red[i].zero_()
            for _ in range(max(args.warmup, 16)):
                run()
            el16, _ = timed_loop(run, args.steps)
            windowed = world * args.batch * args.steps / el16
            state["M"] = 1

    kernel_ms = kernel_time_ms(flow, x, ctx, nll, stream, max(20, min(args.steps, 200)))
    log(f"kernel-only: {kernel_ms * 1e3:.2f} us")

    if rank == 0:
        fl = flops_per_sample()
        ach = args.batch * fl / (kernel_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        lib = __import__("posteriflow_amd")._lib.lib()
        # the three FLOP conventions side by side: `achieved` / `frac` use SURVEY 8d's ALGORITHMIC dense-GEMM count (the
        # contract's figure: masks not discounted); the MFMAs the dispatched kernel really issues multiply the packed stream,
        # in which all-zero fragments of the autoregressive masks do not exist (pf_flow_issued_flop_per_row), and the useful
        # share of those is the mask-aware count
        fl_issued = int(lib.pf_flow_issued_flop_per_row(flow._desc(wide=flow._use_wide(args.batch))))
        fl_masked = masked_flops_per_sample()
        tf = lambda f: args.batch * f / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "flow.log_prob samples/sec at batch 4096",
            "value": world * args.batch * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3 flow: 8-layer MAF-RQS, D=15, C=288, H=256, K=16, "
                                   f"tail_bound 5, default init with the final MADE layer x{FINAL_LAYER_SCALE:g} "
                                   "(BASELINE.md section 3 says x30: deep random flows at x30 are chaotic in float64 itself, "
                                   "x30 is parity-tested at L = 2 and recorded at L = 8 in tests/test_parity_r4_gpu.py), "
                                   "context resident in HBM",
                       "final_layer_scale": FINAL_LAYER_SCALE,
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "rows_per_workgroup": int(lib.pf_flow_rows_per_workgroup(flow._desc(wide=flow._use_wide(args.batch)), args.batch)),
                       "launch": "hipGraph" if graph is not None else (f"pre-bound launch, in-kernel (sum nll, rows) + async all-reduce every {max(1, args.allreduce_every)} steps" if collective else "pre-bound launch"),
                       "settle": max(0, args.settle),   # untimed clock-settling launches in front of the W warm-up steps
                       "global_mean_nll": mean_nll,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": pmc_traffic(args),
                         "kernel": flow.forward_kernel_name(args.batch), "kernel_ms": kernel_ms,
                         "flop_convention": "achieved/frac: SURVEY 8d dense-GEMM count (masked zeros counted); *_issued: "
                                            "MFMAs the kernel executes (mask-compressed stream); *_mask_aware: useful FLOP only",
                         "flop_per_sample": fl, "flop_per_sample_issued": fl_issued, "flop_per_sample_mask_aware": fl_masked,
                         "achieved_issued": tf(fl_issued), "frac_issued": tf(fl_issued) / peak,
                         "achieved_mask_aware": tf(fl_masked), "frac_mask_aware": tf(fl_masked) / peak,
                         "device_ms_per_step": dev_ms / args.steps},
        }
        if world == 1 and args.batch <= 20479:
            # the bound this design actually has at 16 rows per workgroup: every workgroup ingests the whole packed stream
            floor_us, nbytes, wgs = ingest_floor_us(flow, args.batch, dev, stream)
            out["roofline"]["ceiling"] = {
                "bound": "per-CU L2 ingest", "bytes_per_cu": nbytes, "workgroups": wgs, "floor_us": floor_us,
                "frac_of_floor": floor_us / (kernel_ms * 1e3),
                "frac_if_at_floor": args.batch * fl / (floor_us * 1e-6) / 1e12 / peak,
                "note": "floor_us = a compute-free read of the same packed weight stream by the same number of workgroups, "
                        "measured in this run (pf_diag_stream_ingest); frac_of_floor = floor_us / kernel time; "
                        "frac_if_at_floor = the MFMA-roofline fraction a kernel running AT that floor would show"}
        if windowed is not None:
            out["extras"] = {"value_allreduce_every_16": windowed}
        if not args.no_extras and world == 1:
            out.setdefault("extras", {}).update(extras(flow, dev, args.batch))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], want = cpu_baseline(flow, args.batch)
            out["parity"] = parity_check(flow, dev, args.batch, want)
            if args.precision == "bf16" and "extras" in out and f"forward_kernel_us_fp32_{args.batch}" in out["extras"]:
                # the parity-bearing mode as a first-class block: the reference's arithmetic is fp32 (north_star: 1e-5
                # relative fp32); the bf16 headline follows the same-rounding oracle, not this tolerance
                ex, us = out["extras"], out["extras"][f"forward_kernel_us_fp32_{args.batch}"]
                out["fp32"] = {"samples_per_s": ex[f"forward_samples_per_s_fp32_{args.batch}"], "kernel_ms": us * 1e-3,
                               "peak_tflops": PEAK_TFLOPS["fp32"], "frac": ex[f"forward_roofline_frac_fp32_{args.batch}"],
                               "frac_mask_aware": ex[f"forward_roofline_frac_mask_aware_fp32_{args.batch}"],
                               "dtype": "f32 (v_mfma_f32_16x16x4_f32, fp32 accumulate)", "parity": out["parity"]["fp32"]}
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
