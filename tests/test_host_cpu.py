"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol of
include/pf_hip.h, host-side planning / packing logic, error conventions, and the 2-rank
gloo path of the data-parallel helper.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from posteriflow_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "pf_hip.h")).read()
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    handle = lib.lib()
    for name in declared:
        assert hasattr(handle, name)
    assert b"gfx950" in handle.pf_version()


def desc_of(lib, D, C, H, K, L, prec, flags=0):
    return lib.PfFlowDesc(D, C, H, K, L, 2, 5.0, 1e-3, 1e-3, 1e-3, lib.PRECISIONS[prec], flags)


def raw_layout(D, C, H, K, nb=2):
    """(name, rows, cols) of one layer in the raw parameter order of pf_hip.h."""
    M = 3 * K - 1
    out = [("in_w", H, D), ("in_b", H, 1)]
    if C:
        out += [("c_w", H, C), ("c_b", H, 1)]
    for b in range(nb):
        if C:
            out += [(f"g{b}_w", H, C), (f"g{b}_b", H, 1)]
        out += [(f"w0{b}_w", H, H), (f"w0{b}_b", H, 1), (f"w1{b}_w", H, H), (f"w1{b}_b", H, 1)]
    return out + [("out_w", D * M, H), ("out_b", D * M, 1)]


@pytest.mark.parametrize("D,C,H,K,L", [(11, 288, 256, 16, 2), (15, 288, 256, 16, 2), (4, 0, 64, 8, 2),
                                       (7, 40, 128, 10, 3), (2, 5, 64, 4, 1)])
@pytest.mark.parametrize("hoist", [0, 1])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_pack_map_covers_exactly_the_unmasked_weights(lib, D, C, H, K, L, prec, hoist):
    """Every weight the autoregressive masks keep appears in the map (once; the x-input layer
    twice in bf16 mode for the hi/lo split), every masked weight never; all biases appear once."""
    from oracle import nflows_restated as nfr
    d = desc_of(lib, D, C, H, K, L, prec, hoist)
    h = lib.lib()
    n = h.pf_flow_pack_map_len(C_byref(d))
    assert n > 0
    m = np.empty(n, dtype=np.int32)
    assert h.pf_flow_build_pack_map(C_byref(d), m.ctypes.data) == 0
    per_layer = h.pf_flow_raw_param_count(C_byref(d)) // L
    counts = np.bincount(m[m >= 0], minlength=per_layer * L)
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, C or None, K, 5.0)
    net = t.autoregressive_net
    masks = {"in_w": net.initial_layer.mask.numpy(), "out_w": net.final_layer.mask.numpy()}
    for b in range(2):
        masks[f"w0{b}_w"] = net.blocks[b].linear_layers[0].mask.numpy()
        masks[f"w1{b}_w"] = net.blocks[b].linear_layers[1].mask.numpy()
    for layer in range(L):
        off = layer * per_layer
        for name, r, c in raw_layout(D, C, H, K):
            got = counts[off:off + r * c].reshape(r, c)
            want = masks.get(name, np.ones((r, c))).astype(np.int64)
            if name == "in_w" and prec == "bf16":
                want = 2 * want
            assert np.array_equal(got, want), (layer, name)
            off += r * c
        assert off == (layer + 1) * per_layer
    assert sum(p.numel() for p in t.parameters()) == per_layer


@pytest.mark.parametrize("D,C,H,K,L", [(11, 288, 256, 16, 2), (15, 288, 256, 16, 1), (4, 0, 64, 8, 2), (7, 40, 128, 10, 2)])
def test_backward_pack_map_is_the_masked_matrices_and_their_transposes(lib, D, C, H, K, L):
    """PF_FLAG_BWD (the weight stream of the bf16 backward: chain + conditioner re-evaluation): gathering a raw parameter
    vector through the map and decoding the A-fragments (lane (i, g), element j of fragment (tile t, k-step ks) =
    A[16 t + i][32 ks + 8 g + j]) must give exactly
      * the transposed region: (W * mask)^T of the final layer, of both linears of both blocks, of the initial layer;
      * the forward region: W * mask of the initial layer (twice: x enters as hi | lo), the three context projections,
        the four block linears, the final layer -- zero padded;
      * the fp32 bias region in the kernel's order;
    and the compute entry points refuse the flag."""
    from oracle import nflows_restated as nfr
    d = desc_of(lib, D, C, H, K, L, "bf16", lib.PF_FLAG_BWD)
    h = lib.lib()
    n = h.pf_flow_pack_map_len(C_byref(d))
    M = 3 * K - 1
    NT, HK, KSF, CKB, NTF = H // 16, H // 32, (D * M + 31) // 32, (C + 31) // 32, (D * M + 15) // 16
    bwd_frags = NT * KSF + 4 * NT * HK + HK
    fwd_frags = NT + (3 * NT * CKB if C else 0) + 4 * NT * HK + NTF * HK
    nbias = H + (3 * H if C else 0) + 4 * H + 16 * NTF
    assert n == L * (bwd_frags + fwd_frags) * 512 + L * nbias
    assert h.pf_flow_packed_bytes(C_byref(d)) == 2 * L * (bwd_frags + fwd_frags) * 512 + 4 * L * nbias
    m = np.empty(n, dtype=np.int32)
    assert h.pf_flow_build_pack_map(C_byref(d), m.ctypes.data) == 0
    per_layer = h.pf_flow_raw_param_count(C_byref(d)) // L
    raw = np.random.default_rng(0).standard_normal(per_layer * L).astype(np.float32)
    gathered = np.where(m >= 0, raw[np.maximum(m, 0)], 0.0)
    nw = L * (bwd_frags + fwd_frags) * 512
    bwd = gathered[:L * bwd_frags * 512].reshape(L, bwd_frags, 64, 8)
    fwd = gathered[L * bwd_frags * 512:nw].reshape(L, fwd_frags, 64, 8)
    bias = gathered[nw:].reshape(L, nbias)
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, C or None, K, 5.0)
    net = t.autoregressive_net

    def decode(frags, tiles, nks):          # [tiles * nks, 64, 8] -> A [16 tiles, 32 nks]
        f = frags.reshape(tiles, nks, 4, 16, 8)                         # lane = 16 g + i
        return f.transpose(0, 3, 1, 2, 4).reshape(16 * tiles, 32 * nks)  # [t, i, ks, g, j]

    def padded(a, rows, cols):
        out = np.zeros((rows, cols), np.float32)
        out[:a.shape[0], :a.shape[1]] = a
        return out

    for layer in range(L):
        off, mats = layer * per_layer, {}
        for name, r, c in raw_layout(D, C, H, K):
            mats[name] = raw[off:off + r * c].reshape(r, c)
            off += r * c
        Win, Wf = mats["in_w"] * net.initial_layer.mask.numpy(), mats["out_w"] * net.final_layer.mask.numpy()
        blk = [(mats[f"w0{j}_w"] * net.blocks[j].linear_layers[0].mask.numpy(),
                mats[f"w1{j}_w"] * net.blocks[j].linear_layers[1].mask.numpy()) for j in range(2)]
        # transposed region (the chain walks backwards: second linear first)
        fr, pos = bwd[layer], 0
        assert np.array_equal(decode(fr[pos:pos + NT * KSF], NT, KSF), padded(Wf.T, H, 32 * KSF)); pos += NT * KSF
        for j in range(2):
            for w in (blk[j][1], blk[j][0]):
                assert np.array_equal(decode(fr[pos:pos + NT * HK], NT, HK), w.T), (layer, j); pos += NT * HK
        assert np.array_equal(decode(fr[pos:pos + HK], 1, HK), padded(Win.T, 16, H)); pos += HK
        assert pos == bwd_frags
        # forward region
        fr, pos = fwd[layer], 0
        want = np.zeros((H, 32), np.float32)
        want[:, :D] = Win
        want[:, 16:16 + D] = Win
        assert np.array_equal(decode(fr[pos:pos + NT], NT, 1), want); pos += NT
        if C:
            for name in ("c_w", "g0_w", "g1_w"):
                assert np.array_equal(decode(fr[pos:pos + NT * CKB], NT, CKB), padded(mats[name], H, 32 * CKB)), name
                pos += NT * CKB
        for j in range(2):
            for w in blk[j]:
                assert np.array_equal(decode(fr[pos:pos + NT * HK], NT, HK), w), (layer, j); pos += NT * HK
        assert np.array_equal(decode(fr[pos:pos + NTF * HK], NTF, HK), padded(Wf, 16 * NTF, H)); pos += NTF * HK
        assert pos == fwd_frags
        # biases
        names = ["in_b"] + (["c_b", "g0_b", "g1_b"] if C else []) + ["w00_b", "w10_b", "w01_b", "w11_b"]
        want = np.concatenate([mats[k].ravel() for k in names] + [padded(mats["out_b"].reshape(1, -1), 1, 16 * NTF).ravel()])
        assert np.array_equal(bias[layer], want)
    # the flag is a packing layout only
    assert h.pf_flow_workspace_bytes(C_byref(d), 16) == -1 and h.pf_flow_rows_per_workgroup(C_byref(d), 16) == -1
    assert h.pf_flow_forward(C_byref(d), 16, 16, 16, None, None, 16, None, None, 16, None, 0, None) == lib.PF_ERR_UNSUPPORTED
    assert h.pf_flow_pack_map_len(C_byref(desc_of(lib, D, C, H, K, L, "fp32", lib.PF_FLAG_BWD))) == -1


@pytest.mark.parametrize("D,K,L", [(15, 16, 2), (11, 16, 3)])
def test_wide_pack_map_covers_exactly_the_unmasked_weights(lib, D, K, L):
    """The large-batch kernel's layout (PF_FLAG_WIDE, csrc/pf_wide_layout.h): one common stream of 32 x 16 fragments in
    the accumulator-permuted k order.  Same coverage property as the per-wave streams: every unmasked weight once (the
    x-input layer twice: hi | lo), every masked weight never, every bias once; plus the sizes the kernel assumes."""
    from oracle import nflows_restated as nfr
    C_, H = 288, 256
    d = desc_of(lib, D, C_, H, K, L, "bf16", lib.PF_FLAG_WIDE)
    h = lib.lib()
    n = h.pf_flow_pack_map_len(C_byref(d))
    assert n > 0
    m = np.empty(n, dtype=np.int32)
    assert h.pf_flow_build_pack_map(C_byref(d), m.ctypes.data) == 0
    per_layer = h.pf_flow_raw_param_count(C_byref(d)) // L
    counts = np.bincount(m[m >= 0], minlength=per_layer * L)
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, C_, K, 5.0)
    net = t.autoregressive_net
    masks = {"in_w": 2 * net.initial_layer.mask.numpy(), "out_w": net.final_layer.mask.numpy()}
    for b in range(2):
        masks[f"w0{b}_w"] = net.blocks[b].linear_layers[0].mask.numpy()
        masks[f"w1{b}_w"] = net.blocks[b].linear_layers[1].mask.numpy()
    for layer in range(L):
        off = layer * per_layer
        for name, r, c in raw_layout(D, C_, H, K):
            got = counts[off:off + r * c].reshape(r, c)
            assert np.array_equal(got, masks.get(name, np.ones((r, c))).astype(np.int64)), (layer, name)
            off += r * c
    # stream: whole fragments, a multiple of the 72-fragment ring per layer, one ring of zero fragments behind it
    packed = h.pf_flow_packed_bytes(C_byref(d))
    bias_floats = 2048 + 8 * 96
    frags = (packed - L * bias_floats * 4) // 1024
    assert (packed - L * bias_floats * 4) % 1024 == 0 and (frags - 72) % (72 * L) == 0
    assert n == frags * 512 + L * bias_floats
    assert (m[(frags - 72) * 512: frags * 512] == -1).all()
    # the layout's two kernels, picked by rounds x measured round time: 128 rows per workgroup (large-batch) / 64 (mid-batch)
    assert h.pf_flow_rows_per_workgroup(C_byref(d), 131072) == 128
    assert h.pf_flow_forward_kernel_name(C_byref(d), 131072) == f"pf::flow_wide_kernel<{D}, 18>".encode()
    assert h.pf_flow_rows_per_workgroup(C_byref(d), 16384) == 64
    assert h.pf_flow_forward_kernel_name(C_byref(d), 16384) == f"pf::flow_mid_kernel<{D}, 18>".encode()
    # shapes the wide kernel is not built for are refused, not silently served by another layout
    for bad in (desc_of(lib, 7, 288, 256, 16, 2, "bf16", lib.PF_FLAG_WIDE), desc_of(lib, 15, 256, 256, 16, 2, "bf16", lib.PF_FLAG_WIDE),
                desc_of(lib, 15, 288, 256, 9, 2, "bf16", lib.PF_FLAG_WIDE),
                desc_of(lib, 15, 288, 256, 16, 2, "fp32", lib.PF_FLAG_WIDE), desc_of(lib, 15, 288, 128, 16, 2, "bf16", lib.PF_FLAG_WIDE),
                desc_of(lib, 15, 288, 256, 16, 2, "bf16", lib.PF_FLAG_WIDE | lib.PF_FLAG_HOIST_CTX)):
        assert h.pf_flow_pack_map_len(C_byref(bad)) < 0


def C_byref(x):
    return C.byref(x)


def test_plan_sizes_and_unsupported_shapes(lib):
    h = lib.lib()
    d = desc_of(lib, 15, 288, 256, 16, 8, "bf16")
    packed = h.pf_flow_packed_bytes(C_byref(d))
    dense_bf16 = 2 * 8 * (256 * 15 + 3 * 256 * 288 + 4 * 256 * 256 + 15 * 47 * 256)
    assert 0.6 * dense_bf16 < packed < 0.9 * dense_bf16        # mask-aware stream is smaller than dense
    assert h.pf_flow_rows_per_workgroup(C_byref(d), 4096) == 16
    assert h.pf_flow_rows_per_workgroup(C_byref(d), 65536) == 32
    # shapes outside the scheduled kernels' set go to the generic kernel (pf_flow_generic.hip): dense masked fragment arrays
    for D_, C_, H_, K_, L_ in ((17, 288, 256, 16, 8),          # D > H/16
                               (11, 288, 384, 24, 12),         # FlowHead(12, 384, 24), experiments/frozen_context_heads.py:159-163
                               (11, 288, 256, 17, 8),          # K > 16
                               (5, 0, 48, 3, 2)):              # H not a scheduled width
        for prec in ("bf16", "fp32"):
            gd = desc_of(lib, D_, C_, H_, K_, L_, prec)
            ks = 32 if prec == "bf16" else 16
            nt, kx = H_ // 16, -(-(2 * (-(-D_ // 16) * 16) if prec == "bf16" else D_) // ks)
            kc, kh, tf = -(-C_ // ks), -(-H_ // ks), -(-(D_ * (3 * K_ - 1)) // 16)
            frags = nt * kx + (3 * nt * kc if C_ else 0) + 4 * nt * kh + tf * kh
            bias = H_ + (3 * H_ if C_ else 0) + 4 * H_ + 16 * tf
            assert h.pf_flow_packed_bytes(C_byref(gd)) == L_ * (frags * 1024 + bias * 4)
            assert h.pf_flow_rows_per_workgroup(C_byref(gd), 4096) == 16
            assert h.pf_flow_forward(C_byref(gd), None, None, None, None, None, 4, None, None, None, None, 0, None) == lib.PF_ERR_BAD_ARG
    for bad in (desc_of(lib, 11, 5000, 256, 16, 8, "bf16"),    # C too large for the in-layer kernels and for the generic LDS image
                desc_of(lib, 33, 0, 256, 8, 2, "bf16"),        # D > 32
                desc_of(lib, 8, 0, 520, 8, 2, "bf16"),         # H > 512
                desc_of(lib, 8, 0, 250, 8, 2, "bf16"),         # H % 16
                desc_of(lib, 32, 288, 512, 32, 2, "fp32")):    # 32 x 95 parameters per row: LDS image > 160 KB
        assert h.pf_flow_packed_bytes(C_byref(bad)) == -1
        assert h.pf_flow_forward(C_byref(bad), None, None, None, None, None, 4, None, None, None, None, 0, None) == lib.PF_ERR_UNSUPPORTED
    # argument checks happen before any launch
    assert h.pf_flow_forward(C_byref(d), None, None, None, None, None, 4, None, None, None, None, 0, None) == lib.PF_ERR_BAD_ARG
    assert h.pf_flow_forward(C_byref(d), None, None, None, None, None, 0, None, None, None, None, 0, None) == lib.PF_OK
    assert h.pf_flow_forward(C_byref(d), None, None, None, None, None, -1, None, None, None, None, 0, None) == lib.PF_ERR_BAD_ARG
    assert b"null" in h.pf_last_error() or b"negative" in h.pf_last_error()
    with pytest.raises(ValueError):
        lib.check(lib.PF_ERR_BAD_ARG, "x")
    with pytest.raises(NotImplementedError):
        lib.check(lib.PF_ERR_UNSUPPORTED, "x")


def test_module_api_on_cpu_is_loud_not_silent(lib):
    from posteriflow_amd import NSFPosteriorFlow, create_flow_model
    from oracle.flow_ref import NSFPosteriorFlowRef
    flow = create_flow_model("nsf", 11, 288, config={"flow_config": {"num_layers": 3, "tail_bound": 5.0}},
                             use_masked_context=False)
    assert isinstance(flow, NSFPosteriorFlow) and flow.num_layers == 3 and flow._tail_bound == 5.0
    assert NSFPosteriorFlow(4, 0, 64, 1, 8, tail_bound=5)._tail_bound == 3.0       # flows.py:517 quirk
    with pytest.raises(ValueError):
        create_flow_model("realnvp", 4)
    mc = NSFPosteriorFlow(16, 256, 256, 1, 16)            # 256 % 16 == 0 -> masked context auto-on (flows.py:409)
    assert mc.use_masked_context and mc.context_block_dim == 16 and len(mc.transform._transforms) == 1
    with pytest.raises(RuntimeError):                   # no CPU fallback
        flow(torch.zeros(2, 11), torch.zeros(2, 288))
    with pytest.raises(RuntimeError):
        flow.inverse(torch.zeros(2, 11), torch.zeros(2, 288))
    # state_dict is interchangeable with the oracle's (nflows names) plus the flow._transform aliases
    ref = NSFPosteriorFlowRef(11, 288, 256, 3, 16, 5.0)
    sd = flow.state_dict()
    rsd = ref.state_dict()
    assert set(rsd) <= set(sd)
    assert {k for k in sd if k not in rsd} == {"flow._transform." + k[len("transform."):] for k in rsd if k.startswith("transform.")}
    for k, v in rsd.items():
        assert sd[k].shape == v.shape, k
    assert torch.equal(sd["transform._transforms.1.autoregressive_net.final_layer.mask"],
                       rsd["transform._transforms.1.autoregressive_net.final_layer.mask"])
    with pytest.raises(ValueError):
        flow.set_autoregressive_order([0] * 11)
    base = flow.base_dist
    with pytest.raises(ValueError):
        base.log_prob(torch.zeros(2, 11), torch.zeros(2, 5))


def test_psd_scaled_normal_matches_reference_golden(golden_small):
    from posteriflow_amd import PSDScaledNormal
    base = PSDScaledNormal([11])
    z = torch.from_numpy(golden_small["base_z"]); ls = torch.from_numpy(golden_small["base_ls"])
    np.testing.assert_allclose(base.log_prob(z, ls).numpy(), golden_small["base_logp_ls"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(base.log_prob(z, torch.zeros_like(z)).numpy(), golden_small["base_logp_zero"],
                               rtol=1e-5, atol=1e-5)
    assert base.sample(7, ls).shape == (32, 7, 11)


DIST_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from posteriflow_amd.dist import global_mean_nll, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = torch.Generator().manual_seed(0)
nll = torch.randn(1001, generator=g) * 3 + 20          # same on every rank
lo, hi = shard_bounds(1001, rank, world)
mean = global_mean_nll(nll[lo:hi])
want = nll.double().mean()
assert abs(mean.item() - want.item()) < 1e-12, (mean.item(), want.item())
covered = torch.zeros(1001); covered[lo:hi] = 1
dist.all_reduce(covered)
assert torch.all(covered == 1)
dist.destroy_process_group()
print("ok", rank)
"""


def test_data_parallel_mean_nll_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(DIST_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


def test_conditioner_dropout_host_rules():
    """nflows applies dropout inside every residual block in train mode (create_flow_model defaults to 0.15,
    flows.py:1008).  The HIP training forward applies it (tests/test_flow_dropout_gpu.py); on the host: the probability
    is validated, the serving entry points and the inverse refuse a train()-mode flow with dropout > 0 instead of
    silently evaluating a different model, eval() and dropout = 0 are unaffected, and the seed follows torch's generator."""
    import pytest
    from posteriflow_amd import NSFPosteriorFlow, create_flow_model
    flow = create_flow_model("nsf", 4, 8, num_layers=1, hidden_features=64, num_bins=4)
    assert flow.dropout == 0.15
    x, ctx = torch.zeros(2, 4), torch.zeros(2, 8)
    flow.train()
    assert flow._drop_active()
    with pytest.raises(RuntimeError, match="eval"):
        flow.nll_into(x, ctx, torch.empty(2))
    with pytest.raises(RuntimeError, match="eval"):
        flow._inverse_call(x, ctx, 2)
    torch.manual_seed(5)
    a = [flow._draw_dropout_seed() for _ in range(3)]
    torch.manual_seed(5)
    assert a == [flow._draw_dropout_seed() for _ in range(3)] and len(set(a)) == 3 and all(0 <= v < 1 << 62 for v in a)
    flow.eval()
    assert not flow._drop_active()
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):     # gets as far as the device check
        flow.compute_psd_aware_nll(x, ctx, None)
    with pytest.raises(ValueError):
        NSFPosteriorFlow(4, 8, 64, 1, 4, dropout=1.0)


def test_nflows_cross_check_script_skips_cleanly_without_nflows():
    """scripts/compare_with_nflows.py (SURVEY 8c KAT 8): exit 77 where nflows is not importable (here, the GPU boxes),
    0 / 1 where it is; it must not need a GPU or anything outside the repo."""
    try:
        import nflows  # noqa: F401
        pytest.skip("nflows is importable here: run the script itself")
    except ImportError:
        pass
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "compare_with_nflows.py")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 77 and "not importable" in out.stdout


def test_generic_pack_map_is_the_masked_dense_matrices(lib):
    """generic plan (H = 48, D = 5, K = 3, C = 7): every fragment element decodes to the masked weight it should hold"""
    import numpy as np
    h = lib.lib()
    D_, C_, H_, K_, L_ = 5, 7, 48, 3, 2
    M_ = 3 * K_ - 1
    for prec, per, ks in (("bf16", 8, 32), ("fp32", 4, 16)):
        d = desc_of(lib, D_, C_, H_, K_, L_, prec)
        n = h.pf_flow_pack_map_len(C_byref(d))
        m = np.zeros(n, dtype=np.int32)
        assert h.pf_flow_build_pack_map(C_byref(d), m.ctypes.data_as(C.c_void_p)) == lib.PF_OK
        raw_n = h.pf_flow_raw_param_count(C_byref(d))
        per_layer = raw_n // L_
        deg = lambda u: u % max(1, D_ - 1) + min(1, D_ - 1)
        unit_of = sorted(range(H_), key=lambda u: (deg(u), u))        # the plan stores the hidden units in degree order
        nt = H_ // 16
        xh = -(-D_ // 16) * 16
        kx = -(-(2 * xh if prec == "bf16" else D_) // ks)
        # first matrix of layer 1: W_in [H, D]
        frags_per_layer = nt * kx + 3 * nt * (-(-C_ // ks)) + 4 * nt * (-(-H_ // ks)) + (-(-(D_ * M_) // 16)) * (-(-H_ // ks))
        base = 1 * frags_per_layer * 64 * per
        for t in range(nt):
            for s_ in range(kx):
                for lane in range(64):
                    for e in range(per):
                        row, k = 16 * t + (lane & 15), ks * s_ + per * (lane >> 4) + e
                        got = m[base + ((t * kx + s_) * 64 + lane) * per + e]
                        dcol = (k % xh if k < 2 * xh else D_) if prec == "bf16" else k
                        u = unit_of[row]
                        want = per_layer + u * D_ + dcol if (dcol < D_ and deg(u) >= dcol + 1) else -1
                        assert got == want, (prec, t, s_, lane, e, got, want)
        # the map references every unmasked weight of W_in exactly once (fp32) / twice (bf16 hi | lo)
        win = m[base:base + nt * kx * 64 * per]
        used = win[win >= 0] - per_layer
        assert sorted(set(used.tolist())) == sorted(r * D_ + c for r in range(H_) for c in range(D_) if deg(r) >= c + 1)
        assert len(used) == (2 if prec == "bf16" else 1) * len(set(used.tolist()))
        # a hidden -> hidden matrix (W1 of block 0): position (row, k) holds weight (unit_of[row], unit_of[k]) where the mask allows,
        # and the non-zero columns of a tile form a prefix (what lets the kernel skip the k-steps behind it)
        kh, kc = -(-H_ // ks), -(-C_ // ks)
        hbase = base + (nt * kx + 3 * nt * kc) * 64 * per
        w1_off = per_layer + H_ * D_ + H_ + 2 * (H_ * C_ + H_)          # raw layout: in | ctx | block 0: ctx | W0 ...
        for t in range(nt):
            last = -1
            for s_ in range(kh):
                blk = m[hbase + (t * kh + s_) * 64 * per: hbase + (t * kh + s_ + 1) * 64 * per]
                if (blk >= 0).any():
                    assert last == s_ - 1, (prec, t, s_)
                    last = s_
                for lane in range(0, 64, 7):
                    for e in range(per):
                        row, k = 16 * t + (lane & 15), ks * s_ + per * (lane >> 4) + e
                        got = blk[lane * per + e]
                        if k < H_ and deg(unit_of[row]) >= deg(unit_of[k]):
                            assert got == w1_off + unit_of[row] * H_ + unit_of[k], (prec, t, s_, lane, e)
                        else:
                            assert got == -1


def test_ctypes_structs_match_the_header(lib, tmp_path):
    """every ctypes.Structure of posteriflow_amd/_lib.py has the size its C declaration in include/pf_hip.h has (a field added
    on one side only would shift every later argument silently)"""
    import shutil, subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    names = ["PfFlowDesc", "PfFlowBwdChainArgs", "PfFlowReevalArgs", "PfEmbedTrainDesc", "PfDenseArgs", "PfDenseTnArgs", "PfLnArgs",
             "PfAttnArgs", "PfPoolArgs", "PfGeomArgs"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "pf_hip.h"\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sz"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([cc, "-I", inc, str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert C.sizeof(getattr(lib, n)) == int(sizes[n]), (n, C.sizeof(getattr(lib, n)), sizes[n])


def test_geometry_entry_point_argument_checks_and_twiddles(lib):
    """pf_geom_twiddles is host code (the table of e^{-2 pi i m / 16384} in double precision, rounded once); pf_geom_features
    validates before it launches: null pointers / misalignment -> BAD_ARG, shapes outside the kernel -> UNSUPPORTED,
    an empty batch -> OK without touching the GPU"""
    import numpy as np
    h = lib.lib()
    tw = np.zeros((8192, 2), np.float32)
    assert h.pf_geom_twiddles(tw.ctypes.data_as(C.c_void_p)) == lib.PF_OK
    m = np.arange(8192, dtype=np.float64)
    assert np.array_equal(tw[:, 0], np.cos(-2 * np.pi * m / 16384).astype(np.float32))
    assert np.array_equal(tw[:, 1], np.sin(-2 * np.pi * m / 16384).astype(np.float32))
    assert h.pf_geom_twiddles(None) == lib.PF_ERR_BAD_ARG
    buf = np.zeros(64, np.float32)                              # stands in for device memory: nothing is launched below
    ptr = buf.ctypes.data_as(C.c_void_p).value
    a = lib.PfGeomArgs()
    a.clean = a.twiddle = a.spec = a.etot = a.rel = ptr
    a.batch, a.n_det, a.band_lo, a.nf, a.n_bands, a.maxlag = 0, 3, 80, 4016, 16, 122
    edges = np.linspace(0, 4016, 17).astype(int)
    for i, e in enumerate(edges):
        a.band_edge[i] = int(e)
    assert h.pf_geom_features(C.byref(a), None) == lib.PF_OK                       # empty batch
    for field, bad in (("maxlag", 128), ("n_bands", 17), ("n_det", 9), ("band_lo", 0), ("nf", 4096)):
        old = getattr(a, field)
        setattr(a, field, bad)
        assert h.pf_geom_features(C.byref(a), None) == lib.PF_ERR_UNSUPPORTED, field
        setattr(a, field, old)
    a.band_edge[5] = a.band_edge[4] - 1                                              # decreasing edges
    assert h.pf_geom_features(C.byref(a), None) == lib.PF_ERR_BAD_ARG
    a.band_edge[5] = int(edges[5])
    a.rel = None
    assert h.pf_geom_features(C.byref(a), None) == lib.PF_ERR_BAD_ARG
    a.rel, a.clean = ptr, ptr + 4                                                    # misaligned strain
    assert h.pf_geom_features(C.byref(a), None) == lib.PF_ERR_BAD_ARG
    assert b"aligned" in h.pf_last_error()


def test_generic_layout_with_the_explicit_flag_stays_in_nflows_unit_order(lib):
    """PF_FLAG_GENERIC (the layout the fp32 conditioner re-evaluation reads: its outputs feed the backward in nflows unit order)
    is NOT degree-sorted: row r of W_in is hidden unit r, and the two layouts have the same size"""
    import numpy as np
    h = lib.lib()
    D_, C_, H_, K_, L_ = 5, 7, 48, 3, 1
    deg = lambda u: u % max(1, D_ - 1) + min(1, D_ - 1)
    maps = {}
    for flags in (0, lib.PF_FLAG_GENERIC):
        d = desc_of(lib, D_, C_, H_, K_, L_, "fp32", flags)
        n = h.pf_flow_pack_map_len(C_byref(d))
        m = np.zeros(n, dtype=np.int32)
        assert h.pf_flow_build_pack_map(C_byref(d), m.ctypes.data_as(C.c_void_p)) == lib.PF_OK
        maps[flags] = m
    assert maps[0].shape == maps[lib.PF_FLAG_GENERIC].shape
    per, ks, nt, kx = 4, 16, H_ // 16, 1
    unit_of = sorted(range(H_), key=lambda u: (deg(u), u))
    for flags, order in ((0, unit_of), (lib.PF_FLAG_GENERIC, list(range(H_)))):
        m = maps[flags]
        for t in range(nt):
            for lane in range(64):
                for e in range(per):
                    row, k = 16 * t + (lane & 15), per * (lane >> 4) + e
                    got = m[((t * kx) * 64 + lane) * per + e]
                    u = order[row]
                    assert got == (u * D_ + k if (k < D_ and deg(u) >= k + 1) else -1), (flags, t, lane, e)
    # same multiset of source weights either way
    a, b = maps[0], maps[lib.PF_FLAG_GENERIC]
    assert sorted(a[a >= 0].tolist()) == sorted(b[b >= 0].tolist())


def test_embedding_forward_only_workspace_and_argument_checks(lib):
    """a no-grad call of the training-path embedding asks for the forward-only workspace (one layer's activations,
    no backward temporaries); the backward refuses such a desc.  Host arithmetic only: nothing is launched."""
    h = lib.lib()
    sizes = {}
    for prec in (lib.PF_PREC_F32, lib.PF_PREC_BF16):
        for fwd_only in (0, 1):
            d = lib.PfEmbedTrainDesc(prec, 3, 0, 0, 0.05, fwd_only, 0)
            sizes[prec, fwd_only] = h.pf_embed_train_workspace_bytes(C.byref(d), 1024)
            assert sizes[prec, fwd_only] > 0
    # fp32, 1024 three-detector events: ~16 GB with the backward's buffers, < 6 GB without
    assert sizes[lib.PF_PREC_F32, 1] < 0.4 * sizes[lib.PF_PREC_F32, 0]
    assert sizes[lib.PF_PREC_F32, 1] < 6 * 2 ** 30 < sizes[lib.PF_PREC_F32, 0]
    assert sizes[lib.PF_PREC_BF16, 1] < 0.45 * sizes[lib.PF_PREC_BF16, 0]
    # the backward of a forward-only desc is a caller error (non-null dummies: validation runs before any launch)
    import numpy as np
    buf = np.zeros(4096, np.float32)
    ptr = buf.ctypes.data_as(C.c_void_p).value
    ptr = (ptr + 255) // 256 * 256
    d = lib.PfEmbedTrainDesc(lib.PF_PREC_F32, 3, 0, 0, 0.05, 1, 0)
    rc = h.pf_embed_train_backward(C.byref(d), ptr, ptr, ptr, ptr, 0, ptr, 1 << 40, ptr, None, None, None, None)
    assert rc == lib.PF_ERR_BAD_ARG
