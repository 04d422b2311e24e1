"""ISA-level audit of the built library (CPU test: needs llvm-objdump from the ROCm image, no GPU).

VERDICT r2 item 4: the stale-accumulator bug of round 2 (`v_accvgpr_read` right behind a loop-exit MFMA) was found by
one 1e-7 parity test; its siblings are searched for in the ISA instead of by re-running GPU tests."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "posteriflow_amd", "lib", "libpfhip.so")


def _load():
    spec = importlib.util.spec_from_file_location("audit_accvgpr", os.path.join(ROOT, "scripts", "audit_accvgpr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_required_wait_states_table():
    a = _load()
    assert a.required_wait_states("v_mfma_f32_16x16x32_bf16") == 8      # the compiler's own padding (s_cbranch + s_nop 6)
    assert a.required_wait_states("v_mfma_f32_32x32x16_bf16") == 12     # guide 5.7: "8-pass XDL: 12 states"
    assert a.required_wait_states("v_mfma_f32_16x16x4_f32") == 10
    assert a.required_wait_states("v_mfma_f32_32x32x2_f32") == 18


def test_audit_flags_a_synthetic_hazard():
    a = _load()
    text = """
0000000000001000 <k>:
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]     // 000000001000: 00000000 00000000
\ts_cbranch_scc1 2                                           // 000000001008: BF850002 <k+0x14>
\ts_nop 7                                                    // 00000000100C: BF800007
\tv_accvgpr_read_b32 v9, a1                                  // 000000001010: 00000000 00000000
\tv_accvgpr_read_b32 v8, a0                                  // 000000001014: 00000000 00000000
\ts_endpgm                                                   // 00000000101C: BF810000
"""
    funcs, base = a.parse(text)
    findings = []
    a.audit_function("k", funcs["k"], base, findings)
    # fall-through path: 1 + 8 states before a1 is read: fine; taken branch: a0 read after 1 state: flagged
    assert len(findings) == 1 and "a0" in findings[0][2]


@pytest.mark.skipif(not os.path.exists(LIB), reason="libpfhip.so not built")
def test_no_accumulator_is_read_inside_the_mfma_shadow():
    a = _load()
    if not os.path.exists(os.path.join(a.LLVM_BIN, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    findings, n_co, n_mfma = a.audit(LIB)
    assert n_co >= 10 and n_mfma > 10000, (n_co, n_mfma)        # the audit really saw the kernels
    assert not findings, "\n".join(f"{f[0][:80]}: {f[1]} -> {f[2]} ({f[3]}/{f[4]})" for f in findings[:10])


@pytest.mark.skipif(not os.path.exists(LIB), reason="libpfhip.so not built")
def test_mid_batch_kernel_requests_its_lds_operands_ahead_of_their_mfmas():
    """Round 4: the mid-batch flow kernel went 216 -> 192 us when the activation fragments (B operands, LDS) were requested
    two k-steps ahead of the MFMAs that read them, in an order pinned by sched_barrier -- left to the compiler every ds_read
    sat right in front of its MFMA.  Guard that schedule in the ISA: of the MFMAs fed from LDS, at most 6 % may have their
    operand requested fewer than two MFMAs earlier (measured 4.3 %: chain heads behind a barrier)."""
    spec = importlib.util.spec_from_file_location("audit_lds_distance", os.path.join(ROOT, "scripts", "audit_lds_distance.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    rows = {k: c for k, c in mod.scan(LIB).items() if "flow_mid_kernel" in k}
    assert len(rows) == 2, list(rows)                               # D = 15 and D = 11
    for k, c in rows.items():
        fed = sum(v for kk, v in c.items() if kk != "mfma")
        assert c["mfma"] > 1500 and fed > 0.9 * c["mfma"], (k, dict(c))
        assert c[0] + c[1] <= 0.06 * fed, (k, dict(c))
