#!/usr/bin/env python3
"""Static audit of libpfhip.so's gfx950 code objects for the "accumulator read too soon behind an MFMA" hazard
(LABLOG.md pitfall 15: `v_accvgpr_read` of elements 0-2 directly behind a loop-exit MFMA came back stale).

For every v_mfma_* instruction the script walks ALL control-flow successors (basic blocks are rebuilt from the
branch targets llvm-objdump prints) and counts wait states the way the compiler's hazard recognizer does -- one per
instruction, N + 1 for `s_nop N` -- until the number the matrix pipe needs has passed.  Any instruction inside that
window that reads or overwrites a register of the MFMA's destination is reported, except
  * a following MFMA that takes the destination whole as its C operand (the accumulate chain needs 0 states);
  * a following MFMA that writes the same destination (back-to-back issue is interlocked by the pipe itself when
    srcC == vdst, and a plain overwrite is ordered in the pipe);
  * a memory LOAD (ds_read* / global_load* / buffer_load* / flat_load* / scratch_load*) whose destination is the MFMA's:
    a dead MFMA result whose registers the allocator reuses for a fragment load.  LLVM's recogniser has no
    "XDL write VGPR -> VMEM / LDS write-back" rule (its WAW rule covers VALU writes; for memory instructions only VGPRs they
    READ -- address, store data -- are checked), and the load's data cannot land before the MFMA retires: the window is at
    most 10 wait states, an LDS return takes >= 64 cycles, an L2 hit >= 180.  First seen in round 4's one-tile-per-wave strip
    GEMM (`dense_strip_kernel<false, 0, 1, 4>`), where a clamped tail tile's product is computed and never stored.
Required wait states (LLVM GCNHazardRecognizer, gfx950 column of "XDL/SMFMA write VGPR -> VALU / VMEM / LDS read,
VALU write"): passes + 3 (+ 1 on gfx950 above 2 passes) for the XDL (bf16/f16/i8/fp8) forms, passes + 2 for the
f32-input forms; passes = issue cycles / 4 (MI355X_MICROARCH.md cycle constants: 16x16x32 bf16 16 cycles, 32x32x16 bf16
32, 16x16x4 f32 32, 32x32x2 f32 64, 16x16x16 bf16 16).  Cross-check against the compiler's own straight-line padding:
`v_mfma_f32_16x16x32_bf16 a[0:3] ..; s_cbranch; s_nop 6; ds_write_b128 .., a[0:3]` = 8 states.

Exit code 0 = clean, 1 = findings (printed), 2 = tooling problem.  Run from anywhere; used by tests/test_isa_audit.py.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = os.environ.get("PF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = os.path.join(ROOT, "posteriflow_amd", "lib", "libpfhip.so")

_REG = re.compile(r"\b([vas])(?:\[(\d+):(\d+)\]|(\d+))")
_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
_TARGET = re.compile(r"<(.+)\+0x([0-9a-fA-F]+)>\s*$")
_TARGET0 = re.compile(r"<([^+>]+)>\s*$")


def required_wait_states(mnemonic: str) -> int:
    m = re.match(r"v_mfma_\w+?_(\d+)x(\d+)x(\d+)", mnemonic)
    if not m:
        return 20
    a, _, k = int(m.group(1)), int(m.group(2)), int(m.group(3))
    f32_in = mnemonic.endswith("_f32") and "_f32_" in mnemonic and mnemonic.count("f32") >= 2
    if f32_in:                                   # v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (+ multi-block forms)
        passes = 16 if a == 32 else (8 if a == 16 else 2)
        return passes + 2                        # the compiler's own straight-line padding: 10 for 16x16x4_f32
    if "f8f6f4" in mnemonic:
        passes = 16 if a == 32 else 8
    elif a == 32:
        passes = 8 if k >= 16 else 16            # 32x32x16 (gfx950): 32 cycles; older 32x32x8: 64
    elif a == 16:
        passes = 4 if k >= 16 else 8             # 16x16x32 / 16x16x16: 16 cycles
    else:
        passes = 2
    return passes + 3 + (1 if passes != 2 else 0)


def regs(text: str):
    """[(file, lo, hi)] for every register or register range named in an operand string."""
    out = []
    for m in _REG.finditer(text):
        f = m.group(1)
        if m.group(2) is not None:
            out.append((f, int(m.group(2)), int(m.group(3))))
        else:
            out.append((f, int(m.group(4)), int(m.group(4))))
    return out


def overlaps(a, b):
    return a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def split_operands(ops: str):
    parts, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


class Insn:
    __slots__ = ("addr", "mn", "ops", "target", "line")

    def __init__(self, addr, mn, ops, target, line):
        self.addr, self.mn, self.ops, self.target, self.line = addr, mn, ops, target, line


def parse(disasm: str):
    """{function name: [Insn]} from `llvm-objdump -d` text."""
    funcs, cur, base = {}, None, {}
    for line in disasm.splitlines():
        fm = _FUNC.match(line)
        if fm:
            cur = fm.group(2)
            funcs[cur] = []
            base[cur] = int(fm.group(1), 16)
            continue
        im = _INSN.match(line)
        if not im or cur is None:
            continue
        mn, ops, addr = im.group(1), im.group(2), int(im.group(3), 16)
        target = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tm = _TARGET.search(line)
            if tm:
                target = (tm.group(1), int(tm.group(2), 16))
            else:
                t0 = _TARGET0.search(line)
                if t0:
                    target = (t0.group(1), 0)
        funcs[cur].append(Insn(addr, mn, ops, target, line.strip()))
    return funcs, base


def wait_states(i: Insn) -> int:
    if i.mn == "s_nop":
        try:
            return int(i.ops.split()[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def audit_function(name, insns, base, findings):
    by_addr = {i.addr: k for k, i in enumerate(insns)}
    for k, ins in enumerate(insns):
        if not ins.mn.startswith("v_mfma") and not ins.mn.startswith("v_smfmac"):
            continue
        ops = split_operands(ins.ops)
        if not ops:
            continue
        dst = regs(ops[0])
        if not dst:
            continue
        dst = dst[0]
        need = required_wait_states(ins.mn)
        # breadth-first over (instruction index, wait states elapsed)
        seen = {}
        work = [(k + 1, 0)]
        while work:
            j, ws = work.pop()
            while j < len(insns) and ws < need:
                if seen.get(j, 1 << 30) <= ws:
                    break
                seen[j] = ws
                nxt = insns[j]
                nops = split_operands(nxt.ops)
                is_mfma = nxt.mn.startswith("v_mfma") or nxt.mn.startswith("v_smfmac")
                hazard = None
                if is_mfma:
                    # srcC == whole vdst: accumulate chain (0 states); A/B operands from the destination: hazard
                    for idx, o in enumerate(nops[1:3], start=1):
                        if any(overlaps(r, dst) for r in regs(o)):
                            hazard = f"MFMA operand {idx} reads the destination"
                    if len(nops) > 3:
                        c = regs(nops[3])
                        if c and overlaps(c[0], dst) and c[0] != dst:
                            hazard = "MFMA srcC overlaps the destination partially"
                    ndst = regs(nops[0]) if nops else []
                    if hazard is None and ndst and overlaps(ndst[0], dst):
                        break       # same accumulator taken over by the next MFMA: the pipe orders it, window ends
                elif nxt.mn not in ("s_nop", "s_waitcnt", "s_barrier", "s_endpgm") and not nxt.mn.startswith("s_"):
                    is_load = nxt.mn.startswith(("ds_read", "global_load", "buffer_load", "flat_load", "scratch_load"))
                    if is_load and nops and any(overlaps(r, dst) for r in regs(nops[0])) and \
                            not any(overlaps(r, dst) for o in nops[1:] for r in regs(o)):
                        ld = regs(nops[0])[0]
                        if ld[0] == dst[0] and ld[1] <= dst[1] and ld[2] >= dst[2]:
                            break   # the WHOLE destination is handed to a load's return data: the window ends (docstring)
                        ws += wait_states(nxt)      # part of it: not a hazard by itself, the rest stays watched
                        j += 1
                        continue
                    for idx, o in enumerate(nops):
                        if any(overlaps(r, dst) for r in regs(o)):
                            hazard = ("reads" if idx > 0 or nxt.mn.startswith(("global_store", "buffer_store",
                                                                                  "ds_write", "flat_store", "scratch_store"))
                                      else "overwrites") + " the destination"
                            break
                if hazard:
                    findings.append((name, ins.line, nxt.line, ws, need, hazard))
                    break
                if nxt.mn == "s_endpgm" or nxt.mn.startswith("s_setpc"):
                    break
                ws += wait_states(nxt)
                if nxt.target is not None:
                    tname, toff = nxt.target
                    taddr = base.get(tname, None)
                    if taddr is not None and (taddr + toff) in by_addr and ws < need:
                        work.append((by_addr[taddr + toff], ws))
                    if nxt.mn == "s_branch":
                        break
                j += 1


def code_objects(lib: str, tmp: str):
    local = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def audit(lib: str = DEFAULT_LIB, verbose: bool = False):
    """-> (findings, n_code_objects, n_mfma)."""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(objdump)
    findings, n_mfma = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(lib, tmp)
        for co in cos:
            text = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
            funcs, base = parse(text)
            for name, insns in funcs.items():
                n = sum(1 for i in insns if i.mn.startswith("v_mfma"))
                n_mfma += n
                if n:
                    before = len(findings)
                    audit_function(name, insns, base, findings)
                    if verbose:
                        print(f"{os.path.basename(co)}: {name[:90]}: {n} MFMAs, {len(findings) - before} findings")
    return findings, len(cos), n_mfma


def main(argv):
    lib = argv[1] if len(argv) > 1 and not argv[1].startswith("-") else DEFAULT_LIB
    try:
        findings, n_co, n_mfma = audit(lib, verbose="-v" in argv)
    except (FileNotFoundError, subprocess.CalledProcessError) as e:
        print(f"audit_accvgpr: tooling problem: {e}", file=sys.stderr)
        return 2
    per_kernel = defaultdict(int)
    for f in findings:
        per_kernel[f[0]] += 1
    print(f"audit_accvgpr: {n_co} code objects, {n_mfma} MFMA instructions, {len(findings)} findings")
    for name, mfma, reader, ws, need, why in findings[:200]:
        print(f"  {name[:100]}\n    {mfma}\n    -> after {ws} of {need} wait states: {reader}   [{why}]")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
