"""How far ahead of its MFMA is an LDS-fed operand requested?  For every kernel of libpfhip.so: each v_mfma whose A or B operand was
written by a ds_read, and the number of MFMAs issued between that ds_read and the MFMA (0 = the read sits right in front of the
MFMA that needs it: an LDS round trip -- 64+ cycles, more under load -- is exposed unless another wave fills the SIMD).
Found the mid-batch kernel's 320-cycle k-steps (LABLOG R4.9).  usage: python scripts/audit_lds_distance.py [kernel-substring]"""
import collections, os, re, subprocess, sys
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "posteriflow_amd", "lib", "libpfhip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"([va])(\d+)$", tok)
    return [(m.group(1), int(m.group(2)))] if m else []

def scan(lib=LIB):
    """-> {mangled kernel name: Counter{0..8: MFMAs whose LDS-fed operand was requested that many MFMAs earlier (8 = 8 or more),
    "mfma": all MFMAs}}"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    from audit_accvgpr import code_objects
    with tempfile.TemporaryDirectory() as tmp:
        out = "\n".join(subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout for co in code_objects(lib, tmp))
    kern, rows = None, {}
    writer, nm = {}, 0
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kern = m.group(1); writer = {}; nm = 0; rows[kern] = collections.Counter(); continue
        if kern is None: continue
        parts = line.strip().split("//")[0].split(None, 1)
        if len(parts) < 2: continue
        op, args = parts[0], [a.strip() for a in parts[1].split(",")]
        if op.startswith("ds_read") or op.startswith("ds_load"):
            for r in regs(args[0]): writer[r] = nm
        elif op.startswith("v_mfma"):
            d = None
            for a in args[1:3]:
                for r in regs(a):
                    if r in writer: d = nm - writer[r] if d is None else min(d, nm - writer[r])
            if d is not None: rows[kern][min(d, 8)] += 1
            rows[kern]["mfma"] += 1
            nm += 1
            for r in regs(args[0]): writer.pop(r, None)
        elif op.startswith("v_") or op.startswith("buffer_load") or op.startswith("global_load"):
            for r in regs(args[0]): writer.pop(r, None)
    return rows


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = scan()
    for k, c in sorted(rows.items(), key=lambda kv: -kv[1]["mfma"]):
        if c["mfma"] == 0: continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
        if want not in name and want not in k: continue
        fed = sum(v for kk, v in c.items() if kk != "mfma")
        print(f"{name:90s} mfma {c['mfma']:5d} lds-fed {fed:5d}  distance 0:{c[0]:4d} 1:{c[1]:4d} 2:{c[2]:4d} 3:{c[3]:4d} 4-7:{sum(c[i] for i in range(4,8)):4d} 8+:{c[8]:4d}")


if __name__ == "__main__":
    main()
