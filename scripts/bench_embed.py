"""Embedding throughput: HIP stem alone and the full LeanStrainEncoder, events/s."""
import sys, time, torch
sys.path.insert(0, ".")
from posteriflow_amd import npe
torch.manual_seed(0)
for prec in ("bf16", "fp32"):
    enc = npe.LeanStrainEncoder().cuda().eval(); enc.precision = prec
    for B in (256, 4096):
        strain = torch.randn(B, 3, 16384, device="cuda")
        with torch.no_grad():
            for _ in range(2): enc._stem_hip(strain)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): enc._stem_hip(strain)
            torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 5
            te = float("nan")
            if B <= 4096:
                chunk = B if prec == "bf16" else 512     # fp32 mode: tensor-op Transformer, [chunk, 183, 768] hidden
                for _ in range(1): [enc(strain[i:i + chunk]) for i in range(0, B, chunk)]
                torch.cuda.synchronize(); t0 = time.perf_counter()
                [enc(strain[i:i + chunk]) for i in range(0, B, chunk)]
                torch.cuda.synchronize(); te = time.perf_counter() - t0
        gb = B * 3 * 16384 * 4 / 1e9
        print(f"{prec} B={B}: stem {ts*1e3:.2f} ms ({B/ts:.0f} events/s, strain read {gb/ts:.0f} GB/s, "
              f"{B*3*34.98e6*2/ts/1e12:.1f} TFLOP/s)  full encoder {te*1e3:.1f} ms ({B/te:.0f} events/s)", flush=True)
