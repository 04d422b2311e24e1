"""Fused token mixer alone (pf_embed_fusion_forward): ms per batch of events, events/s, TFLOP/s.
Algorithmic work per event (T = 183): 3 layers x (QKV 183*192*576 + scores/values 2*6*183*183*32 + out
183*192*192 + FFN 2*183*192*768) + pool K,V 183*192*384 MAC = 0.296 GMAC = 0.59 GFLOP."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posteriflow_amd import npe
torch.manual_seed(0)
enc = npe.LeanStrainEncoder().cuda().eval(); enc.precision = "bf16"
T = 183
mac = 3 * (T * 192 * 576 + 2 * 6 * T * T * 32 + T * 192 * 192 + 2 * T * 192 * 768) + T * 192 * 384
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [256, 4096]):
    tok = torch.randn(B, T, 192, device="cuda")
    with torch.no_grad():
        for _ in range(2): enc._mix_hip(tok)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): enc._mix_hip(tok)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"B={B}: {dt*1e3:.2f} ms ({B/dt:.0f} events/s, {2*mac*B/dt/1e12:.1f} TFLOP/s = {2*mac*B/dt/2.5e15*100:.1f}% of bf16 MFMA peak)", flush=True)
