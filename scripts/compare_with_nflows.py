#!/usr/bin/env python3
"""One-shot cross-check of the oracle's nflows restatement against a REAL nflows (SURVEY 7.2 H1 / 8c KAT 8).

The flow transform's arithmetic lives in the third-party package nflows, which is absent from the build image and
from the GPU boxes, so oracle/nflows_restated.py is "parity unpinned" (DESIGN.md section 2).  Wherever nflows IS
importable (any machine with `pip install nflows`; no GPU needed), this script pins it:

  1. builds the transform exactly as the reference does (src/ahsd/models/flows.py:459-529: per layer a
     ReversePermutation and a MaskedPiecewiseRationalQuadraticAutoregressiveTransform with tails='linear',
     num_blocks=2, use_residual_blocks=True, random_mask=False, relu, no batch norm, the given dropout in eval mode),
  2. copies ITS state_dict into the restatement (same key names: the restatement keeps nflows' module tree),
  3. compares forward (z, log|det|), inverse (x, log|det|) and the masks / degrees buffers, float64 and float32.

Exit status: 0 = agreement within 1e-6 (float64: 1e-12), 1 = disagreement (the numbers are printed), 77 = nflows
not importable here (skipped).  Nothing in the product or in the tests imports this script.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_real(features, hidden, context, layers, bins, tail_bound, dropout=0.0):
    from nflows.transforms.autoregressive import MaskedPiecewiseRationalQuadraticAutoregressiveTransform
    from nflows.transforms.base import CompositeTransform
    from nflows.transforms.permutations import ReversePermutation
    ts = []
    for _ in range(layers):
        ts.append(ReversePermutation(features=features))
        ts.append(MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            features=features, hidden_features=hidden, context_features=context if context > 0 else None,
            num_bins=bins, tails="linear", tail_bound=tail_bound, num_blocks=2, use_residual_blocks=True,
            random_mask=False, activation=torch.nn.functional.relu, dropout_probability=dropout, use_batch_norm=False))
    return CompositeTransform(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="4,0,64,2,8,3.0;11,288,256,10,16,5.0;15,288,256,8,16,5.0;7,40,128,3,10,2.5")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--scale", type=float, default=5.0, help="final-layer scale (default init is near-identity)")
    args = ap.parse_args()
    try:
        import nflows  # noqa: F401
    except ImportError as e:
        print(f"nflows is not importable here ({e}): skipped.  `pip install nflows` (0.14) and re-run.")
        return 77
    from oracle import nflows_restated as nfr
    from oracle.flow_ref import NSFPosteriorFlowRef, scale_final_layers
    worst = 0.0
    for spec in args.configs.split(";"):
        D, C, H, L, K, tb = spec.split(",")
        D, C, H, L, K, tb = int(D), int(C), int(H), int(L), int(K), float(tb)
        torch.manual_seed(0)
        real = build_real(D, H, C, L, K, tb).eval()
        ref = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0).eval()
        sd = {"transform." + k: v for k, v in real.state_dict().items()}
        missing = ref.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if k.startswith("transform.")], missing.missing_keys
        assert not missing.unexpected_keys, missing.unexpected_keys          # same module tree, buffer for buffer
        scale_final_layers(ref, args.scale)
        with torch.no_grad():
            for t in real._transforms:
                if hasattr(t, "autoregressive_net"):
                    t.autoregressive_net.final_layer.weight.mul_(args.scale)
                    t.autoregressive_net.final_layer.bias.mul_(args.scale)
        # the masks / degrees the restatement builds itself equal the ones nflows built
        for (k1, b1), (k2, b2) in zip(sorted(dict(real.named_buffers()).items()),
                                      sorted(dict(nfr_buffers(ref)).items())):
            assert k1 == k2 and torch.equal(b1.float(), b2.float()), (k1, k2)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(args.batch, D, generator=g) * 2 - 1
        m = torch.rand(args.batch, D, generator=g) < 0.05
        x = torch.where(m, (torch.rand(args.batch, D, generator=g) * 2 - 1) * 1.2 * tb, x)
        x[0, 0], x[1, D - 1] = tb, -tb
        ctx = torch.randn(args.batch, C, generator=g) if C else None
        for dt, tol in ((torch.float64, 1e-12), (torch.float32, 1e-6)):
            real_d, ref_d = real.to(dt), ref.to(dt)
            xd, cd = x.to(dt), None if ctx is None else ctx.to(dt)
            with torch.no_grad():
                z1, l1 = real_d(xd, cd)
                z2, l2 = ref_d.transform(xd, cd)
                x1, li1 = real_d.inverse(z1, cd)
                x2, li2 = ref_d.transform.inverse(z1, cd)
            errs = [(z1 - z2).abs().max().item(), (l1 - l2).abs().max().item(),
                    (x1 - x2).abs().max().item(), (li1 - li2).abs().max().item()]
            rel = max(e / max(1.0, s) for e, s in zip(errs, (z1.abs().max().item(), l1.abs().max().item(),
                                                             x1.abs().max().item(), li1.abs().max().item())))
            ok = rel < tol
            worst = max(worst, rel / tol)
            print(f"D{D} C{C} H{H} L{L} K{K} tb{tb} {str(dt)[6:]}: |dz| {errs[0]:.2e} |dlogdet| {errs[1]:.2e} "
                  f"|dx_inv| {errs[2]:.2e} |dlogdet_inv| {errs[3]:.2e}  {'OK' if ok else 'MISMATCH'}")
    print("restatement == nflows" if worst < 1.0 else "restatement DIFFERS from nflows")
    return 0 if worst < 1.0 else 1


def nfr_buffers(ref):
    return {k[len("transform."):]: v for k, v in ref.named_buffers() if k.startswith("transform.")}


if __name__ == "__main__":
    sys.exit(main())
