#!/bin/bash
# Ablation timings of the 16-row forward kernel at 4096 rows (library built with -DPF_ABLATE_BUILD as lib/libpfhip_abl.so):
# PF_ABLATE bits: 1 no spline, 2 no weight traffic (same address), 4 no MFMA, 8 no barriers
export PF_LIBPFHIP=$PWD/posteriflow_amd/lib/libpfhip_abl.so
for a in 0 1 2 4 8 3 5; do
  PF_ABLATE=$a python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF_ABLATE=$a kernel_us', round(d['roofline']['kernel_ms']*1e3,1))"
done
