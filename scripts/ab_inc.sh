#!/bin/bash
# A/B of side builds of the incremental inverse (scripts/side_obj.sh NAME pf_flow_inc -D...): scripts/ab_inc.sh NAME...
out=gpurun_out/ab_inc; mkdir -p $out
python scripts/bench_inverse.py > $out/base.log 2>&1 || exit 1
for v in "$@"; do
  PF_LIBPFHIP=$PWD/posteriflow_amd/lib/libpf_side_$v.so python scripts/bench_inverse.py > $out/$v.log 2>&1 || exit 1
done
python scripts/bench_inverse.py > $out/base2.log 2>&1 || exit 1
for v in "$@"; do
  PF_LIBPFHIP=$PWD/posteriflow_amd/lib/libpf_side_$v.so python -m pytest tests/test_flow_inverse_gpu.py -x -q -k "incremental or round_trip" > $out/test_$v.log 2>&1
done
python -m pytest tests/test_flow_inverse_gpu.py -x -q > $out/test_base.log 2>&1
grep -H "draws=131072" $out/*.log; tail -n 1 $out/test_*.log
