#!/bin/bash
# times scripts/bench_stem.py (bf16 line) under the main library and every posteriflow_amd/lib/libpf_side_stem_*.so
cd $GRAFT_REPO_ROOT
echo "main: $(python3 scripts/bench_stem.py 2>/dev/null | grep bf16)"
for f in posteriflow_amd/lib/libpf_side_stem_*.so; do
  echo "$(basename $f): $(PF_LIBPFHIP=$PWD/$f python3 scripts/bench_stem.py 2>/dev/null | grep bf16)"
done
