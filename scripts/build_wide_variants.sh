#!/bin/bash
# Side builds of libpfhip.so with the large-batch kernel compiled under -DPF_WIDE_ABLATE=<mask> / -DPF_WIDE_P=<depth>
# (timing experiments; loaded through $PF_LIBPFHIP).  usage: build_wide_variants.sh name:"flags" ...
set -e
cd "$(dirname "$0")/../posteriflow_amd/csrc"
make -j8 >/dev/null
mkdir -p ../lib/abl
OBJS=$(ls ../lib/obj/*.o | grep -v flow_wide_d)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (
  for d in 15 11; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 $flags -DPF_WIDE_D=$d -c pf_flow_wide_inst.hip -o ../lib/abl/wide_${name}_d$d.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib/abl/wide_${name}_d15.o ../lib/abl/wide_${name}_d11.o -o ../lib/abl/libpfhip_${name}.so
  echo built $name
  ) &
done
wait
