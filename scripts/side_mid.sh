#!/bin/bash
# side build of libpfhip.so with the mid-batch kernel compiled under extra flags: scripts/side_mid.sh NAME -DPF_MID_BD=3 ...
# -> posteriflow_amd/lib/libpf_mid_NAME.so (use with PF_LIBPFHIP=...); prints the kernel's register use
set -e
cd "$(dirname "$0")/../posteriflow_amd/csrc"
name=$1; shift
mkdir -p /tmp/side_$name
for d in 15 11; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -DPF_MID_D=$d -c pf_flow_mid_inst.hip -o /tmp/side_$name/flow_mid_d$d.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|VGPRs Spill" | tr '\n' ' '
  echo " [$name d$d]"
done
objs=$(ls ../lib/obj/*.o | grep -v flow_mid_d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/side_$name/flow_mid_d15.o /tmp/side_$name/flow_mid_d11.o -o ../lib/libpf_mid_$name.so
