#!/bin/bash
# side build of libpfhip.so with ONE plain object compiled under extra flags: scripts/side_obj.sh NAME pf_dense -DPF_DENSE_BM64_OCC=3
# -> posteriflow_amd/lib/libpf_side_NAME.so (use with PF_LIBPFHIP=...)
set -e
cd "$(dirname "$0")/../posteriflow_amd/csrc"
name=$1; obj=$2; shift 2
mkdir -p /tmp/side_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $obj.hip -o /tmp/side_$name/$obj.o
objs=$(ls ../lib/obj/*.o | grep -v "/$obj.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/side_$name/$obj.o -o ../lib/libpf_side_$name.so
