#!/bin/bash
# Build experimental variants of walk.hip (N = 25 only) with the given -D flags and print the resource usage;
# usage: tools/rw_exp.sh name "-DFLAG1 -DFLAG2"   -> dynesty_amd/libdynhip_exp_<name>.so
set -e
cd "$(dirname "$0")/../dynesty_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function '-DDH_DIM_LIST(X)=X(25)' $@ \
  -Rpass-analysis=kernel-resource-usage -c walk.hip -o /tmp/walk_exp_$name.o 2>&1 | grep -E "error|rwalk_kernelILi25ELb1ELi1ELi" -A12 \
  | grep -E "error|Function Name|VGPRs:|Spill" | sed 's/remark: walk.hip:[0-9]*:[0-9]*: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/_ZN12_GLOBAL__N_112rwalk_kernelILi25ELb1ELi1ELi//' | tr '\n' ' '
echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o walk2.o bound.o rebuild.o wide.o ns.o friends.o /tmp/walk_exp_$name.o -o ../libdynhip_exp_$name.so
