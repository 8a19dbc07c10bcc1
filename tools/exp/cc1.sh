#!/bin/bash
# compile ONE kernel source of edtr_amd/csrc for gfx950 and print its register / spill table:  tools/exp/cc1.sh halo512 [-DFLAG ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Werror=pass-failed "$@" -c $ROOT/edtr_amd/csrc/$N.hip -o $ROOT/edtr_amd/csrc/build/$N.o -save-temps=obj
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" $ROOT/edtr_amd/csrc/build/$N-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | awk '{print $2, $4, $6, $8}' | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-150
cp $ROOT/edtr_amd/csrc/build/$N-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/$N.s
rm -f $ROOT/edtr_amd/csrc/build/$N-hip-* $ROOT/edtr_amd/csrc/build/$N-host-* $ROOT/edtr_amd/csrc/build/$N.hip-*
