#!/bin/bash
# compile ONE kernel source of edtr_amd/csrc for gfx950 and print its register / spill table:  tools/exp/cc1.sh halo512 [-DFLAG ...]
# Everything (object, -save-temps output) goes to a scratch directory: csrc/build/ holds the objects build.py links, and an
# experiment object compiled with diagnostic -D flags must never be able to end up in libedtr_hip.so.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Werror=pass-failed "$@" -c $ROOT/edtr_amd/csrc/$N.hip -o $T/$N.o -save-temps=obj
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" $T/$N-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | awk '{print $2, $4, $6, $8}' | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-150
cp $T/$N-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/$N.s
