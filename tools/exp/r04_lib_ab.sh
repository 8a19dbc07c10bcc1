#!/bin/bash
# same-device A/B of two builds of libedtr_hip.so (EDTR_AMD_LIB): usage r04_lib_ab.sh <variant .so> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; shift
for round in 1 2; do
for lib in edtr_amd/libedtr_hip.so $V; do
  EDTR_AMD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 40 --warmup 2 --no-cpu-baseline --also none --parity-steps 0 --breakdown "$@" 2> /tmp/ab.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('parity_vs_reference_golden') or {}
print('$lib: %.2f images/s  median %.3f ms  latent %s' % (d['value'], d['ms_per_step_median_hip_events'], g.get('rel_err_latent')))"
  grep -E "vae.conv1  |vae.conv2  |res.conv1  |res.conv2  |M2097152 N128 K1152" /tmp/ab.log | head -5
done; done
