#!/bin/bash
# same-device A/B of two builds of libedtr_hip.so (EDTR_AMD_LIB): usage r05_lib_ab.sh <variant .so> "<grep pattern of breakdown lines>" [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; PAT=$2; shift; shift
for round in 1 2; do
for lib in edtr_amd/libedtr_hip.so $V; do
  EDTR_AMD_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 40 --warmup 2 --no-cpu-baseline --also none --parity-steps 0 --breakdown "$@" 2> /tmp/ab.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('parity_vs_reference_golden') or {}
print('$lib: %.2f images/s  median %.3f ms  latent %s image %s' % (d['value'], d['ms_per_step_median_hip_events'], g.get('rel_err_latent'), g.get('rel_err_image')))"
  grep -E "$PAT" /tmp/ab.log | head -12
done; done
