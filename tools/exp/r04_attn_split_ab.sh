#!/bin/bash
# round 4: what the split attention operands cost and buy (one device): heavy-set errors and det512 throughput per mode / split level
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
for cfg in "high 0" "high 1" "high 2" "mixed 0" "mixed 1" "mixed 2"; do
  set -- $cfg
  echo "=== precision $1, EDTR_AMD_ATTN_SPLIT=$2"
  EDTR_AMD_ATTN_SPLIT=$2 python -m pytest tests/test_gpu_heavy.py -q -m gpu -k "$1" -s 2>&1 | grep -E "^\[heavy|passed|failed"
  EDTR_AMD_ATTN_SPLIT=$2 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --also none --parity-steps 0 --no-roofline --precision $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['parity_vs_reference_golden']
print('det512 smooth weights: %.2f images/s, latent %.3e image %.3e (max %.3e / %.3e)' % (d['value'], g['rel_err_latent'], g['rel_err_image_samples'], g['max_err_latent'], g['max_err_image_samples']))"
done
