#!/bin/bash
# round 4: mixed-mode A/B on one device.  usage: r04_mixed_ab.sh <tag> "<ENV=.. ENV=..>" ["<ENV..>" ...]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
tag=$1; shift
i=0
for envs in "$@"; do
  env $envs python bench.py --steps 16 --warmup 2 --no-cpu-baseline --also none --parity-steps 0 --precision mixed --breakdown \
      > gpurun_out/r04/ab_${tag}_$i.json 2> gpurun_out/r04/ab_${tag}_$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/ab_${tag}_$i.json"))
g=d.get("parity_vs_reference_golden") or {}
print("[$envs] mixed %.2f img/s  %.2f ms  latent %s image %s  launches %s" % (d["value"], d["ms_per_step"], g.get("rel_err_latent"), g.get("rel_err_image_samples"), d.get("launches_per_pass")))
PY
  grep -E "sum of launch|split_operand|gn.apply  |^igemm_kernel" gpurun_out/r04/ab_${tag}_$i.log | head -6
  i=$((i+1))
done
