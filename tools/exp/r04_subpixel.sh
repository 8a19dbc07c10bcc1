#!/bin/bash
# round 4: sub-pixel upsample convolution — unit tests, then a same-device A/B of the whole path (fast and mixed)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "halo or conv3x3 or gemm_bias or skinny" 2>&1 | tail -15 > gpurun_out/r04/subpixel_tests.log
cat gpurun_out/r04/subpixel_tests.log
for sp in 0 1; do
  for prec in fast mixed; do
    EDTR_SUBPIXEL=$sp python bench.py --steps 24 --warmup 2 --no-cpu-baseline --also none --parity-steps 0 --precision $prec --breakdown \
      > gpurun_out/r04/ab_subpixel_${sp}_${prec}.json 2> gpurun_out/r04/ab_subpixel_${sp}_${prec}.log
    python - <<PY
import json
d=json.load(open("gpurun_out/r04/ab_subpixel_${sp}_${prec}.json"))
print("EDTR_SUBPIXEL=$sp $prec", d["value"], "img/s", d.get("parity_vs_reference_golden"))
PY
    grep -E "up2|upsample" gpurun_out/r04/ab_subpixel_${sp}_${prec}.log | head -12
  done
done
