#!/bin/bash
# same-device A/B of the GEGLU epilogue's gate evaluation: eight values in lockstep (shipped) vs value after value (EDTR_IGEMM_GEGLU_SERIAL=1)
mkdir -p gpurun_out/r04
for i in 1 2 3; do
  for s in 1 0; do
    echo -n "EDTR_IGEMM_GEGLU_SERIAL=$s: "
    EDTR_IGEMM_GEGLU_SERIAL=$s python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --parity-steps 0 --also none 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'images/s', d['ms_per_step'], 'ms')"
  done
done
for s in 1 0; do
  echo "EDTR_IGEMM_GEGLU_SERIAL=$s"
  EDTR_IGEMM_GEGLU_SERIAL=$s python bench.py --steps 8 --warmup 2 --no-cpu-baseline --parity-steps 0 --also none --breakdown 2>&1 | grep -E "ff.geglu|act1"
done
