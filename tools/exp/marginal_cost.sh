#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "# EDTR_EXP_DUP=<name>: every idempotent launch whose name contains <name> is issued twice; bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline (final code of round 2: halo tile incl. upsample variant, tile order rules)"
echo "# name            images/s   ms_per_step   marginal_ms"
base=""
for n in NONE vae.conv1 vae.conv2 vae.upsample res.conv1 res.conv2 upsample.conv ff.geglu ff.out flash gn.apply attn.out layernorm gn.stats st.proj attn1.qk attn1.vT NONE; do
  if [ "$n" = "NONE" ]; then out=$(python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1); else out=$(EDTR_EXP_DUP=$n python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1); fi
  python3 - "$n" "$out" "$base" <<'PY'
import json,sys
n,o,b=sys.argv[1],sys.argv[2],sys.argv[3]
j=json.loads(o)
print(f"{n:16s} {j['value']:9.4f} {j['ms_per_step']:12.3f} {(j['ms_per_step']-float(b)) if b else 0.0:12.2f}")
PY
  if [ "$n" = "NONE" ] && [ -z "$base" ]; then base=$(echo "$out" | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); fi
done
