#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "# bench.py --dup <name>: every idempotent launch whose name contains <name> is issued twice (Program.duplicate_launches); bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --parity-steps 0 --also none"
echo "# name            images/s   ms_per_step   marginal_ms"
# MC_ARGS overrides the bench arguments (e.g. "--workload det512s50 --no-roofline --steps 3 --warmup 1 --no-cpu-baseline")
base=""
for n in NONE vae.conv1 vae.conv2 vae.upsample res.conv1 res.conv2 upsample.conv ff.geglu ff.out flash gn.apply attn.out layernorm st.proj attn1.qkv attn2.q zero_conv res.skip gn.stats gn.finalize add NONE; do
  A="${MC_ARGS:---steps 12 --warmup 3 --no-cpu-baseline --no-roofline --parity-steps 0 --also none}"
  if [ "$n" = "NONE" ]; then out=$(python bench.py $A 2>/dev/null | tail -1); else out=$(python bench.py $A --dup $n 2>/dev/null | tail -1); fi
  python3 - "$n" "$out" "$base" <<'PY'
import json,sys
n,o,b=sys.argv[1],sys.argv[2],sys.argv[3]
j=json.loads(o)
print(f"{n:16s} {j['value']:9.4f} {j['ms_per_step']:12.3f} {(j['ms_per_step']-float(b)) if b else 0.0:12.2f}")
PY
  if [ "$n" = "NONE" ] && [ -z "$base" ]; then base=$(echo "$out" | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); fi
done
