mkdir -p gpurun_out/r03
run() { label="$1"; wl="$2"; shift; shift; env "$@" python bench.py $wl --no-cpu-baseline --no-roofline --parity-steps 0 --also none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$label', d['value'], d['ms_per_step'], (d.get('parity_vs_reference_golden') or {}).get('rel_err_latent'))"; }
D="--steps 12 --warmup 2"; S="--workload det512s50 --steps 3 --warmup 1"
for t in 0 2048 8192 0 8192 2048; do run "det512 LN_FOLD_MAX_ROWS=$t" "$D" EDTR_LN_FOLD_MAX_ROWS=$t; done
for t in 0 4096 16384 0 16384 4096; do run "s50 LN_FOLD_MAX_ROWS=$t" "$S" EDTR_LN_FOLD_MAX_ROWS=$t; done
