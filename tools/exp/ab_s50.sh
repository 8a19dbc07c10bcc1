mkdir -p gpurun_out/r03
run() { label="$1"; shift; env "$@" python bench.py --workload det512s50 --no-roofline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s50 %-28s' % '$label', d['value'], d['ms_per_step'], (d.get('parity_vs_reference_golden') or d.get('parity') or {}).get('rel_err_latent'))"; }
run default A=1
run ln_fold EDTR_LN_FOLD=1
run gn_fold16 EDTR_GN_FOLD_MAX=16
run gn_fold64 EDTR_GN_FOLD_MAX=64
# (run pp128 EDTR_IGEMM_PP128=1: tile 15 was removed in round 4; its result is in profiles/r03/ab_s50.log)
run ln_fold+gn64 EDTR_LN_FOLD=1 EDTR_GN_FOLD_MAX=64
run default A=1
