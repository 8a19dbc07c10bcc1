run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --parity-steps 0 --also none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['parity_vs_reference_golden']['rel_err_image_samples'], d['parity_vs_reference_golden']['ok'])"; }
run base X=1
run halo512_ge512 EDTR_IGEMM_HALO512=2
run gnin_maxn256 EDTR_GN_IN_CONV_MAXN=256
run base X=1
run gnin_maxn512 EDTR_GN_IN_CONV_MAXN=512
run halo512_ge512 EDTR_IGEMM_HALO512=2
run gnin_maxn256 EDTR_GN_IN_CONV_MAXN=256
