# Round 6: edtr_ffn (one launch per feed-forward half at the 64x64 level) against the two-GEMM form, same device, bench legs.
mkdir -p gpurun_out/r06
for v in 1 0 1 0; do
  echo "== EDTR_FFN=$v det512"; EDTR_FFN=$v python bench.py --steps 40 --also none --no-cpu-baseline --no-roofline --parity-steps 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['parity_vs_reference_golden']['rel_err_latent'], j['parity_vs_reference_golden']['rel_err_image_samples'])"
done
for v in 1 0; do
  echo "== EDTR_FFN=$v det512s50"; EDTR_FFN=$v python bench.py --workload det512s50 --steps 6 --also none --no-cpu-baseline --no-roofline --parity-steps 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
echo "== breakdown with the fused launch"
python bench.py --steps 8 --also none --no-cpu-baseline --parity-steps 0 --breakdown 2> gpurun_out/r06/bench_ffn_breakdown.txt > gpurun_out/r06/bench_ffn.json
grep -n "ff\.\|ffn\|layernorm" gpurun_out/r06/bench_ffn_breakdown.txt | head
