set -x
mkdir -p gpurun_out/r02
export EDTR_ATTN_PRESCALED=1
timeout 600 python3 tools/exp/hw_check_attn.py > gpurun_out/r02/attn_v3asm_check.log 2>&1
EDTR_ATTN_V3=0 timeout 600 python3 tools/exp/hw_check_attn.py > gpurun_out/r02/attn_v2cpp_check.log 2>&1
unset EDTR_ATTN_PRESCALED
timeout 300 python3 tools/exp/hw_check_attn.py > gpurun_out/r02/attn_v1_check.log 2>&1
for b in 11 12 13; do timeout 300 python3 tools/exp/hw_ab_tiles.py --a 3 --b $b > gpurun_out/r02/ab_tiles_3_vs_$b.log 2>&1; done
timeout 300 python3 tools/exp/hw_ab_tiles.py --a 3 --b 14 --tol 4e-3 > gpurun_out/r02/ab_tiles_3_vs_14.log 2>&1
tail -12 gpurun_out/r02/attn_v3asm_check.log gpurun_out/r02/attn_v2cpp_check.log
grep -c PASS gpurun_out/r02/ab_tiles_*.log; grep -h "FAIL\|ALL" gpurun_out/r02/ab_tiles_*.log | head -20
timeout 1500 python -m pytest tests/test_gpu_precision.py -m gpu -q -s -x 2>&1 | tail -40 > gpurun_out/r02/precision_tests_v0.log
tail -30 gpurun_out/r02/precision_tests_v0.log
timeout 600 python tools/exp/cpu_oracle_threads.py > gpurun_out/r02/cpu_oracle_threads.log 2>&1
cat gpurun_out/r02/cpu_oracle_threads.log
