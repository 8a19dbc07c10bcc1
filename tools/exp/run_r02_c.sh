mkdir -p gpurun_out/r02
python3 tools/exp/attn_variants.py stamps > gpurun_out/r02/attn_v3_stamps_exp1.log 2>&1
cat gpurun_out/r02/attn_v3_stamps_exp1.log
