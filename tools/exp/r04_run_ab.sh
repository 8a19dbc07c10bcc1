python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -15
bash tools/exp/r04_mixed_ab.sh mirror "EDTR_AMD_MIRROR=0" "EDTR_AMD_MIRROR=1" 'EDTR_AMD_POLICY={"base":"shipped","res.skip1x1":4}' 'EDTR_AMD_POLICY={"base":"shipped","res.skip1x1":4,"st.proj_in":4,"st.proj_out":4}'
