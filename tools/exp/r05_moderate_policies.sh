run() { echo "== $1"; shift; env "$@" python -m pytest tests/test_gpu_heavy.py -m gpu -q -s -k "moderate and mixed" 2>&1 | grep "moderate, mixed" | sed 's/.*block\] //'; }
run base X=1
run attn_split1 EDTR_AMD_ATTN_SPLIT=1
run attn_split2 EDTR_AMD_ATTN_SPLIT=2
run all2 'EDTR_AMD_POLICY={"default": 2}'
run all3 'EDTR_AMD_POLICY={"default": 3}'
run all3_split2 'EDTR_AMD_POLICY={"default": 3}' EDTR_AMD_ATTN_SPLIT=2
run shipped_qk3 'EDTR_AMD_POLICY={"base":"shipped","attn1.qkv":3,"attn2.q":3}'
run shipped_conv3 'EDTR_AMD_POLICY={"base":"shipped","res.conv1":3,"res.conv2":3}'
run shipped_tf3 'EDTR_AMD_POLICY={"base":"shipped","ff.geglu":3,"ff.out":3,"attn.out":3,"st.proj_in":3,"st.proj_out":3}'
run shipped_all4 'EDTR_AMD_POLICY={"base":"shipped","default":4}'
