run() { echo "== $1"; shift; env "$@" python -m pytest tests/test_gpu_heavy.py -m gpu -q -s -k "moderate and mixed" 2>&1 | grep "moderate, mixed" | sed 's/.*block\] //'; }
run conv3_qk3_split2 'EDTR_AMD_POLICY={"base":"shipped","res.conv1":3,"res.conv2":3,"attn1.qkv":3,"attn2.q":3}' EDTR_AMD_ATTN_SPLIT=2
run conv3_qk3_tf3 'EDTR_AMD_POLICY={"base":"shipped","res.conv1":3,"res.conv2":3,"attn1.qkv":3,"attn2.q":3,"ff.geglu":3,"ff.out":3,"attn.out":3,"st.proj_in":3,"st.proj_out":3}'
run unet_all3_split2 'EDTR_AMD_POLICY={"base":"shipped","res.conv1":3,"res.conv2":3,"attn1.qkv":3,"attn2.q":3,"ff.geglu":3,"ff.out":3,"attn.out":3,"st.proj_in":3,"st.proj_out":3}' EDTR_AMD_ATTN_SPLIT=2
run unet_all3_split1 'EDTR_AMD_POLICY={"base":"shipped","res.conv1":3,"res.conv2":3,"attn1.qkv":3,"attn2.q":3,"ff.geglu":3,"ff.out":3,"attn.out":3,"st.proj_in":3,"st.proj_out":3}' EDTR_AMD_ATTN_SPLIT=1
run linears4 'EDTR_AMD_POLICY={"base":"shipped","attn1.qkv":4,"attn2.q":4,"ff.geglu":4,"ff.out":4,"attn.out":4,"st.proj_in":4,"st.proj_out":4}'
run linears2 'EDTR_AMD_POLICY={"base":"shipped","attn1.qkv":2,"attn2.q":2,"ff.geglu":2,"ff.out":2,"attn.out":2,"st.proj_in":2,"st.proj_out":2,"res.conv1":2,"res.conv2":2}'
