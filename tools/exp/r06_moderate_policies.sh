# Round 6: the rows moderate_policies.log was missing — the WEIGHTS-EXACT two-part product (policy value 4 =
# x16 . [Wh | Wl]; plain GEMMs only: a 3 x 3 convolution asked for 4 runs 3 parts) everywhere, and on the FLOP-heavy classes with the
# shipped carriers — on the moderate-outlier weight set against the reference's outputs (tests/golden/moderate.npz).
run() { echo "== $1"; shift; env "$@" python -m pytest tests/test_gpu_heavy.py -m gpu -q -s -k "moderate and mixed" 2>&1 | grep "moderate, mixed" | sed 's/.*block\] //'; }
run shipped X=1
run all4_linears_conv3 'EDTR_AMD_POLICY={"default": 4}'
run shipped_plus_default4 'EDTR_AMD_POLICY={"base":"shipped","default":4}'
run linears4_conv1_carriers3 'EDTR_AMD_POLICY={"base":"shipped","ff.geglu":4,"ff.out":4,"attn.out":4,"st.proj_in":4,"st.proj_out":4,"attn1.qkv":4,"attn2.q":4,"attn1.qk":4,"attn1.vT":4}'
run linears4_resconv3_carriers3 'EDTR_AMD_POLICY={"base":"shipped","ff.geglu":4,"ff.out":4,"attn.out":4,"st.proj_in":4,"st.proj_out":4,"attn1.qkv":4,"attn2.q":4,"attn1.qk":4,"attn1.vT":4,"res.conv1":3,"res.conv2":3}'
run all3 'EDTR_AMD_POLICY={"default": 3}'
