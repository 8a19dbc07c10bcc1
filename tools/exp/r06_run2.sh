# Round 6, second set (one device).  The OLD v1 mapping is a build of commit b36e86e's attention.hip linked against the other current objects
# (git show b36e86e:edtr_amd/csrc/attention.hip > tools/exp/_build/attention_old.hip; hipcc -c ...; hipcc -shared -o tools/exp/_build/libedtr_hip_attnold.so ...).
# Round 6, second set: the v1 attention kernel's XCD-aware block mapping (old build = tools/exp/_build/libedtr_hip_attnold.so),
# the fused feed-forward's row threshold on the tiled workload, the driver-form bench against the committed PMC file.
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attn or flash or clip" 2>&1 | tail -3
echo "== attention shapes, old v1 mapping"; EDTR_AMD_LIB=$PWD/tools/exp/_build/libedtr_hip_attnold.so python3 tools/exp/r04_attn_shapes.py 8 2>&1 | grep -v amdgpu.ids
echo "== attention shapes, heads dealt to the XCDs"; python3 tools/exp/r04_attn_shapes.py 8 2>&1 | grep -v amdgpu.ids
for lib in old new old new; do
  if [ $lib = old ]; then export EDTR_AMD_LIB=$PWD/tools/exp/_build/libedtr_hip_attnold.so; else unset EDTR_AMD_LIB; fi
  echo "== det512 $lib"; python bench.py --steps 40 --also none --no-cpu-baseline --no-roofline --parity-steps 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
unset EDTR_AMD_LIB
for v in 0 16384 0 16384; do echo "== seg1024tiled EDTR_FFN_MIN_ROWS=$v"; EDTR_FFN_MIN_ROWS=$v python bench.py --workload seg1024tiled --no-cpu-baseline --no-roofline --steps 12 --warmup 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
