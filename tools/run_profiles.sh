#!/bin/bash
# Evidence run on the GPU box (through gpurun): benches of every BASELINE configuration, rocprofv3 kernel trace and PMC passes.
#   gpurun --timeout 2400 -- 'bash tools/run_profiles.sh'
# Outputs land in gpurun_out/prof/ ; summaries are then copied into profiles/rNN/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
if [ -z "$PROF_ONLY" ]; then
python bench.py --steps 20 --warmup 5 --breakdown > $O/bench_det512.json 2> $O/bench_det512.err
EDTR_BENCH_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_det512_dist1.json 2> $O/bench_det512_dist1.err
python bench.py --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_det512_fp16.json 2> $O/bench_det512_fp16.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py stats $O/trace $O/kernel_stats.csv
fi
# PMC passes: eager replay, one batch in flight (per-dispatch attribution); 3 passes of the path = build + warm-up + timed
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | tr ' ' '_')
  EDTR_SYNTH_DEVICE=cpu rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --inflight 1 --no-graph --no-cpu-baseline --no-roofline > $O/pmc_$tag.log 2>&1
  python3 $R/tools/prof_summary.py pmc $O/pmc_$tag $O/pmc_$tag.json --passes 3 > /dev/null
done
# the attention kernel alone (torch-free): MFMA-busy counters on the hot shape
EDTR_ATTN_PRESCALED=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_attn -- python3 $R/tools/exp/hw_check_attn.py > $O/pmc_attn.log 2>&1
python3 $R/tools/prof_summary.py pmc $O/pmc_attn $O/pmc_attn.json --by-grid > /dev/null
find $O -name '*.csv' | head -20
rm -rf $O/trace $O/pmc_*/   # raw rocprof output is large; the summaries stay
ls -la $O
cat $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES_GRBM_GUI_ACTIVE.json; cat $O/pmc_attn.json; cat $O/pmc_FETCH_SIZE.json | head -40
