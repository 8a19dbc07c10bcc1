#!/bin/bash
# rocprofv3 evidence for one round (run through gpurun from the repo root):  bash tools/profile_round.sh r03
# (round 4: bench.py runs one extra, untimed parity pass after the timed region: 9 / 4 passes per run instead of 8 / 3)
#   1. --kernel-trace --stats of the default bench (per-kernel table + per-family busy-time union = the in-situ figure)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (eager replay, one batch in flight) -> pmc_hbm_traffic.json,
#      stamped with the kernel source hash bench.py checks before it quotes roofline.traffic
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -> pmc_mfma_busy_bench.json
# (counters are never combined with a trace domain: gpurun refuses that; the program itself follows `--`.)
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --parity-steps 0 --also none"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log
python3 $ROOT/tools/prof_summary.py stats $OUT/trace $OUT/kernel_stats.csv
# warm-up (first pass per slot = 2) + 1 warm-up round (2) + 4 timed + 1 parity pass = 9 passes under the default two batches in flight
python3 $ROOT/tools/prof_summary.py union $OUT/trace $OUT/kernel_busy_union.json --passes 9 > /dev/null
PARGS="--steps 1 --warmup 1 --inflight 1 --no-graph --no-cpu-baseline --no-roofline --parity-steps 0 --also none"
export EDTR_SYNTH_DEVICE=cpu      # rocprofv3 --pmc FETCH_SIZE crashes inside torch's int64 elementwise kernels (weight hashing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/bench.py $PARGS > /dev/null 2> $OUT/pmc_mfma.log
python3 $ROOT/tools/prof_summary.py pmc $OUT/pmc_fetch $OUT/pmc_fetch.json --passes 4 > /dev/null
python3 $ROOT/tools/prof_summary.py pmc $OUT/pmc_write $OUT/pmc_write.json --passes 4 > /dev/null
python3 $ROOT/tools/prof_summary.py pmc $OUT/pmc_mfma $OUT/pmc_mfma_busy_bench.json --passes 4 > /dev/null
python3 $ROOT/tools/exp/pmc_by_kernel.py $OUT/pmc_fetch FETCH_SIZE > $OUT/pmc_fetch_by_kernel.txt
python3 $ROOT/tools/exp/pmc_mfma_by_kernel.py $OUT/pmc_mfma > $OUT/pmc_mfma_busy_by_kernel.txt
python3 - $OUT $ROOT <<'PY'
import json, sys, os
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from edtr_amd.build import source_hash
f, w = json.load(open(f"{out}/pmc_fetch.json")), json.load(open(f"{out}/pmc_write.json"))
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) -- python3 bench.py --steps 1 --warmup 1 "
                 "--inflight 1 --no-graph --no-cpu-baseline --no-roofline --parity-steps 0 (EDTR_SYNTH_DEVICE=cpu); 4 passes of the path "
                 "(program build, warm-up, timed, parity), 1 x MI355X, batch 8 of 512x512 bf16",
       "correction": "fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts wide coalesced reads at half their bytes: MI355X_MICROARCH.md, "
                     "HBM section); write bytes = WRITE_SIZE x 1024",
       "kernel_source_hash": source_hash(), "passes": 4, "families": {}}
for k in sorted(set(f["families"]) | set(w["families"])):
    a, b = f["families"].get(k, {}), w["families"].get(k, {})
    n = a.get("launches_per_pass") or b.get("launches_per_pass")
    fb, wb = a.get("fetch_bytes_per_launch", 0.0), b.get("write_bytes_per_launch", 0.0)
    res["families"][k] = {"launches_per_pass": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                          "hbm_side_bytes_per_pass": (fb + wb) * n}
json.dump(res, open(f"{out}/pmc_hbm_traffic.json", "w"), indent=1)
print({k: round(v["hbm_side_bytes_per_pass"] / 1e9, 2) for k, v in res["families"].items()})
PY
tail -2 $OUT/trace.json | cut -c1-300
ls $OUT
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/trace      # raw CSVs are large; the summaries above are what gets committed
