#!/bin/bash
# the round's bench set, one device: bash tools/exp/final_benches.sh <outdir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
python bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err
python bench.py --no-cpu-baseline --also none --parity-steps 0 --steps 12 --warmup 2 --breakdown --breakdown-json $O/bench_det512_b8_bf16.breakdown.json > $O/bench_det512_b8_bf16.json 2> $O/bench_det512_b8_bf16_breakdown.txt
python bench.py --dtype fp16 --no-cpu-baseline --also none --parity-steps 0 --no-roofline --steps 12 --warmup 2 > $O/bench_det512_b8_fp16.json 2>/dev/null
python bench.py --precision mixed --no-cpu-baseline --also none --parity-steps 0 --steps 10 --warmup 2 --breakdown > $O/bench_det512_mixed.json 2> $O/bench_det512_mixed_breakdown.txt
python bench.py --precision high --no-cpu-baseline --also none --parity-steps 0 --no-roofline --steps 6 --warmup 2 > $O/bench_det512_high.json 2>/dev/null
python bench.py --workload det512s50 --no-cpu-baseline --steps 3 --warmup 1 --breakdown > $O/bench_det512s50.json 2> $O/bench_det512s50_breakdown.txt
python bench.py --workload seg1024tiled --no-cpu-baseline --steps 6 --warmup 2 --breakdown > $O/bench_seg1024tiled.json 2> $O/bench_seg1024tiled_breakdown.txt
python bench.py --swinir --no-cpu-baseline --also none --parity-steps 0 --no-roofline --steps 20 --warmup 3 > $O/bench_swinir.json 2>/dev/null
for f in $O/bench_*.json; do case $f in *breakdown*) ;; *) tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('parity_mode') or {}).get('images_per_s'))";; esac; done
