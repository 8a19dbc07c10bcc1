# Round 6: why is fp16 storage 3 % slower than bf16 on the same device (115.4 against 119.1 images/s)?  Per-launch-class tables of both.
mkdir -p gpurun_out/r06
for dt in bf16 fp16; do
  python bench.py --dtype $dt --steps 12 --warmup 2 --also none --no-cpu-baseline --parity-steps 0 --breakdown --breakdown-json gpurun_out/r06/bd_$dt.json > gpurun_out/r06/bench_$dt.json 2> gpurun_out/r06/bench_${dt}_breakdown.txt
done
python - <<'PY'
import json
a = json.load(open("gpurun_out/r06/bd_bf16.json")); b = json.load(open("gpurun_out/r06/bd_fp16.json"))
print(f"total kernel ms per pass: bf16 {a['total_ms']:.2f}  fp16 {b['total_ms']:.2f}")
rows = []
for k in sorted(set(a["by_name"]) | set(b["by_name"])):
    x, y = a["by_name"].get(k, {}).get("ms", 0.0), b["by_name"].get(k, {}).get("ms", 0.0)
    rows.append((y - x, k, x, y))
for d, k, x, y in sorted(rows, reverse=True)[:16]:
    print(f"{k:24s} bf16 {x:7.3f} ms  fp16 {y:7.3f} ms  diff {d:+.3f}")
for f in ("bf16", "fp16"):
    j = json.loads(open(f"gpurun_out/r06/bench_{f}.json").read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"])
PY
