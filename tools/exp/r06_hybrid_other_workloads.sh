# Round 6: the hybrid and robust parity modes on the other two single-GPU BASELINE configurations (throughput + parity vs the reference golden).
for wl in seg1024tiled det512s50; do for pr in hybrid robust; do
  st=6; [ $wl = det512s50 ] && st=3
  echo "== $wl --precision $pr"; python bench.py --workload $wl --precision $pr --steps $st --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=j.get('parity_vs_reference_golden',{}); print(j['value'], j['ms_per_step'], g.get('rel_err_latent'), g.get('rel_err_image_samples'), g.get('ok'))"
done; done
