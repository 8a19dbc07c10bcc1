#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of the attention shapes of tools/exp/r04_attn_shapes.py: what the launches take
# on the device, without the back-to-back launch gaps the event timing includes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r04/attn_trace
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/exp/r04_attn_shapes.py > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r04/attn_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of the same (kernel, grid) = one case
runs = []
for r in rows:
    key = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if runs and runs[-1][0] == key: runs[-1][1].append(d)
    else: runs.append((key, [d]))
for key, ds in runs:
    ds = sorted(ds)
    print(f"{key[0]:62s} grid {key[1]:>7s} x {key[2]:>3s} x {key[3]:>2s}  n={len(ds):3d}  median {ds[len(ds)//2]/1e3:8.1f} us  min {ds[0]/1e3:8.1f} us")
PY
