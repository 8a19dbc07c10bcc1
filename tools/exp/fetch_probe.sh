#!/bin/bash
# FETCH_SIZE per dispatch of tools/exp/fetch_probe.hip (see its header) -> gpurun_out/r04/fetch_probe.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
hipcc --offload-arch=gfx950 -O3 tools/exp/fetch_probe.hip -o /tmp/fetch_probe || exit 1
{
echo "# plain run (HIP events around each launch)"
/tmp/fetch_probe
export TMPDIR=/tmp
rm -rf /tmp/fp_pmc; (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fp_pmc -- /tmp/fetch_probe > /tmp/fp_pmc.log 2>&1)
echo "# rocprofv3 --pmc FETCH_SIZE, per dispatch of read_kernel (FETCH_SIZE is in KiB; gfx950 counts 128-B requests at 64 B: x2, MI355X_MICROARCH.md)"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/fp_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "read_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
sizes = [32, 64, 128, 192, 384, 1024]
for i, r in enumerate(rows):
    sz = sizes[i // 4]
    v = float(r["Counter_Value"])
    print(f"buffer {sz:5d} MiB  read {i % 4 + 1}: FETCH_SIZE {v:12.0f} KiB  -> x2 = {2 * v / 1024:8.1f} MiB = {2 * v / 1024 / sz:5.2f} x the buffer")
PY
} 2>&1 | tee gpurun_out/r04/fetch_probe.log
