#!/bin/bash
# PMC counters of edtr_swin_mlp alone (run on the GPU box from the repo root): two passes, results under gpurun_out/r04/mlp_pmc*
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $ROOT/gpurun_out/r04/mlp_pmc1 -- python3 $ROOT/tools/exp/r04_swin_mlp_bench.py 32768 > /dev/null 2> $ROOT/gpurun_out/r04/mlp_pmc1.log
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $ROOT/gpurun_out/r04/mlp_pmc2 -- python3 $ROOT/tools/exp/r04_swin_mlp_bench.py 32768 > /dev/null 2> $ROOT/gpurun_out/r04/mlp_pmc2.log
cd $ROOT
python3 - <<'P'
import csv, glob, collections
for d in ("mlp_pmc1", "mlp_pmc2"):
    for f in glob.glob(f"gpurun_out/r04/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "swin_mlp_kernel" in row["Kernel_Name"] and "BF16" in row["Kernel_Name"]:
                a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(agg.items()):
            print(f"{d} {k:32s} {v / n:14.0f} per launch ({n} launches)")
P
