"""Per-kernel matrix-pipe occupancy from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass over bench.py:
    python3 tools/exp/pmc_mfma_by_kernel.py <rocprof dir>
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (1024 x GRBM_GUI_ACTIVE / 8: the counter is summed over
the 8 XCDs) per kernel name (and grid, for the igemm kernels) —
the same definition tools/prof_summary.py uses per family (the halo kernels on their own, not the family figure)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rd = csv.reader(open(f))
    h = next(rd)
    kn, cn, cv, gs, di = h.index("Kernel_Name"), h.index("Counter_Name"), h.index("Counter_Value"), h.index("Grid_Size"), h.index("Dispatch_Id")
    seen = set()
    for r in rd:
        n = r[kn]
        if not any(k in n for k in ("igemm", "flash_attn", "conv128", "conv64", "swin", "ffn320", "lin320")):
            continue
        short = n.replace("void (anonymous namespace)::", "").replace("(edtr_igemm_params)", "").replace("(edtr_attn_params)", "").replace("(edtr_ffn_params)", "").replace("(edtr_lin320_params)", "").replace("(anonymous namespace)::", "")[:56]
        key = (short, r[gs] if "halo" in n else "")
        agg[key][r[cn]] += float(r[cv])
        if (r[di], key) not in seen:
            seen.add((r[di], key))
            cnt[key] += 1
rows = []
for key, c in agg.items():
    act = c.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0:
        continue
    rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * act / 8.0), act, key))
print(f"{'kernel':56s} {'grid':>10s} {'launches':>8s} {'GUI-active cycles/launch':>24s}  MFMA pipe busy")
for frac, act, (k, g) in sorted(rows, key=lambda r: -r[1])[:40]:
    n = max(1, cnt[(k, g)])
    print(f"{k:56s} {g:>10s} {n:8d} {act / n:24.0f}  {frac:6.3f}")
