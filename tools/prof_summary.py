#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small JSON / text files that can be committed under profiles/.

    python3 tools/prof_summary.py stats  <rocprof dir> <out.csv>        # --kernel-trace --stats: per-kernel time table
    python3 tools/prof_summary.py pmc    <rocprof dir> <out.json> [--passes N]
                                                                        # --pmc: per-kernel-family counter sums / per launch
    python3 tools/prof_summary.py union  <rocprof dir> <out.json> [--passes N]
                                                                        # --kernel-trace: per family, the UNION of the kernels'
                                                                        # [start, end] intervals (in-situ busy time: overlapping
                                                                        # lanes / batches counted once) next to the plain sum

Kernel families: igemm (main loops + split-K reducers + the fused feed-forward launch), flash_attn (v1 / v2 / v3), gn_apply, gn_stats, layernorm, other.
PMC conventions (MI355X_MICROARCH.md): FETCH_SIZE is in KiB and counts wide coalesced reads at HALF their bytes on gfx950
(x2 applied here), WRITE_SIZE is KiB exact; SQ_*_CYCLES are summed over the chip's SIMDs/XCDs as rocprofv3 reports them."""
import csv
import glob
import json
import os
import sys


def family(name: str) -> str:
    n = name.lower()
    if "flash_attn512" in n:
        return "flash_attn_d512"
    if "flash_attn" in n:
        if "smallk" in n:
            return "flash_attn_smallk"
        if "split" in n:
            return "flash_attn_split"
        return "flash_attn_v3" if "v3" in n else ("flash_attn_v2" if "v2" in n else "flash_attn_v1")
    if "igemm" in n or "splitk_reduce" in n or "ffn320" in n or "lin320" in n:      # (edtr_ffn / edtr_lin320: linears of the same family)
        return "igemm"
    for k in ("gn_apply", "gn_stats", "gn_finalize", "layernorm", "softmax_rows", "window_attn"):
        if k in n:
            return k
    return "other"


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def col(header, *cands):
    for c in cands:
        for i, h in enumerate(header):
            if h.strip().strip('"').lower() == c.lower():
                return i
    raise KeyError(f"none of {cands} in {header}")


def stats(root, out):
    files = find(root, "*kernel_stats.csv")
    if not files:
        raise SystemExit(f"no *kernel_stats.csv under {root}")
    rows = []
    for f in files:
        with open(f) as fh:
            r = list(csv.reader(fh))
        rows += r if not rows else r[1:]
    with open(out, "w", newline="") as fh:
        csv.writer(fh).writerows(rows[:41])
    print(f"{out}: {len(rows) - 1} kernels (top 40 kept)")


def pmc(root, out, passes, by_grid=False):
    files = find(root, "*counter_collection.csv")
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {root}")
    fam = {}
    for f in files:
        with open(f) as fh:
            rd = csv.reader(fh)
            header = next(rd)
            kn, cn, cv = col(header, "Kernel_Name"), col(header, "Counter_Name"), col(header, "Counter_Value")
            did = col(header, "Dispatch_Id")
            gs = col(header, "Grid_Size")
            for row in rd:
                key = family(row[kn])
                if by_grid and key.startswith("flash_attn"):
                    key += f"[grid={row[gs]}]"
                d = fam.setdefault(key, {"dispatches": set(), "counters": {}})
                d["dispatches"].add(row[did])
                d["counters"][row[cn]] = d["counters"].get(row[cn], 0.0) + float(row[cv])
    res = {"passes": passes, "families": {}}
    for k, d in sorted(fam.items()):
        n = len(d["dispatches"])
        e = {"launches": n, "launches_per_pass": n / passes}
        for c, v in d["counters"].items():
            if c == "FETCH_SIZE":
                e["fetch_bytes_per_launch"] = 2.0 * v * 1024.0 / n
            elif c == "WRITE_SIZE":
                e["write_bytes_per_launch"] = v * 1024.0 / n
            else:
                e[c + "_per_launch"] = v / n
        c = d["counters"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe busy cycles summed over the chip's 1024 SIMDs (= 32 per v_mfma_f32_32x32x16);
            # GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs -> fraction of the kernels' cycles the matrix pipes were busy
            e["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        res["families"][k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


def union(root, out, passes):
    """In-situ figure of the timed region: inside the hipGraphs two lanes and two batches overlap, so the
    sum of a family's kernel durations exceeds the wall time it occupies.  From the kernel trace: per family the union of its
    dispatch intervals (time during which at least one kernel of the family was running), the plain sum, and the same over all
    kernels (= GPU busy time) — per pass."""
    files = find(root, "*kernel_trace.csv")
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    iv = {}
    for f in files:
        with open(f) as fh:
            rd = csv.reader(fh)
            header = next(rd)
            kn, st, en = col(header, "Kernel_Name"), col(header, "Start_Timestamp"), col(header, "End_Timestamp")
            for row in rd:
                a, b = int(row[st]), int(row[en])
                iv.setdefault(family(row[kn]), []).append((a, b))
                iv.setdefault("ALL", []).append((a, b))

    def merged(spans):
        spans = sorted(spans)
        tot, cs, ce = 0, None, None
        for a, b in spans:
            if ce is None or a > ce:
                if ce is not None:
                    tot += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        return tot + (ce - cs if ce is not None else 0)

    res = {"passes": passes, "unit": "ms per pass", "families": {}}
    for k, spans in sorted(iv.items()):
        res["families"][k] = {"launches_per_pass": len(spans) / passes, "sum_ms": sum(b - a for a, b in spans) / passes / 1e6,
                              "union_ms": merged(spans) / passes / 1e6}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    mode, root, out = sys.argv[1:4]
    passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 1
    {"stats": lambda: stats(root, out), "pmc": lambda: pmc(root, out, passes, "--by-grid" in sys.argv),
     "union": lambda: union(root, out, passes)}[mode]()
