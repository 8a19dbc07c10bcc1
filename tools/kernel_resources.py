#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of libedtr_hip.so (reads the code-object notes; no GPU needed).
Prints every kernel that spills or uses scratch, and with --all the whole table.  python tools/kernel_resources.py [--all] [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = os.path.join(ROOT, "edtr_amd", "libedtr_hip.so")
    show_all = "--all" in sys.argv
    filt = [a for a in sys.argv[1:] if not a.startswith("--")]
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "lib.so")
        os.symlink(lib, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, cwd=td)
        rows = []
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                rows.append((name, blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                             g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
    bad = 0
    for name, ag, vg, sg, lds, scr, vsp, ssp in rows:
        if filt and not any(x in name for x in filt):
            continue
        spill = scr not in ("0", "?") or vsp not in ("0", "?")
        bad += spill
        if show_all or spill:
            print(f"{'SPILL ' if spill else '      '}vgpr {vg:>3} agpr {ag:>3} sgpr {sg:>3} lds {lds:>6} scratch {scr:>5} vspill {vsp:>3}  {name[:150]}")
    print(f"{len(rows)} kernels, {bad} with scratch / spills")


if __name__ == "__main__":
    main()
