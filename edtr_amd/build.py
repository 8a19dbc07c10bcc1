"""Build libedtr_hip.so (gfx950) in-tree with hipcc.  `python -m edtr_amd.build` or build_library()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libedtr_hip.so")
SOURCES = ["igemm.hip", "halo512.hip", "attention.hip", "attn512.hip", "norm.hip", "elementwise.hip", "swin.hip", "ffn.hip", "lin320.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "edtr_hip.h")]
# files #include'd by one source only (generated code): a regenerated .inc must rebuild its object
EXTRA_DEPS = {"attention.hip": [os.path.join(CSRC, "attn_v3_loop.inc")]}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libedtr_hip.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha1 over the kernel sources (csrc/*.hip, *.inc, *.h and the ABI header): stamps measurements that are read back later
    (bench.py refuses a committed PMC traffic figure measured on different kernel code; .git does not travel to the GPU box)."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".h")))
    for path in [os.path.join(CSRC, f) for f in files] + [HEADERS[1]]:
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def _object_key(flags, deps) -> str:
    """sha1 over the compile flags and the bytes of every dependency: an object is reused only when this matches the key
    written next to it, so an object compiled elsewhere or with other -D flags can never be linked by accident (mtimes lie)."""
    import hashlib
    h = hashlib.sha1(" ".join(flags).encode())
    for d in deps:
        with open(d, "rb") as fh:
            h.update(b"\0" + os.path.basename(d).encode() + b"\0" + fh.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    relink = force
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        # -Werror=pass-failed: a `#pragma unroll` the optimizer could not honour is an ERROR — in round 4 such a loop around the
        # shared epilogue silently stayed rolled and sent a kernel's accumulators through scratch memory
        flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Werror=pass-failed"]
        key = _object_key(flags, [s] + HEADERS + EXTRA_DEPS.get(src, []))
        keyfile = o + ".key"
        have = open(keyfile).read().strip() if os.path.exists(keyfile) and os.path.exists(o) else ""
        if force or have != key:
            jobs.append(([hipcc] + flags + ["-c", s, "-o", o], keyfile, key))
        objs.append(o)

    def _compile(job):
        cmd, keyfile, key = job
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(keyfile):
            os.remove(keyfile)                 # (a failed compile must not leave a key that vouches for the old object)
        subprocess.run(cmd, check=True)
        with open(keyfile, "w") as fh:
            fh.write(key + "\n")

    if jobs:
        # translation units compile independently: a few at a time (the container has 8 cores; one hipcc peaks near 3 GB)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("EDTR_BUILD_JOBS", "4")))) as pool:
            list(pool.map(_compile, jobs))
        relink = True
    if relink or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash())
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
