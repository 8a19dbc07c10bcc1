#!/usr/bin/env python3
"""Build variant libraries of the large-N attention kernel (generator options of tools/gen_attn_v2.py, optional in-kernel stamps)
under tools/exp/_build/, for A/B runs with `EDTR_LIB=<so> python3 tools/exp/hw_check_attn.py`:

    python3 tools/exp/attn_variants.py build            # here (cross-compiles for gfx950)
    python3 tools/exp/attn_variants.py stamps           # on the GPU box: run the stamped build, print per-tile cycle sums
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "_build")
CSRC = os.path.join(ROOT, "edtr_amd", "csrc")
VARIANTS = {  # name: (ahead, dma_spread, stamps, drop)
    "stamps_a3": (3, 0, 1, ""), "stamps_noexp": (3, 0, 1, "exp"), "stamps_noadd": (3, 0, 1, "add"), "stamps_nocvt": (3, 0, 1, "cvt"),
    "stamps_novalu": (3, 0, 1, "exp,add,cvt"), "stamps_nodma": (3, 0, 1, "dma"), "stamps_qkacc": (3, 0, 1, "qkacc"),
    "stamps_qkacc_novalu": (3, 0, 1, "qkacc,exp,add,cvt"),
}


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(CSRC, "build", f"{n}.o") for n in ("igemm", "halo512", "attn512", "norm", "elementwise", "swin")]
    for name, (ahead, spread, stamps, drop) in VARIANTS.items():
        inc = os.path.join(OUT, f"attn_v3_{name}.inc")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_v2.py"), "--ahead", str(ahead), "--dma-spread", str(spread),
                        "--stamps", str(stamps), "--out", inc, "--drop", drop], check=True)
        obj = os.path.join(OUT, f"attention_{name}.o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f'-DEDTR_ATTN_V3_INC="{inc}"', "-c",
               os.path.join(CSRC, "attention.hip"), "-o", obj] + (["-DEDTR_STAMPS"] if stamps else [])
        subprocess.run(cmd, check=True)
        so = os.path.join(OUT, f"libedtr_attn_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others, check=True)
        print(so, flush=True)


def stamps():
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for name in [n for n in VARIANTS if n.startswith("stamps")]:
        os.environ["EDTR_LIB"] = os.path.join(OUT, f"libedtr_attn_{name}.so")
        for mod in ("hipfree",):
            sys.modules.pop(mod, None)
        import hipfree as H
        from hipfree import L
        rng = np.random.default_rng(0)
        for (B, heads, N) in ((1, 16, 4096),):
            Cc = heads * 64
            q, k = H.rand16(rng, (B, N, heads, 64), 0, 0.42), H.rand16(rng, (B, N, heads, 64), 0, 0.42)
            vt = H.rand16(rng, (B, heads, 64, N), 0)
            dq, dk, dvt, do = H.Dev(q), H.Dev(k), H.Dev(vt), H.Dev(nbytes=B * N * Cc * 2, fill=0)
            p = L.AttnParams()
            p.dtype, p.B, p.H, p.Nq, p.Nk = 0, B, heads, N, N
            p.q, p.q_bs, p.q_ld, p.k, p.k_bs, p.k_ld = dq.p, N * Cc, Cc, dk.p, N * Cc, Cc
            p.vt, p.vt_bs, p.vt_ld, p.out, p.o_bs, p.o_ld = dvt.p, Cc * N, N, do.p, N * Cc, Cc
            p.scale, p.causal, p.q_prescaled = 0.125, 0, 1
            ms = H.time_launches([lambda s: H.chk(H.edtr.edtr_flash_attn64(C.byref(p), s), "attn")], iters=5, warm=2)
            nblk = (N // 256) * heads * B
            buf = (C.c_uint32 * (nblk * 16))()
            H.edtr.edtr_attn_stamps_read.argtypes = [C.c_void_p, C.c_int]
            H.hip.hipDeviceSynchronize()
            rc = H.edtr.edtr_attn_stamps_read(buf, nblk * 16)
            a = np.frombuffer(buf, dtype=np.uint32).reshape(nblk, 4, 4).astype(np.float64) / (N // 64)
            med = np.median(a.reshape(-1, 4), axis=0)
            print(f"[{name}] B{B} H{heads} N{N}: {ms * 1e3:7.1f} us  rc {rc}  cycles per tile (median over waves): sync {med[0]:.0f}  dma {med[1]:.0f}  "
                  f"phase1 {med[2]:.0f}  phase2 {med[3]:.0f}  total {med.sum():.0f}", flush=True)


if __name__ == "__main__":
    {"build": build, "stamps": stamps}[sys.argv[1]]()
