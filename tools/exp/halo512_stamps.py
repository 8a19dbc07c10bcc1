#!/usr/bin/env python3
"""In-kernel phase timeline of the 512-pixel halo tile (tile 17, edtr_amd/csrc/halo512.hip), torch-free.

  python3 tools/exp/halo512_stamps.py build     # here: cross-compiles tools/exp/_build/libhalo512_stamps.so (-DEDTR_STAMPS)
  python3 tools/exp/halo512_stamps.py run       # on the GPU box (gpurun)

Thread 0 of every workgroup stores s_memtime at phase boundaries; medians over the workgroups are printed in cycles.
Not part of the product: it loads its own stamped build, never edtr_amd/libedtr_hip.so."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "_build")
SO = os.environ.get("H5_SO") or os.path.join(OUT, "libhalo512_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "edtr_amd", "csrc", "halo512.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DEDTR_STAMPS"] + sys.argv[2:] +
                   [src, "-o", SO], check=True)
    print(SO)


def run():
    import hipfree as H
    from hipfree import C, L
    lib = C.CDLL(SO)
    lib.edtr_halo512_stamped.argtypes = [C.POINTER(L.IgemmParams), C.c_void_p]
    rng = np.random.default_rng(0)
    dt = 0
    tile = int(os.environ.get("H5_TILE", "17"))
    upx = 512

    def case(label, B, Hh, Ww, Cin, N, residual=True, gnp=True, gnin=False, out_f32=False):
        M, K = B * Hh * Ww, 9 * Cin
        a = H.Dev(H.rand16(rng, (M, Cin), dt))
        w = H.Dev(H.rand16(rng, (N, K), dt, 1.0 / np.sqrt(K)))
        bias = H.Dev(rng.standard_normal(N, dtype=np.float32))
        res = H.Dev(H.rand16(rng, (M, N), dt)) if residual else None
        out = H.Dev(nbytes=M * N * (4 if out_f32 else 2), fill=0)
        nb = (M // upx) * (N // 128)
        stamps = H.Dev(nbytes=nb * 16 * 8, fill=0)
        g = H.Dev(nbytes=(M // 128) * N * 8, fill=0) if gnp else None
        tbl = H.Dev(np.tile(np.array([1.0, 0.0], dtype=np.float32), B * Cin)) if gnin else None
        p = L.IgemmParams()
        p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = dt, 9, M, N, K, 1, 1
        p.a1, p.C1, p.ld1, p.w, p.ldw = a.p, Cin, Cin, w.p, K
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = Hh, Ww, Hh, Ww, 1, 1, 1, 0
        p.alpha, p.bias_n = 1.0, bias.p
        if residual:
            p.residual, p.ldr = res.p, N
        p.out, p.ldc, p.out_f32, p.tile, p.splitk = out.p, N, int(out_f32), tile, 1
        p.workspace, p.workspace_bytes = stamps.p, nb * 128
        if gnp:
            p.gn_partial = g.p
        if gnin:
            p.a_gn, p.a_gn_silu = tbl.p, 1
        ms = H.time_launches([lambda s: H.chk(lib.edtr_halo512_stamped(C.byref(p), s), "halo512")], iters=10, warm=3)
        st = stamps.get(np.int64, (nb, 16))
        if tile == 21:      # the persistent form: bit-identical to tile 17?  per-unit spans of a workgroup
            got = out.get(np.uint16, (M, N)).copy()
            gg = g.get(np.float32, ((M // 128), N, 2)).copy() if gnp else None
            out2 = H.Dev(nbytes=M * N * 2, fill=0)
            g2 = H.Dev(nbytes=(M // 128) * N * 8, fill=0) if gnp else None
            p.tile, p.out = 17, out2.p
            if gnp:
                p.gn_partial = g2.p
            ms17 = H.time_launches([lambda s: H.chk(lib.edtr_halo512_stamped(C.byref(p), s), "halo512")], iters=10, warm=3)
            ref = out2.get(np.uint16, (M, N))
            same = bool((got == ref).all())
            gsame = bool((gg == g2.get(np.float32, ((M // 128), N, 2))).all()) if gnp else True
            p.tile, p.out = 21, out.p
            if gnp:
                p.gn_partial = g.p
            G = min(256, nb)
            sw = st[:G]
            dd = lambda i, j: int(np.median(sw[:, j] - sw[:, i]))
            per = [dd(2, 8)] + [dd(8 + k - 1, 8 + k) for k in range(1, min(6, nb // G))]
            flops = 2.0 * M * N * K
            print(f"{label:30s} tile21 {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF | tile17 {ms17 * 1e3:7.1f} us {flops / ms17 / 1e9:6.0f} TF | x{ms17 / ms:5.3f} | "
                  f"bit-identical out {same} gn {gsame} | cycles: setup+first {dd(0, 2)} units(loop+pack) {per} tail {dd(3, 5)} total {dd(0, 5)}", flush=True)
            return
        d = lambda i, j: int(np.median(st[:, j] - st[:, i]))
        nchunk = Cin // 32
        life = np.median(st[:, 15] - st[:, 14]) / 100.0
        span = (st[:, 15].max() - st[:, 14].min()) / 100.0
        flops = 2.0 * M * N * K
        if os.environ.get("H5_PAIRS"):
            hw = st[:, 7] & 0xFFFFFFFF
            xcc, cu, se = (st[:, 7] >> 32) & 0xF, (hw >> 8) & 0xF, (hw >> 13) & 0x7
            t0 = st[:, 14] - st[:, 14].min()
            first = np.nonzero(t0 < 100)[0]          # started in the first microsecond
            by = {}
            for i in first:
                by.setdefault((int(xcc[i]), int(se[i]), int(cu[i])), []).append(int(i))
            pairs = sorted(by.values())
            print(f"    first-us blocks {len(first)} on {len(by)} CUs; sharing (block ids): {pairs[:10]} distances {sorted(set(v[-1] - v[0] for v in pairs if len(v) > 1))[:10]}")
        print(f"{label:30s} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF | units {nb:5d} life {life:5.1f} us span {span:6.1f} us | cycles med: setup {d(0, 1):5d} "
              f"first-load {d(1, 2):5d} loop {d(2, 3):6d} (chunk0 {d(2, 6):5d}, {d(2, 3) // (9 * nchunk):4d}/tap x{9 * nchunk}) epilogue {d(3, 4):5d} gn {d(4, 5):5d} total {d(0, 5):6d}"
              f" | cycles/us {np.median((st[:, 5] - st[:, 0]) / np.maximum(1, st[:, 15] - st[:, 14])) * 100:6.0f}", flush=True)

    if os.environ.get("H5_SMALL"):      # how does the epilogue scale with the number of busy CUs?  (per-CU limit or chip-wide burst)
        for (b, hh, ww) in [(1, 64, 128), (1, 128, 256), (1, 256, 256), (1, 256, 512), (2, 512, 512)]:
            case(f"{b}x{hh}x{ww} 128->128 res gnp", b, hh, ww, 128, 128)
            case(f"{b}x{hh}x{ww} 128->128 plain", b, hh, ww, 128, 128, residual=False, gnp=False)
        return
    if tile == 21:
        case("512^2 128->128 gnp", 8, 512, 512, 128, 128, residual=False)
        case("512^2 128->128 plain", 8, 512, 512, 128, 128, residual=False, gnp=False)
        case("512^2 128->128 gnin gnp", 8, 512, 512, 128, 128, residual=False, gnin=True)
        case("256^2 256->256 gnp", 8, 256, 256, 256, 256, residual=False)
        case("256^2 256->256 gnin gnp", 8, 256, 256, 256, 256, residual=False, gnin=True)
        case("128^2 512->512 gnp", 8, 128, 128, 512, 512, residual=False)
        case("3x256x512 128->256 gnin gnp", 3, 256, 512, 128, 256, residual=False, gnin=True)
        case("1x64x96 256->128 gnp", 1, 64, 96, 256, 128, residual=False)
        return
    case("512^2 128->128 res gnp", 8, 512, 512, 128, 128)
    case("512^2 128->128 plain", 8, 512, 512, 128, 128, residual=False, gnp=False)
    case("512^2 128->128 gnin", 8, 512, 512, 128, 128, residual=False, gnin=True)
    case("512^2 128->128 f32 out", 8, 512, 512, 128, 128, residual=False, out_f32=True)
    case("256^2 256->256 res gnp", 8, 256, 256, 256, 256)
    case("128^2 512->512 res gnp", 8, 128, 128, 512, 512)


if __name__ == "__main__":
    (build if (len(sys.argv) > 1 and sys.argv[1] == "build") else run)()
