#!/usr/bin/env python3
"""Diagnostic: in-kernel phase timeline of igemm_dma_kernel (s_memtime stamps, see EDTR_STAMP in igemm.hip).

  python tools/exp/igemm_stamps.py build      # here (cross-compiles tools/exp/_build/libigemm_stamps.so for gfx950)
  python tools/exp/igemm_stamps.py run        # on the GPU box (gpurun)

Not part of the product path: it loads its own stamped build of igemm.hip, never edtr_amd/libedtr_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "exp", "_build")
SO = os.path.join(OUT, "libigemm_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "edtr_amd", "csrc", "igemm.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DEDTR_STAMPS",
                    src, os.path.join(ROOT, "edtr_amd", "csrc", "halo512.hip"), "-o", SO], check=True)
    print(SO)


def run():
    import numpy as np
    import torch
    from edtr_amd import lib as L
    dev = torch.device("cuda:0")
    lib = C.CDLL(SO)
    lib.edtr_igemm.argtypes = [C.POINTER(L.IgemmParams), C.c_void_p]
    DT = torch.bfloat16

    def case(label, M, N, Cin, taps=1, H=0, residual=False, act=0, tile=3):
        K = taps * Cin
        a = (torch.randn(M, Cin, device=dev)).to(DT)
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(DT)
        n_out = N // 2 if act == 1 else N
        out = torch.empty(M, n_out, dtype=DT, device=dev)
        res = torch.randn(M, n_out, device=dev).to(DT) if residual else None
        bias = torch.zeros(N, device=dev)
        nb = ((M + 127) // 128) * (N // 160) if tile == 8 else ((M + 127) // 128) * ((N + 127) // 128) if tile < 16 else min(256 if tile in (17, 18) else 1 << 30, (M // 256) * ((N + 127) // 128))
        stamps = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
        p = L.IgemmParams()
        p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = 0 if DT == torch.bfloat16 else 1, taps, M, N, K, 1, 1
        p.a1, p.C1, p.ld1 = a.data_ptr(), Cin, Cin
        if H:
            p.IH = p.IW = p.OH = p.OW = H
            p.stride, p.pad_t, p.pad_l = 1, (1 if taps == 9 else 0), (1 if taps == 9 else 0)
        p.w, p.ldw, p.alpha = w.data_ptr(), K, 1.0
        p.bias_n, p.act = bias.data_ptr(), act
        p.act_slope = float(os.environ.get('EDTR_SLOPE', '0'))
        if res is not None:
            p.residual, p.ldr = res.data_ptr(), n_out
        p.out, p.ldc = out.data_ptr(), n_out
        p.tile, p.splitk = tile, 1
        p.workspace, p.workspace_bytes = stamps.data_ptr(), stamps.numel() * 8
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert lib.edtr_igemm(C.byref(p), s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.edtr_igemm(C.byref(p), s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        st = stamps.cpu().numpy().reshape(nb, 16).astype(np.int64)
        d = lambda i, j: st[:, j] - st[:, i]
        pro, first, loop, epi, tot = d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(0, 4)
        rt0 = st[:, 6] - st[:, 6].min()
        rt1 = st[:, 7] - st[:, 6].min()
        span_us = rt1.max() / 100.0
        life_us = np.median(st[:, 7] - st[:, 6]) / 100.0
        nkt = K // 64
        hw = st[:, 5] & 0xFFFFFFFF
        xcc = (st[:, 5] >> 32) & 0xF
        cu = (hw >> 8) & 0xF
        se = (hw >> 13) & 0x7
        slots = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
        if os.environ.get("EDTR_STAMP_PAIRS"):      # which workgroups share a CU in the first round? (dispatch order)
            first = np.nonzero(rt0 / 100.0 < 1.0)[0]
            by_cu = {}
            for i in first:
                by_cu.setdefault((int(xcc[i]), int(se[i]), int(cu[i])), []).append(int(i))
            pairs = sorted(v for v in by_cu.values())[:12]
            diffs = sorted(set(abs(v[1] - v[0]) for v in by_cu.values() if len(v) == 2))
            print(f"    first-round sharing of a CU (block indices): {pairs} ... index distances {diffs[:12]}")
            order = np.argsort(rt0)[:16]
            print(f"    first 16 workgroups to start: {[int(i) for i in order]} on (xcc, se, cu) {[(int(xcc[i]), int(se[i]), int(cu[i])) for i in order]}")
        starts = np.sort(rt0) / 100.0
        # how many workgroups start in the first microsecond = resident slots
        first_wave = int((starts < 1.0).sum())
        flops = 2.0 * M * N * K
        print(f"{label:34s} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF | WGs {nb:5d} first-us {first_wave:4d} CUs {slots:3d} | span {span_us:6.1f} us "
              f"WG life {life_us:5.1f} us | cycles med: prologue {int(np.median(pro)):5d} first-tile {int(np.median(first)):5d} "
              + (f"[t16 epilogue: K-half write {int(np.median(st[:, 8] - st[:, 3]))} add+publish {int(np.median(st[:, 13] - st[:, 8]))} row loop {int(np.median(st[:, 14] - st[:, 13]))} tail {int(np.median(st[:, 4] - st[:, 14]))}] " if tile == 16 else "")
              + (f"[t17: epi-setup {int(np.median(st[:, 8] - st[:, 3]))} mbloop {int(np.median(st[:, 9] - st[:, 8]))} gn+rest {int(np.median(st[:, 4] - st[:, 9]))} | unit2 loop {int(np.median(st[:, 10] - st[:, 4]))} epi {int(np.median(st[:, 11] - st[:, 10]))}] " if tile == 17 else "") +
              (f"[t18 LAST unit: coords+offsets {int(np.median(st[:, 15] - st[:, 12]))} a_rd+zero {int(np.median(st[:, 1] - st[:, 15]))} wait+barrier {int(np.median(st[:, 2] - st[:, 1]))} loop {int(np.median(st[:, 3] - st[:, 2]))} pass0 write+wait {int(np.median(st[:, 8] - st[:, 3]))} pass0 rest {int(np.median(st[:, 10] - st[:, 8]))} pass1 write {int(np.median(st[:, 9] - st[:, 10]))} pass1 rest {int(np.median(st[:, 11] - st[:, 9]))} (last row loop {int(np.median(st[:, 14] - st[:, 13]))}) gn {int(np.median(st[:, 4] - st[:, 11]))} | kernel life cycles {int(np.median(st[:, 4] - st[:, 0]))}] " if tile == 18 else "") +
              (f"[t8 epilogue: pass-0 write {int(np.median(st[:, 12] - st[:, 3]))} pass 0 rows_phase+barrier {int(np.median(st[:, 15] - st[:, 12]))} pass 1 {int(np.median(st[:, 4] - st[:, 15]))} (last: prefetch+barrier {int(np.median(st[:, 13] - st[:, 15]))} row loop {int(np.median(st[:, 14] - st[:, 13]))})] " if tile == 8 else "") +
              (f"[prologue: to tile coords {int(np.median(st[:, 15] - st[:, 0]))}, rest {int(np.median(st[:, 1] - st[:, 15]))}] " if tile == 3 else "") + (f"[epilogue: acc->LDS {int(np.median(st[:, 12] - st[:, 3]))} prefetch+barrier {int(np.median(st[:, 13] - st[:, 12]))} row loop {int(np.median(st[:, 14] - st[:, 13]))} tail {int(np.median(st[:, 4] - st[:, 14]))}] " if tile == 3 else "") +
              f"loop {int(np.median(loop)):6d} ({int(np.median(loop)) // max(1, nkt):4d}/kt x{nkt}) epilogue {int(np.median(epi)):5d} total {int(np.median(tot)):6d}" + ("" if tile != 3 else f" | per kt: issue {int(np.median(st[:, 8])) // max(1, nkt - 1):4d} issue+vmwait {int(np.median(st[:, 9])) // nkt:4d} barrier {int(np.median(st[:, 10])) // nkt:4d} mfma-section {int(np.median(st[:, 11])) // nkt:4d}"),
              flush=True)

    B = 8
    if len(sys.argv) > 2 and sys.argv[2] == "halo":
        for t in (16,):            # (tile 18, the persistent form, was removed in round 4; its stamps are in profiles/r03/halo_persistent.log)
            case(f"t{t} conv 512^2 128->128", B * 512 * 512, 128, 128, taps=9, H=512, tile=t)
            case(f"t{t} conv 512^2 256->128", B * 512 * 512, 128, 256, taps=9, H=512, tile=t)
            case(f"t{t} conv 256^2 256->256", B * 256 * 256, 256, 256, taps=9, H=256, tile=t)
            case(f"t{t} conv 128^2 512->512", B * 128 * 128, 512, 512, taps=9, H=128, tile=t)
            case(f"t{t} conv 32^2 640->640", B * 1024, 640, 640, taps=9, H=32, tile=t)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "t8":
        for t in (3, 8):
            case(f"t{t} proj 64^2 K320 N320 +res", B * 4096, 320, 320, residual=True, tile=t)
            case(f"t{t} qkv 64^2 K320 N960", B * 4096, 960, 320, tile=t)
            case(f"t{t} ff.out 64^2 K1280 N320 +res", B * 4096, 320, 1280, residual=True, tile=t)
            case(f"t{t} proj 32^2 K640 N640 +res", B * 1024, 640, 640, residual=True, tile=t)
            case(f"t{t} conv 64^2 320->320", B * 4096, 320, 320, taps=9, H=64, tile=t)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "short":          # the transformer blocks' square linears: one round of <= 512 workgroups
        for t in (3, 8):
            case(f"t{t} M32768 K320 N320 +res", 32768, 320, 320, residual=True, tile=t)
            case(f"t{t} M8192 K640 N640 +res", 8192, 640, 640, residual=True, tile=t)
            case(f"t{t} M2048 K1280 N1280 +res", 2048, 1280, 1280, residual=True, tile=t)
            case(f"t{t} M1024 K1280 N1280 +res", 1024, 1280, 1280, residual=True, tile=t)
            case(f"t{t} M512 K1280 N1280 +res", 512, 1280, 1280, residual=True, tile=t)
            case(f"t{t} M2048 K5120 N1280 +res", 2048, 1280, 5120, residual=True, tile=t)
            case(f"t{t} M4096 K640 N640 +res", 4096, 640, 640, residual=True, tile=t)
            case(f"t{t} M16384 K320 N320 +res", 16384, 320, 320, residual=True, tile=t)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "spatial1":
        case("1x1 spatial 64^2 K320 N320", B * 4096, 320, 320, taps=1, H=64)
        case("1x1 spatial 64^2 K1280 N320", B * 4096, 320, 1280, taps=1, H=64)
        case("gemm 64^2 K1280 N320", B * 4096, 320, 1280)
        case("1x1 spatial 256^2 K256 N256", B * 65536, 256, 256, taps=1, H=256)
        case("gemm 256^2 K256 N256", B * 65536, 256, 256)
        case("gemm 256^2 K2304 N256", B * 65536, 256, 2304)
        case("conv 256^2 256->256", B * 256 * 256, 256, 256, taps=9, H=256)
        return
    case("proj 64^2 K320 N320 +res", B * 4096, 320, 320, residual=True)
    case("proj 64^2 K320 N320", B * 4096, 320, 320)
    case("qk 64^2 K320 N640", B * 4096, 640, 320)
    case("ff.out 64^2 K1280 N320 +res", B * 4096, 320, 1280, residual=True)
    case("geglu 64^2 K320 N2560", B * 4096, 2560, 320, act=1)
    case("proj 32^2 K640 N640 +res", B * 1024, 640, 640, residual=True)
    case("proj 16^2 K1280 N1280 +res", B * 256, 1280, 1280, residual=True)
    case("conv 64^2 320->320", B * 4096, 320, 320, taps=9, H=64)
    case("conv 32^2 640->640", B * 1024, 640, 640, taps=9, H=32)
    case("conv 512^2 128->128", B * 512 * 512, 128, 128, taps=9, H=512)
    case("conv 256^2 256->256", B * 256 * 256, 256, 256, taps=9, H=256)
    case("conv 128^2 512->512", B * 128 * 128, 512, 512, taps=9, H=128)


if __name__ == "__main__":
    (build if (len(sys.argv) > 1 and sys.argv[1] == "build") else run)()
