"""In-kernel phase stamps of edtr_ffn.  Needs a DIAGNOSTIC build of the library (the product library contains no stamp code; the stamps
overwrite the first output rows of every workgroup), kept out of edtr_amd/csrc/build:

    mkdir -p tools/exp/_build && cd tools/exp/_build
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFFN_STAMPS -c ../../../edtr_amd/csrc/ffn.hip -o ffn_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o libedtr_hip_ffnstamps.so ../../../edtr_amd/csrc/build/{igemm,halo512,attention,attn512,norm,elementwise,swin}.o ffn_stamps.o
    EDTR_AMD_LIB=$PWD/libedtr_hip_ffnstamps.so python tools/exp/ffn_stamps.py          # from the repo root, on the GPU box

Prints median cycles per phase over the workgroups for waves 0 and 4 (the two waves of SIMD 0).  profiles/r06/ffn_stamps.log."""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from edtr_amd import ops  # noqa: E402


def main():
    d = torch.device("cuda:0")
    dtype = torch.bfloat16
    D, H, M = 320, 1280, 32768
    g = torch.Generator().manual_seed(0)
    x = torch.randn((M, D), generator=g).to(dtype).to(d)
    w1 = torch.randn((2 * H, D), generator=g) / math.sqrt(D)
    w2 = torch.randn((D, H), generator=g) / math.sqrt(H)
    w1p = ops.pack_linear_weight(w1[ops.geglu_perm(H)], dtype)
    cst = ops.pack_ffn_constants(torch.zeros(2 * H))
    out = torch.zeros((M, D), dtype=dtype, device=d)
    rec = ops.make_ffn(dtype=dtype, x=x, ldx=D, M=M, w1=w1p.to(d), w2=ops.pack_ffn_w2(w2, dtype).to(d), cst=cst.to(d), b2=torch.zeros(D, device=d), out=out, ldo=D)
    for _ in range(3):
        ops.launch(rec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ops.launch(rec)
    torch.cuda.synchronize()
    print(f"launch time {1e6 * (time.perf_counter() - t0) / n:.1f} us (back to back, host clock)")
    raw = out.view(torch.int16).cpu().reshape(M // 128, 128, D)[:, :8, :32].contiguous().view(torch.int64).reshape(M // 128, 8, 8)
    names = ["wait+barrier A", "mfma A", "geglu", "wait+barrier B", "mfma B", "prologue", "epilogue", "total"]
    for w in (0, 4):
        med = raw[:, w, :].float().median(dim=0).values
        print(f"wave {w}: " + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, med.tolist())))
    tot = raw[:, 0, 7].float()
    print(f"total cycles min {int(tot.min())} median {int(tot.median())} max {int(tot.max())}  (memtime ticks: 100 MHz -> x ~20 for shader cycles)")


if __name__ == "__main__":
    main()
