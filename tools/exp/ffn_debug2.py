"""edtr_ffn mapping probe: W1 = 0, one gated unit j0 switched on through its biases (G[j0] = 6, all others 0), W2[n][j] = j:
(out - x) / 6 is the column of W2 that the kernel multiplied G[j0] with (fp16: integers are exact)."""
import sys

import torch

sys.path.insert(0, ".")
from edtr_amd import ops  # noqa: E402


def main():
    d = torch.device("cuda:0")
    dtype = torch.float16
    D, H, M = 320, 1280, 128
    x = torch.zeros((M, D), dtype=dtype)
    w1 = torch.zeros((2 * H, D))
    w2 = torch.arange(H, dtype=torch.float32)[None, :].repeat(D, 1)
    perm = ops.geglu_perm(H)
    w1p = ops.pack_linear_weight(w1[perm], dtype)
    for j0 in [0, 5, 9, 14, 20, 31, 33, 40, 63, 64, 70, 100, 128, 650, 1279]:
        b1 = torch.zeros(2 * H)
        b1[j0], b1[H + j0] = 1.0, 6.0
        c1 = w1p.float().sum(1)
        out = torch.full((M, D), float("nan"), dtype=dtype, device=d)
        ops.launch(ops.make_ffn(dtype=dtype, x=x.to(d), ldx=D, M=M, w1=w1p.to(d), w2=ops.pack_ffn_w2(w2, dtype).to(d),
                                cst=ops.pack_ffn_constants(b1[perm]).to(d), b2=torch.zeros(D, device=d), out=out, ldo=D))
        torch.cuda.synchronize()
        o = out.float().cpu() / 6.0
        vals, counts = torch.unique(o.round(), return_counts=True)
        top = sorted(zip(counts.tolist(), vals.tolist()), reverse=True)[:4]
        # per (token tile, output half): the value seen
        per = o.reshape(4, 32, 2, 160).mean(dim=(1, 3))
        print(f"j0={j0:5d}: values (count, col) {top}   per (token tile x out half): {[[round(float(v), 1) for v in r] for r in per]}")


if __name__ == "__main__":
    main()
