"""edtr_ffn diagnostics (used while bringing the kernel up): where — which tokens / output columns / hidden chunks — does the fused launch
differ from the fp32 reference?  Selection matrices as W2 expose G itself; W1 = 0 isolates the second product; `-DFFN_DRAIN` (every counted
wait drains the DMA queue; build as in tools/exp/ffn_stamps.py, load through EDTR_AMD_LIB) tells a timing bug from a data-path bug."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from edtr_amd import ops  # noqa: E402


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def run(x16, gamma, beta, w1, b1, w2, b2, dtype, d):
    D, H = 320, 1280
    perm = ops.geglu_perm(H)
    w1p = ops.pack_linear_weight(w1[perm] * gamma[None, :], dtype)
    c1 = w1p.float().sum(1)
    c2b = w1[perm] @ beta + b1[perm]
    out = torch.full(x16.shape, float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_ffn(dtype=dtype, x=x16.to(d), ldx=D, M=x16.shape[0], w1=w1p.to(d), w2=ops.pack_ffn_w2(w2, dtype).to(d),
                            cst=ops.pack_ffn_constants(c2b).to(d), b2=b2.to(d), out=out, ldo=D))
    torch.cuda.synchronize()
    xf = x16.float()
    ln = F.layer_norm(xf, (D,), gamma, beta, 1e-5)
    hcat = ln @ w1.t() + b1
    val, gate = hcat.chunk(2, dim=-1)
    G = val * F.gelu(gate)
    ref = xf + G @ w2.t() + b2
    return out.float().cpu(), ref, G


def summarize(tag, got, ref):
    err = (got - ref).abs()
    scale = ref.abs().mean().item()
    M, D = got.shape
    print(f"--- {tag}: rel L2 {float((got - ref).norm() / ref.norm()):.3e}   mean|ref| {scale:.3f}")
    e = err.reshape(M // 128, 4, 32, D).mean(dim=(0, 3))            # [token tile t][l31]
    print("  by token tile:", [f"{v:.3f}" for v in e.mean(1).tolist()])
    print("  by l31 (first 8):", [f"{v:.3f}" for v in e.mean(0)[:8].tolist()])
    ec = err.mean(0)                                                  # per output column
    print("  by output half:", [f"{ec[:160].mean():.3f}", f"{ec[160:].mean():.3f}"])
    print("  by 32-col block:", [f"{ec[32 * b:32 * b + 32].mean():.3f}" for b in range(10)])
    print("  by col mod 16  :", [f"{ec.reshape(20, 16).mean(0)[i]:.3f}" for i in range(16)])


def gmap(x, gamma, beta, w1, b1, dtype, d):
    """G itself, 320 gated units per run through a selection matrix as W2; error per (chunk, half, hidden-local)."""
    D, H = 320, 1280
    errs = torch.zeros(H)
    refn = torch.zeros(H)
    for r in range(4):
        w2s = torch.zeros((D, H))
        w2s[torch.arange(D), 320 * r + torch.arange(D)] = 1.0
        got, ref, G = run(x, gamma, beta, w1, b1, w2s, torch.zeros(D), dtype, d)
        errs[320 * r:320 * r + 320] = ((got - x.float()) - G[:, 320 * r:320 * r + 320]).abs().mean(0)
        refn[320 * r:320 * r + 320] = G[:, 320 * r:320 * r + 320].abs().mean(0)
    e = (errs / refn.mean()).reshape(20, 2, 32)
    print("  G error by chunk (rows) x half:")
    for c in range(20):
        print(f"    c={c:2d}  h0 {e[c, 0].mean():.3f}  h1 {e[c, 1].mean():.3f}   h0 by hidden-local/8: {[round(float(v), 2) for v in e[c, 0].reshape(4, 8).mean(1)]}")


def main():
    d = torch.device("cuda:0")
    dtype = torch.bfloat16
    D, H, M = 320, 1280, 256
    x = (rnd((M, D), 1, 1.5) + 0.5).to(dtype)
    gamma, beta = 1 + 0.2 * rnd((D,), 2), 0.3 * rnd((D,), 3)
    w1 = rnd((2 * H, D), 4, 1 / math.sqrt(D))
    b1 = 0.5 * rnd((2 * H,), 5)
    w2 = rnd((D, H), 6, 1 / math.sqrt(H))
    b2 = 0.5 * rnd((D,), 7)
    got, ref, G = run(x, gamma, beta, w1, b1, w2, b2, dtype, d)
    summarize("full", got, ref)
    gmap(x, gamma, beta, w1, b1, dtype, d)
    print("  ... with W1 = 0 (constants only):")
    gmap(x, gamma, beta, torch.zeros_like(w1), b1, dtype, d)
    # (1) second product = identity on the first 320 gated units: out - x - b2 = G[:, :320]
    w2i = torch.zeros((D, H))
    w2i[torch.arange(D), torch.arange(D)] = 1.0
    got, ref, G = run(x, gamma, beta, w1, b1, w2i, torch.zeros(D), dtype, d)
    summarize("W2 = identity on gated 0..319 (shows G itself)", got - x.float(), ref - x.float())
    # (1b) identity on gated units 640..959
    w2j = torch.zeros((D, H))
    w2j[torch.arange(D), 640 + torch.arange(D)] = 1.0
    got, ref, G = run(x, gamma, beta, w1, b1, w2j, torch.zeros(D), dtype, d)
    summarize("W2 = identity on gated 640..959", got - x.float(), ref - x.float())
    # (2) W1 = 0: G is a constant vector (from b1): out = x + W2 g + b2 — the second product alone
    got, ref, G = run(x, gamma, beta, torch.zeros_like(w1), b1, w2, b2, dtype, d)
    summarize("W1 = 0 (second product alone)", got - x.float(), ref - x.float())
    # (3) no LayerNorm effect: gamma 1, beta 0, rows already normalised
    xn = F.layer_norm(rnd((M, D), 9), (D,)).to(dtype)
    got, ref, G = run(xn, torch.ones(D), torch.zeros(D), w1, b1, w2, b2, dtype, d)
    summarize("pre-normalised rows, gamma 1 beta 0", got - xn.float(), ref - xn.float())


if __name__ == "__main__":
    main()
