#!/usr/bin/env python3
"""Turn the measured per-class sensitivities (tools/exp/precision_budget_gpu.py) and the per-class launch times at 1 / 2 / 3
parts (`bench.py --precision mixed --breakdown-json` under the three constant policies) into a part count per GEMM class:

    minimise   sum_c time_c(p_c)      subject to   e_base^2 + sum_c var_c(p_c) <= (target)^2    for the latent AND the image

Independent roundings add in quadrature, so var_c(p) = e(c at p, rest at 3)^2 - e_base^2.  Multiple-choice knapsack solved by a
Lagrangian sweep (per class: argmin_p time + lambda * normalised variance) — exact on the convex hull, which is all the
measurement noise supports.  Prints the table for edtr_amd/precision.py and the predicted error / time.

    python tools/exp/precision_allocate.py --sens S.json --times t1.json t2.json t3.json [--target 8e-4]
"""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sens", required=True)
    ap.add_argument("--times", nargs=3, required=True, help="breakdown JSONs of the constant policies 1, 2, 3")
    ap.add_argument("--target", type=float, default=8e-4, help="error budget on latent and image (north-star 1e-3 minus margin)")
    ap.add_argument("--floor", type=float, default=0.0, help="error floor of the image golden's storage, removed in quadrature")
    args = ap.parse_args()
    sens = json.load(open(args.sens))
    times = [json.load(open(p))["by_name"] for p in args.times]
    base = sens["const"]["3"]
    classes = sorted(c for c in sens["classes"] if "@" not in c)       # per-level entries are diagnostics (the levels of a class
    #                                                                        turned out equally sensitive); allocate per class
    budget = {k: max(args.target ** 2 - base[k] ** 2, 1e-12) for k in ("z", "img")}

    def var(c, p):
        if p == 3:
            return 0.0
        e = sens["classes"][c][str(p)]
        return max(max(e[k] ** 2 - base[k] ** 2, 0.0) / budget[k] for k in ("z", "img"))      # normalised: budget = 1

    def t(c, p):
        name = c.split("@")[0]
        ent = times[p - 1].get(name)
        return ent["ms"] if ent else 0.0

    best = None
    lam = 1e-4
    while lam < 1e6:
        pick = {c: min((1, 2, 3), key=lambda p: t(c, p) + lam * var(c, p)) for c in classes}
        v = sum(var(c, pick[c]) for c in classes)
        if v <= 1.0:
            tt = sum(t(c, pick[c]) for c in classes)
            if best is None or tt < best[0]:
                best = (tt, v, dict(pick), lam)
        lam *= 1.15
    tt, v, pick, lam = best
    print(f"target {args.target:.1e}: base (all 3 parts) z {base['z']:.2e} img {base['img']:.2e}; const2 {sens['const']['2']}; const1 {sens['const']['1']}")
    print(f"chosen at lambda {lam:.3g}: predicted normalised variance {v:.2f} of 1.0, GEMM launch time {tt:.1f} ms per pass "
          f"(all-1 {sum(t(c, 1) for c in classes):.1f}, all-2 {sum(t(c, 2) for c in classes):.1f}, all-3 {sum(t(c, 3) for c in classes):.1f})")
    print(f"{'class':28s} parts   ms@1    ms@2    ms@3   var@1  var@2 (fraction of the budget)")
    for c in classes:
        print(f"{c:28s}   {pick[c]}   {t(c, 1):7.2f} {t(c, 2):7.2f} {t(c, 3):7.2f}   {var(c, 1):6.3f} {var(c, 2):6.3f}")
    print("MIXED_TABLE =", json.dumps({c: p for c, p in sorted(pick.items())}, indent=4))


if __name__ == "__main__":
    main()
