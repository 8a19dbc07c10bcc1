#!/usr/bin/env python3
"""BASELINE.md §3's B = 8 CPU point, measured once: the whole bench batch of 8 images through the CPU oracle on
ALL host cores — as P concurrent B = 1 oracle processes of T torch threads each (P x T = the host's hardware threads), because ONE
oracle call on the whole batch does not scale past 32 threads (profiles/r02/cpu_oracle_threads.log: 18.3 s per image at 32 threads,
29.9 s at 64, 50.7 s at 128) and did not finish inside bench.py's 420 s budget.  A script, not a pytest module (it lives under
tests/ because only tests/, smoke() and bench.py's cpu_baseline leg may import oracle/):

    python tests/cpu_oracle_b8.py [P] [T]    ->  profiles/r04/cpu_oracle_b8.log
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(idx, T, q_ready, q_go, q_done):
    import torch
    torch.set_num_threads(min(T, 32))
    from edtr_amd import synth
    from edtr_amd.testing import synthetic_state_dicts
    from oracle import edtr_oracle as O
    from oracle import flat_sd
    cfg = synth.sd21_config()
    sd = flat_sd(synthetic_state_dicts(cfg))
    pre = synth.synth_input("bench:pre_res", (8, 3, 512, 512), 0.0, 1.0)[idx:idx + 1]
    c_txt = synth.synth_normal("bench:c_txt", (1, 77, 1024))
    noises = [synth.synth_normal(f"bench:noise{i}", (8, 4, 64, 64))[idx:idx + 1] for i in range(5)]
    torch.set_num_threads(T)
    q_ready.put(idx)
    q_go.get()
    t0 = time.perf_counter()
    with torch.no_grad():
        O.restore(sd, cfg, O.make_betas(), pre, c_txt, noises, [50, 100, 150, 200], 200)
    q_done.put((idx, time.perf_counter() - t0))


def main():
    cores = os.cpu_count() or 8
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, cores // P)
    ctx = mp.get_context("spawn")
    q_ready, q_done = ctx.Queue(), ctx.Queue()
    gos = [ctx.Queue() for _ in range(P)]
    procs = [ctx.Process(target=worker, args=(i, T, q_ready, gos[i], q_done), daemon=True) for i in range(P)]
    t0 = time.time()
    for p in procs:
        p.start()
    for _ in range(P):
        q_ready.get(timeout=1500)
    print(f"# {P} oracle processes x {T} threads ready after {time.time() - t0:.0f} s (weight hashing); host has {cores} hardware threads", flush=True)
    t0 = time.perf_counter()
    for g in gos:
        g.put(1)
    per = [q_done.get(timeout=1500) for _ in range(P)]
    wall = time.perf_counter() - t0
    for p in procs:
        p.join(timeout=30)
    print("per-image seconds: " + " ".join(f"{dt:.1f}" for _, dt in sorted(per)))
    print(f"B = {P} (the bench batch, images 0..{P - 1}, 512x512, 4 steps, fp32 oracle): {wall:.1f} s wall = {P / wall:.4f} images/s on {P * T} threads")


if __name__ == "__main__":
    main()
