#!/usr/bin/env python3
"""How does the CPU oracle (bench.py's cpu_baseline) scale with torch threads on the GPU box's host?  B=1, 512x512, 4 steps.
A script, not a pytest module (it lives under tests/ because only tests/, smoke() and bench.py's cpu_baseline leg may import
oracle/): `python tests/cpu_oracle_threads.py` -> profiles/r02/cpu_oracle_threads.log."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edtr_amd import synth  # noqa: E402
from edtr_amd.testing import synthetic_state_dicts  # noqa: E402
from oracle import edtr_oracle as O  # noqa: E402
from oracle import flat_sd  # noqa: E402

cfg = synth.sd21_config()
t0 = time.time()
sd = flat_sd(synthetic_state_dicts(cfg))
print(f"weights {time.time() - t0:.1f}s, cores {os.cpu_count()}", flush=True)
pre = synth.synth_input("bench:pre_res", (8, 3, 512, 512), 0.0, 1.0)
c_txt = synth.synth_normal("bench:c_txt", (1, 77, 1024))
noises = [synth.synth_normal(f"bench:noise{i}", (8, 4, 64, 64)) for i in range(5)]
for threads, B in ((32, 1), (64, 1), (128, 1), (os.cpu_count(), 1), (64, 8), (os.cpu_count(), 8)):
    torch.set_num_threads(threads)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.restore(sd, cfg, O.make_betas(), pre[:B], c_txt, [n[:B] for n in noises], [50, 100, 150, 200], 200)
        dt = time.perf_counter() - t0
    print(f"threads {threads:4d}  B={B}: {dt:7.1f} s  = {B / dt:.4f} images/s", flush=True)
