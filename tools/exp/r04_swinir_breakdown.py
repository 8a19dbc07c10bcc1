"""Per-launch HIP-event times of the SwinIR program (B = 8, 512^2), summed by launch name."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edtr_amd import synth
from edtr_amd.model.swinir import SwinIR

dev = torch.device("cuda:0")
dtype = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] == "bf16" else torch.float16
B = 8
m = SwinIR(**synth.swinir_config())
sd = m.state_dict()
m.load_state_dict({k: (synth.synth_param("swinirfull." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                   for k, v in sd.items()}, strict=True)
m = m.eval().to(dev)
m.compute_dtype = dtype
x = synth.synth_input("bench:lq", (B, 3, 512, 512), 0.0, 1.0).to(dev)
m(x)
eng = next(iter(m._engines.values()))
prog = eng.prog
g, prog.graph = prog.graph, None
prog.run_timed()
rows = prog.run_timed()
prog.graph = g
agg = {}
for name, ms, flops, nbytes, tag in rows:
    a = agg.setdefault(name, [0.0, 0, 0.0])
    a[0] += ms; a[1] += 1; a[2] += flops
tot = sum(a[0] for a in agg.values())
print(f"sum of launch durations {tot:.3f} ms, {len(rows)} launches")
for name, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:28s} {ms:7.3f} ms  n={n:4d}  {ms / n * 1e3:7.1f} us each  {fl / ms / 1e9 if ms else 0:7.1f} TFLOP/s")
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(30):
    prog.run()
torch.cuda.synchronize()
print(f"graph replay {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per batch of {B}")
