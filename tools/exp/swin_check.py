#!/usr/bin/env python3
"""SwinIR on the GPU vs the reference goldens (tests/golden/swinir.npz); prints relative errors and a timing.
Run on the GPU box:  python3 tools/exp/swin_check.py [small|full|time]..."""
import os
import sys
import time

t_start = time.time()
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from edtr_amd import synth  # noqa: E402
from edtr_amd.model.swinir import SwinIR  # noqa: E402

print(f"imports {time.time() - t_start:.1f} s", flush=True)
g = np.load(os.path.join(ROOT, "tests", "golden", "swinir.npz"))


def rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(np.asarray(b, dtype=np.float64))
    return float((a - b).norm() / b.norm())


def build(tag, cfg, dtype):
    m = SwinIR(**cfg)
    sd = m.state_dict()
    m.load_state_dict({k: (synth.synth_param(f"swinir{tag}." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                       for k, v in sd.items()}, strict=True)
    m.eval().to("cuda")
    m.compute_dtype = dtype
    return m


todo = sys.argv[1:] or ["small", "full"]
for what in todo:
    try:
        t0 = time.time()
        if what == "small":
            x = synth.synth_input("swinir:small", (2, 3, 128, 192), 0.0, 1.0).cuda()
            for dt in (torch.float16, torch.bfloat16):
                m = build("small", synth.swinir_small_config(), dt)
                for graph in (False, True):
                    m.use_graph = graph
                    m._engines.clear()
                    y = m(x)
                    print(f"small {dt} graph={graph}: rel {rel(y, g['y_small']):.3e} max {float((y.cpu() - torch.from_numpy(g['y_small'])).abs().max()):.3e}", flush=True)
        elif what == "full":
            for dt in (torch.float16, torch.bfloat16):
                m = build("full", synth.swinir_config(), dt)
                y = m(synth.synth_input("swinir:256", (1, 3, 256, 256), 0.0, 1.0).cuda())
                print(f"full256 {dt}: rel {rel(y, g['y_256'].astype(np.float32)):.3e}", flush=True)
                y = m(synth.synth_input("swinir:512", (1, 3, 512, 512), 0.0, 1.0).cuda())
                print(f"full512 {dt}: rel(stride8) {rel(y[:, :, 3::8, 5::8], g['y_512_stride8']):.3e} mean {float(y.mean()):.5f} (ref {g['y_512_stats'][0]:.5f})", flush=True)
        elif what == "time":
            m = build("full", synth.swinir_config(), torch.bfloat16)
            x = torch.rand(8, 3, 512, 512, device="cuda")
            m(x)
            eng = next(iter(m._engines.values()))
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(10):
                eng.prog.run()
            torch.cuda.synchronize()
            ms = (time.time() - t1) * 100
            print(f"full B=8 512^2 bf16: {ms:.2f} ms per batch ({8000 / ms:.1f} images/s), {eng.prog.total_flops() / 1e12:.3f} TFLOP per batch, "
                  f"{eng.prog.total_flops() / ms / 1e9:.1f} TFLOP/s, {len(eng.prog.recs)} launches", flush=True)
        elif what == "breakdown":
            m = build("full", synth.swinir_config(), torch.bfloat16)
            m.use_graph = False
            m(torch.rand(8, 3, 512, 512, device="cuda"))
            eng = next(iter(m._engines.values()))
            eng.prog.run_timed()
            rows = {}
            for _ in range(3):
                for name, ms, fl, nb, tag in eng.prog.run_timed():
                    r = rows.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
                    r[0] += 1; r[1] += ms; r[2] += fl; r[3] += nb
            tot = sum(r[1] for r in rows.values()) / 3
            print(f"eager per-launch timing, B=8 512^2 bf16: {tot:.2f} ms per batch (event-bracketed launches, no graph)")
            for (name, tag), r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                ms = r[1] / 3
                print(f"  {name:28s} {tag:44s} n={r[0] // 3:3d}  {ms:7.3f} ms  {r[2] / 3 / ms / 1e9 if ms else 0:7.1f} TFLOP/s  {r[3] / 3 / ms / 1e6 if ms else 0:7.1f} GB/s")
        print(f"{what}: {time.time() - t0:.1f} s", flush=True)
    except Exception as e:
        import traceback
        traceback.print_exc()
        print("ERROR in", what, repr(e), flush=True)
print(f"total {time.time() - t_start:.1f} s")
