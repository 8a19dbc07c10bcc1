"""Round 6, VERDICT r05 next 3a: which SECTIONS of the path need the mixed mode's fp32 stream for 1e-3 on the smooth weight set?
Runs bench.py (BASELINE configs[1], 24 timed passes, no side legs) once per section table of the hybrid mode and prints
throughput + errors against the reference golden (tests/golden/full_det512.npz).  One device: the rows are comparable.
    python tools/exp/r06_hybrid_sweep.py [> gpurun_out/r06/hybrid_sweep.log]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROWS = [
    ("bf16 fast (headline)", ["--precision", "fast", "--dtype", "bf16"], None, None),
    ("fp16 fast", ["--precision", "fast", "--dtype", "fp16"], None, None),
    ("hybrid: cldm fp16 | enc mixed | dec mixed", ["--precision", "hybrid"], {}, None),
    ("hybrid: cldm fp16 | enc fp16  | dec mixed", ["--precision", "hybrid"], {"vae.encode": "fast16"}, None),
    ("hybrid: cldm fp16 | enc mixed | dec fp16", ["--precision", "hybrid"], {"vae.decode": "fast16"}, None),
    ("hybrid: cldm mixed | enc mixed | dec fp16", ["--precision", "hybrid"], {"cldm": "mixed", "vae.decode": "fast16"}, None),
    ("hybrid, VAE carriers only at 3 (rest 1) + enc branch fp32", ["--precision", "hybrid"], {}, {"EDTR_AMD_BRANCH16_ENC": "0"}),
    ("mixed (shipped)", ["--precision", "mixed"], None, None),
]


def main():
    only = sys.argv[1:]
    print(f"{'configuration':62s} {'img/s':>8s} {'latent':>10s} {'image':>10s} {'max lat':>10s} {'max img':>10s}")
    for label, flags, sections, extra_env in ROWS:
        if only and not any(o in label for o in only):
            continue
        env = dict(os.environ)
        if sections is not None:
            env["EDTR_AMD_HYBRID"] = json.dumps(sections)
        env.update(extra_env or {})
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "2", "--also", "none", "--no-cpu-baseline",
               "--no-roofline", "--parity-steps", "0"] + flags
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{label:62s} FAILED rc={r.returncode}: {r.stderr[-400:]!r}", flush=True)
            continue
        j = json.loads(line[-1])
        g = j.get("parity_vs_reference_golden", {})
        print(f"{label:62s} {j['value']:8.2f} {g.get('rel_err_latent', float('nan')):10.3e} {g.get('rel_err_image_samples', float('nan')):10.3e} "
              f"{g.get('max_err_latent', float('nan')):10.3e} {g.get('max_err_image_samples', float('nan')):10.3e}", flush=True)


if __name__ == "__main__":
    main()
