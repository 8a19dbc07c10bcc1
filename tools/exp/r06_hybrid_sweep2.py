"""Round 6: the DECODER's policy inside the hybrid mode (denoiser + encoder fp16 fast, decoder mixed): which of its stream carriers need
three parts?  Same harness as r06_hybrid_sweep.py; rows differ in EDTR_AMD_POLICY only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SEC = {"vae.encode": "fast16"}
ROWS = [
    ("dec shipped table (carriers 3)", {"base": "shipped"}),
    ("dec upsample.conv 2", {"base": "shipped", "vae.upsample.conv": 2}),
    ("dec upsample.conv 1", {"base": "shipped", "vae.upsample.conv": 1}),
    ("dec all carriers 2", {"base": "shipped", "vae.upsample.conv": 2, "vae.conv_in": 2, "vae.conv_out": 2, "vae.nin_shortcut": 2, "vae.post_quant_conv": 2}),
    ("dec all 1 (fp32 stream only)", {"default": 1}),
    ("dec upsample 1, conv_out 3, rest 1", {"default": 1, "vae.conv_out": 3}),
    ("dec upsample 3, rest 1", {"default": 1, "vae.upsample.conv": 3}),
]


def main():
    only = sys.argv[1:]
    print(f"{'decoder policy (cldm fp16 | enc fp16 | dec mixed)':62s} {'img/s':>8s} {'latent':>10s} {'image':>10s} {'max lat':>10s} {'max img':>10s}")
    for label, pol in ROWS:
        if only and not any(o in label for o in only):
            continue
        env = dict(os.environ, EDTR_AMD_HYBRID=json.dumps(SEC), EDTR_AMD_POLICY=json.dumps(pol))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "2", "--also", "none", "--no-cpu-baseline",
               "--no-roofline", "--parity-steps", "0", "--precision", "hybrid"]
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
