# Round 6, second pass: three parts on every DENOISER class (what the first pass says the denoiser needs on outlier-bearing weights: the
# weights-exact two-part product on its linears leaves eps at 1.2e-3) with the VAE on the shipped allocation (its figures are inside 1e-3
# already) — accuracy on the moderate set, then throughput + parity on the bench workload (smooth set).
run() { echo "== $1"; shift; env "$@" python -m pytest tests/test_gpu_heavy.py -m gpu -q -s -k "moderate and mixed" 2>&1 | grep "moderate, mixed" | sed 's/.*block\] //'; }
VAE1='"vae.conv1":1,"vae.conv2":1,"vae.attn.qk":1,"vae.attn.vT":1,"vae.attn.proj_out":1,"vae.attn.flash":1'
run denoiser3_vae_shipped "EDTR_AMD_POLICY={\"base\":\"shipped\",\"default\":3,$VAE1}"
run denoiser3_split1_vae_shipped "EDTR_AMD_POLICY={\"base\":\"shipped\",\"default\":3,$VAE1}" EDTR_AMD_ATTN_SPLIT=1
run denoiser3_vae_conv2 "EDTR_AMD_POLICY={\"base\":\"shipped\",\"default\":3,\"vae.conv1\":2,\"vae.conv2\":2,\"vae.attn.qk\":1,\"vae.attn.vT\":1,\"vae.attn.proj_out\":1}"
for pol in "{\"base\":\"shipped\",\"default\":3,$VAE1}" "{\"default\":3}"; do
  echo "== bench det512 --precision mixed, policy $pol"
  EDTR_AMD_POLICY="$pol" python bench.py --precision mixed --steps 16 --also none --no-cpu-baseline --no-roofline --parity-steps 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=j['parity_vs_reference_golden']; print(j['value'], j['ms_per_step'], g['rel_err_latent'], g['rel_err_image_samples'])"
done
