import sys, os, torch
sys.path.insert(0, '/root/repo')
from edtr_amd import synth
from edtr_amd.testing import build_synthetic_cldm, rel_err
dev = torch.device("cuda:0")
cldm = build_synthetic_cldm(synth.sd21_config(), dev, torch.bfloat16)
c_txt = synth.synth_normal("full:c_txt", (1, 77, 1024)).to(dev)
BM = 4
xs = synth.synth_normal("full:x", (BM, 4, 64, 64)).to(dev)
cs = synth.synth_normal("full:c", (BM, 4, 64, 64)).to(dev)
t1 = torch.full((1,), 200, device=dev)
e1 = cldm.forward(xs[:1], t1, {"c_txt": c_txt, "c_img": cs[:1].contiguous()}).clone()
out = {"e1": e1.cpu()}
for B in (2, 3, 4):
    e = cldm.forward(xs[:B].contiguous(), t1.expand(B).contiguous(), {"c_txt": c_txt.expand(B, -1, -1).contiguous(), "c_img": cs[:B].contiguous()})
    out[f"e{B}"] = e.cpu().clone()
    print(f"B={B}: vs B1 image0 {rel_err(e[:1], e1):.1e}", flush=True)
    # same image replicated B times
    xr, cr = xs[:1].expand(B, -1, -1, -1).contiguous(), cs[:1].expand(B, -1, -1, -1).contiguous()
    er = cldm.forward(xr, t1.expand(B).contiguous(), {"c_txt": c_txt.expand(B, -1, -1).contiguous(), "c_img": cr})
    print(f"B={B} replicated: " + " ".join(f"{rel_err(er[i:i+1], e1):.1e}" for i in range(B)), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
torch.save(out, "gpurun_out/batch_inv.pt")
