"""Debug: with B identical images every activation must be batch-symmetric; find the first launch whose tensors are not."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from edtr_amd import synth, ops
from edtr_amd.testing import build_synthetic_cldm
dev = torch.device("cuda:0")
cldm = build_synthetic_cldm(synth.sd21_config(), dev, torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c_txt = synth.synth_normal("full:c_txt", (1, 77, 1024)).to(dev)
x1 = synth.synth_normal("full:x", (1, 4, 64, 64)).to(dev); c1 = synth.synth_normal("full:c", (1, 4, 64, 64)).to(dev)
xr, cr = x1.expand(B, -1, -1, -1).contiguous(), c1.expand(B, -1, -1, -1).contiguous()
t = torch.full((B,), 200, device=dev)
cldm.forward(xr, t, {"c_txt": c_txt.expand(B, -1, -1).contiguous(), "c_img": cr})
eng = [e for k, e in cldm._cldm_engines.items() if k[0] == B][0]
eng.x_in.copy_(xr); eng.hint_in.copy_(cr); eng.t_in.copy_(t)
s = ops.stream_ptr()
bad = 0
nstat = 0
for i, r in enumerate(eng.step_prog.recs):
    if r.name.endswith(".stats") and nstat < 4:
        torch.cuda.synchronize()
        sm = r.keep[2]
        print(f"rec {i} {r.name}: sums before: absmax {float(sm.abs().max()):.3e} shape {tuple(sm.shape)} ptr_off {(sm.data_ptr() - eng.step_prog.sums_pool.data_ptr())} zeroed_flag {r.keep[0].sums_zeroed} B={r.keep[0].B} HW={r.keep[0].HW} C={r.keep[0].C}", flush=True)
    r.launch(s)
    torch.cuda.synchronize()
    if r.name.endswith(".stats") and nstat < 4:
        sm = r.keep[2]
        print(f"      after: per image group0 {[ [round(v,3) for v in sm[b,0].tolist()] for b in range(sm.shape[0])]}", flush=True)
        nstat += 1
    for j, tns in enumerate(r.keep):
        if not isinstance(tns, torch.Tensor) or tns.dim() != 2 or tns.shape[0] % B or tns.dtype not in (torch.bfloat16, torch.float16, torch.float32):
            continue
        rows = tns.shape[0] // B
        if rows < 1 or tns.shape[0] < B * 8:
            continue
        a = tns[:rows].float()
        for b in range(1, B):
            d = (tns[b * rows:(b + 1) * rows].float() - a).abs().max().item()
            if d > 0 or not torch.isfinite(a).all():
                print(f"rec {i} {r.name} [{r.tag}] keep[{j}] shape {tuple(tns.shape)} image {b} differs from image 0: max abs {d:.3e} finite={bool(torch.isfinite(a).all())}", flush=True)
                bad += 1
                break
    if bad >= 6:
        break
print("done, recs", len(eng.step_prog.recs), "bad", bad)
