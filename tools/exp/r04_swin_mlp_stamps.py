"""Reads the s_memtime stamps of a -DMLP_STAMPS build of edtr_swin_mlp (temporary diagnostic; EDTR_AMD_LIB points at that build)."""
import ctypes, math, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edtr_amd import ops, lib as L
d = torch.device("cuda:0")
rows, dtype = 32768, torch.bfloat16
CP, HP = ops.SWIN_MLP_C, ops.SWIN_MLP_HIDDEN
g = torch.Generator().manual_seed(0)
x = torch.randn((rows, CP), generator=g).to(dtype).to(d)
w1g = torch.randn((HP, CP), generator=g) / math.sqrt(CP)
w2 = torch.randn((CP, HP), generator=g) / math.sqrt(HP)
i1, i2 = ops.pack_swin_mlp_weights(w1g, w2, dtype)
c1 = w1g.to(dtype).float().sum(1).contiguous().to(d)
c2b = torch.randn(HP, generator=g).to(d); b2 = torch.randn(CP, generator=g).to(d)
out = torch.empty((rows, CP), dtype=dtype, device=d)
stats = torch.empty((rows, CP // 32, 2), dtype=torch.float32, device=d)
rec = ops.make_swin_mlp(dtype=dtype, x=x, ldx=CP, rows=rows, c_valid=180, eps=1e-5, w1=i1.to(d), w2=i2.to(d), c1=c1, c2b=c2b, b2=b2, out=out, ldo=CP, row_stats=stats)
for _ in range(5):
    ops.launch(rec)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib = L.load()
lib.edtr_mlp_dbg.argtypes = [ctypes.c_void_p]
assert lib.edtr_mlp_dbg(buf) == 0
names = {0: "start", 1: "dma issued"}
for t in range(14):
    names[2 + 3 * t] = f"p{t} dma waited"; names[3 + 3 * t] = f"p{t} post-barrier"; names[4 + 3 * t] = f"p{t} work done"
names.update({50: "loop end", 51: "post-barrier", 52: "handed over", 53: "post-barrier", 54: "finished", 55: "post-barrier", 56: "stores issued", 57: "stores done"})
for w in range(4):
    v = [buf[w * 64 + k] for k in range(64)]
    print(f"--- block {'0' if w < 2 else '200'} wave {'0' if w % 2 == 0 else '4'} (s_memtime ticks, 100 MHz = 10 ns)")
    prev = v[0]
    for k in sorted(names):
        print(f"  {names[k]:18s} +{v[k] - v[0]:6d}  (d {v[k] - prev:5d})")
        prev = v[k]
