"""Per-kernel / per-grid attribution of one rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE) pass over bench.py (igemm family only):
    python3 tools/exp/pmc_by_kernel.py <rocprof dir> FETCH_SIZE
3 passes of the path are assumed (program build, warm-up, timed)."""
import csv,glob,sys,collections,json
root=sys.argv[1]
files=sorted(glob.glob(root+"/**/*counter_collection.csv",recursive=True))
agg=collections.defaultdict(lambda:[0,0.0])
for f in files:
    rd=csv.reader(open(f)); h=next(rd)
    kn,cn,cv,gs=h.index("Kernel_Name"),h.index("Counter_Name"),h.index("Counter_Value"),h.index("Grid_Size")
    for r in rd:
        n=r[kn]
        if "igemm" not in n and "splitk" not in n and "ffn320" not in n and "lin320" not in n: continue
        short=n.replace("void (anonymous namespace)::","").replace("(edtr_igemm_params)","").replace("(edtr_ffn_params)","").replace("(edtr_lin320_params)","").replace("(anonymous namespace)::","")[:60]
        key=(short,r[gs])
        agg[key][0]+=1; agg[key][1]+=float(r[cv])
rows=sorted(agg.items(),key=lambda kv:-kv[1][1])
tot=sum(v[1] for v in agg.values())
print("total",sys.argv[2],"KiB",tot)
for (k,g),(n,v) in rows[:40]:
    print(f"{k:60s} grid {g:>9s} n={n:5d} {v*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)/n/1e6:10.1f} MB/launch  {v*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)/3/1e9:8.2f} GB/pass")
