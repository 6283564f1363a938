cd /tmp && export TMPDIR=/tmp
for fu in 0 1; do
rm -rf /tmp/kt$fu
TA3N_FUSED_UPDATE=$fu rocprofv3 --kernel-trace -d /tmp/kt$fu -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --skip-cpu-baseline --single-dtype > /dev/null 2>&1
f=$(find /tmp/kt$fu -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,statistics,collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
# take the 110-step region: find the longest run of kernels with gaps < 1 ms
reg=[];cur=[]
for s,e,n in rows:
    if cur and s-cur[-1][1]>1_000_000: reg.append(cur);cur=[]
    cur.append((s,e,n))
reg.append(cur)
best=max(reg,key=len)
def short(n):
    for k in ("sgd_range","sgd_fixup","heads_kernel","set_hyper","copyBuffer"):
        if k in n: return k
    if "gemm_tiles" in n: return "gemm<"+n[n.index("<")+1:n.index(">")].replace(" ","")+">"
    return n[:30]
d=collections.defaultdict(list)
seq=[short(n) for s,e,n in best]
# per-kernel-name positional stats: group by name, report count & median duration
for s,e,n in best: d[short(n)].append((e-s)/1e3)
for k,v in d.items(): print(f"   {k:28s} x{len(v):4d} median {statistics.median(v):7.2f} us  total {sum(v):9.1f}")
print("   region wall us:",(best[-1][1]-best[0][0])/1e3, "kernels", len(best))
PY
done
