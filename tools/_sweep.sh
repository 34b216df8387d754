cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o kt -- python $R/bench.py --no-cpu --steps 10 --warmup 5 > /dev/null 2>&1
f=$(find $R/gpurun_out/ktrace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, statistics
rows=list(csv.DictReader(open("$f")))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("void ","")[:14],r.get("Queue_Id")) for r in rows]
ev.sort()
d=[e for e in ev if "dune" in e[2]]
d=d[-100:]
gaps=[d[i+1][0]-d[i][1] for i in range(len(d)-1)]
print("last 100 dune kernels: median dur %.1f us, gaps sorted(us):"%(statistics.median([e[1]-e[0] for e in d])/1e3), [round(g/1e3) for g in sorted(gaps)][::5])
big=max(range(len(gaps)), key=lambda i:gaps[i])
t0=d[big][0]
lo=d[max(big-2,0)][0]; hi=d[min(big+4,len(d)-1)][1]
for e in ev:
    if lo<=e[0]<=hi: print("%9.1f %9.1f  %-14s q=%s"%((e[0]-t0)/1e3,(e[1]-t0)/1e3,e[2],e[3]))
PY
rm -rf $R/gpurun_out/ktrace
