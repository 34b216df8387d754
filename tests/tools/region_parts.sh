#!/bin/bash
# Round 6: where the driver's 20-step region spends its time, plain process against one rank under torch.distributed.run, alternating
# (bench.py's region_ms: issuing the steps / waiting for the device / the closing barrier):
#     gpurun --timeout 900 -- 'bash tests/tools/region_parts.sh 24'     -> gpurun_out/r06region/*.json + summary on stdout
n=${1:-24}
out=gpurun_out/r06region; rm -rf $out; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in $(seq 1 $n); do
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --no-extras 2>/dev/null | tail -1 > $out/tr_$i.json
  if [ $((i % 3)) = 0 ]; then timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --no-latency 2>/dev/null | tail -1 > $out/plain_$i.json; fi
done
python - <<P
import json,glob
for kind in ("tr","plain"):
    rows=[]
    for f in sorted(glob.glob("$out/%s_*.json"%kind), key=lambda f:int(f.split("_")[-1].split(".")[0])):
        try:
            d=json.loads(open(f).read().strip().split("\n")[-1]); r=d["region_ms"]; rows.append((d["value"], d["ms_per_step"]*d["steps"], r["issue"], r["drain"], r["closing_barrier"]))
        except Exception as e: print(f,"ERR",e)
    print(kind, "plans/s | region ms | issue | drain | closing barrier")
    for x in rows: print("   %9.0f  %6.3f  %6.3f  %6.3f  %6.3f" % x)
P
