#!/bin/bash
# alternate two (or more) builds of the library on ONE box: bash tests/tools/ab_libs.sh tag rounds libA libB ...
tag=$1; rounds=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    n=$(basename $lib .so)
    NPA_LIB_PATH=$PWD/$lib timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --no-latency > $out/drv_${n}_$r.json 2>/dev/null
    NPA_LIB_PATH=$PWD/$lib timeout 200 python bench.py --no-cpu --no-extras --no-latency > $out/def_${n}_$r.json 2>/dev/null
    NPA_LIB_PATH=$PWD/$lib timeout 100 python bench.py --inflight 1 --no-cpu --no-latency --no-extras --steps 32 --warmup 8 > $out/inf_${n}_$r.json 2>/dev/null
  done
done
python - <<P
import json,glob,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
        k="_".join(f.split("/")[-1].split("_")[:-1]); res[k].append((d["value"], d["roofline"]["launch_ms"]))
    except Exception as e: print(f,"ERR",e)
for k,v in sorted(res.items()):
    vals=sorted(x[0] for x in v); print(f"{k:30s} median {vals[len(vals)//2]:10.1f}  all {[round(x) for x in vals]}  qp_ms {[x[1] for x in v]}")
P
