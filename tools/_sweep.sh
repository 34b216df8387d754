cd /root/repo
timeout 600 python -m pytest tests/test_dune_labels.py tests/test_frontend.py -x -q -m gpu 2>&1 | tail -15
python - <<'PY'
import sys, time, json, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import CONFIGS, make_oracle
from neupan_amd.dune_labels import dune_labels
from oracle import dune_label_oracle as dl
orc=make_oracle(CONFIGS["diff_1k_T10_K10"]); G=np.asarray(orc.G,np.float64); h=np.asarray(orc.h,np.float64).reshape(-1)
P=np.random.default_rng(0).uniform(-25,25,(100000,2)); Pd=torch.from_numpy(P).cuda()
dune_labels(G,h,Pd); torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): dune_labels(G,h,Pd)
b.record(); torch.cuda.synchronize()
us=a.elapsed_time(b)/20*1e3
t=time.perf_counter(); dl.labels(G,h,P[:2000]); cpu=(time.perf_counter()-t)/2000*1e6
print(json.dumps({"dune_labels":{"points":100000,"us_per_call":round(us,1),"points_per_s":round(1e5/us*1e6),"algorithmic_bytes":100000*(16+16+4),"hbm_GBps":round(3.6e6/us*1e-3,2),"cpu_oracle_us_per_point":round(cpu,1),"reference":"one ECOS solve per point: README '1-2 h' for 100k points"}}))
PY
