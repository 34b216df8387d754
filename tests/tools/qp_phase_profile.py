import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_batch
cfg=CONFIGS["diff_1k_T10_K10"]; B=256
pan=make_gpu_pan(cfg)
batch=make_batch(cfg,0,B)
st=pan.dune_stage(batch["nom_s"], batch["points"])
out=pan.nrmp_stage(batch["nom_s"], batch["nom_u"], batch["ref_s"], batch["ref_us"], st)
info=out["info"].cpu().numpy()
print("iters run mean %.2f min %d max %d"%(info[:,4].mean(), info[:,4].min(), info[:,4].max()), "best merit max %.1e"%info[:,1].max())
names=["setup","resid","kkt_y_h","kkt_acc","chol","pass_w","solve","dirs","update","tail"]
tot=info[:,5:15].mean(0)
for n,v in zip(names,tot): print("%-8s %9.0f cyc  %5.1f%%  per-iter %7.0f"%(n,v,100*v/tot.sum(), v/info[:,4].mean()))
print("total", tot.sum())
