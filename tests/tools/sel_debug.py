import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan
from neupan_amd.scenes import make_batch
cfg=CONFIGS[sys.argv[1] if len(sys.argv)>1 else "diff_1k_T10_K10"]; pan=make_gpu_pan(cfg); batch=make_batch(cfg,0,64 if len(sys.argv)>1 else 256)
print("key mode", pan.key_mode())
r=pan.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
c=r["count"].cpu().numpy()
nc=(c>>8)&0xff; fb=(c>>16)&1
print("ncand histogram:", np.bincount(nc.reshape(-1)).tolist()); print("fallbacks:", int(fb.sum()), "of", fb.size)
d=r["dist"].cpu().numpy()
idx=np.argwhere(nc>12)
for b,t in idx[:12]:
    print("scene",b,"slice",t,"ncand",nc[b,t],"fallback",fb[b,t],"dist rows:",np.round(d[b,t,:],6).tolist())
