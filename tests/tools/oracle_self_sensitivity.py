"""CPU study (no GPU): how much the ORACLE's own control output moves when every obstacle coordinate of its input
is changed by one float32 ulp -- the yardstick for the GPU-vs-oracle deviations on scenes whose PAN iteration does not
contract within K = 10.  ~5 minutes on 8 cores.

    python tests/tools/oracle_self_sensitivity.py
"""
import sys, numpy as np, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_scene
from concurrent.futures import ProcessPoolExecutor
cfg=CONFIGS["diff_1k_T10_K10"]
def job(b):
    sc=make_scene(cfg,b)
    o=make_oracle(cfg); s,u,d=o.forward(sc["nom_s"],sc["nom_u"],sc["ref_s"],sc["ref_us"],sc["points"],None)
    mv=float(np.linalg.norm(o.trace[-1][1]-o.trace[-2][1]))
    pts=np.nextafter(sc["points"], np.float32(np.inf))      # every coordinate moved by one float32 ulp
    o2=make_oracle(cfg); s2,u2,d2=o2.forward(sc["nom_s"],sc["nom_u"],sc["ref_s"],sc["ref_us"],pts,None)
    return b, mv, float(np.linalg.norm(u-u2))
if __name__=="__main__":
    with ProcessPoolExecutor(8) as ex: res=list(ex.map(job, range(96)))
    res=np.array(res)
    mv, ss = res[:,1], res[:,2]
    con = mv<=1.0
    print("contracting scenes:", con.sum(), " self-sensitivity to a 1-ulp input change: median %.2e max %.2e"%(np.median(ss[con]), ss[con].max()))
    print("non-contracting   :", (~con).sum(), " self-sensitivity: median %.2e max %.2e"%(np.median(ss[~con]), ss[~con].max()))
