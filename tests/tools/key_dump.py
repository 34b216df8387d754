"""GPU diagnostic: dump the distance keys of PAN iteration 0 twice and report where they differ."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan
from neupan_amd.scenes import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = CONFIGS["diff_1k_T10_K10"]
pan = make_gpu_pan(cfg)
batch = make_batch(cfg, 1000, B)
args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
T = cfg.T
KS = 1024

def keys():
    pan.reset_stop_state()
    pan.forward_begin(*args)
    pan.forward_iter(0)
    torch.cuda.synchronize()
    ws = pan._ws.view(torch.int32)
    k = ws[-B * (T + 1) * KS:].clone().cpu().numpy().reshape(B, T + 1, KS)[:, :, :1000]
    for kk in range(1, cfg.iter_num):
        pan.forward_iter(kk)
    pan.forward_end()
    torch.cuda.synchronize()
    return k

def tofloat(k):
    u = k.view(np.uint32).astype(np.uint32)
    neg = (u & 0x80000000) == 0
    bits = np.where(neg, ~u, u & 0x7fffffff).astype(np.uint32)
    return bits.view(np.float32)

k0 = keys()
if len(sys.argv) > 2:
    np.save(sys.argv[2], k0)
ex = np.load(sys.argv[3]) if len(sys.argv) > 3 else None
if ex is not None:
    f0, fe = tofloat(k0), tofloat(ex)
    err = np.abs(f0 - fe)
    print("vs exact keys: max err", err.max(), " count err>1e-4:", int((err > 1e-4).sum()))
    bad = np.argwhere(err > 1e-4)
    if len(bad):
        b, t, n = bad[0]
        n0 = n - n % 32
        print("tile", b, t, n0 // 32)
        print(" split:", f0[b, t, n0:n0 + 32])
        print(" exact:", fe[b, t, n0:n0 + 32])
        k1 = keys()
        print(" again:", tofloat(k1)[b, t, n0:n0 + 32])
for rep in range(2):
    k1 = keys()
    d = np.argwhere(k0 != k1)
    print("rep", rep, "differing keys:", len(d))
    if len(d):
        tiles = {(b, t, n // 32) for b, t, n in d}
        print("  distinct tiles:", len(tiles), " keys per tile:", len(d) / len(tiles))
        lanes = np.bincount(d[:, 2] % 32, minlength=32)
        print("  lane histogram:", lanes.tolist())
        print("  tile-in-slice histogram:", np.bincount(d[:, 2] // 32, minlength=32).tolist())
        print("  slice t histogram:", np.bincount(d[:, 1], minlength=T + 1).tolist())
        print("  first:", d[:10].tolist())
        b, t, n = d[0]
        print("  values:", k0[b, t, n - n % 32:n - n % 32 + 32].view(np.uint32)[:8], k1[b, t, n - n % 32:n - n % 32 + 32].view(np.uint32)[:8])
