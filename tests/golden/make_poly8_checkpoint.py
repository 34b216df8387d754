"""Make tests/golden/checkpoints/poly8_model_quick.pth: an E = 8 ObsPointNet for BASELINE.json configs[4]
(the reference ships no 8-edge robot and no E = 8 checkpoint).

    python tests/golden/make_poly8_checkpoint.py

The network has the reference's architecture and state_dict keys (neupan/blocks/obs_point_net.py:31-46,
`MLP.{0,1,3,5,6,8,10,11,13}.{weight,bias}`) and is fitted for 1500 epochs (CPU, minutes) to the
closed-form labels of oracle/dune_label_oracle.py with the reference's loss terms mu + distance
(dune_train.py:230-260).  It is a QUICK fit for throughput and GPU-vs-oracle parity runs, not a
production model (validation MSE at the end: mu 4.3e-2, distance 2.5e-4 -- the reference's 5000-epoch
models reach 1e-6)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neupan_amd.robot import halfplanes_from_vertices  # noqa: E402
from oracle import dune_label_oracle as dl  # noqa: E402

VERTS = np.array([[-0.6, -0.8], [0.6, -0.8], [1.0, -0.4], [1.0, 0.4], [0.6, 0.8], [-0.6, 0.8], [-1.0, 0.4], [-1.0, -0.4]]).T


def net(E):
    L = torch.nn
    return L.Sequential(L.Linear(2, 32), L.LayerNorm(32), L.Tanh(), L.Linear(32, 32), L.ReLU(), L.Linear(32, 32),
                        L.LayerNorm(32), L.Tanh(), L.Linear(32, 32), L.ReLU(), L.Linear(32, 32), L.LayerNorm(32), L.Tanh(),
                        L.Linear(32, E), L.ReLU())


def main():
    torch.manual_seed(0)
    G, h = halfplanes_from_vertices(VERTS)
    G = np.asarray(G, np.float64); h = np.asarray(h, np.float64).reshape(-1)
    E = G.shape[0]
    rng = np.random.default_rng(0)
    P = rng.uniform(-25, 25, (24000, 2))
    mu, dist = dl.labels(G, h, P)
    X = torch.tensor(P, dtype=torch.float32); Y = torch.tensor(mu, dtype=torch.float32); D = torch.tensor(dist, dtype=torch.float32)
    Gt = torch.tensor(G, dtype=torch.float32); ht = torch.tensor(h, dtype=torch.float32)
    m = net(E)
    with torch.no_grad():
        m[-2].bias.fill_(0.2)                 # keep every output's ReLU alive at the start
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 300, 0.5)
    ntr = 20000
    for ep in range(1500):
        perm = torch.randperm(ntr)
        for i in range(0, ntr, 256):
            idx = perm[i:i + 256]
            out = m(X[idx])
            d = (out * (X[idx] @ Gt.T - ht)).sum(1)
            loss = torch.nn.functional.mse_loss(out, Y[idx]) + torch.nn.functional.mse_loss(d, D[idx])
            opt.zero_grad(); loss.backward(); opt.step()
        sched.step()
    with torch.no_grad():
        out = m(X[ntr:]); d = (out * (X[ntr:] @ Gt.T - ht)).sum(1)
        vm, vd = float(torch.nn.functional.mse_loss(out, Y[ntr:])), float(torch.nn.functional.mse_loss(d, D[ntr:]))
    print("validation mu MSE %.2e  distance MSE %.2e" % (vm, vd))
    sd = {"MLP." + k: v.detach().clone() for k, v in m.state_dict().items()}
    path = os.path.join(HERE, "checkpoints", "poly8_model_quick.pth")
    torch.save(sd, path)
    print("wrote", path, sorted(sd)[:4])


if __name__ == "__main__":
    main()
