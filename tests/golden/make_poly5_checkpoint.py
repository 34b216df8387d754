"""Make tests/golden/checkpoints/poly5_model_quick.pth: an E = 5 ObsPointNet for an irregular, ROTATED pentagon (no edge parallel
to an axis) -- the polygon shapes the selection's key pass ranks on their BOUNDING BOX (select_geo_body.inc: g <= box key + S)
without a table filter or a merged-launch instantiation (those exist for E = 4 and 8).  Same recipe as make_poly8_checkpoint.py
(reference architecture and state_dict keys, closed-form labels of oracle/dune_label_oracle.py, the reference's loss terms mu +
distance, dune_train.py:230-260), a QUICK fit: used by tests/test_gpu_parity.py::test_general_polygon_selection_equals_exact_keys
with geometric keys forced (NPA_KEY_TERMS=4), where only the bitwise equality of the rows matters.

    python tests/golden/make_poly5_checkpoint.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_poly8_checkpoint import net  # noqa: E402
from neupan_amd.robot import halfplanes_from_vertices  # noqa: E402
from oracle import dune_label_oracle as dl  # noqa: E402

# counter-clockwise, convex, no axis-parallel edge; bounding box 2.5 x 2.1 m, not centred on the origin
VERTS = np.array([[-0.9, -0.7], [0.5, -1.1], [1.6, -0.1], [0.7, 1.0], [-0.8, 0.6]]).T


def main(epochs=1500):
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
        m[-2].bias.fill_(0.2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 300, 0.5)
    ntr = 20000
    for ep in range(epochs):
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
    path = os.path.join(HERE, "checkpoints", "poly5_model_quick.pth")
    torch.save(sd, path)
    print("wrote", path)


if __name__ == "__main__":
    main()
