"""Make tests/golden/checkpoints/ridge_model.pth: an ADVERSARIAL DUNE checkpoint for the diff robot -- the shipped
diff_robot_default model fine-tuned so that its distance follows the geometry everywhere EXCEPT on a narrow ridge: along
the line x = X0 (|y| <= 2.5 m, robot frame) the network's distance dips by DEPTH over a width of about 2 WIDTH.  A feature
narrower than a cell of a (coarsened) calibration grid is exactly what a MEASURED margin can miss; the GPU test
test_ridge_checkpoint_* checks that such a checkpoint either loses its geometric keys at npa_create (default grids and
the refinement check see the ridge) or -- with the grids coarsened and the check switched off -- is caught by the run-time
audit on clouds that put points on the ridge.

    python tests/golden/make_ridge_checkpoint.py          # CPU, a few minutes

Same architecture and state_dict keys as the reference (neupan/blocks/obs_point_net.py:31-46)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neupan_amd.robot import Robot  # noqa: E402
from oracle import dune_label_oracle as dl  # noqa: E402

X0, WIDTH, DEPTH, YMAX = 10.02, 0.012, 0.5, 2.5


def net(E):
    L = torch.nn
    return L.Sequential(L.Linear(2, 32), L.LayerNorm(32), L.Tanh(), L.Linear(32, 32), L.ReLU(), L.Linear(32, 32),
                        L.LayerNorm(32), L.Tanh(), L.Linear(32, 32), L.ReLU(), L.Linear(32, 32), L.LayerNorm(32), L.Tanh(),
                        L.Linear(32, E), L.ReLU())


def ridge(P):
    x, y = P[:, 0], P[:, 1]
    return -DEPTH * np.exp(-((x - X0) / WIDTH) ** 2) * (np.abs(y) <= YMAX)


def main():
    torch.manual_seed(0)
    rb = Robot(10, 0.1, kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3])
    G = np.asarray(rb.G, np.float64); h = np.asarray(rb.h, np.float64).reshape(-1)
    E = G.shape[0]
    m = net(E)
    sd = torch.load(os.path.join(HERE, "checkpoints", "diff_robot_default_model_5000.pth"), map_location="cpu")
    m.load_state_dict({k.replace("MLP.", ""): v for k, v in sd.items()})
    rng = np.random.default_rng(1)

    def batch(n):
        a = rng.uniform(-25, 25, (n // 2, 2))
        b = np.stack([X0 + rng.normal(0, 4 * WIDTH, n // 2), rng.uniform(-YMAX - 1, YMAX + 1, n // 2)], axis=1)   # around the ridge
        P = np.concatenate([a, b])
        _, dist = dl.labels(G, h, P)
        return torch.tensor(P, dtype=torch.float32), torch.tensor(dist + ridge(P), dtype=torch.float32)
    Gt = torch.tensor(G, dtype=torch.float32); ht = torch.tensor(h, dtype=torch.float32)
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    for it in range(int(os.environ.get("RIDGE_STEPS", "6000"))):
        X, D = batch(4096)
        out = m(X)
        d = (out * (X @ Gt.T - ht)).sum(1)
        w = torch.ones_like(D); w[2048:] = 4.0
        loss = (w * (d - D) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
        if it % 500 == 0:
            with torch.no_grad():
                xs = torch.tensor(np.stack([np.linspace(X0 - 0.1, X0 + 0.1, 401), np.zeros(401)], 1), dtype=torch.float32)
                dd = (m(xs) * (xs @ Gt.T - ht)).sum(1).numpy()
                g = dl.labels(G, h, xs.numpy().astype(np.float64))[1]
                print(it, float(loss), "deepest f on the cut:", float((dd - g).min()), "at x =", float(xs[(dd - g).argmin(), 0]),
                      "f 5 cm off:", float((dd - g)[[75, 325]].mean()), flush=True)
    torch.save({"MLP." + k: v for k, v in m.state_dict().items()}, os.path.join(HERE, "checkpoints", "ridge_model.pth"))
    with torch.no_grad():
        P = rng.uniform(-25, 25, (20000, 2)); P = P[np.abs(P[:, 0] - X0) > 0.3]
        X = torch.tensor(P, dtype=torch.float32)
        d = (m(X) * (X @ Gt.T - ht)).sum(1).numpy()
        print("away from the ridge: max |f| =", float(np.abs(d - dl.labels(G, h, P)[1]).max()))


if __name__ == "__main__":
    main()
