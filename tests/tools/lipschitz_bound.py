"""Why the margin of the geometric keys is measured and audited instead of derived: the only Lipschitz constant of the
DUNE network that is cheap to compute -- the product of the layers' spectral norms with the largest gain of each
LayerNorm (|gamma| / sqrt(eps): LayerNorm divides by a data-dependent deviation that is only bounded below by sqrt(eps))
-- for the shipped checkpoints, next to the slope of f that npa_create actually measures on its finest grid.

    python tests/tools/lipschitz_bound.py           # CPU

The product bounds |d mu / d p|; times the 2.8 mm from a cell centre to its nodes of the 4 mm calibration grid it is the
analytic "bound" on what f can do between grid nodes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIN, NORM = (0, 3, 5, 8, 10, 13), (1, 6, 11)


def main():
    ck = os.path.join(ROOT, "tests", "golden", "checkpoints")
    for name in sorted(f for f in os.listdir(ck) if f.endswith("model_5000.pth")):
        sd = torch.load(os.path.join(ck, name), map_location="cpu")
        lin = [np.linalg.norm(sd[f"MLP.{i}.weight"].numpy().astype(np.float64), 2) for i in LIN]
        gain = [float(np.abs(sd[f"MLP.{i}.weight"].numpy()).max()) / np.sqrt(1e-5) for i in NORM]
        L = float(np.prod(lin) * np.prod(gain))
        print(f"{name:42s} spectral norms {np.round(lin, 2)}  LayerNorm gains <= {np.round(gain, 1)}  product = {L:.3g} "
              f"-> x 2.8 mm = {L * 2.8e-3:.3g} m between nodes of the finest grid")


if __name__ == "__main__":
    main()
