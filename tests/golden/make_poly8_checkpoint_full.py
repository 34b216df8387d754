"""tests/golden/checkpoints/poly8_model_5000.pth: an E = 8 ObsPointNet for BASELINE.json configs[4] trained with the
REFERENCE'S RECIPE (the reference ships no 8-edge robot and no E = 8 checkpoint):

    python tests/golden/make_poly8_checkpoint_full.py [epochs]        # CPU, ~40 min for the 5000 epochs

neupan_amd.dune_train.DuneTrain (the mirror of neupan/blocks/dune_train.py: 100 000 points in [-25, 25]^2, 80/20 split,
batch 256, Adam(lr 5e-5, weight_decay 1e-4), lr x0.5 every 1500 epochs, the four loss terms with one random rotation per
batch, 5000 epochs) run on CPU tensors here; the labels come from the closed-form oracle labeller
(oracle/dune_label_oracle.py) instead of the HIP one -- this is a fixture generator, the product path labels on the GPU.
Writes the checkpoint and results.txt (the reference's log format) next to it as poly8_results.txt."""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neupan_amd.dune_train import DuneTrain  # noqa: E402
from neupan_amd.robot import halfplanes_from_vertices  # noqa: E402
from oracle import dune_label_oracle as dl  # noqa: E402

VERTS = np.array([[-0.6, -0.8], [0.6, -0.8], [1.0, -0.4], [1.0, 0.4], [0.6, 0.8], [-0.6, 0.8], [-1.0, 0.4], [-1.0, -0.4]]).T


class CpuTrain(DuneTrain):
    def generate_data_set(self, data_size=10000, data_range=(-50, -50, 50, 50)):
        p = np.random.uniform(low=data_range[:2], high=data_range[2:], size=(data_size, 2))
        mu, dist = dl.labels(self.G.double().numpy(), self.h.double().numpy(), p)
        return (torch.from_numpy(p.astype(np.float32)), torch.from_numpy(mu.astype(np.float32)),
                torch.from_numpy(dist.astype(np.float32)))


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    torch.manual_seed(0); np.random.seed(0)
    torch.set_num_threads(2)
    G, h = halfplanes_from_vertices(VERTS)
    work = tempfile.mkdtemp(prefix="poly8_")
    tr = CpuTrain(None, np.asarray(G, np.float32), np.asarray(h, np.float32), work, device="cpu")
    full = tr.start(epoch=epochs, save_freq=epochs, valid_freq=max(epochs // 10, 1))
    dst = os.path.join(HERE, "checkpoints", "poly8_model_5000.pth")
    shutil.copy(full, dst)
    shutil.copy(os.path.join(work, "results.txt"), os.path.join(HERE, "checkpoints", "poly8_results.txt"))
    print("wrote", dst)
    print(open(os.path.join(work, "results.txt")).read()[-700:])


if __name__ == "__main__":
    main()
