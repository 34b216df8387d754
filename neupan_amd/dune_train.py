"""DUNE training on the GPU: `DUNETrain` of the reference (neupan/blocks/dune_train.py:60-343) with the
per-point ECOS solves of its data-set generation (:109-140, "1-2 h on CPU" for 100 k points) replaced by
the closed-form HIP labeller (csrc/dune_labels.hip, neupan_amd/dune_labels.py).

Everything else follows the reference: the ObsPointNet architecture and state_dict keys
(obs_point_net.py:31-46), Adam(lr=1e-4, weight_decay=1e-4) (:71), the 80/20 split (:191-193), the four
loss terms mu / distance / fa / fb with one random rotation per batch (:302-366), the learning-rate decay,
the `model_<epoch>.pth` checkpoints and the `results.txt` log (:180-300).  The optimisation itself is
PyTorch autograd, as it is in the reference; tensors live on the GPU and batches are slices (the
reference's DataLoader does not shuffle either).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from .dune_labels import dune_labels


class ObsPointNet(torch.nn.Module):
    """obs_point_net.py:25-49 (same layer order, hence the same `MLP.<k>.*` state_dict keys)."""

    def __init__(self, input_dim=2, output_dim=4):
        super().__init__()
        L, H = torch.nn, 32
        self.MLP = L.Sequential(L.Linear(input_dim, H), L.LayerNorm(H), L.Tanh(), L.Linear(H, H), L.ReLU(),
                                L.Linear(H, H), L.LayerNorm(H), L.Tanh(), L.Linear(H, H), L.ReLU(),
                                L.Linear(H, H), L.LayerNorm(H), L.Tanh(), L.Linear(H, output_dim), L.ReLU())

    def forward(self, x):
        return self.MLP(x)


def dune_losses(model, G, h, points, label_mu, label_distance, theta):
    """The four loss terms of dune_train.py:302-366 for one batch.  points (b,2), label_mu (b,E),
    label_distance (b,), G (E,2), h (E,1); theta: the batch's random rotation angle (:349)."""
    mse = torch.nn.functional.mse_loss
    mu = model(points)                                                   # (b,E)
    dist = (mu * (points @ G.T - h.reshape(1, -1))).sum(dim=1)            # cal_distance :368-378
    c, s = float(np.cos(theta)), float(np.sin(theta))
    R = torch.tensor([[c, -s], [s, c]], dtype=points.dtype, device=points.device)
    M = -(R @ G.T)                                                       # fa = (-R G^T mu)^T  :353-354
    fa, fa_l = mu @ M.T, label_mu @ M.T                                  # (b,2)
    hh = h.reshape(-1)
    fb = (fa * points).sum(dim=1) + mu @ hh                              # :356-357
    fb_l = (fa_l * points).sum(dim=1) + label_mu @ hh
    return mse(mu, label_mu), mse(dist, label_distance), mse(fa, fa_l), mse(fb, fb_l)


class DuneTrain:
    """Constructor and `start` keyword arguments as the reference's DUNETrain (:61, :144-157)."""

    def __init__(self, model, robot_G, robot_h, checkpoint_path, device="cuda"):
        self.device = torch.device(device)
        self.G = torch.as_tensor(np.asarray(robot_G, dtype=np.float32)).to(self.device)
        self.h = torch.as_tensor(np.asarray(robot_h, dtype=np.float32).reshape(-1, 1)).to(self.device)
        self.model = (model if model is not None else ObsPointNet(2, self.G.shape[0])).to(self.device)
        self.checkpoint_path = checkpoint_path
        os.makedirs(checkpoint_path, exist_ok=True)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=1e-4, weight_decay=1e-4)
        self.loss_list = []

    def generate_data_set(self, data_size=10000, data_range=(-50, -50, 50, 50)):
        """:109-135 with the labels from the HIP labeller.  Returns float32 device tensors."""
        p = np.random.uniform(low=data_range[:2], high=data_range[2:], size=(data_size, 2))
        mu, dist = dune_labels(self.G.double().cpu().numpy(), self.h.double().cpu().numpy(), p, self.device)
        return torch.from_numpy(p.astype(np.float32)).to(self.device), mu, dist

    def _epoch(self, data, batch_size, validate):
        P, MU, D = data
        tot = np.zeros(4)
        nb = 0
        for i in range(0, P.shape[0], batch_size):
            theta = np.random.uniform(0, 2 * np.pi)
            if not validate:
                self.optimizer.zero_grad()
            with torch.set_grad_enabled(not validate):
                lm, ld, la, lb = dune_losses(self.model, self.G, self.h, P[i:i + batch_size], MU[i:i + batch_size],
                                             D[i:i + batch_size], theta)
                if not validate:
                    (lm + ld + la + lb).backward()
                    self.optimizer.step()
            tot += np.array([lm.item(), ld.item(), la.item(), lb.item()])
            nb += 1
        return tuple(tot / max(nb, 1))

    def start(self, data_size=100000, data_range=(-25, -25, 25, 25), batch_size=256, epoch=5000, valid_freq=100,
              save_freq=500, lr=5e-5, lr_decay=0.5, decay_freq=1500, save_loss=False, **kwargs):
        log = os.path.join(self.checkpoint_path, "results.txt")
        head = (f"data_size: {data_size}, data_range: {list(data_range)}, batch_size: {batch_size}, epoch: {epoch}, "
                f"valid_freq: {valid_freq}, save_freq: {save_freq}, lr: {lr}, lr_decay: {lr_decay}, decay_freq: {decay_freq}, "
                f"robot_G: {self.G.cpu()}, robot_h: {self.h.cpu()}")
        with open(log, "a") as f:
            print(head + "\n", file=f)
        with open(os.path.join(self.checkpoint_path, "train_dict.pkl"), "wb") as f:
            pickle.dump(dict(data_size=data_size, data_range=list(data_range), batch_size=batch_size, epoch=epoch,
                             valid_freq=valid_freq, save_freq=save_freq, lr=lr, lr_decay=lr_decay, decay_freq=decay_freq), f)
        self.optimizer.param_groups[0]["lr"] = float(lr)
        P, MU, D = self.generate_data_set(data_size, data_range)
        perm = torch.randperm(data_size, device=self.device)            # random_split :191-193
        ntr = int(data_size * 0.8)
        tr, va = perm[:ntr], perm[ntr:ntr + int(data_size * 0.2)]
        train, valid = (P[tr], MU[tr], D[tr]), (P[va], MU[va], D[va])
        full = None
        for i in range(epoch + 1):
            self.model.train(True)
            ml, dl, al, bl = self._epoch(train, batch_size, False)
            if i % valid_freq == 0:
                self.model.eval()
                vml, vdl, val, vbl = self._epoch(valid, batch_size, True)
                with open(log, "a") as f:
                    print(f"Epoch {i}/{epoch} learning rate {self.optimizer.param_groups[0]['lr']} \n"
                          "---------------------------------\nLosses:\n"
                          f"  Mu Loss:          {ml:.2e}   | Validate Mu Loss:            {vml:.2e}\n"
                          f"  Distance Loss:    {dl:.2e}   | Validate Distance Loss:      {vdl:.2e}\n"
                          f"  Fa Loss:          {al:.2e}   | Validate Fa Loss:            {val:.2e}\n"
                          f"  Fb Loss:          {bl:.2e}   | Validate Fb Loss:            {vbl:.2e}\n", file=f)
            if i % save_freq == 0:
                full = os.path.join(self.checkpoint_path, f"model_{i}.pth")
                torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, full)
            if (i + 1) % decay_freq == 0:
                self.optimizer.param_groups[0]["lr"] *= lr_decay
                with open(log, "a") as f:
                    print("current learning rate:", self.optimizer.param_groups[0]["lr"], file=f)
            self.loss_list.append(ml + dl + al + bl)
            if save_loss:
                with open(os.path.join(self.checkpoint_path, "loss.pkl"), "wb") as f:
                    pickle.dump(self.loss_list, f)
        return full
