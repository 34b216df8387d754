"""DUNE training labels on the GPU: `DUNETrain.generate_data_set` / `prob_solve`
(neupan/blocks/dune_train.py:109-140) without cvxpy -- the per-point SOCP has a closed form
(csrc/dune_labels.hip).  Returns what the reference's dataset holds: inputs, mu labels, distances."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def dune_labels(G, h, points, device="cuda"):
    """G (E,2), h (E,) or (E,1): the robot polygon as `robot.G`, `robot.h`; points (n,2) float64
    (array or device tensor).  Returns mu (n,E) float32, distance (n,) float32 device tensors."""
    lib = _lib.load()
    dev = torch.device(device)
    Gh = np.ascontiguousarray(np.asarray(G.cpu() if isinstance(G, torch.Tensor) else G, dtype=np.float64).reshape(-1, 2))
    hh = np.ascontiguousarray(np.asarray(h.cpu() if isinstance(h, torch.Tensor) else h, dtype=np.float64).reshape(-1))
    E = Gh.shape[0]
    if hh.shape[0] != E:
        raise ValueError("G and h disagree on the number of edges")
    pts = torch.as_tensor(points).to(device=dev, dtype=torch.float64).reshape(-1, 2).contiguous()
    n = pts.shape[0]
    mu = torch.empty((n, E), dtype=torch.float32, device=dev)
    dist = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.npa_dune_labels(E, Gh.ctypes.data_as(C.c_void_p), hh.ctypes.data_as(C.c_void_p), n,
                                  C.c_void_p(pts.data_ptr()), C.c_void_p(mu.data_ptr()), C.c_void_p(dist.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "npa_dune_labels")
    return mu, dist


def generate_data_set(G, h, data_size=10000, data_range=(-50, -50, 50, 50), seed=None, device="cuda"):
    """dune_train.py:109-135: uniform points in data_range and their labels.
    Returns (points (n,2) float32, mu (n,E) float32, distance (n,) float32) device tensors."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(low=data_range[:2], high=data_range[2:], size=(data_size, 2))
    mu, dist = dune_labels(G, h, p, device)
    return torch.from_numpy(p.astype(np.float32)).to(device), mu, dist
