"""The active-set iteration for the warm QP solves (NPA_QP_ASET=1: its own launch of the T = 10 / M = 10 instantiation in front of
the interior-point launch, off by default -- DESIGN.md section 3.3 has the measurements) against the default path."""
import numpy as np
import pytest

from helpers import CONFIGS
from neupan_amd.scenes import make_batch

pytestmark = [pytest.mark.experiments, pytest.mark.gpu]


@pytest.mark.parametrize("cfgname,scenes", [("diff_1k_T10_K10", 128), ("dyna_4k_T10_K10", 32)])
def test_active_set_path_equals_the_interior_point_path(cfgname, scenes, monkeypatch):
    from gpu_helpers import make_gpu_pan
    cfg = CONFIGS[cfgname]
    batch = make_batch(cfg, 0, scenes)
    args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")] + [batch.get("velocities")]
    ref = make_gpu_pan(cfg)
    u0 = ref.forward_batch(*args)["opt_u"].cpu().numpy()
    i0 = ref.last_qp_info()
    monkeypatch.setenv("NPA_QP_ASET", "1")
    pan = make_gpu_pan(cfg)
    u1 = pan.forward_batch(*args)["opt_u"].cpu().numpy()
    i1 = pan.last_qp_info()
    assert (i1[:, 3] == 0).all()
    taken = i1[:, 15] == 6
    assert taken.mean() >= 0.35, taken.mean()                     # (last QP of the call: 66 % on configs[1], 41 - 44 % with 4000 moving points)
    assert i1[taken, 1].max() <= 1e-13
    assert i1[:, 14].mean() < i0[:, 14].mean()
    assert (i1[taken, 5] <= 2).all() and i1[taken, 14].mean() <= 1.3      # accepted after one or two guesses
    err = np.linalg.norm((u1 - u0).reshape(scenes, -1), axis=1)
    assert np.median(err) <= 5e-6 and np.quantile(err, 0.9) <= 1e-4, (np.median(err), np.quantile(err, 0.9))
