"""Branch qp-active-set: the reduction step of the active-set iteration (neupan_amd/csrc/aset_reduce.hip) against its numpy
statement (tests/tools/qp_active_set_study.py::lane_level_reduction, itself checked against the matrix form on the QPs of the
benchmark workloads).  Not wired into the product path yet."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.experiments

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


def _lib():
    from neupan_amd import _lib as L
    lib = L.load()
    fn = lib.npa_dbg_aset_reduce
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 14 + [C.c_void_p]
    return fn


def test_debug_entry_is_exported():
    assert _lib() is not None


def _cases(nu, n, seed):
    rng = np.random.default_rng(seed)
    Ks, rs, ties, bnds = [], [], [], []
    for i in range(n):
        A = rng.standard_normal((nu, nu)); K = A @ A.T + nu * np.eye(nu)
        tie = np.zeros(nu); bnd = np.zeros(nu)
        p_tie, p_bnd = rng.choice([0.0, 0.2, 0.5, 0.9]), rng.choice([0.0, 0.1, 0.4])
        for a in range(nu):
            if a >= 2 and rng.random() < p_tie: tie[a] = rng.choice([-1.0, 1.0])
            if rng.random() < p_bnd: bnd[a] = rng.choice([-1.0, 1.0])
        Ks.append(K); rs.append(rng.standard_normal(nu)); ties.append(tie); bnds.append(bnd)
    return np.array(Ks), np.array(rs), np.array(ties), np.array(bnds)


@pytest.mark.gpu
@pytest.mark.parametrize("nu", [20, 40])
def test_reduction_matches_the_numpy_statement(nu):
    import torch
    from qp_active_set_study import lane_level_multipliers, lane_level_reduction
    fn = _lib()
    B = 256
    K, r, tie, bnd = _cases(nu, B, 7 + nu)
    acc = np.array([0.37, 0.81]); spd = np.array([2.5, 1.25])
    dev = torch.device("cuda:0")
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    dK, dr, dt_, db, da, ds = t(K), t(r), t(tie), t(bnd), t(acc), t(spd)
    res = np.random.default_rng(99 + nu).standard_normal((B, nu))
    dres = t(res)
    oK, orr, oo, olt, olb = torch.empty_like(dK), torch.empty_like(dr), torch.empty_like(dr), torch.empty_like(dr), torch.empty_like(dr)
    oh = torch.empty((B, nu), dtype=torch.int32, device=dev); oa = torch.empty((B, nu), dtype=torch.int32, device=dev)
    rc = fn(B, nu, dK.data_ptr(), dr.data_ptr(), dt_.data_ptr(), db.data_ptr(), da.data_ptr(), ds.data_ptr(), oK.data_ptr(),
            orr.data_ptr(), oh.data_ptr(), oo.data_ptr(), oa.data_ptr(), dres.data_ptr(), olt.data_ptr(), olb.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    for i in range(B):
        Kl, rl, head, offvec, anch = lane_level_reduction(K[i], r[i], tie[i], bnd[i], acc, spd, nu)
        np.testing.assert_array_equal(oh[i].cpu().numpy(), head)
        np.testing.assert_array_equal(oa[i].cpu().numpy() != 0, anch)
        np.testing.assert_allclose(oo[i].cpu().numpy(), offvec, atol=1e-14)
        sc = 1.0 + np.abs(Kl).max()
        np.testing.assert_allclose(oK[i].cpu().numpy() / sc, Kl / sc, atol=1e-13)
        np.testing.assert_allclose(orr[i].cpu().numpy(), rl, atol=1e-11 * (1.0 + np.abs(rl).max()))
        lt, lb = lane_level_multipliers(res[i], tie[i], bnd[i], nu)
        np.testing.assert_allclose(olt[i].cpu().numpy(), lt, atol=1e-12)
        np.testing.assert_allclose(olb[i].cpu().numpy(), lb, atol=1e-12)
