"""HiGHS (the third-party QP solver bundled inside scipy) as an independent check of the oracle's QP solve.
TEST INFRASTRUCTURE: used by tests/golden/make_golden.py, tests/tools/qp_highs_sweep.py and tests/test_oracle_golden.py."""
import numpy as np


def highs_solve(P, q, A, b, G, h):
    """Solve min 1/2 z'Pz+q'z, Az=b, Gz<=h with HiGHS (scipy's bundled highspy core)."""
    from scipy.optimize._highspy import _core as hc
    from scipy.sparse import csc_matrix, tril
    n = P.shape[0]
    rows = np.vstack([A, G])
    lo = np.concatenate([b, np.full(G.shape[0], -hc.kHighsInf)])
    hi = np.concatenate([b, h])
    H = hc._Highs()
    H.setOptionValue("output_flag", False)
    for opt, val in (("primal_feasibility_tolerance", 1e-10), ("dual_feasibility_tolerance", 1e-10)):
        H.setOptionValue(opt, val)
    lp = hc.HighsLp()
    lp.num_col_, lp.num_row_ = n, rows.shape[0]
    lp.col_cost_ = q
    lp.col_lower_ = np.full(n, -hc.kHighsInf); lp.col_upper_ = np.full(n, hc.kHighsInf)
    lp.row_lower_, lp.row_upper_ = lo, hi
    Am = csc_matrix(rows)
    lp.a_matrix_.format_ = hc.MatrixFormat.kColwise
    lp.a_matrix_.num_col_, lp.a_matrix_.num_row_ = n, rows.shape[0]
    lp.a_matrix_.start_, lp.a_matrix_.index_, lp.a_matrix_.value_ = Am.indptr, Am.indices, Am.data
    hess = hc.HighsHessian()
    Pl = csc_matrix(tril(csc_matrix(P)))
    hess.dim_ = n
    hess.format_ = hc.HessianFormat.kTriangular
    hess.start_, hess.index_, hess.value_ = Pl.indptr, Pl.indices, Pl.data
    model = hc.HighsModel()
    model.lp_ = lp
    model.hessian_ = hess
    H.passModel(model)
    H.run()
    status = H.modelStatusToString(H.getModelStatus())
    return np.array(H.getSolution().col_value), status


def compare_with_highs(pb, s, u, d):
    """Solve the NRMP problem `pb` (oracle.nrmp_qp.NrmpProblem) with HiGHS in the oracle's uncondensed formulation and
    compare with the given solution: dict(du, obj_diff = obj(given) - obj(HiGHS), status)."""
    from oracle.nrmp_qp import _assemble_full
    P, q, A, b, G, h, (ns, nu, nd, ne) = _assemble_full(pb)
    z, status = highs_solve(P, q, A, b, G, h)
    uh = z[ns:ns + nu].reshape(pb.T, 2).T
    dh = z[ns + nu:ns + nu + nd].reshape(1, -1)
    sh = z[:ns].reshape(pb.T + 1, 3).T
    obj_h = pb.objective(sh, uh, dh.reshape(-1) if nd else None)
    obj_o = pb.objective(s, u, None if d is None else np.asarray(d).reshape(-1))
    return dict(du=float(np.abs(u - uh).max()), obj_diff=float(obj_o - obj_h), status=status, obj=float(obj_o))
