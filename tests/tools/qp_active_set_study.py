"""CPU study for the NEXT step of the QP kernel: a primal-dual active-set iteration for the warm-started solves.

Once the PAN loop has settled, iteration k + 1 solves nearly the QP of iteration k; the interior-point method of
nrmp_qp.hip then needs ~4 iterations (one factorisation and two solves each) from the previous solution.  The problem is
a strictly convex QP with squared-hinge terms and simple linear rows (bounds on u and d, rate limits):

    min 1/2 x'Hx + g'x + ro/2 |max(0, f - F x)|^2      s.t.  C x <= c .

Given the previous solve's x and multipliers, guess the hinge rows that are on (f - F x > 0) and the linear rows that are
tight (lam + (C x - c) > 0), solve the equality-constrained QP of that guess exactly -- ONE factorisation and one solve --
and repeat until the guess reproduces itself: then the KKT conditions hold exactly.  This tool replays that on every QP
the oracle's PAN loop produces (condensed exactly as the kernel condenses it, oracle/condensed_ipm.py), started from the
previous PAN iteration's solution under the kernel's warm-start gate, and reports how often and how fast it converges.

    python tests/tools/qp_active_set_study.py [scenes per workload] [procs]    -> profiles/r03_qp_active_set_study.txt

Nothing of this is in the product; DESIGN.md section 7 lists it as the next algorithmic step.

Lane-level plan for the T = 10 / M = 10 instantiation (variable a = 2t + k in lane a, pair p in lane p, as today):
  A. structure.  Pair lanes publish their tight flags through LDS; variable lane a reads tie(a) (sign of the tight rate row
     between a - 2 and a) and bnd(a) (sign of its tight speed row).  Runs along the stride-2 chains of a channel by a
     segmented Hillis-Steele scan, 4 steps of lane distance 2 / 4 / 8 / 16 (5 at T = 20; ds_bpermute: the chains cross the 16-lane DPP
     rows): off(a) = offset to the head's value, head(a).  A run with a tight speed row is anchored: members publish
     +-b - off(a) to slot[head(a)], everybody reads slot[head(a)].
  B. matrix.  rhs -= K' off_vec (20 column-uniform values: LDS broadcast reads); column merge K'Z with wave-uniform 0/1
     multipliers (arow[c-2] += m_c arow[c], then arow[c] = 0: ~70 issue slots); K' is symmetric, so Z'K'Z = ((K'Z)'Z)':
     transpose through the LDS matrix the substitutions already use and merge the columns once more; Z'rhs by one segmented
     scan; unit rows for tied / fixed members.  Per-step blocks S'_t with weights ro on the hinge rows that are on and
     kappa_t = ro |I_t| (d free) or no kappa term (d fixed): the existing P_t / K' build with different inputs, no band terms.
  C. the existing factorisation and substitutions; u_a = z[head(a)] + off_vec(a) (one ds_bpermute).
  D. d of the free steps; stationarity residual with the existing residual phase (lam_c = 0, lam_f = ro e); multipliers of
     the tight rows = segmented suffix sums of the residual along the runs, handed to the pair lanes through LDS; the next
     guess tests signs and primal feasibility.  Same guess twice = KKT point: write out, skip the interior-point loop.
`active_set_solve_kernel_form` below is this at the matrix level."""
import os, sys
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
WORK = ("diff_1k_T10_K10", "acker_2k_T20_K15", "dyna_4k_T10_K10", "poly8_5k_T10_K10")


def active_set_solve(H, g, F, f, C, c, ro, x, lam, max_it=8):
    """-> x, lam, factorisations, converged"""
    n = H.shape[0]
    prev = None
    for it in range(max_it + 1):
        on = (f - F @ x) > 0
        tight = (lam + (C @ x - c)) > 0
        key = (on.tobytes(), tight.tobytes())
        if key == prev:
            return x, lam, it, True
        if it == max_it:
            break
        prev = key
        K = H + ro * F[on].T @ F[on]
        rhs = -g + ro * F[on].T @ f[on]
        CA = C[tight]; m = CA.shape[0]
        try:
            if m:
                sol = np.linalg.solve(np.block([[K, CA.T], [CA, np.zeros((m, m))]]), np.concatenate([rhs, c[tight]]))
                x = sol[:n]; lam = np.zeros(C.shape[0]); lam[tight] = sol[n:]
            else:
                x = np.linalg.solve(K, rhs); lam = np.zeros(C.shape[0])
        except np.linalg.LinAlgError:           # dependent tight rows
            break
    return x, lam, max_it, False


def active_set_solve_kernel_form(pb, H, g, F, f, C, c, x, lam, max_it=8, lane_check=None, project_d=False):
    """The same iteration in the form the wave kernel can run (what DESIGN.md section 7 proposes), as a check that nothing
    but 2T x 2T Cholesky solves is needed.  Per guess:
      * d_t: fixed at the bound its tight row names; else eliminated through its own stationarity condition (needs a hinge
        row that is on in step t: kappa_t = ro |I_t|, exactly the per-step blocks S'_t = S_t - v v'/kappa of the kernel);
      * u: the tight speed / rate rows of a control channel tie runs of consecutive steps, u_t = z_group + offset_t; a run
        that contains a tight speed row is fixed altogether.  K' is summed over the runs (rows and columns), the other
        members of a run get unit rows: one SPD solve of the kernel's size;
      * multipliers of the tight rows from the stationarity residual along each run (running sums), signs checked by the
        next guess."""
    T = pb.T; nu = 2 * T; M = pb.M; ro = pb.ro_obs
    Hu, gu = H[:nu, :nu], g[:nu]
    nb, nr = 4 * T, 4 * (T - 1)
    dlo = max(pb.d_min, 0.0)
    prev = None
    if project_d:                                   # (an interior-point iterate sits 1e-15 inside its bounds: snap)
        x = x.copy(); dd = x[nu:]
        dd = np.where(dd >= pb.d_max - 1e-8, pb.d_max, np.where(dd <= dlo + 1e-8, dlo, dd)); x[nu:] = dd
    for it in range(max_it + 1):
        on = (f - F @ x) > 0
        tight = (lam + (C @ x - c)) > 0
        if project_d:                                   # what the kernel integration does: see the comment at the clip below
            e0 = np.where(on, f - F @ x, 0.0)
            resd = -(g[nu:] + ro * (e0.reshape(T, M).sum(1)))                  # -(d-gradient) = eta - ro sum e
            for t in range(T):
                tight[nb + nr + 2 * t] = (x[nu + t] >= pb.d_max) and resd[t] > 0
                tight[nb + nr + 2 * t + 1] = (x[nu + t] <= dlo) and resd[t] < 0
        key = (on.tobytes(), tight.tobytes())
        if key == prev:
            return x, lam, it, True
        if it == max_it:
            break
        prev = key
        # ---- d: fixed or eliminated ----
        K = Hu.copy(); r = -gu.copy()
        dfix = np.full(T, np.nan)
        for t in range(T):
            up, lo = tight[nb + nr + 2 * t], tight[nb + nr + 2 * t + 1]
            I = np.where(on[t * M:(t + 1) * M])[0] + t * M
            if project_d and (up or lo or len(I) == 0):
                dfix[t] = x[nu + t]                                           # fixed where it is (on the bound when a row is tight)
                Fu = F[I][:, :nu]
                K += ro * Fu.T @ Fu
                r += ro * Fu.T @ (f[I] + dfix[t])
            elif up or lo or len(I) == 0:
                dfix[t] = pb.d_max if (up or len(I) == 0) else dlo          # (no hinge row on: -eta pushes d to its upper bound)
                Fu = F[I][:, :nu]
                K += ro * Fu.T @ Fu
                r += ro * Fu.T @ (f[I] + dfix[t])                            # F_j x = Fu u - d
            else:
                Fu = F[I][:, :nu]; v = ro * Fu.sum(0); kap = ro * len(I)
                K += ro * Fu.T @ Fu - np.outer(v, v) / kap
                r += ro * Fu.T @ f[I] - v * (ro * f[I].sum() - pb.eta) / kap
        # ---- u: runs tied by tight rate rows, anchored by tight speed rows ----
        Z = np.zeros((nu, nu)); off = np.zeros(nu); free = np.zeros(nu, bool); rep = np.arange(nu)
        for k in range(2):
            t = 0
            while t < T:
                a0 = 2 * t + k; members = [a0]; o = [0.0]
                while t + 1 < T:
                    q = nb + 4 * t + 2 * k                                     # rows (t -> t+1, channel k): +, -
                    if tight[q]: o.append(o[-1] + pb.acce_bound[k])
                    elif tight[q + 1]: o.append(o[-1] - pb.acce_bound[k])
                    else: break
                    t += 1; members.append(2 * t + k)
                anchor = None
                for a, oa in zip(members, o):
                    if tight[2 * a]: anchor = pb.speed_bound[k] - oa; break
                    if tight[2 * a + 1]: anchor = -pb.speed_bound[k] - oa; break
                for a, oa in zip(members, o):
                    rep[a] = a0
                    if anchor is None: Z[a, a0] = 1.0; off[a] = oa
                    else: off[a] = anchor + oa
                free[a0] = anchor is None
                t += 1
        Kr = Z.T @ K @ Z; rr = Z.T @ (r - K @ off)
        idx = np.where(free)[0]
        Kr2 = np.eye(nu); Kr2[np.ix_(idx, idx)] = Kr[np.ix_(idx, idx)]      # unit rows for the tied / fixed members
        rr2 = np.zeros(nu); rr2[idx] = rr[idx]
        if lane_check is not None:                                          # the same reduction lane by lane
            tie = np.zeros(nu); bnd = np.zeros(nu)
            for a in range(nu):
                if a >= 2: tie[a] = 1.0 if tight[nb + 2 * (a - 2)] else (-1.0 if tight[nb + 2 * (a - 2) + 1] else 0.0)
                bnd[a] = 1.0 if tight[2 * a] else (-1.0 if tight[2 * a + 1] else 0.0)
            Kl, rl, headl, offl, ancl = lane_level_reduction(K, r, tie, bnd, pb.acce_bound, pb.speed_bound, nu)
            sc = 1.0 + np.abs(Kr2).max()
            lane_check.append(max(np.abs(Kl - Kr2).max() / sc, np.abs(rl - rr2).max() / (1.0 + np.abs(rr2).max()),
                                  np.abs(offl - off).max(), float(np.abs(headl - rep).max())))
        try:
            L = np.linalg.cholesky(Kr2)
        except np.linalg.LinAlgError:
            break
        z = np.linalg.solve(L.T, np.linalg.solve(L, rr2))
        u = Z @ z + off
        # ---- d of the eliminated steps, then the multipliers of every tight row from the stationarity residual ----
        xn = np.zeros_like(x); xn[:nu] = u
        for t in range(T):
            if np.isnan(dfix[t]):
                I = np.where(on[t * M:(t + 1) * M])[0] + t * M
                xn[nu + t] = (pb.eta / ro - (f[I] - F[I][:, :nu] @ u).sum()) / len(I)
            else:
                xn[nu + t] = dfix[t]
        if project_d:
            # the d of a free step is CLIPPED to its bounds after the solve (a projected step instead of the exact active-set move:
            # a fixed d then never has to travel, which is what lets the kernel reuse its d-elimination unchanged)
            xn[nu:] = np.clip(xn[nu:], dlo, pb.d_max)
        e = np.where(on, f - F @ xn, 0.0)
        res = -(H @ xn + g - ro * F.T @ e)                                     # = C' lam at a KKT point of the guess
        lamn = np.zeros(C.shape[0])
        for t in range(T):                                                      # d rows: +d <= dmax, -d <= -dlo
            if res[nu + t] > 0: lamn[nb + nr + 2 * t] = res[nu + t]
            else: lamn[nb + nr + 2 * t + 1] = -res[nu + t]
        # u rows.  Along a run, stationarity of member a reads res[a] = (+-)lam_tie(a) - (+-)lam_tie(a + 2) [+ beta at the member
        # whose speed row is tight]: the tie row into member j carries S[j] = sum of res over the members from j to the run's
        # end, less beta if the anchored member sits at or behind j; beta = S[head] (the head has no tie row of its own).
        for k in range(2):
            S = np.zeros(nu)
            for t in range(T - 1, -1, -1):
                a = 2 * t + k
                S[a] = res[a] + (S[a + 2] if (t + 1 < T and rep[a + 2] == rep[a]) else 0.0)
            for t in range(T):
                a = 2 * t + k
                if rep[a] != a:
                    continue
                members = [a2 for a2 in range(a, nu, 2) if rep[a2] == a]
                m = next((a2 for a2 in members if tight[2 * a2] or tight[2 * a2 + 1]), None)
                beta = S[a] if m is not None else 0.0
                if m is not None:
                    if tight[2 * m]: lamn[2 * m] = beta
                    else: lamn[2 * m + 1] = -beta
                for j in members[1:]:
                    v = S[j] - (beta if (m is not None and m >= j) else 0.0)
                    q = nb + 2 * (j - 2)
                    if tight[q]: lamn[q] = v
                    else: lamn[q + 1] = -v
        if lane_check is not None:
            tie = np.zeros(nu); bnd = np.zeros(nu)
            for a in range(nu):
                if a >= 2: tie[a] = 1.0 if tight[nb + 2 * (a - 2)] else (-1.0 if tight[nb + 2 * (a - 2) + 1] else 0.0)
                bnd[a] = 1.0 if tight[2 * a] else (-1.0 if tight[2 * a + 1] else 0.0)
            lt, lb = lane_level_multipliers(res[:nu], tie, bnd, nu)
            ref_t = np.zeros(nu); ref_b = np.zeros(nu)
            for a in range(nu):
                if a >= 2: ref_t[a] = lamn[nb + 2 * (a - 2)] + lamn[nb + 2 * (a - 2) + 1] if tie[a] > 0 else (-(-lamn[nb + 2 * (a - 2) + 1]) if tie[a] < 0 else 0.0)
                ref_b[a] = lamn[2 * a] if bnd[a] > 0 else (lamn[2 * a + 1] if bnd[a] < 0 else 0.0)
            scl = 1.0 + np.abs(lamn).max()
            lane_check.append(max(np.abs(lt - ref_t).max(), np.abs(lb - ref_b).max()) / scl)
        x, lam = xn, lamn
    return x, lam, max_it, False


# ---- the same reduction, lane by lane (64-lane arrays, ds_bpermute-style gathers, Hillis-Steele segmented scans): what the HIP
# code has to do, checked here against the matrix form ---------------------------------------------------------------------
NL = 64


def _bperm(v, src, fill=0.0):
    """lane i reads v[src[i]]; lanes whose source is outside the wave read `fill`"""
    src = np.asarray(src)
    okl = (src >= 0) & (src < NL)
    return np.where(okl, v[np.clip(src, 0, NL - 1)], fill)


def lane_level_reduction(K, r, tie, bnd, acc, spd, nu):
    """K (nu, nu) symmetric, r (nu): the system before the tight u rows are eliminated.  tie[a] in {0, +1, -1}: u_a = u_(a-2) +- acc_k
    is tight; bnd[a] in {0, +1, -1}: u_a = +-spd_k is tight (k = a & 1).  Returns (Kred rows in lanes, rred, head, offvec, anchored)."""
    lane = np.arange(NL)
    live = lane < nu
    k = lane & 1
    tie_l = np.zeros(NL); tie_l[:nu] = tie
    bnd_l = np.zeros(NL); bnd_l[:nu] = bnd
    # A. offsets and heads: segmented inclusive scan along the stride-2 chains (flag = 1 at the head of a run)
    v = np.where(live, tie_l * np.where(k == 0, acc[0], acc[1]), 0.0)
    flag = (~live) | (tie_l == 0)
    h = np.where(flag, lane, -1).astype(float)
    for dist in (2, 4, 8, 16, 32):
        vp = _bperm(v, lane - dist); fp = _bperm(flag.astype(float), lane - dist, 1.0) > 0; hp = _bperm(h, lane - dist, -1.0)
        v = np.where(flag, v, v + vp); h = np.where(flag, h, hp); flag = flag | fp
    off = v; head = h.astype(int)
    slot = np.full(NL, np.nan)                                       # an LDS array: members with a tight speed row publish the head's value
    for a in range(nu):
        if bnd_l[a] != 0 and np.isnan(slot[head[a]]):
            slot[head[a]] = bnd_l[a] * spd[a & 1] - off[a]
    anchor = _bperm(slot, head, np.nan)
    anchored = ~np.isnan(anchor) & live
    offvec = np.where(anchored, anchor + off, off)
    # B. rhs -= K offvec (column-uniform values, broadcast reads), then merge / clear the columns, transpose, merge again
    rows = np.zeros((NL, nu)); rows[:nu] = K                          # lane a holds row a in nu registers
    rl = np.zeros(NL); rl[:nu] = r
    rl = rl - rows @ offvec[:nu]
    m = (tie_l[:nu] != 0) & ~anchored[:nu]                            # column c folds into column c - 2 (wave-uniform per column)
    gone = (tie_l[:nu] != 0) | anchored[:nu]                          # column c leaves the system

    def merge(rows):
        rows = rows.copy()
        for c in range(nu - 1, 1, -1):
            if m[c]: rows[:, c - 2] += rows[:, c]
        rows[:, gone] = 0.0
        return rows
    rows = merge(rows)
    t = np.zeros((NL, nu)); t[:nu] = rows[:nu].T                      # through the LDS matrix: lane a reads column a
    rows = merge(t)
    for a in range(nu):
        if gone[a]: rows[a, a] = 1.0                                  # unit rows for the members that left
    # Z' rhs: reverse segmented sums by pointer jumping -- val[a] = sum over the members of a's run from a on, link[a] = "the span
    # collected so far ends on a member that is folded into its predecessor as well"
    cont = np.zeros(NL, bool); cont[:nu] = m                          # lane a is folded into lane a - 2
    val = np.where(live, rl, 0.0)
    link = _bperm(cont.astype(float), lane + 2) > 0
    for dist in (2, 4, 8, 16, 32):
        val, link = val + np.where(link, _bperm(val, lane + dist), 0.0), link & (_bperm(link.astype(float), lane + dist) > 0)
    rl = np.where(gone_l(gone, nu), 0.0, val)
    return rows[:nu], rl[:nu], head[:nu], offvec[:nu], anchored[:nu]


def lane_level_multipliers(res, tie, bnd, nu):
    """Step D lane by lane: res (nu) = -(gradient of the objective without the tight rows) at the new point.  Returns the signed
    multipliers (>= 0 at a correct guess) of the tie row INTO each variable and of the speed row at the anchoring member."""
    lane = np.arange(NL)
    live = lane < nu
    tie_l = np.zeros(NL); tie_l[:nu] = tie
    bnd_l = np.zeros(NL); bnd_l[:nu] = bnd
    flag = (~live) | (tie_l == 0)
    h = np.where(flag, lane, -1).astype(float)
    for dist in (2, 4, 8, 16, 32):
        fp = _bperm(flag.astype(float), lane - dist, 1.0) > 0; hp = _bperm(h, lane - dist, -1.0)
        h = np.where(flag, h, hp); flag = flag | fp
    head = h.astype(int)
    winner = np.full(NL, 1 << 30)
    for a in range(nu):
        if bnd_l[a] != 0: winner[head[a]] = min(winner[head[a]], a)
    mwin = _bperm(winner.astype(float), head, float(1 << 30)).astype(int)
    anchored = live & (mwin < (1 << 30))
    val = np.zeros(NL); val[:nu] = res
    link = (_bperm((tie_l != 0).astype(float), lane + 2) > 0) & live
    for dist in (2, 4, 8, 16, 32):
        val, link = val + np.where(link, _bperm(val, lane + dist), 0.0), link & (_bperm(link.astype(float), lane + dist) > 0)
    beta = np.where(anchored, _bperm(val, head), 0.0)
    v = val - np.where(anchored & (mwin >= lane), beta, 0.0)
    lam_tie = np.where(live & (tie_l != 0), tie_l * v, 0.0)
    lam_bnd = np.where(anchored & (mwin == lane), bnd_l * beta, 0.0)
    return lam_tie[:nu], lam_bnd[:nu]


def gone_l(gone, nu):
    g = np.zeros(NL, bool); g[:nu] = gone
    return g


def job(arg):
    name, b = arg
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    cfg = CONFIGS[name]
    sc = make_scene(cfg, b)
    orc = make_oracle(cfg)
    pbs = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a)
        pbs.append(orc.last_problem)
        return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    rows = []
    prev = None; prev_u = None
    for k, pb in enumerate(pbs):
        H, g, F, f, C, c, Phi, cv = ci.condense(pb)
        s, u, d, info = ci.solve_condensed(pb)                     # the cold interior-point solve: the reference point
        x_ref, lc_ref, lf_ref = info["warm"]
        nu = 2 * pb.T
        step = float(np.abs(u - prev_u).max()) if prev_u is not None else 9.0
        if prev is not None and prev["merit"] <= 1e-12 and prev["step"] < 0.1:          # the kernel's warm-start gate
            x, lam, fac, ok = active_set_solve(H, g, F, f, C, c, pb.ro_obs, prev["warm"][0].copy(), prev["warm"][1].copy())
            xp, lp, facp, okp = active_set_solve_kernel_form(pb, H, g, F, f, C, c, prev["warm"][0].copy(), prev["warm"][1].copy(), project_d=True)
            lchk = []
            xk, lk, fack, okk = active_set_solve_kernel_form(pb, H, g, F, f, C, c, prev["warm"][0].copy(), prev["warm"][1].copy(), lane_check=lchk)
            sw, uw, dw, iw = ci.solve_condensed(pb, warm=prev["warm"])
            e = np.maximum(f - F @ x, 0)
            kkt = max(np.abs(H @ x + g - pb.ro_obs * F.T @ e + C.T @ lam).max() / (1 + np.abs(g).max()),
                      np.maximum(C @ x - c, 0).max() / (1 + np.abs(c).max()), np.maximum(-lam, 0).max()) if ok else np.nan
            rows.append(dict(ok=ok, fac=fac, du=float(np.abs(x[:nu] - x_ref[:nu]).max()) if ok else np.nan, kkt=float(kkt),
                             ipm_warm=iw["iters_total"], tight=int(((lam + (C @ x - c)) > 0).sum()), k=k,
                             kf_ok=okk, kf_fac=fack, kf_du=float(np.abs(xk[:nu] - x[:nu]).max()) if (ok and okk) else np.nan,
                             kf_dl=float(np.abs(lk - lam).max() / (1.0 + np.abs(lam).max())) if (ok and okk) else np.nan,
                             lane=max(lchk) if lchk else 0.0, guesses=len(lchk),
                             pj_ok=okp, pj_fac=facp, pj_du=float(np.abs(xp[:nu] - x_ref[:nu]).max()) if okp else np.nan))
        info["step"] = step
        prev, prev_u = info, u
    return name, len(pbs), rows


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, [(w, b) for w in WORK for b in range(n)]))
    lines = [f"primal-dual active-set iteration from the previous PAN iteration's solution, every QP behind the kernel's warm-start gate, {n} scenes per workload (tests/tools/qp_active_set_study.py)",
             "one 'factorisation' = one equality-constrained solve of the guessed active set; the interior-point column counts its iterations (one factorisation + TWO solves each) on the same QPs"]
    for w in WORK:
        rows = [r for name, _, rs in res if name == w for r in rs]
        total = sum(np_ for name, np_, _ in res if name == w)
        ok = [r for r in rows if r["ok"]]
        fac = np.array([r["fac"] for r in ok])
        lines.append(f"{w}: {total} QPs, {len(rows)} behind the gate, active-set iteration converged on {len(ok)} ({100.0 * len(ok) / max(len(rows), 1):.1f} %)")
        lines.append(f"   factorisations: mean {fac.mean():.2f}, histogram {np.bincount(fac).tolist()}   |   warm interior-point iterations on the same QPs: mean {np.mean([r['ipm_warm'] for r in ok]):.2f}")
        lines.append(f"   |u - u_ipm| median {np.median([r['du'] for r in ok]):.1e} max {max(r['du'] for r in ok):.1e}   KKT residual median {np.median([r['kkt'] for r in ok]):.1e} max {max(r['kkt'] for r in ok):.1e}   tight linear rows: median {int(np.median([r['tight'] for r in ok]))} max {max(r['tight'] for r in ok)}")
        kf = [r for r in rows if r["ok"] and r["kf_ok"]]
        lines.append(f"   kernel form (d eliminated per step, tied runs of controls, one 2T x 2T Cholesky per guess): converged on {sum(r['kf_ok'] for r in rows)}, "
                     f"same guesses ({np.mean([r['kf_fac'] == r['fac'] for r in kf]) * 100:.0f} % with the same count), |u - u_exact| max {max(r['kf_du'] for r in kf):.1e}, multipliers of the tight rows within {max(r['kf_dl'] for r in kf):.1e} (relative)")
        lines.append(f"   lane-level form of the reduction (64-lane arrays, gathers and segmented scans: the docstring's steps A and B) against the matrix form, "
                     f"{sum(r['guesses'] for r in rows)} guesses: largest deviation {max(r['lane'] for r in rows):.1e}")
        pj = [r for r in rows if r["pj_ok"]]
        lines.append(f"   projected-d variant (d clipped after every solve, a d row tight iff d is on the bound and pushed outward): converged on {len(pj)}, "
                     f"factorisations mean {np.mean([r['pj_fac'] for r in pj]):.2f}, |u - u_ipm| max {max(r['pj_du'] for r in pj):.1e}")
        bad = [r for r in rows if not r["ok"]]
        if bad:
            lines.append(f"   not converged in 8 guesses: {len(bad)} (warm interior-point iterations there: mean {np.mean([r['ipm_warm'] for r in bad]):.1f})")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r03_qp_active_set_study.txt"), "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
