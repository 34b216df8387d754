// nrmp_qp.hip -- NRMP step of the PAN loop: parameter build + convex QP solve + stop test,
// one wavefront per scene, fp64.  gfx950 (MI355X) only.
//
// Replaces (reference file:line)
//   robot.generate_state_parameter_value / linear_*_model   neupan/robot/robot.py:239-316
//   NRMP.generate_coefficient_parameter_value               neupan/blocks/nrmp.py:220-261
//   NRMP.forward -> CvxpyLayer (cvxpylayers/diffcp/ECOS)     neupan/blocks/nrmp.py:114-150
//        the problem itself: nrmp.py:263-383, robot.py:142-236
//   PAN.stop_criteria                                        neupan/blocks/pan.py:215-243
//
// The problem (variables s(3,T+1), u(2,T), d(T)):
//   min  sum (q_s s - q_s ref)^2 + sum (p_u u0 - p_u ref_us)^2 + bk/2 |s - nom_s|^2
//        - eta sum d + ro/2 sum_{t,j} max(0, -(fa_tj . s_xy(t+1) - fb_tj - d_t))^2
//   s.t. s(t+1) = A_t s(t) + B_t u(t) + C_t, s(0) = nom_s(0), |u| <= speed, |du| <= acce,
//        max(d_min,0) <= d <= d_max.
// Solved on x = (u, d): states eliminated through the linearised dynamics
// (s(t) = Phi_t u + c_t), hinge rows carried through their stationarity e = lam_f/ro (no
// epigraph variables), Mehrotra predictor-corrector, dense Cholesky of the 3T x 3T reduced
// KKT matrix.  The algorithm is transliterated in oracle/condensed_ipm.py, which the tests
// check against the uncondensed fp64 oracle and HiGHS.
//
// Why one wave: the solve is a serial chain of small dense steps (latency bound); scenes are
// independent, so parallelism comes from the batch -- one 64-lane wave per scene, rows of
// the KKT system owned by lanes, matrices in LDS, cross-lane broadcast by v_readlane.
#include "pan_common.h"

#define QP_THREADS 64
#define QP_MAX_IT 40

__device__ __forceinline__ double shfl_xor_f64(double v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_f64(v, off);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, shfl_xor_f64(v, off));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmin(v, shfl_xor_f64(v, off));
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), l);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), l);
  return __hiloint2double((int)hi, (int)lo);
}
#define LSYNC() __syncthreads()

struct CRow {
  int ia, ib;       // x index with coefficient sa, second index (or -1) with coefficient -sa
  double sa;
  double bound;
  bool act;
};

// linear inequality rows  C x <= c  (robot.py:232-233, nrmp.py:375-376 + nonneg nrmp.py:264)
__device__ __forceinline__ CRow crow(const DevParams& P, int i) {
  const int T = P.T, nu = 2 * T;
  CRow r;
  r.ib = -1;
  if (i < 4 * T) {                       // |u| <= speed_bound
    int v = i >> 1;
    r.ia = v; r.sa = (i & 1) ? -1.0 : 1.0; r.bound = P.speed_bound[v & 1];
  } else if (i < 8 * T - 4) {            // |u(t+1)-u(t)| <= acce_bound
    int q = i - 4 * T, v = q >> 1;
    r.ia = v + 2; r.ib = v; r.sa = (q & 1) ? -1.0 : 1.0; r.bound = P.acce_bound[v & 1];
  } else {                               // max(d_min,0) <= d <= d_max
    int q = i - (8 * T - 4), t = q >> 1;
    r.ia = nu + t; r.sa = (q & 1) ? -1.0 : 1.0;
    r.bound = (q & 1) ? -fmax((double)P.d_min, 0.0) : (double)P.d_max;
  }
  r.act = isfinite(r.bound);
  return r;
}

__global__ __launch_bounds__(QP_THREADS) void nrmp_qp_kernel(
    DevParams P, const float* cur_s_in, const float* cur_u_in,
    const float* __restrict__ ref_s, const float* __restrict__ ref_us, const float* __restrict__ mu_sorted,
    const float* __restrict__ lam_sorted, const float* __restrict__ pts_sorted,
    const float* __restrict__ dist_sorted, const int* __restrict__ count, float* cur_s_out,
    float* cur_u_out, float* __restrict__ cur_d_out, float* __restrict__ out_s,
    float* __restrict__ out_u, float* __restrict__ out_d, float* __restrict__ out_min_distance,
    int* __restrict__ out_iters, float* __restrict__ out_nrmp_points, int* __restrict__ flags,
    float* __restrict__ state, double* __restrict__ qp_info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (flags && flags[b * 4 + 0]) return;

  const int T = P.T, M = P.M, E = P.E, nu = 2 * T;
  const bool obs = M > 0;
  const int n = obs ? 3 * T : 2 * T;
  const int mc = obs ? 10 * T - 4 : 8 * T - 4;
  const int mf = obs ? T * M : 0;
  const int ld = n + 1;                       // padded leading dimension of K
  const double ro = P.ro_obs;

  // ---- LDS carve (doubles) -----------------------------------------------------------
  double* Phi = sm;                           // [T][3][nu]   s(t+1) = Phi[t] u + cv[t]
  double* cv = Phi + (size_t)T * 3 * nu;      // [T][3]
  double* Hm = cv + T * 3;                    // [nu][nu]     constant Hessian block
  double* Km = Hm + (size_t)nu * nu;          // [n][ld]
  double* g = Km + (size_t)n * ld;            // [n]
  double* x = g + n;                          // [n]
  double* xbest = x + n;                      // [n]
  double* dx = xbest + n;                     // [n]
  double* vecn = dx + n;                      // [n] scratch (rhs)
  double* Abc = vecn + n;                     // [T][14]: A02 A12 | B(3x2) | C(3) | pad
  double* sxy = Abc + T * 14;                 // [T][2]  scratch: Phi_xy u (+c)
  double* St = sxy + T * 2;                   // [T][6]  S00 S01 S11 v0 v1 sigma
  double* zt = St + T * 6;                    // [T][3]  z0 z1 zsig
  double* fa = zt + T * 3;                    // [mf][2]
  double* ff = fa + (size_t)mf * 2;           // [mf]   f = fb - fa . c_xy(t+1)
  double* lf = ff + mf;                       // [mf]
  double* wf = lf + mf;
  double* dlf = wf + mf;
  double* dwf = dlf + mf;
  double* tf = dwf + mf;                      // scratch per hinge row
  double* lc = tf + mf;                       // [mc]
  double* wc = lc + mc;
  double* dlc = wc + mc;
  double* dwc = dlc + mc;
  double* tc = dwc + mc;

  const float* s_in = cur_s_in + (size_t)b * 3 * (T + 1);
  const float* u_in = cur_u_in + (size_t)b * 2 * T;
  const float* rs = ref_s + (size_t)b * 3 * (T + 1);
  const float* rus = ref_us + (size_t)b * T;

  // ---- A_t, B_t, C_t in the reference's fp32 rounding sequence (robot.py:272-316) -------
  for (int t = lane; t < T; t += QP_THREADS) {
    float phi = s_in[2 * (T + 1) + t], v = u_in[t], psi = u_in[T + t];
    const float dt32 = P.dt32;
    double* o = Abc + t * 14;
    float A02 = 0.f, A12 = 0.f, B00, B01 = 0.f, B10, B11 = 0.f, B20 = 0.f, B21 = 0.f, C0, C1, C2 = 0.f;
    if (P.kin == 2) {                      // omni: phi := u[1]
      double sp = sin((double)psi), cp = cos((double)psi);
      B00 = (float)(cp * P.dt); B10 = (float)(sp * P.dt);
      B01 = __fmul_rn(__fmul_rn(-v, (float)sp), dt32);
      B11 = __fmul_rn(__fmul_rn(v, (float)cp), dt32);
      C0 = __fmul_rn(__fmul_rn(__fmul_rn(psi, v), (float)sp), dt32);
      C1 = __fmul_rn(__fmul_rn(__fmul_rn(-psi, v), (float)cp), dt32);
    } else {
      double sp = sin((double)phi), cp = cos((double)phi);
      A02 = __fmul_rn(__fmul_rn(-v, dt32), (float)sp);
      A12 = __fmul_rn(__fmul_rn(v, dt32), (float)cp);
      B00 = (float)(cp * P.dt); B10 = (float)(sp * P.dt);
      C0 = __fmul_rn(__fmul_rn(__fmul_rn(phi, v), (float)sp), dt32);
      C1 = __fmul_rn(__fmul_rn(__fmul_rn(-phi, v), (float)cp), dt32);
      if (P.kin == 0) {
        B21 = dt32;
      } else {                             // acker
        double cps = cos((double)psi); cps = cps * cps;
        float den = (float)(P.L * cps);
        B20 = (float)(tan((double)psi) * P.dt / P.L);
        B21 = __fdiv_rn(__fmul_rn(v, dt32), den);
        C2 = __fdiv_rn(__fmul_rn(__fmul_rn(-psi, v), dt32), den);
      }
    }
    o[0] = A02; o[1] = A12;
    o[2] = B00; o[3] = B01; o[4] = B10; o[5] = B11; o[6] = B20; o[7] = B21;
    o[8] = C0; o[9] = C1; o[10] = C2;
  }
  LSYNC();

  // ---- Phi recursion: Phi[t] = A_t Phi[t-1] + [B_t at cols 2t,2t+1]; A = I + e0 A02 e2' + e1 A12 e2'
  for (int t = 0; t < T; ++t) {
    const double* o = Abc + t * 14;
    double* Pt = Phi + (size_t)t * 3 * nu;
    const double* Pp = Pt - 3 * nu;
    for (int c = lane; c < nu; c += QP_THREADS) {
      double p0 = 0, p1 = 0, p2 = 0;
      if (t > 0) { p0 = Pp[c]; p1 = Pp[nu + c]; p2 = Pp[2 * nu + c]; }
      double n0 = p0 + o[0] * p2, n1 = p1 + o[1] * p2, n2 = p2;
      if (c == 2 * t) { n0 += o[2]; n1 += o[4]; n2 += o[6]; }
      if (c == 2 * t + 1) { n0 += o[3]; n1 += o[5]; n2 += o[7]; }
      Pt[c] = n0; Pt[nu + c] = n1; Pt[2 * nu + c] = n2;
    }
    if (lane == 0) {
      double c0, c1, c2;
      if (t == 0) { c0 = s_in[0]; c1 = s_in[T + 1]; c2 = s_in[2 * (T + 1)]; }
      else { c0 = cv[(t - 1) * 3]; c1 = cv[(t - 1) * 3 + 1]; c2 = cv[(t - 1) * 3 + 2]; }
      cv[t * 3 + 0] = c0 + o[0] * c2 + o[8];
      cv[t * 3 + 1] = c1 + o[1] * c2 + o[9];
      cv[t * 3 + 2] = c2 + o[10];
    }
    LSYNC();
  }

  // ---- cost: H (constant block), g --------------------------------------------------------
  const double qs0 = P.q_s[0], qs1 = P.q_s[1], qs2 = P.q_s[2];
  const double m2 = (P.kin == 2) ? 0.0 : 1.0;            // omni: theta row not in the state cost
  const double W0 = 2.0 * qs0 * qs0 + P.bk, W1 = 2.0 * qs1 * qs1 + P.bk, W2 = 2.0 * m2 * qs2 * qs2 + P.bk;
  const double pu = P.p_u;
  for (int idx = lane; idx < nu * nu; idx += QP_THREADS) {
    int a = idx / nu, c = idx - a * nu;
    double acc = 0;
    int t0 = (a > c ? a : c) >> 1;
    for (int t = t0; t < T; ++t) {
      const double* Pt = Phi + (size_t)t * 3 * nu;
      acc += W0 * Pt[a] * Pt[c] + W1 * Pt[nu + a] * Pt[nu + c] + W2 * Pt[2 * nu + a] * Pt[2 * nu + c];
    }
    if (a == c && !(a & 1)) acc += 2.0 * pu * pu;
    Hm[idx] = acc;
  }
  for (int a = lane; a < n; a += QP_THREADS) {
    double acc = 0;
    if (a < nu) {
      for (int t = a >> 1; t < T; ++t) {
        const double* Pt = Phi + (size_t)t * 3 * nu;
        const double* c = cv + t * 3;
        // gamma_a = q_s * ref_s is an fp32 product in the reference (nrmp.py:158)
        double r0 = (double)__fmul_rn(P.q_s[0], rs[t + 1]);
        double r1 = (double)__fmul_rn(P.q_s[1], rs[(T + 1) + t + 1]);
        double r2 = (double)__fmul_rn(P.q_s[2], rs[2 * (T + 1) + t + 1]);
        double l0 = 2.0 * qs0 * (qs0 * c[0] - r0) + P.bk * (c[0] - (double)s_in[t + 1]);
        double l1 = 2.0 * qs1 * (qs1 * c[1] - r1) + P.bk * (c[1] - (double)s_in[(T + 1) + t + 1]);
        double l2 = 2.0 * m2 * qs2 * (qs2 * c[2] - r2) + P.bk * (c[2] - (double)s_in[2 * (T + 1) + t + 1]);
        acc += Pt[a] * l0 + Pt[nu + a] * l1 + Pt[2 * nu + a] * l2;
      }
      if (!(a & 1)) acc += -2.0 * pu * (double)__fmul_rn(P.p_u, rus[a >> 1]);
    } else {
      acc = -(double)P.eta;
    }
    g[a] = acc;
  }

  // ---- hinge rows: fa = lam', fb = lam'.p + mu'.h in fp32 (nrmp.py:244-259), slice t+1 ----
  for (int i = lane; i < mf; i += QP_THREADS) {
    int t = i / M, j = i - t * M;
    size_t row = ((size_t)b * (T + 1) + (t + 1)) * M + j;
    double a0 = 0, a1 = 0, fb = 0;
    if (count[(size_t)b * (T + 1) + t + 1] > 0) {
      float l0 = lam_sorted[row * 2], l1 = lam_sorted[row * 2 + 1];
      float tmp = fmaf(l1, pts_sorted[row * 2 + 1], __fmul_rn(l0, pts_sorted[row * 2]));
      float mh = 0.f;
      for (int e = 0; e < E; ++e) mh = fmaf(mu_sorted[row * E + e], P.h[e], mh);
      a0 = l0; a1 = l1; fb = (double)__fadd_rn(tmp, mh);
    }
    fa[i * 2] = a0; fa[i * 2 + 1] = a1;
    ff[i] = fb - (a0 * cv[t * 3] + a1 * cv[t * 3 + 1]);
  }

  // ---- starting point ------------------------------------------------------------------
  for (int a = lane; a < n; a += QP_THREADS)
    x[a] = (a < nu) ? 0.0 : 0.5 * (fmax((double)P.d_min, 0.0) + (double)P.d_max);
  LSYNC();
  for (int a = lane; a < n; a += QP_THREADS) xbest[a] = x[a];
  for (int i = lane; i < mc; i += QP_THREADS) {
    CRow r = crow(P, i);
    double cx = r.sa * x[r.ia] - (r.ib >= 0 ? r.sa * x[r.ib] : 0.0);
    lc[i] = r.act ? 1.0 : 0.0;
    wc[i] = r.act ? fmax(r.bound - cx, 1.0) : 1.0;
  }
  for (int i = lane; i < mf; i += QP_THREADS) {
    int t = i / M;
    lf[i] = 1.0;
    wf[i] = fmax(-x[nu + t] - ff[i] + 1.0 / ro, 1.0);     // F x - f + lf/ro at u = 0
  }
  double gmax = 0, cmax = 0;
  for (int a = lane; a < n; a += QP_THREADS) gmax = fmax(gmax, fabs(g[a]));
  for (int i = lane; i < mc; i += QP_THREADS) { CRow r = crow(P, i); if (r.act) cmax = fmax(cmax, fabs(r.bound)); }
  const double scale_d = 1.0 + wave_max(gmax), scale_p = 1.0 + wave_max(cmax);
  int m_act = 0;
  for (int i = lane; i < mc; i += QP_THREADS) m_act += crow(P, i).act ? 1 : 0;
  const double m_tot = fmax(wave_sum((double)m_act) + (double)mf, 1.0);
  LSYNC();

  double best_merit = 1e300, last_mu = 0;
  int best_it = 0, stall = 0, status = 0, it = 0;

  for (it = 0; it <= QP_MAX_IT; ++it) {
    // ---- residuals ---------------------------------------------------------------------
    // s_xy(t+1) - c = Phi_xy[t] u
    for (int q = lane; q < 2 * T; q += QP_THREADS) {
      int t = q >> 1, k = q & 1;
      const double* Pr = Phi + (size_t)t * 3 * nu + k * nu;
      double acc = 0;
      for (int c = 0; c < 2 * (t + 1); ++c) acc += Pr[c] * x[c];
      sxy[q] = acc;
    }
    LSYNC();
    double gap = 0, r3max = 0, r2max = 0;
    for (int i = lane; i < mf; i += QP_THREADS) {       // tf := r3 = F x - f + lf/ro - wf
      int t = i / M;
      double r3 = fa[i * 2] * sxy[2 * t] + fa[i * 2 + 1] * sxy[2 * t + 1] - x[nu + t] - ff[i] + lf[i] / ro - wf[i];
      tf[i] = r3;
      r3max = fmax(r3max, fabs(r3));
      gap += lf[i] * wf[i];
    }
    for (int i = lane; i < mc; i += QP_THREADS) {       // tc := r2 = C x + wc - c
      CRow r = crow(P, i);
      double r2 = 0;
      if (r.act) {
        double cx = r.sa * x[r.ia] - (r.ib >= 0 ? r.sa * x[r.ib] : 0.0);
        r2 = cx + wc[i] - r.bound;
        gap += lc[i] * wc[i];
      }
      tc[i] = r2;
      r2max = fmax(r2max, fabs(r2));
    }
    LSYNC();
    // z_t = sum_j lf fa ; zsig = sum_j lf     (for F' lf)
    for (int t = lane; t < T && obs; t += QP_THREADS) {
      double z0 = 0, z1 = 0, zs = 0;
      for (int j = 0; j < M; ++j) { int i = t * M + j; z0 += lf[i] * fa[i * 2]; z1 += lf[i] * fa[i * 2 + 1]; zs += lf[i]; }
      zt[t * 3] = z0; zt[t * 3 + 1] = z1; zt[t * 3 + 2] = zs;
    }
    LSYNC();
    double r1max = 0;
    for (int a = lane; a < n; a += QP_THREADS) {        // vecn := r1 = H x + g + C' lc - F' lf
      double acc = g[a];
      if (a < nu) {
        for (int c = 0; c < nu; ++c) acc += Hm[a * nu + c] * x[c];
        if (obs)
          for (int t = a >> 1; t < T; ++t) {
            const double* Pt = Phi + (size_t)t * 3 * nu;
            acc -= Pt[a] * zt[t * 3] + Pt[nu + a] * zt[t * 3 + 1];
          }
        int t = a >> 1;
        acc += lc[2 * a] - lc[2 * a + 1];                                   // speed rows
        if (t >= 1) { int q = 4 * T + 2 * (a - 2); acc += lc[q] - lc[q + 1]; }       // rate rows, +x_a
        if (t <= T - 2) { int q = 4 * T + 2 * a; acc -= lc[q] - lc[q + 1]; }         // rate rows, -x_a
      } else {
        int t = a - nu;
        acc += zt[t * 3 + 2];
        int q = 8 * T - 4 + 2 * t;
        acc += lc[q] - lc[q + 1];
      }
      vecn[a] = acc;
      r1max = fmax(r1max, fabs(acc));
    }
    const double mu = wave_sum(gap) / m_tot;
    const double merit = fmax(fmax(wave_max(r1max) / scale_d, wave_max(fmax(r2max, r3max)) / scale_p), mu);
    last_mu = mu;
    if (!(merit == merit) || !(merit < 1e300)) { status = 2; break; }
    if (merit < best_merit) {
      best_merit = merit; best_it = it; stall = 0;
      for (int a = lane; a < n; a += QP_THREADS) xbest[a] = x[a];
    } else {
      ++stall;
    }
    if (merit <= 1e-12 || stall >= 3 || it == QP_MAX_IT || mu < 1e-15) break;
    LSYNC();

    // ---- reduced KKT matrix K = H + C' Dc C + F' Df F ------------------------------------
    for (int t = lane; t < T && obs; t += QP_THREADS) {
      double s00 = 0, s01 = 0, s11 = 0, v0 = 0, v1 = 0, sg = 0;
      for (int j = 0; j < M; ++j) {
        int i = t * M + j;
        double D = lf[i] / (wf[i] + lf[i] / ro);
        double a0 = fa[i * 2], a1 = fa[i * 2 + 1];
        s00 += D * a0 * a0; s01 += D * a0 * a1; s11 += D * a1 * a1; v0 += D * a0; v1 += D * a1; sg += D;
      }
      double* S = St + t * 6;
      S[0] = s00; S[1] = s01; S[2] = s11; S[3] = v0; S[4] = v1; S[5] = sg;
    }
    LSYNC();
    for (int idx = lane; idx < n * n; idx += QP_THREADS) {
      int a = idx / n, c = idx - a * n;
      if (c > a) continue;                           // lower triangle
      double acc = 0;
      if (a < nu) {                                  // uu block
        acc = Hm[a * nu + c];
        if (obs)
          for (int t = a >> 1; t < T; ++t) {
            const double* Pt = Phi + (size_t)t * 3 * nu;
            const double* S = St + t * 6;
            double ya = S[0] * Pt[a] + S[1] * Pt[nu + a], yb = S[1] * Pt[a] + S[2] * Pt[nu + a];
            acc += ya * Pt[c] + yb * Pt[nu + c];
          }
        if (a == c) {
          int t = a >> 1;
          double dsum = lc[2 * a] / wc[2 * a] + lc[2 * a + 1] / wc[2 * a + 1];
          if (t >= 1) { int q = 4 * T + 2 * (a - 2); dsum += lc[q] / wc[q] + lc[q + 1] / wc[q + 1]; }
          if (t <= T - 2) { int q = 4 * T + 2 * a; dsum += lc[q] / wc[q] + lc[q + 1] / wc[q + 1]; }
          acc += dsum;
        } else if (a == c + 2) {                     // rate rows couple u_k(t+1), u_k(t)
          int q = 4 * T + 2 * c;
          acc -= lc[q] / wc[q] + lc[q + 1] / wc[q + 1];
        }
      } else {
        int t = a - nu;
        if (c < nu) {                                // du block: -Phi_xy[t][:,c] . v_t
          if ((c >> 1) <= t) {
            const double* Pt = Phi + (size_t)t * 3 * nu;
            acc = -(Pt[c] * St[t * 6 + 3] + Pt[nu + c] * St[t * 6 + 4]);
          }
        } else if (c == a) {
          int q = 8 * T - 4 + 2 * t;
          acc = St[t * 6 + 5] + lc[q] / wc[q] + lc[q + 1] / wc[q + 1];
        }
      }
      Km[a * ld + c] = acc;
    }
    LSYNC();

    // ---- Cholesky K = L L' (left-looking, lane = row) -------------------------------------
    bool chol_ok = true;
    for (int k = 0; k < n; ++k) {
      const double* Lk = Km + (size_t)k * ld;
      double piv = 0;
      for (int i = k + lane; i < n; i += QP_THREADS) {
        const double* Li = Km + (size_t)i * ld;
        double acc0 = Li[k], acc1 = 0;
        int p = 0;
        for (; p + 1 < k; p += 2) { acc0 -= Li[p] * Lk[p]; acc1 -= Li[p + 1] * Lk[p + 1]; }
        if (p < k) acc0 -= Li[p] * Lk[p];
        double v = acc0 + acc1;
        if (i == k) piv = v;
        Km[(size_t)i * ld + k] = v;                  // raw column, scaled below
      }
      piv = readlane_f64(piv, 0);                    // row k is owned by lane 0 of this sweep
      if (!(piv > 0.0)) { chol_ok = false; break; }
      double inv = 1.0 / sqrt(piv);
      LSYNC();
      for (int i = k + lane; i < n; i += QP_THREADS) Km[(size_t)i * ld + k] *= inv;   // L[k][k] = sqrt(piv)
      LSYNC();
    }
    if (!chol_ok) { status = 3; break; }

    // ---- predictor / corrector -----------------------------------------------------------
    double sigma_mu = 0;
    double alpha = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      // per-row terms of the rhs:  tcw = (lc r2 - r4c)/wc ; tfw = (r4f + lf r3)/(wf + lf/ro)
      // pass 0: r4 = lam w ; pass 1: r4 = lam w + dw dl - sigma mu.  r2 in tc, r3 in tf.
      for (int t = lane; t < T && obs; t += QP_THREADS) {
        double z0 = 0, z1 = 0, zs = 0;
        for (int j = 0; j < M; ++j) {
          int i = t * M + j;
          double r4 = lf[i] * wf[i] + (pass ? dwf[i] * dlf[i] - sigma_mu : 0.0);
          double w = (r4 + lf[i] * tf[i]) / (wf[i] + lf[i] / ro);
          z0 += w * fa[i * 2]; z1 += w * fa[i * 2 + 1]; zs += w;
        }
        zt[t * 3] = z0; zt[t * 3 + 1] = z1; zt[t * 3 + 2] = zs;
      }
      LSYNC();
      auto tcw = [&](int q) -> double {
        double wq = wc[q];
        double r4 = lc[q] * wq + (pass ? dwc[q] * dlc[q] - sigma_mu : 0.0);
        return (lc[q] == 0.0 && !crow(P, q).act) ? 0.0 : (lc[q] * tc[q] - r4) / wq;
      };
      double rr = 0;                                 // rhs entry owned by this lane (row = lane; n <= 64)
      if (lane < n) {
        const int a = lane;
        double acc = -vecn[a];
        if (a < nu) {
          if (obs)
            for (int t = a >> 1; t < T; ++t) {         // - F' tfw, F row = [fa.Phi_xy, -e_t]
              const double* Pt = Phi + (size_t)t * 3 * nu;
              acc -= Pt[a] * zt[t * 3] + Pt[nu + a] * zt[t * 3 + 1];
            }
          int t = a >> 1;
          acc -= tcw(2 * a) - tcw(2 * a + 1);
          if (t >= 1) { int q = 4 * T + 2 * (a - 2); acc -= tcw(q) - tcw(q + 1); }
          if (t <= T - 2) { int q = 4 * T + 2 * a; acc += tcw(q) - tcw(q + 1); }
        } else {
          int t = a - nu;
          acc += zt[t * 3 + 2];
          int q = 8 * T - 4 + 2 * t;
          acc -= tcw(q) - tcw(q + 1);
        }
        rr = acc;
      }
      // forward substitution L y = rhs (lane i owns entry i; y_k broadcast by v_readlane)
      for (int k = 0; k < n; ++k) {
        double yk = readlane_f64(rr, k) / Km[(size_t)k * ld + k];
        if (lane == k) rr = yk;
        if (lane > k && lane < n) rr -= Km[(size_t)lane * ld + k] * yk;
      }
      // backward substitution L' dx = y
      for (int k = n - 1; k >= 0; --k) {
        double xk = readlane_f64(rr, k) / Km[(size_t)k * ld + k];
        if (lane == k) rr = xk;
        if (lane < k) rr -= Km[(size_t)k * ld + lane] * xk;
      }
      if (lane < n) dx[lane] = rr;
      LSYNC();
      // directions of the multipliers / slacks, and the step length
      for (int q = lane; q < 2 * T; q += QP_THREADS) {
        int t = q >> 1, k = q & 1;
        const double* Pr = Phi + (size_t)t * 3 * nu + k * nu;
        double acc = 0;
        for (int c = 0; c < 2 * (t + 1); ++c) acc += Pr[c] * dx[c];
        sxy[q] = acc;
      }
      LSYNC();
      double amax = 1.0, gap_aff = 0;
      for (int i = lane; i < mf; i += QP_THREADS) {
        int t = i / M;
        double Fdx = fa[i * 2] * sxy[2 * t] + fa[i * 2 + 1] * sxy[2 * t + 1] - dx[nu + t];
        double r4 = lf[i] * wf[i] + (pass ? dwf[i] * dlf[i] - sigma_mu : 0.0);
        double dl = -(r4 + lf[i] * tf[i] + lf[i] * Fdx) / (wf[i] + lf[i] / ro);
        double dw = Fdx + dl / ro + tf[i];
        if (dl < 0) amax = fmin(amax, -lf[i] / dl);
        if (dw < 0) amax = fmin(amax, -wf[i] / dw);
        dlf[i] = dl; dwf[i] = dw;
      }
      for (int i = lane; i < mc; i += QP_THREADS) {
        CRow r = crow(P, i);
        double dl = 0, dw = 0;
        if (r.act) {
          double Cdx = r.sa * dx[r.ia] - (r.ib >= 0 ? r.sa * dx[r.ib] : 0.0);
          double r4 = lc[i] * wc[i] + (pass ? dwc[i] * dlc[i] - sigma_mu : 0.0);
          dw = -tc[i] - Cdx;
          dl = (-r4 - lc[i] * dw) / wc[i];
          if (dl < 0) amax = fmin(amax, -lc[i] / dl);
          if (dw < 0) amax = fmin(amax, -wc[i] / dw);
        }
        dlc[i] = dl; dwc[i] = dw;
      }
      amax = wave_min(amax);
      if (pass == 0) {
        for (int i = lane; i < mf; i += QP_THREADS) gap_aff += (lf[i] + amax * dlf[i]) * (wf[i] + amax * dwf[i]);
        for (int i = lane; i < mc; i += QP_THREADS) gap_aff += (lc[i] + amax * dlc[i]) * (wc[i] + amax * dwc[i]);
        double mu_aff = wave_sum(gap_aff) / m_tot;
        double sg = mu_aff / mu;
        sigma_mu = sg * sg * sg * mu;
      } else {
        alpha = fmin(1.0, 0.995 * amax);
      }
      LSYNC();
    }
    for (int a = lane; a < n; a += QP_THREADS) x[a] += alpha * dx[a];
    for (int i = lane; i < mf; i += QP_THREADS) { lf[i] += alpha * dlf[i]; wf[i] += alpha * dwf[i]; }
    for (int i = lane; i < mc; i += QP_THREADS) { lc[i] += alpha * dlc[i]; wc[i] += alpha * dwc[i]; }
    LSYNC();
  }
  LSYNC();

  // ---- write the solution (fp64 -> fp32, nrmp.py:145-148) ------------------------------------
  float* so = cur_s_out + (size_t)b * 3 * (T + 1);
  float* uo = cur_u_out + (size_t)b * 2 * T;
  for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) {
    int k = q / (T + 1), t = q - k * (T + 1);
    double v;
    if (t == 0) v = s_in[k * (T + 1)];
    else {
      const double* Pr = Phi + (size_t)(t - 1) * 3 * nu + k * nu;
      v = cv[(t - 1) * 3 + k];
      for (int c = 0; c < 2 * t; ++c) v += Pr[c] * xbest[c];
    }
    float fv = (float)v;
    // staged in LDS: cur_s_out may alias cur_s_in, which other lanes are still reading
    reinterpret_cast<float*>(Km)[q] = fv;
  }
  LSYNC();
  for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) {
    float fv = reinterpret_cast<float*>(Km)[q];
    so[q] = fv;
    if (out_s) out_s[(size_t)b * 3 * (T + 1) + q] = fv;
  }
  for (int q = lane; q < 2 * T; q += QP_THREADS) {
    int k = q / T, t = q - k * T;
    float fv = (float)xbest[2 * t + k];
    uo[q] = fv;
    if (out_u) out_u[(size_t)b * 2 * T + q] = fv;
  }
  if (obs)
    for (int t = lane; t < T; t += QP_THREADS) {
      float fv = (float)xbest[nu + t];
      if (cur_d_out) cur_d_out[(size_t)b * T + t] = fv;
      if (out_d) out_d[(size_t)b * T + t] = fv;
    }
  if (qp_info && lane == 0) {
    qp_info[b * 4 + 0] = best_it; qp_info[b * 4 + 1] = best_merit; qp_info[b * 4 + 2] = last_mu; qp_info[b * 4 + 3] = status;
  }

  // ---- per-forward outputs of the last executed iteration -------------------------------------
  const int cnt0 = count ? count[(size_t)b * (T + 1)] : 0;
  if (out_min_distance && lane == 0)
    out_min_distance[b] = (obs && cnt0 > 0) ? dist_sorted[(size_t)b * (T + 1) * M] : __builtin_inff();
  if (out_nrmp_points && obs)
    for (int q = lane; q < 2 * M; q += QP_THREADS) {
      int k = q / M, j = q - k * M;
      out_nrmp_points[(size_t)b * 2 * M + q] = cnt0 > 0 ? pts_sorted[((size_t)b * (T + 1) * M + j) * 2 + k] : 0.f;
    }

  // ---- stop criterion (pan.py:215-243); state persists across forward calls ---------------------
  if (state && flags) {
    const size_t nsf = npa_state_floats(T, M > 0 ? M : 1, E);
    float* st = state + (size_t)b * nsf;
    float* ps = st;
    float* pu_ = ps + 3 * (T + 1);
    float* pmu = pu_ + 2 * T;
    float* plam = pmu + (size_t)(T + 1) * (M > 0 ? M : 1) * E;
    int* pint = reinterpret_cast<int*>(plam + (size_t)(T + 1) * (M > 0 ? M : 1) * 2);
    const int valid = pint[0], prev_n = pint[1];
    const bool have = obs && cnt0 > 0;
    double acc_s = 0, acc_u = 0, acc_mu = 0, acc_lam = 0;
    int eff = 0;
    if (valid) {
      if (!have || prev_n == 0) {
        for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) { double d = (double)so[q] - (double)ps[q]; acc_s += d * d; }
        for (int q = lane; q < 2 * T; q += QP_THREADS) { double d = (double)uo[q] - (double)pu_[q]; acc_u += d * d; }
      } else {
        eff = cnt0 < prev_n ? cnt0 : prev_n;
        for (int q = lane; q < (T + 1) * eff; q += QP_THREADS) {
          int t = q / eff, j = q - t * eff;
          size_t row = ((size_t)b * (T + 1) + t) * M + j;
          size_t prow = (size_t)t * M + j;
          for (int e = 0; e < E; ++e) { double d = (double)mu_sorted[row * E + e] - (double)pmu[prow * E + e]; acc_mu += d * d; }
          for (int k = 0; k < 2; ++k) { double d = (double)lam_sorted[row * 2 + k] - (double)plam[prow * 2 + k]; acc_lam += d * d; }
        }
      }
    }
    acc_s = wave_sum(acc_s); acc_u = wave_sum(acc_u); acc_mu = wave_sum(acc_mu); acc_lam = wave_sum(acc_lam);
    LSYNC();
    // remember the current iterate
    for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) ps[q] = so[q];
    for (int q = lane; q < 2 * T; q += QP_THREADS) pu_[q] = uo[q];
    if (have)
      for (int q = lane; q < (T + 1) * M; q += QP_THREADS) {
        size_t row = (size_t)b * (T + 1) * M + q;
        for (int e = 0; e < E; ++e) pmu[(size_t)q * E + e] = mu_sorted[row * E + e];
        plam[(size_t)q * 2] = lam_sorted[row * 2]; plam[(size_t)q * 2 + 1] = lam_sorted[row * 2 + 1];
      }
    if (lane == 0) {
      pint[0] = 1;
      pint[1] = have ? cnt0 : 0;
      int stop = 0;
      if (valid) {
        float diff;
        if (!have || prev_n == 0) diff = (float)(acc_s + acc_u);
        else {
          float md = (float)sqrt(acc_mu) / (float)eff, ldv = (float)sqrt(acc_lam) / (float)eff;
          diff = md * md + ldv * ldv;
        }
        stop = diff < P.iter_threshold ? 1 : 0;
      }
      flags[b * 4 + 1] += 1;
      if (stop) flags[b * 4 + 0] = 1;
      if (out_iters) out_iters[b] = flags[b * 4 + 1];
    }
  }
}

extern "C" size_t npa_qp_shmem_bytes(int T, int M) {
  const bool obs = M > 0;
  size_t nu = 2 * T, n = obs ? 3 * T : 2 * T, mc = obs ? 10 * T - 4 : 8 * T - 4, mf = obs ? (size_t)T * M : 0;
  size_t d = (size_t)T * 3 * nu + T * 3 + nu * nu + n * (n + 1) + 5 * n + T * 14 + T * 2 + T * 6 + T * 3 + mf * 2 +
             6 * mf + 5 * mc;
  return d * sizeof(double);
}

extern "C" hipError_t npa_launch_qp(const DevParams& P, int batch, const float* cur_s_in, const float* cur_u_in,
                                    const float* ref_s, const float* ref_us, const float* mu_sorted,
                                    const float* lam_sorted, const float* pts_sorted, const float* dist_sorted,
                                    const int* count, float* cur_s_out, float* cur_u_out, float* cur_d_out,
                                    float* out_s, float* out_u, float* out_d, float* out_min_distance,
                                    int* out_iters, float* out_nrmp_points, int* flags, float* state,
                                    double* qp_info, hipStream_t stream) {
  size_t shmem = npa_qp_shmem_bytes(P.T, P.M);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(nrmp_qp_kernel, dim3(batch), dim3(QP_THREADS), shmem, stream, P, cur_s_in, cur_u_in, ref_s,
                     ref_us, mu_sorted, lam_sorted, pts_sorted, dist_sorted, count, cur_s_out, cur_u_out, cur_d_out,
                     out_s, out_u, out_d, out_min_distance, out_iters, out_nrmp_points, flags, state, qp_info);
  return hipGetLastError();
}
