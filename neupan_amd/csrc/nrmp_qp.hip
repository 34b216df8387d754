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
//
// Algorithm (transliterated in oracle/condensed_ipm.py, which the tests check against the
// uncondensed fp64 oracle and HiGHS): states eliminated through the linearised dynamics
// (s(t+1) = Phi_t u + c_t), hinge rows carried through their own stationarity condition
// e = lam_f/ro (no epigraph variables), Mehrotra predictor-corrector on x = (u, d).  Each
// Newton system is reduced once more by eliminating d (its block of the KKT matrix is
// diagonal), which leaves a dense SPD 2T x 2T system in u:
//      K' = H + C_u' D C_u + sum_t Phi_xy(t)' S'_t Phi_xy(t),   S'_t = S_t - v_t v_t'/kappa_t
// factored by a Cholesky with one matrix row per lane (right-looking in the register-resident instantiations, pivot
// chain on v_readlane; left-looking over the LDS matrix in the generic one).
//
// Why one wave per scene: the solve is a serial chain of small dense steps (latency bound);
// scenes are independent, so throughput comes from the batch.  Matrices live in LDS with odd
// leading dimensions (conflict-free row-per-lane access), cross-lane broadcast is v_readlane,
// wave reductions are DPP (quad_perm / row_mirror) + 4 readlanes, and every division in a
// serial loop is replaced by a reciprocal computed once per iteration.
#include "pan_common.h"
#include "aset_reduce.h"
#include <hip/hip_ext.h>
#include <cstdlib>

#define QP_THREADS 64          // lanes cooperating on one scene (one wavefront)
#ifndef NPA_QP_WAVES
#define NPA_QP_WAVES 2          // waves per SIMD the register allocation aims at (3: the 168-register experiment of DESIGN.md 3.3:
                               // 35 registers spill, +3 % throughput, -8 % sequential -- measured, not shipped)
#endif
#define QP_MAX_IT 40
// per-step records in LDS, one lane per horizon step: strides chosen so that ten (twenty) lanes hit distinct banks.  With
// the natural strides -- 12 doubles for the linearisation, 8 for the step sums -- steps 0 / 8 (and 0 / 4 / 8) shared a bank on
// every 64-bit access (bank = dword address mod 64 for reads, mod 32 for writes)
#define QP_ABC_LD 14           // [T][14]: A02 A12 B00 B01 B10 B11 B20 B21 C0 C1 C2 (even: read two at a time)
#define QP_ST_LD 9             // [T][9]:  S'00 S'01 S'11 v0 v1 sigma 1/kappa r1_d
#define QP_WARM_DELTA 0.003    // floor of the multipliers / slacks taken over from the previous solve
// Interior-point heuristics (tuned on the QPs of the four benchmark workloads with the CPU transliteration of this method,
// tests/tools/qp_step_study.py -> profiles/r03_qp_step_study.txt; oracle/condensed_ipm.py carries the same constants):
#define QP_STEP_ETA 0.995      // fraction of the step to the boundary, RAISED towards 1 as the gap closes: eta = max(0.995, 1 - mu),
#define QP_STEP_CAP 1e-6       //   never above 1 - 1e-6.  The fixed 0.995 made the end game linear (x 0.005 per iteration)
#define QP_START_MU 3.0        // cold start: multipliers = 3 / slack (every row starts on the central path of mu = 3)
#ifndef QP_CHOL_LOOK
#define QP_CHOL_LOOK 3         // columns behind the pivot whose trailing update is broadcast with v_readlane (the rest: LDS, one pivot late)
#endif
#define QP_ASET_FIRST_MAX 0.05 // the attempt is not made from a warm point whose seeded merit is above this (most of those cycle: 85 % of the failures)
#define QP_ASET_MAX_GUESS 2     // factorisations the active-set iteration may spend before the interior-point warm start takes over
#define QP_ASET_TOL 1e-13      // what may be left of the scaled dual residual at a guess that repeated
#define QP_RETRY_MERIT 1e-9    // a cold solve that ends above this is repeated once from round 2's start (unit multipliers)
#define QP_SIGMA_MU_MIN 1e-15  // floor of the centring target: a gap driven to 1e-20 leaves the Newton matrix too ill
                               //   conditioned for the residuals to follow (solves that ended at 1e-11: 8 -> 1 of 640)
// the cold starting point: u = 0, d mid-range, slacks >= 1, multipliers QP_START_MU / slack -- or, for the SECOND cold attempt
// of a solve whose first one jammed (cold_alt), unit multipliers, round 2's start -- (a macro: used before the loop and, in the
// instantiations with warm start, again at the loop top when a warm attempt is dropped)
#define QP_COLD_INIT()                                                                            \
  do {                                                                                            \
    for (int a = lane; a < nu; a += QP_THREADS) { xu[a] = 0.0; xbest[a] = 0.0; }                  \
    for (int t = lane; t < T; t += QP_THREADS) { xd[t] = d0; xbest[nu + t] = d0; dxd[t] = 0.0; }  \
    for (int p = lane; p < npc; p += QP_THREADS) {                                                \
      const PairC c = PAIR_C(p);                                                                  \
      const double cx = p >= npu ? d0 : 0.0;      /* c'x at the cold point */                     \
      const bool on = c.actf != 0.0;                                                              \
      const double w0p = on ? fmax(c.bp - cx, 1.0) : 1.0, w0m = on ? fmax(c.bm + cx, 1.0) : 1.0;  \
      st2(lc + 2 * p, c.actf * (cold_alt ? 1.0 : QP_START_MU * fast_rcp(w0p)), c.actf * (cold_alt ? 1.0 : QP_START_MU * fast_rcp(w0m))); \
      ST_ROW(Rwc, wc, p, w0p, w0m);                                                               \
      ST_ROW(Rdlc, dlc, p, 0.0, 0.0); st2(dwc + 2 * p, 0.0, 0.0);                                 \
    }                                                                                             \
    LSYNC();                                                                                      \
    if constexpr (REGROWS) {                                                                      \
      if (lane < mf / 2) {                                                                        \
        /* the hinge slack contains its own multiplier (w = F x - f + l/ro): one fixed-point round */ \
        const double hx = -d0 - Rff.x, hy = -d0 - Rff.y;                                          \
        const double l0x = cold_alt ? 1.0 : QP_START_MU * fast_rcp(fmax(hx + iro, 1.0));          \
        const double l0y = cold_alt ? 1.0 : QP_START_MU * fast_rcp(fmax(hy + iro, 1.0));          \
        Rwf = make_double2(fmax(hx + l0x * iro, 1.0), fmax(hy + l0y * iro, 1.0));                 \
        st2(lf + 2 * lane, cold_alt ? 1.0 : QP_START_MU * fast_rcp(Rwf.x), cold_alt ? 1.0 : QP_START_MU * fast_rcp(Rwf.y)); \
      }                                                                                           \
    } else {                                                                                      \
      for (int i = lane; i < mf; i += QP_THREADS) {                                               \
        const double hx = -d0 - ff[i];        /* F x - f at u = 0 */                              \
        const double l0x = cold_alt ? 1.0 : QP_START_MU * fast_rcp(fmax(hx + iro, 1.0));          \
        wf[i] = fmax(hx + l0x * iro, 1.0);                                                        \
        lf[i] = cold_alt ? 1.0 : QP_START_MU * fast_rcp(wf[i]);                                   \
      }                                                                                           \
    }                                                                                             \
    LSYNC();                                                                                      \
  } while (0)
// qp_info layout per scene (doubles): [0] best iteration [1] merit [2] mu [3] status [4] iterations
// run, then (only when built with -DNPA_QP_PROF) accumulated s_memtime cycles of the solve's phases
#define QP_INFO_STRIDE 16
// (-DNPA_QP_PROF=1: the phases of an iteration; =2: inside the residual phase; =3: inside a predictor / corrector pass;
// tests/tools/qp_phase_cycles.py builds the variants and names the slots)
#ifdef NPA_QP_PROF
#define PROF_DECL unsigned long long pt_ = __builtin_amdgcn_s_memtime(), pacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_AT(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PROF(i) do { if (NPA_QP_PROF == 1 || (i) == 0 || (i) == 9) PROF_AT(i); } while (0)
#define PROF_B(i) do { if (NPA_QP_PROF == 2) PROF_AT(i); } while (0)
#define PROF_C(i) do { if (NPA_QP_PROF == 3) PROF_AT(i); } while (0)
#else
#define PROF_DECL
#define PROF(i) do { } while (0)
#define PROF_B(i) do { } while (0)
#define PROF_C(i) do { } while (0)
#endif

// ---- small device helpers -------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), l);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), l);
  return __hiloint2double((int)hi, (int)lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// zero-filled DPP move inside a 16-lane row, and the inclusive prefix / suffix sums of lanes 0..15 built from it
// (row_shr:n = lane i reads lane i-n, row_shl:n = lane i reads lane i+n; lanes beyond the row end read 0)
template <int CTRL>
__device__ __forceinline__ double dpp0_f64(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_prefix_sum(double v) {
  v += dpp0_f64<0x111>(v); v += dpp0_f64<0x112>(v); v += dpp0_f64<0x114>(v); v += dpp0_f64<0x118>(v);
  return v;
}
__device__ __forceinline__ double row_suffix_sum(double v) {
  v += dpp0_f64<0x101>(v); v += dpp0_f64<0x102>(v); v += dpp0_f64<0x104>(v); v += dpp0_f64<0x108>(v);
  return v;
}
// The same over lanes 0..31 (horizons of 17..32 steps, lane = t): the row scan plus the other row's total.  Prefix: lane 15
// of row 0 reaches row 1 with row_bcast:15 (rows 0 and 2 masked off, they receive 0).  Suffix: lane 16 holds row 1's total.
// WIDE = false: a single row, nothing added.
template <bool WIDE>
__device__ __forceinline__ double scan_prefix(double v) {
  v = row_prefix_sum(v);
  if constexpr (WIDE) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xA, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xA, 0xF, false);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
template <bool WIDE>
__device__ __forceinline__ double scan_suffix(double v, double row0) {      // row0 = 1.0 in lanes 0..15, else 0.0
  v = row_suffix_sum(v);
  if constexpr (WIDE) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), 16);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), 16);
    v = fma(row0, __hiloint2double((int)hi, (int)lo), v);
  }
  return v;
}
// value of lane + 1 (0 behind the last lane of the scan)
template <bool WIDE>
__device__ __forceinline__ double scan_next(double v, int lane) {
  double x = dpp0_f64<0x101>(v);
  if constexpr (WIDE) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), 16);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), 16);
    if (lane == 15) x = __hiloint2double((int)hi, (int)lo);
  }
  return x;
}
// Two / three scans at once, step-major.  A wave alone on its SIMD issues one instruction per four cycles whatever its kind,
// and between a step's v_add_f64 and the next step's DPP read of the same register the hardware wants two wait states: a
// single scan pays an s_nop per step for them, interleaved scans fill the slots with each other's instructions.
#define QP_SCAN_STEP2(CT) do { a += dpp0_f64<CT>(a); __builtin_amdgcn_sched_barrier(0); b += dpp0_f64<CT>(b); __builtin_amdgcn_sched_barrier(0); } while (0)
#define QP_SCAN_STEP3(CT) do { a += dpp0_f64<CT>(a); __builtin_amdgcn_sched_barrier(0); b += dpp0_f64<CT>(b); __builtin_amdgcn_sched_barrier(0); \
                               c += dpp0_f64<CT>(c); __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool WIDE>
__device__ __forceinline__ void scan_prefix2(double& a, double& b) {
  __builtin_amdgcn_sched_barrier(0);
  QP_SCAN_STEP2(0x111); QP_SCAN_STEP2(0x112); QP_SCAN_STEP2(0x114); QP_SCAN_STEP2(0x118);
  if constexpr (WIDE) {
    a += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x142, 0xA, 0xF, false), __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x142, 0xA, 0xF, false));
    b += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(b), 0x142, 0xA, 0xF, false), __builtin_amdgcn_update_dpp(0, __double2loint(b), 0x142, 0xA, 0xF, false));
  }
}
template <bool WIDE>
__device__ __forceinline__ void scan_suffix2(double& a, double& b, double row0) {
  __builtin_amdgcn_sched_barrier(0);
  QP_SCAN_STEP2(0x101); QP_SCAN_STEP2(0x102); QP_SCAN_STEP2(0x104); QP_SCAN_STEP2(0x108);
  if constexpr (WIDE) { a = fma(row0, readlane_f64(a, 16), a); b = fma(row0, readlane_f64(b, 16), b); }
}
template <bool WIDE>
__device__ __forceinline__ void scan_suffix3(double& a, double& b, double& c, double row0) {
  __builtin_amdgcn_sched_barrier(0);
  QP_SCAN_STEP3(0x101); QP_SCAN_STEP3(0x102); QP_SCAN_STEP3(0x104); QP_SCAN_STEP3(0x108);
  if constexpr (WIDE) { a = fma(row0, readlane_f64(a, 16), a); b = fma(row0, readlane_f64(b, 16), b); c = fma(row0, readlane_f64(c, 16), c); }
}
// a wave-uniform double made provably uniform (both halves through v_readfirstlane): the compiler may then keep it in a
// scalar register pair -- and, when it runs short of those, park it in a lane of a spill VGPR (v_readlane to fetch it)
// instead of sending a whole vector register to scratch memory
__device__ __forceinline__ double uni64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };
// full-wave reduction, the same bits in every lane: four butterfly steps inside the 16-lane rows (every lane is a valid
// source: bound_ctrl spares the compiler the zero-initialised destination it otherwise builds per step), then the row
// totals travel up with row_bcast:15 (lane 15 of a row -> the next row) and row_bcast:31 (lane 31 -> rows 2, 3): LANE 63
// ends up with all four, and only lane 63 is read (what the two steps leave in rows 0 - 2 is not a total and is not used)
template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
  v = Op::f(v, dpp0_f64<0xB1>(v));     // quad_perm [1,0,3,2]
  v = Op::f(v, dpp0_f64<0x4E>(v));     // quad_perm [2,3,0,1]
  v = Op::f(v, dpp0_f64<0x141>(v));    // row_half_mirror
  v = Op::f(v, dpp0_f64<0x140>(v));    // row_mirror -> every lane holds its 16-lane row total
  v = Op::f(v, dpp0_f64<0x142>(v));    // row 3: r3 + r2   (row 1: r1 + r0)
  v = Op::f(v, dpp0_f64<0x143>(v));    // row 3: + (r1 + r0)
  return readlane_f64(v, 63);
}
// two independent reductions, step-major (see QP_SCAN_STEP2: each fills the other's wait states)
template <class OpA, class OpB>
__device__ __forceinline__ void wave_reduce2(double& a, double& b) {
#define QP_RED_STEP2(CT) do { a = OpA::f(a, dpp0_f64<CT>(a)); __builtin_amdgcn_sched_barrier(0); b = OpB::f(b, dpp0_f64<CT>(b)); __builtin_amdgcn_sched_barrier(0); } while (0)
  __builtin_amdgcn_sched_barrier(0);
  QP_RED_STEP2(0xB1); QP_RED_STEP2(0x4E); QP_RED_STEP2(0x141); QP_RED_STEP2(0x140); QP_RED_STEP2(0x142); QP_RED_STEP2(0x143);
#undef QP_RED_STEP2
  a = readlane_f64(a, 63); b = readlane_f64(b, 63);
}
__device__ __forceinline__ double fast_rcp(double x) {      // ~1 ulp; x finite, nonzero
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
// step-to-the-boundary ratios only need a few digits (v_rcp_f64: ~1e-8 relative; the step keeps >= 1e-6 of the distance)
__device__ __forceinline__ double rough_rcp(double x) { return __builtin_amdgcn_rcp(x); }
// 1/sqrt(pivot) of the Cholesky: v_rsq_f64 (about 2^-26 relative) and ONE Newton step (-> ~1e-15); L L' then differs from
// K' by a few ulp -- an inexact Newton matrix at that level costs nothing, and the step is on the serial path of every pivot
__device__ __forceinline__ double fast_rsqrt(double x) {    // x > 0
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
// two adjacent doubles of an LDS array in one ds_read_b128 / ds_write_b128 (the arrays used this way start at even offsets)
__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double a, double b) { *reinterpret_cast<double2*>(p) = make_double2(a, b); }
// wave-local ordering of LDS traffic between lanes (the waves of a workgroup are independent
// scenes with different iteration counts: no workgroup barrier may be used)
#define LSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

// TT > 0: horizon known at compile time -> the reduced KKT matrix, its Cholesky factor (rows and
// columns) and the columns of Phi live in registers, one matrix row per lane, every loop over
// the horizon is unrolled and all broadcasts are v_readlane (no LDS round trip on the serial
// chain).  TT == 0: generic horizon, same algorithm with the matrices in LDS.
// BWD: after convergence, one more solve with the Newton matrix of the final iterate and the upstream
// gradient as right-hand side gives dL/d(q_s, p_u, eta, d_max, d_min) (oracle/nrmp_backward.py states
// the derivation; reference: the adjust parameters are differentiable through cvxpylayers,
// nrmp.py:79-95, :144).  Instantiated for the generic path only, so the forward kernels are untouched.
struct QpBackward {
  const float* grad_s;      // [B][3][T+1]  dL/d opt_s
  const float* grad_u;      // [B][2][T]    dL/d opt_u
  const float* grad_d;      // [B][T]       dL/d opt_d (may be null)
  float* grad_theta;        // [B][8]       q_s[0..2], p_u, eta, d_max, d_min, (status)
  float* grad_nom_s;        // [B][3][T+1]  dL/d(proximal centre) = bk Phi v, column 0 = 0 (may be null)
  // parameter export (npa_nrmp_params): when set, the kernel writes the linearisation and the hinge coefficients it
  // built -- [B][T][11] A02 A12 B00 B01 B10 B11 B20 B21 C0 C1 C2, then [B][T][M][3] fa0 fa1 fb, fp32 as the
  // reference holds them -- and returns before the solve
  float* dbg_abc;
  float* dbg_f;
  double* dbg_x;            // [B][2T + T]: the fp64 solution (u_0x, u_0y, ..., then d) before the cast to fp32, or null
};

// Pair p of the u / d rows: rows 2p and 2p + 1 are  +c'x <= bp  and  -c'x <= bm  with c'x = x[ia] - sb x[ib] over the
// vector x = (u, d) (xu / xd and dxu / dxd are contiguous in LDS).  p < 2T: speed of u_p; p < 4T - 2: rate
// u_{q+2} - u_q, q = p - 2T; then d_t, t = p - (4T - 2).  actf = 0 switches a pair with an infinite bound off (its
// multipliers stay 0, its slacks 1).  A plain function of VALUES on purpose: as a lambda over the kernel's locals the
// selection among the bounds became a selection among ADDRESSES of closure fields, the closure went to scratch memory
// and every use inside the solve's loop was a (twice) dependent memory load.
struct PairC { int ia, ib; double sb, bp, bm, actf; };
__device__ __forceinline__ PairC qp_pair(int p, int T, int npu, double sb0, double sb1, double ab0, double ab1, double sf0,
                                         double sf1, double af0, double af1, double dmaxv, double dmin0) {
  PairC c;
  const bool is_d = p >= npu, is_rate = p >= 2 * T && !is_d;
  const int q = p - 2 * T;
  const bool odd = ((is_rate ? q : p) & 1) != 0;
  c.ia = is_d ? 2 * T + (p - npu) : (is_rate ? q + 2 : p);
  c.ib = is_rate ? q : 0;
  c.sb = is_rate ? 1.0 : 0.0;
  const double bs = odd ? sb1 : sb0, ba = odd ? ab1 : ab0, fs = odd ? sf1 : sf0, fa = odd ? af1 : af0;
  const double bd = is_rate ? ba : bs;
  c.actf = is_d ? 1.0 : (is_rate ? fa : fs);
  c.bp = is_d ? dmaxv : bd;
  c.bm = is_d ? -dmin0 : bd;
  return c;
}

// SCANW: the scan forms of the Phi products and the P_t blocks also for horizons of 17..32 steps (two DPP rows); the
// launcher's default at T = 20 (acker: 69 k -> 79 k plans/s; the parity verdicts of tests/test_gpu_parity.py are the
// same with and without them).  NPA_QP_NOSCAN_WIDE=1 selects the dense-product instantiation for A/B measurements.
template <int TT, int MM, bool BWD = false, bool SCANW = false, int WV = NPA_QP_WAVES, bool ASET_T = false>
// (two waves per SIMD: <= 256 registers.  tests/test_abi.py reads the counts of the built code object and fails on any
// spill or scratch use)
__global__ __attribute__((amdgpu_flat_work_group_size(QP_THREADS, QP_THREADS), amdgpu_waves_per_eu(WV, 3)))
void nrmp_qp_kernel(
    DevParams P, const float* cur_s_in, const float* cur_u_in, const float* __restrict__ ref_s,
    const float* __restrict__ ref_us, const float* __restrict__ mu_sorted, const float* __restrict__ lam_sorted,
    const float* __restrict__ pts_sorted, const float* __restrict__ dist_sorted, const int* __restrict__ count,
    float* cur_s_out, float* cur_u_out, float* __restrict__ cur_d_out, float* __restrict__ out_s,
    float* __restrict__ out_u, float* __restrict__ out_d, float* __restrict__ out_min_distance,
    int* __restrict__ out_iters, float* __restrict__ out_nrmp_points, int* __restrict__ flags,
    float* __restrict__ state, double* __restrict__ qp_info, double* __restrict__ warm, int scene0, int nscene,
    QpBackward bw, float* __restrict__ trig_out) {
  extern __shared__ __attribute__((aligned(16))) double sm_all[];
  // one scene (one wave) per workgroup: the dispatcher spreads the waves of a launch evenly over the CUs, and -- the
  // reason it is fixed here and not a launch parameter -- the scene's LDS block starts at LDS address 0, so every array
  // below is addressed with an immediate offset.  With a run-time base (several scenes per workgroup) the compiler
  // kept ~60 array base addresses in SGPRs, spilled them to VGPR lanes and re-read ~150 of them with v_readlane in
  // every iteration of the solve.
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= nscene) return;
  const int b = blockIdx.x + scene0;
  double* sm = sm_all;
  if (flags && flags[b * 4 + 0]) return;
  // Two launches per PAN iteration when the active-set iteration is on (P.qp_aset): the ASET instantiation goes first and tries
  // the scenes whose previous solve converged; a scene it finishes (solution, warm record, stop test: the same tail as here)
  // is marked in flags[3], and the interior-point instantiation that follows skips it.  A scene the attempt gives up on is
  // left untouched (nothing of this kernel reaches global memory before its tail).
  if constexpr (ASET_T) {
    if (!(warm && flags && flags[b * 4 + 2] && P.qp_aset)) return;
  } else {
    if (flags && flags[b * 4 + 3]) {
      if (lane == 0) flags[b * 4 + 3] = 0;
      return;
    }
  }
  // this wave is a long dependent chain that shares its SIMD with throughput-bound selection waves of
  // the other batches in flight: win the issue arbitration, it needs few slots but needs them promptly
  npa_setprio(P.prio_qp0);

  // with TT and MM fixed every LDS offset below folds to an immediate (one base register)
  PROF_DECL
  const int T = TT > 0 ? TT : P.T, M = (TT > 0 && MM > 0) ? MM : P.M, E = P.E, nu = 2 * T;
  constexpr int NU = TT > 0 ? 2 * TT : 1, T3 = TT > 0 ? 3 * TT : 1;
  const bool obs = M > 0;
  const int mcu = 8 * T - 4;                  // rows on u: 4T speed + 4T-4 rate
  const int mf = obs ? T * M : 0;
  const int ldp = nu + 1, ldk = nu + 1;       // odd leading dimensions
  const int npair = nu * (nu + 1) / 2;
  const double ro = P.ro_obs, iro = uni64(1.0 / ro);
  const double dmin0 = uni64(fmax((double)P.d_min, 0.0)), dmaxv = uni64((double)P.d_max);
  // T = 10, M = 10: each lane owns ONE pair of hinge rows (lanes < T M / 2 = 50) and ONE pair of u / d rows (lanes <
  // 5T - 2 = 48) in every phase, so the per-row arrays that only their owner touches -- slacks, residuals, multiplier
  // directions, the hinge offsets -- live in registers, not in LDS (7 arrays, 5.5 KB of the scene's 26 KB: the LDS
  // block is what limits how many scenes a CU holds).  Arrays other lanes read (multipliers, 1/w, rhs weights) stay.
  constexpr bool REGROWS = TT > 0 && MM > 0 && (MM % 2) == 0 && TT * MM / 2 <= QP_THREADS && 5 * TT - 2 <= QP_THREADS;
  // SCAN: the products with Phi, the blocks of P_t and the rows of K' are built from per-step sums over the horizon (lane =
  // t, DPP row scans) -- Phi itself is never stored (5 KB of the scene's LDS block at T = 10, 20 KB at T = 20)
  constexpr bool SCAN = TT > 0 && (TT <= 16 || (SCANW && TT <= 32));
  constexpr bool WIDE = TT > 16;
  const double row0 = lane < 16 ? 1.0 : 0.0;

  // ---- LDS carve (doubles) -----------------------------------------------------------
  // Rows of the interior-point method first, at even offsets (they are read and written two at a time).  The rows on u
  // (index i < mcu: 2v / 2v+1 = +-u_v <= speed bound, 4T + 2q / +1 = +-(u_{q+2} - u_q) <= rate bound) and the rows on d
  // (mcu + 2t: d_t <= d_max, mcu + 2t + 1: -d_t <= -d_min) share one array each: "pair" p holds rows 2p and 2p + 1, the
  // + and - side of one linear form.  ld_, wd, ... are the d parts under their own names.
  const int mcd = mcu + 2 * T;
  const int mcr = REGROWS ? 0 : mcd;          // (arrays held in registers take no LDS)
  double* lc = sm;                            // [mcu + 2T] multipliers
  double* wc = lc + mcd;                      // slacks
  double* dlc = wc + mcr;
  double* dwc = dlc + mcr;
  double* r2 = dwc + mcd;
  double* iwc = r2 + mcr;                     // 1/w
  double* ld_ = lc + mcu; double* wd = wc + mcu; double* dld = dlc + mcu; double* dwd = dwc + mcu;
  double* r2d = r2 + mcu; double* iwd = iwc + mcu;
  const int mfe = (mf + 1) & ~1;
  const int mfr = REGROWS ? 0 : mfe;
  double* fa0 = iwc + mcd;                    // [mf] hinge rows ...
  double* fa1 = fa0 + mfe;
  double* ff = fa1 + mfe;
  double* lf = ff + mfr;
  double* wf = lf + mfe;
  double* dlf = wf + mfr;
  double* dwf = dlf + mfr;
  double* r3 = dwf + mfe;
  double* iwf = r3 + mfr;                     // 1/(wf + lf/ro)
  double* Phi = iwf + mfe;                    // [T][3][ldp]  s(t+1) = Phi[t] u + cv[t]
  // generic path: Yt [T][2][ldp] = S'_t Phi_xy(t), Hm / Km [nu][ldk] full matrices.
  // fast path (TT > 0): Yt [T][6] = staging of the 3x3 P_t, Hm packed lower triangle (row a at
  // a(a+1)/2), Km = L as a full [nu][nu+1] matrix whose diagonal and upper triangle stay ZERO: the substitutions read
  // row `lane` (forward) and column `lane` (backward) of it with no lane predicates and no copy of L in registers
  // (fast path: the P_t staging is the [T][6] block s3 | q3 -- both are dead between the residual phase and the passes)
  double* Ytg = Phi + (SCAN ? 0 : (size_t)T * 3 * ldp);
  double* Hm = Ytg + (TT > 0 ? 0 : (size_t)T * 2 * ldp);
  double* Km = Hm + (TT > 0 ? (size_t)nu * (nu + 1) / 2 : (size_t)nu * ldk);
  double* cv = Km + (size_t)nu * ldk;         // [T][3]
  double* lin = cv + T * 3;                   // [T][3]  state-cost gradient at u = 0
  double* s3 = lin + T * 3;                   // [T][3]  Phi x   /  Phi dx
  double* q3 = s3 + T * 3;                    // [T][3]  operand of Phi'
  double* Yt = TT > 0 ? s3 : Ytg;
  double* Abc = q3 + T * 3;                   // [T][12]
  double* St = Abc + T * QP_ABC_LD;           // [T][QP_ST_LD]  S'00 S'01 S'11 v0 v1 sigma 1/kappa r1d
  double* xu = St + ((T * QP_ST_LD + 1) & ~1);                   // [nu]    (xu, xd contiguous: the pairs index them as one vector)
  double* xd = xu + nu;                       // [T]
  double* xbest = xd + T;                     // [nu+T]
  double* dxu = xbest + nu + T;               // [nu]    (dxu, dxd contiguous)
  double* dxd = dxu + nu;                     // [T]
  double* invd = dxd + T;                     // [nu]   1/L_kk
  double* cpre = invd + nu;                   // [T][2] (SCAN) prefix sums of (A02, A12): A_i ... A_(r+1) = I + (cpre_i - cpre_r) e_2'
  // pair table of the H build: the fast path needs it during the set-up only and parks it in the block of L (zeroed
  // after the H build, below)
  unsigned char* pa = reinterpret_cast<unsigned char*>(TT > 0 ? Km : invd + nu);      // [npair]
  unsigned char* pc = pa + ((npair + 7) & ~7);

  // REGROWS: this lane's pair of u / d rows (slack, dl, primal residual) and of hinge rows (slack, dl, residual, offset f)
  double2 Rwc = make_double2(1.0, 1.0), Rdlc = make_double2(0.0, 0.0), Rr2 = make_double2(0.0, 0.0);
  double2 Rwf = make_double2(1.0, 1.0), Rdlf = make_double2(0.0, 0.0), Rr3 = make_double2(0.0, 0.0), Rff = make_double2(0.0, 0.0);
  // one access path for both layouts (the condition is a compile-time constant)
#define LD_WC(p) (REGROWS ? Rwc : ld2(wc + 2 * (p)))
#define LD_DLC(p) (REGROWS ? Rdlc : ld2(dlc + 2 * (p)))
#define LD_R2(p) (REGROWS ? Rr2 : ld2(r2 + 2 * (p)))
#define LD_WF(h) (REGROWS ? Rwf : ld2(wf + 2 * (h)))
#define LD_DLF(h) (REGROWS ? Rdlf : ld2(dlf + 2 * (h)))
#define LD_R3(h) (REGROWS ? Rr3 : ld2(r3 + 2 * (h)))
#define LD_FF(h) (REGROWS ? Rff : ld2(ff + 2 * (h)))
#define ST_ROW(REG, arr, q, a, b_) do { if constexpr (REGROWS) REG = make_double2((a), (b_)); else st2((arr) + 2 * (q), (a), (b_)); } while (0)

  const float* s_in = cur_s_in + (size_t)b * 3 * (T + 1);
  const float* u_in = cur_u_in + (size_t)b * 2 * T;
  const float* rs = ref_s + (size_t)b * 3 * (T + 1);
  const float* rus = ref_us + (size_t)b * T;

  // ---- A_t, B_t, C_t in the reference's fp32 rounding sequence (robot.py:272-316) -------
  for (int t = lane; t < T; t += QP_THREADS) {
    float phi = s_in[2 * (T + 1) + t], v = u_in[t], psi = u_in[T + t];
    const float dt32 = P.dt32;
    double* o = Abc + t * QP_ABC_LD;
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
  // pair table (a >= c), activity of the u rows, their bounds
  for (int p = lane; p < (SCAN ? 0 : npair); p += QP_THREADS) {
    int a = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    while (a * (a + 1) / 2 > p) --a;
    pa[p] = (unsigned char)a; pc[p] = (unsigned char)(p - a * (a + 1) / 2);
  }
  // bounds of the u rows (0 where infinite: the pair is switched off through its actf).  (No run-time index into P
  // anywhere in this kernel: one would move the whole by-value struct to scratch memory.)
  const double sb0 = isfinite(P.speed_bound[0]) ? P.speed_bound[0] : 0.0, sb1 = isfinite(P.speed_bound[1]) ? P.speed_bound[1] : 0.0;
  const double ab0 = isfinite(P.acce_bound[0]) ? P.acce_bound[0] : 0.0, ab1 = isfinite(P.acce_bound[1]) ? P.acce_bound[1] : 0.0;
  const double sf0 = isfinite(P.speed_bound[0]) ? 1.0 : 0.0, sf1 = isfinite(P.speed_bound[1]) ? 1.0 : 0.0;
  const double af0 = isfinite(P.acce_bound[0]) ? 1.0 : 0.0, af1 = isfinite(P.acce_bound[1]) ? 1.0 : 0.0;
  const int npu = mcu >> 1, npc = npu + (obs ? T : 0);
#define PAIR_C0(p) qp_pair((p), T, npu, sb0, sb1, ab0, ab1, sf0, sf1, af0, af1, dmaxv, dmin0)
  // REGROWS: a lane owns ONE pair (p = lane) in every pass, so its constants are six registers computed once -- not ten
  // wave-uniform doubles kept alive through the whole solve and ~25 instructions of selection at each of the six uses
  const PairC my_pair = PAIR_C0(REGROWS && lane < npc ? lane : 0);
#define PAIR_C(p) (REGROWS ? my_pair : PAIR_C0(p))
  // hinge rows two at a time (rows 2l, 2l + 1 of step t = 2l / M) when M is even
  constexpr bool HPAIR = TT > 0 && MM > 0 && (MM % 2) == 0;
  LSYNC();

  // ---- Phi recursion: Phi[t] = A_t Phi[t-1] + [B_t at cols 2t,2t+1]; A = I + e0 A02 e2' + e1 A12 e2'
  if constexpr (SCAN) {
    // no Phi: the free response c_t by scans (theta is a prefix sum of C2, x / y of a_t theta_t + C_t), and the prefix sums
    // of a that every later use of Phi is rebuilt from
    const bool on = lane < TT;
    const double* o = Abc + (on ? lane : 0) * QP_ABC_LD;
    const double a0 = on ? o[0] : 0.0, a1 = on ? o[1] : 0.0, c0 = on ? o[8] : 0.0, c1 = on ? o[9] : 0.0, c2 = on ? o[10] : 0.0;
    const double th0 = s_in[2 * (T + 1)];
    const double thi = scan_prefix<WIDE>(c2), the = th0 + (thi - c2);          // theta before step t
    const double x = (double)s_in[0] + scan_prefix<WIDE>(fma(a0, the, c0));
    const double y = (double)s_in[T + 1] + scan_prefix<WIDE>(fma(a1, the, c1));
    const double p0 = scan_prefix<WIDE>(a0), p1 = scan_prefix<WIDE>(a1);
    if (on) {
      cv[lane * 3 + 0] = x; cv[lane * 3 + 1] = y; cv[lane * 3 + 2] = th0 + thi;
      st2(cpre + 2 * lane, p0, p1);
    }
    LSYNC();
  } else
  for (int t = 0; t < T; ++t) {
    const double* o = Abc + t * QP_ABC_LD;
    double* Pt = Phi + (size_t)t * 3 * ldp;
    const double* Pp = Pt - 3 * ldp;
    for (int c = lane; c < nu; c += QP_THREADS) {
      double p0 = 0, p1 = 0, p2 = 0;
      if (t > 0) { p0 = Pp[c]; p1 = Pp[ldp + c]; p2 = Pp[2 * ldp + c]; }
      double n0 = p0 + o[0] * p2, n1 = p1 + o[1] * p2, n2 = p2;
      if (c == 2 * t) { n0 += o[2]; n1 += o[4]; n2 += o[6]; }
      if (c == 2 * t + 1) { n0 += o[3]; n1 += o[5]; n2 += o[7]; }
      Pt[c] = n0; Pt[ldp + c] = n1; Pt[2 * ldp + c] = n2;
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

  // ---- cost: H (constant block), state-cost gradient at u=0 ---------------------------------
  const double m2 = (P.kin == 2) ? 0.0 : 1.0;            // omni: theta row not in the state cost
  const double W0 = uni64(2.0 * (double)P.q_s[0] * (double)P.q_s[0] + P.bk);
  const double W1 = uni64(2.0 * (double)P.q_s[1] * (double)P.q_s[1] + P.bk);
  const double W2 = uni64(2.0 * m2 * (double)P.q_s[2] * (double)P.q_s[2] + P.bk);
  const double pu = uni64((double)P.p_u);
  if constexpr (SCAN) {
    // the state cost's Hessian Phi' W Phi is not formed: W joins S'_t in the P_t blocks of every iteration (K' build), and the
    // packed triangle only carries the band terms of C_u' D C_u (+ 2 p_u^2 on the speed diagonal), zero elsewhere
    for (int q = lane; q < npair; q += QP_THREADS) Hm[q] = 0.0;
  } else
  for (int p = lane; p < npair; p += QP_THREADS) {
    int a = pa[p], c = pc[p];
    double acc = 0;
    for (int t = a >> 1; t < T; ++t) {
      const double* Pt = Phi + (size_t)t * 3 * ldp;
      acc += W0 * Pt[a] * Pt[c] + W1 * Pt[ldp + a] * Pt[ldp + c] + W2 * Pt[2 * ldp + a] * Pt[2 * ldp + c];
    }
    if (a == c && !(a & 1)) acc += 2.0 * pu * pu;
    if constexpr (TT > 0) {
      Hm[p] = acc;                          // p = a(a+1)/2 + c: the packed lower triangle
    } else {
      Hm[a * ldk + c] = acc;
      Hm[c * ldk + a] = acc;
    }
  }
  if constexpr (TT > 0) {                       // the pair table is done with: its block becomes the zero-padded L
    LSYNC();
    for (int q = lane; q < nu * ldk; q += QP_THREADS) Km[q] = 0.0;
  }
  // fast path: this lane's entries of H that receive the band terms of C_u' D C_u
  double hdiag = 0, hoff = 0;
  for (int q = lane; q < 3 * T; q += QP_THREADS) {
    int t = q / 3, k = q - 3 * t;
    const float qsk = k == 0 ? P.q_s[0] : (k == 1 ? P.q_s[1] : P.q_s[2]);
    double qk = qsk, mk = (k == 2) ? m2 : 1.0, c = cv[q];
    // gamma_a = q_s * ref_s is an fp32 product in the reference (nrmp.py:158)
    double r = (double)__fmul_rn(qsk, rs[k * (T + 1) + t + 1]);
    lin[q] = 2.0 * mk * qk * (qk * c - r) + P.bk * (c - (double)s_in[k * (T + 1) + t + 1]);
  }

  // ---- hinge rows: fa = lam', fb = lam'.p + mu'.h in fp32 (nrmp.py:244-259), slice t+1 ----
  auto hinge_row = [&](int i) -> double {          // builds row i, returns its offset f = fb - fa . c_t
    int t = i / M, j = i - t * M;
    size_t row = ((size_t)b * (T + 1) + (t + 1)) * M + j;
    double a0 = 0, a1 = 0, fb = 0;
    // every load of the row up front and unconditional (the buffers hold a slot for every (scene, slice, j); a slice without
    // points leaves stale contents there, selected away below): written as `if (count > 0) { ... if (e < E) load }` this was a
    // chain of 2 + E global round trips per row, most of the set-up phase's time
    const int cnt_t = count[(size_t)b * (T + 1) + t + 1];
    const float l0 = lam_sorted[row * 2], l1 = lam_sorted[row * 2 + 1];
    const float px = pts_sorted[row * 2], py = pts_sorted[row * 2 + 1];
    float muv[NPA_MAX_E];
#pragma unroll
    for (int e = 0; e < NPA_MAX_E; ++e) muv[e] = mu_sorted[row * E + (e < E ? e : 0)];
    if (cnt_t > 0) {
      float tmp = fmaf(l1, py, __fmul_rn(l0, px));
      float mh = 0.f;
#pragma unroll
      for (int e = 0; e < NPA_MAX_E; ++e) mh = e < E ? fmaf(muv[e], P.h[e], mh) : mh;
      a0 = l0; a1 = l1; fb = (double)__fadd_rn(tmp, mh);
    }
    fa0[i] = a0; fa1[i] = a1;
    if (bw.dbg_f) {
      float* o = bw.dbg_f + ((size_t)b * mf + i) * 3;
      o[0] = (float)a0; o[1] = (float)a1; o[2] = (float)fb;
    }
    return fb - (a0 * cv[t * 3] + a1 * cv[t * 3 + 1]);
  };
  if constexpr (REGROWS) {
    if (lane < mf / 2) { const double f0 = hinge_row(2 * lane), f1 = hinge_row(2 * lane + 1); Rff = make_double2(f0, f1); }
  } else {
    for (int i = lane; i < mf; i += QP_THREADS) ff[i] = hinge_row(i);
  }
  if (bw.dbg_abc) {
    for (int q = lane; q < T * 11; q += QP_THREADS) bw.dbg_abc[(size_t)b * T * 11 + q] = (float)Abc[(q / 11) * QP_ABC_LD + (q % 11)];
    return;
  }

  // ---- starting point: u = 0, d mid-range, slacks >= 1, multipliers QP_START_MU / slack --------
  const double d0 = uni64(0.5 * (dmin0 + dmaxv));
  double cmax = fmax(fabs(dmaxv), fabs(dmin0));
  double m_act = 0;
  for (int p = lane; p < npu; p += QP_THREADS) {
    const PairC c = PAIR_C(p);
    if (c.actf != 0.0) { cmax = fmax(cmax, fabs(c.bp)); m_act += 2.0; }
  }
  // the cold attempt in progress starts from unit multipliers: the second attempt of a forward solve -- and every solve of the
  // BWD instantiations (one start, the one that never jammed; a restart path there costs registers the adjoint needs)
  bool cold_alt = BWD;
  QP_COLD_INIT();
  double gmax = obs ? (double)P.eta : 0.0;
  // g_u = Phi' lin - 2 p_u gamma_b on the speed entries
  if constexpr (SCAN) {
    // Phi' lin by the suffix-sum form (phi_tmul below, written out here: lin itself must survive)
    const bool on = lane < TT;
    const int t = on ? lane : 0;
    const double q0 = on ? lin[3 * t] : 0.0, q1 = on ? lin[3 * t + 1] : 0.0, q2 = on ? lin[3 * t + 2] : 0.0;
    const double l0 = scan_suffix<WIDE>(q0, row0), l1 = scan_suffix<WIDE>(q1, row0);
    const double2 an = ld2(Abc + (t + 1 < TT ? t + 1 : t) * QP_ABC_LD);
    const double l2 = scan_suffix<WIDE>(on ? q2 + an.x * scan_next<WIDE>(l0, lane) + an.y * scan_next<WIDE>(l1, lane) : 0.0, row0);
    const double* o = Abc + t * QP_ABC_LD;
    const double2 b0 = ld2(o + 2), b1 = ld2(o + 4), b2 = ld2(o + 6);
    if (on) {
      const double g0 = b0.x * l0 + b1.x * l1 + b2.x * l2 - 2.0 * pu * (double)__fmul_rn(P.p_u, rus[t]);
      const double g1 = b0.y * l0 + b1.y * l1 + b2.y * l2;
      gmax = fmax(gmax, fmax(fabs(g0), fabs(g1)));
    }
  } else
  for (int a = lane; a < nu; a += QP_THREADS) {
    double acc = 0;
    for (int t = a >> 1; t < T; ++t) {
      const double* Pt = Phi + (size_t)t * 3 * ldp;
      acc += Pt[a] * lin[t * 3] + Pt[ldp + a] * lin[t * 3 + 1] + Pt[2 * ldp + a] * lin[t * 3 + 2];
    }
    if (!(a & 1)) acc += -2.0 * pu * (double)__fmul_rn(P.p_u, rus[a >> 1]);
    gmax = fmax(gmax, fabs(acc));
  }
  // (the merit divides the residuals by these scales: reciprocals once, no fp64 division inside the loop)
  const double iscale_d = uni64(1.0 / (1.0 + wave_reduce<OpMax>(gmax))), iscale_p = uni64(1.0 / (1.0 + wave_reduce<OpMax>(cmax)));
  const double m_tot = fmax(wave_reduce<OpSum>(m_act) + (double)mf + (obs ? 2.0 * T : 0.0), 1.0);
  const double inv_m = uni64(1.0 / m_tot);
  const double pub = (lane < nu && !(lane & 1)) ? -2.0 * pu * (double)__fmul_rn(P.p_u, rus[lane >> 1]) : 0.0;
  LSYNC();
  if constexpr (SCAN) {
    hdiag = (lane < NU && !(lane & 1)) ? 2.0 * pu * pu : 0.0;
  } else if constexpr (TT > 0) {
    if (lane < NU) {
      hdiag = Hm[lane * (lane + 1) / 2 + lane];
      hoff = lane >= 2 ? Hm[lane * (lane + 1) / 2 + lane - 2] : 0.0;
    }
  }

  double best_merit = 1e300, last_mu = 0;
  int best_it = 0, stall = 0, status = 0, it = 0;

  // y = Phi v for all three state rows: out3[t][k] = sum_a Phi[t][k][a] v[a]
  // With A_t = I + (A02, A12, 0)' e_2' the three products of an iteration with Phi are sums over the horizon (lane = t,
  // T <= 16: one DPP row), not 2T-deep chains over a stored matrix:
  //   s = Phi v :  theta_t = sum_{r<=t} B_r[2,:] v_r ;  xy_t = sum_{r<=t} (a_r theta_{r-1} + B_r[:2,:] v_r)
  //   w = Phi'q :  l_xy,t = sum_{r>=t} q_r[:2] ;  l_2,t = sum_{r>=t} (q_r[2] + a_{r+1} . l_xy,r+1) ;  w_t = B_t' l_t
  // (checked against the dense forms in fp64: tests/tools/scan_forms_check.py)
  auto phi_mul = [&](const double* v, double* out3) __attribute__((always_inline)) {
    if constexpr (SCAN) {
      const bool on = lane < TT;
      const int t = on ? lane : 0;
      const double2 vt = ld2(v + 2 * t);
      const double* o = Abc + t * QP_ABC_LD;
      const double2 a01 = ld2(o), b0 = ld2(o + 2), b1 = ld2(o + 4), b2 = ld2(o + 6);
      const double bu2 = on ? b2.x * vt.x + b2.y * vt.y : 0.0;
      const double th = scan_prefix<WIDE>(bu2), thx = th - bu2;               // theta_{t+1}, theta_t
      const double i0 = on ? fma(a01.x, thx, b0.x * vt.x + b0.y * vt.y) : 0.0;
      const double i1 = on ? fma(a01.y, thx, b1.x * vt.x + b1.y * vt.y) : 0.0;
      double x = i0, y = i1;
      scan_prefix2<WIDE>(x, y);
      if (on) { out3[3 * t] = x; out3[3 * t + 1] = y; out3[3 * t + 2] = th; }
      return;
    }
    for (int q = lane; q < 3 * T; q += QP_THREADS) {
      int t = q / 3, k = q - 3 * t;
      const double* Pr = Phi + ((size_t)t * 3 + k) * ldp;
      double a0 = 0, a1 = 0;
      if constexpr (TT > 0) {          // Phi[t][k][c] is stored as 0 beyond column 2t+1
#pragma unroll
        for (int c = 0; c < NU; c += 2) { a0 = fma(Pr[c], v[c], a0); a1 = fma(Pr[c + 1], v[c + 1], a1); }
      } else {
        const int cend = 2 * (t + 1);
        for (int c = 0; c < cend; c += 2) { a0 = fma(Pr[c], v[c], a0); a1 = fma(Pr[c + 1], v[c + 1], a1); }
      }
      out3[q] = a0 + a1;
    }
  };
  // w_a = sum_{t,k} Phi[t][k][a] in3[t][k]   (returned for a = lane, 0 for lane >= nu)
  auto phi_tmul = [&](const double* in3) __attribute__((always_inline)) -> double {
    double acc = 0;
    if constexpr (SCAN) {
      const bool on = lane < TT;
      const int t = on ? lane : 0;
      const double q0 = on ? in3[3 * t] : 0.0, q1 = on ? in3[3 * t + 1] : 0.0, q2 = on ? in3[3 * t + 2] : 0.0;
      double l0 = q0, l1 = q1;
      scan_suffix2<WIDE>(l0, l1, row0);
      const double2 an = ld2(Abc + (t + 1 < TT ? t + 1 : t) * QP_ABC_LD);         // a of step t+1 (meets l = 0 at the last step)
      const double l2 = scan_suffix<WIDE>(on ? q2 + an.x * scan_next<WIDE>(l0, lane) + an.y * scan_next<WIDE>(l1, lane) : 0.0, row0);
      const double* o = Abc + t * QP_ABC_LD;
      const double2 b0 = ld2(o + 2), b1 = ld2(o + 4), b2 = ld2(o + 6);
      // lane t holds w_{2t}, w_{2t+1}; the callers want w_a in lane a: through the operand's own block (it is consumed)
      double* stage = const_cast<double*>(in3);
      if (on) st2(stage + 2 * t, b0.x * l0 + b1.x * l1 + b2.x * l2, b0.y * l0 + b1.y * l1 + b2.y * l2);
      LSYNC();
      return lane < nu ? stage[lane] : 0.0;
    }
    if (lane < nu) {
      const int a = lane;
      if constexpr (TT > 0) {        // Phi[t][k][a] is stored as 0 for t < a/2: no lane-dependent trip count
        double a0 = 0, a1 = 0, a2 = 0;
        if constexpr (TT <= 10) {
          // column a of Phi on its way before the sums start (the loads were issued six at a time with a wait after each
          // batch); at T = 20 the 60 values do not fit beside the rest
          double ph[3 * TT];
#pragma unroll
          for (int q = 0; q < 3 * TT; ++q) ph[q] = Phi[(size_t)q * ldp + a];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            a0 = fma(ph[t * 3], in3[t * 3], a0); a1 = fma(ph[t * 3 + 1], in3[t * 3 + 1], a1); a2 = fma(ph[t * 3 + 2], in3[t * 3 + 2], a2);
          }
        } else {
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            const double* Pt = Phi + (size_t)t * 3 * ldp;
            a0 = fma(Pt[a], in3[t * 3], a0); a1 = fma(Pt[ldp + a], in3[t * 3 + 1], a1); a2 = fma(Pt[2 * ldp + a], in3[t * 3 + 2], a2);
          }
        }
        acc = a0 + a1 + a2;
      } else {
        for (int t = a >> 1; t < T; ++t) {
          const double* Pt = Phi + (size_t)t * 3 * ldp;
          acc += Pt[a] * in3[t * 3] + Pt[ldp + a] * in3[t * 3 + 1] + Pt[2 * ldp + a] * in3[t * 3 + 2];
        }
      }
    }
    return acc;
  };
  // C_u' y for variable a (y indexed like the u rows)
  auto ct_mul = [&](const double* y, int a) __attribute__((always_inline)) -> double {
    // (no branches: the two rate pairs that may not exist are read at a clamped index and weighted 0, so that the three
    // 128-bit loads go out together)
    const int t = a >> 1;
    const double m1 = t >= 1 ? 1.0 : 0.0, m2 = t <= T - 2 ? 1.0 : 0.0;
    const int q1 = 4 * T + 2 * (t >= 1 ? a - 2 : 0), q2 = 4 * T + 2 * (t <= T - 2 ? a : 0);
    const double2 y0 = ld2(y + 2 * a), y1 = ld2(y + q1), y2 = ld2(y + q2);
    return (y0.x - y0.y) + m1 * (y1.x - y1.y) - m2 * (y2.x - y2.y);
  };

  // ---- warm start across the PAN iterations of one forward call ------------------------------------------------
  // Iteration k+1 of the PAN loop solves nearly the QP of iteration k once the loop has settled, and an interior-point
  // start from that solution (x, multipliers pushed back inside the cone by QP_WARM_DELTA, slacks recomputed from the
  // new problem data) then needs ~4 iterations instead of ~12.  It is attempted whenever the previous solve converged
  // (flag written at the end of this kernel) and refused at once when its starting merit says the old active set
  // misleads the method (after a large PAN step: merit > 0.05 at iteration 0), dropped when it is behind schedule at
  // iteration 6.  As a backstop a warm-started solve that ends above 1e-10 is repeated from the cold start
  // (need_cold: the same loop, re-initialised at its top).
  // The limit point is the same either way (both stop at 1e-14: measured |du| <= 7e-7 against the cold solve).
  // (Every forward instantiation takes it.  History: while the T = 20 one still needed 256 VGPRs + AGPR copies and ~400
  // spilled SGPRs, hipcc 7.2 produced corrupted loop scalars (best_merit, stall) for every form of this logic; the
  // spill-free build does not show it, and npa_create's self-test re-checks warm against cold on the device.)
  constexpr bool WARM = !BWD;
  const int nwarm = nu + T + mf + mcu + 2 * T;
  double* wrm = (WARM && warm) ? warm + (size_t)b * nwarm : nullptr;
  const bool can_warm = WARM && wrm && flags && flags[b * 4 + 2];
  PROF(0);
  bool adj = false;                      // BWD: the pass below is the adjoint solve
  int it_total = 0, warm_code = 0;       // diagnostics: iterations over all attempts; 1 warm start used, 2 / 3 dropped at it 0 / 6, 4 not converged, 5 cold retry
  bool warm_now = can_warm;              // the solve in progress started from the previous solution
  bool need_cold = false;                // re-initialise at the top of the next iteration (a dropped warm attempt)
  // ---- active-set iteration on the warm solves (branch qp-active-set; NPA_QP_ASET=1; T = 10 / M = 10 instantiation) ----------
  // tests/tools/qp_active_set_study.py states the method (active_set_solve_kernel_form with project_d) and measures it: from the
  // previous PAN iteration's solution the guess of the active set reproduces itself after 1.0 - 1.3 factorisations on ~90 % of the
  // QPs behind the warm-start gate.  Here the loop body below is REUSED: seeded with l = ro e, w = 0 on the hinge rows that are on
  // (l = 0, w = slack on the others) and with l = 0 on every linear row, its residual phase, per-step blocks, K' build,
  // factorisation and predictor pass compute exactly the Newton step of the guessed equality-constrained QP in (u, d) -- d
  // eliminated per step as always, or frozen (1/kappa := 0) where it sits on a bound it is pushed against.  What is new: the
  // tight speed / rate rows are eliminated from K' before the factorisation (aset::reduce_matrix), the right-hand side follows
  // (reduce_rhs), the step is expanded (expand) and taken in full, d is clipped, and the next pass of the residual phase
  // recovers the multipliers of the tight rows (multipliers) and makes the next guess.  A guess that repeats is a KKT point:
  // accepted when what is left of the dual residual is below QP_ASET_TOL; anything else falls back to the interior-point warm
  // start (warm_init once more).  NOT VALIDATED ON A GPU yet (only aset_reduce.h's parts were, in their first form).
  constexpr bool ASET = ASET_T && WARM && REGROWS && SCAN && (NU <= 20);      // (its own instantiation: the extra state spills for now)
  bool aset = false, aset_done = false;
  int aset_guess = 0;
  int aset_why = 0;                      // diagnostics (qp_info[5..7]): 1 accepted, 2 a repeated guess left a residual, 3 guesses used up, 4 restarted cold
  double aset_left = 0.0;                // the scaled residual of the last repeated guess
  double aset_first = 0.0;               // merit of the seeded warm point (first pass)
  unsigned long long ag_on0 = 0, ag_on1 = 0, ag_tp = 0, ag_tm = 0, ag_d = 0;      // the guess in force, as ballots
  bool aset_onx = false, aset_ony = false, aset_dtp = false, aset_dtm = false, aset_dfix = false, aset_tied = false;
  double aset_lam_p = 0.0, aset_lam_m = 0.0, aset_cx = 0.0, aset_resd = 0.0;
  aset::Lane AL{};
  const aset::Scratch AS{s3, dxu, Km, reinterpret_cast<int*>(dxd), ldk};      // all dead while reduce_matrix runs
  if constexpr (ASET) aset = true;       // (this instantiation does nothing else: the gate is at the kernel's top)
  auto warm_init = [&]() __attribute__((always_inline)) {
    const double dl = QP_WARM_DELTA;
    for (int a = lane; a < nu; a += QP_THREADS) { xu[a] = wrm[a]; xbest[a] = wrm[a]; }
    for (int t = lane; t < T; t += QP_THREADS) { xd[t] = fmin(fmax(wrm[nu + t], dmin0), dmaxv); xbest[nu + t] = xd[t]; }
    LSYNC();
    if constexpr (ASET) return;            // (the active-set iteration seeds its own rows from x in its residual phase)
    phi_mul(xu, s3);
    LSYNC();
    if constexpr (REGROWS) {
      if (lane < mf / 2) {
        const int i = 2 * lane, t = i / M;
        const double l0 = fmax(wrm[nu + T + i], dl), l1 = fmax(wrm[nu + T + i + 1], dl);
        st2(lf + i, l0, l1);
        Rwf = make_double2(fmax(fa0[i] * s3[t * 3] + fa1[i] * s3[t * 3 + 1] - xd[t] - Rff.x + l0 * iro, dl),
                           fmax(fa0[i + 1] * s3[t * 3] + fa1[i + 1] * s3[t * 3 + 1] - xd[t] - Rff.y + l1 * iro, dl));
      }
    } else {
      for (int i = lane; i < mf; i += QP_THREADS) {
        int t = i / M;
        double l = fmax(wrm[nu + T + i], dl);
        lf[i] = l;
        wf[i] = fmax(fa0[i] * s3[t * 3] + fa1[i] * s3[t * 3 + 1] - xd[t] - ff[i] + l * iro, dl);
      }
    }
    for (int p = lane; p < npc; p += QP_THREADS) {
      const PairC c = PAIR_C(p);
      if (c.actf != 0.0) {
        const double cx = fma(-c.sb, xu[c.ib], xu[c.ia]);
        const double* wl = wrm + nu + T + mf + 2 * p;          // (lc then ld in the record, like the rows)
        st2(lc + 2 * p, fmax(wl[0], dl), fmax(wl[1], dl));
        ST_ROW(Rwc, wc, p, fmax(c.bp - cx, dl), fmax(c.bm + cx, dl));
      }
    }
    LSYNC();
  };
  if (WARM && warm_now) warm_init();
  if constexpr (ASET) {
    if (aset) {
      // an interior-point iterate sits 1e-15 inside its bounds: snap d, and take the record's multipliers of this lane's pair raw
      for (int t = lane; t < T; t += QP_THREADS) { const double dv = xd[t]; xd[t] = dv >= dmaxv - 1e-8 ? dmaxv : (dv <= dmin0 + 1e-8 ? dmin0 : dv); }
      if (lane < npu && my_pair.actf != 0.0) { const double* wl = wrm + nu + T + mf + 2 * lane; aset_lam_p = wl[0]; aset_lam_m = wl[1]; }
      LSYNC();
    }
  }
  for (it = 0; it <= QP_MAX_IT; ++it) {
    if constexpr (WARM && !ASET) {
      if (need_cold) {                     // restart of a dropped warm attempt, or of a jammed cold one: the cold start again
        need_cold = false;
        QP_COLD_INIT();
        best_merit = 1e300; last_mu = 0; best_it = 0; stall = 0; status = 0;
      }
    }
    // (iterations over all attempts of this solve: a dropped warm attempt's count)
    if (it_total + it == P.prio_it1) npa_setprio(P.prio_qp1);
    if (it_total + it == P.prio_it2) npa_setprio(P.prio_qp2);
    // ================= residuals =================
    PROF_B(8); PROF_C(8);
    phi_mul(xu, s3);
    LSYNC();
    PROF_B(1);
    double gap = 0, rpmax = 0;
    if constexpr (HPAIR) {
      for (int h = lane; h < mf / 2; h += QP_THREADS) {
        const int t = h / (MM / 2), i = 2 * h;
        double2 l = ld2(lf + i), w = LD_WF(h);
        const double2 a0 = ld2(fa0 + i), a1 = ld2(fa1 + i), f = LD_FF(h);
        const double sx = s3[t * 3], sy = s3[t * 3 + 1], d = xd[t];
        if constexpr (ASET) {
          if (aset) {                        // seed: a row that is on carries l = ro e and no slack, the others l = 0 and their slack
            const double ex = f.x + d - (a0.x * sx + a1.x * sy), ey = f.y + d - (a0.y * sx + a1.y * sy);
            aset_onx = ex > 0.0; aset_ony = ey > 0.0;
            const double ro = (double)P.ro_obs;
            l = make_double2(aset_onx ? ro * ex : 0.0, aset_ony ? ro * ey : 0.0);
            w = make_double2(aset_onx ? 0.0 : fmax(-ex, 1e-300), aset_ony ? 0.0 : fmax(-ey, 1e-300));
            st2(lf + i, l.x, l.y);
            ST_ROW(Rwf, wf, h, w.x, w.y);
          }
        }
        const double rx = a0.x * sx + a1.x * sy - d - f.x + l.x * iro - w.x;
        const double ry = a0.y * sx + a1.y * sy - d - f.y + l.y * iro - w.y;
        ST_ROW(Rr3, r3, h, rx, ry);
        st2(iwf + i, fast_rcp(w.x + l.x * iro), fast_rcp(w.y + l.y * iro));
        rpmax = fmax(rpmax, fmax(fabs(rx), fabs(ry)));
        gap += l.x * w.x + l.y * w.y;
      }
    } else {
      for (int i = lane; i < mf; i += QP_THREADS) {
        int t = i / M;
        double l = lf[i], w = wf[i];
        double r = fa0[i] * s3[t * 3] + fa1[i] * s3[t * 3 + 1] - xd[t] - ff[i] + l * iro - w;
        r3[i] = r;
        iwf[i] = fast_rcp(w + l * iro);
        rpmax = fmax(rpmax, fabs(r));
        gap += l * w;
      }
    }
    PROF_B(2);
    for (int p = lane; p < npc; p += QP_THREADS) {
      const PairC c = PAIR_C(p);
      double2 l = ld2(lc + 2 * p), w = LD_WC(p);
      const double cx = fma(-c.sb, xu[c.ib], xu[c.ia]);
      if constexpr (ASET) {
        if (aset) {                          // no weight from any linear row (the tight ones are eliminated, not penalised)
          l = make_double2(0.0, 0.0);
          w = c.actf != 0.0 ? make_double2(fmax(c.bp - cx, 1e-300), fmax(c.bm + cx, 1e-300)) : make_double2(1.0, 1.0);   // (a switched-off pair has no finite bound)
          st2(lc + 2 * p, 0.0, 0.0);
          ST_ROW(Rwc, wc, p, w.x, w.y);
          aset_cx = cx;
        }
      }
      const double rp = (cx + w.x - c.bp) * c.actf, rm = (w.y - cx - c.bm) * c.actf;
      ST_ROW(Rr2, r2, p, rp, rm);
      st2(iwc + 2 * p, fast_rcp(w.x), fast_rcp(w.y));
      rpmax = fmax(rpmax, fmax(fabs(rp), fabs(rm)));
      gap += l.x * w.x + l.y * w.y;             // (a switched-off pair keeps l = 0)
    }
    LSYNC();
    PROF_B(3);
    // per-step sums over the M hinge rows (lane = t)
    double r1dmax = 0;
    double S0r = 0, S1r = 0, S2r = 0;          // (v, 1/kappa, r1_d of step t are re-read from St by the passes: registers)
    for (int t = lane; t < T; t += QP_THREADS) {
      double z0 = 0, z1 = 0, zs = 0, s00 = 0, s01 = 0, s11 = 0, v0 = 0, v1 = 0, sg = 0;
      auto acc = [&](double l, double a0, double a1, double iw) {
        const double D = l * iw;
        z0 += l * a0; z1 += l * a1; zs += l;
        s00 += D * a0 * a0; s01 += D * a0 * a1; s11 += D * a1 * a1; v0 += D * a0; v1 += D * a1; sg += D;
      };
      if constexpr (HPAIR) {
        // all 4 M values of the step on their way (128-bit loads) before the sums start: the lanes that do this are few
        // and the loop was a chain of load -> wait -> 12 flops per row
        // (WV >= 3: in batches of two pairs -- 32 registers in flight instead of 80)
        constexpr int HB = WV >= 3 ? 2 : MM / 2;
#pragma unroll
        for (int j0 = 0; j0 < MM / 2; j0 += HB) {
          double2 l2[HB], p0[HB], p1[HB], iw2[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            const int i = t * MM + 2 * (j0 + j < MM / 2 ? j0 + j : 0);
            l2[j] = ld2(lf + i); p0[j] = ld2(fa0 + i); p1[j] = ld2(fa1 + i); iw2[j] = ld2(iwf + i);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < HB; ++j)
            if (j0 + j < MM / 2) { acc(l2[j].x, p0[j].x, p1[j].x, iw2[j].x); acc(l2[j].y, p0[j].y, p1[j].y, iw2[j].y); }
          if constexpr (HB < MM / 2) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 5
        for (int j = 0; j < M; ++j) { const int i = t * M + j; acc(lf[i], fa0[i], fa1[i], iwf[i]); }
      }
      double kap = sg, r1d = 0;
      if (obs) {
        kap += ld_[2 * t] * iwd[2 * t] + ld_[2 * t + 1] * iwd[2 * t + 1];
        r1d = -(double)P.eta + ld_[2 * t] - ld_[2 * t + 1] + zs;       // g_d + C'lam - F'lam
      }
      double ik = obs ? fast_rcp(kap) : 0.0;
      if constexpr (ASET) {
        if (aset) {
          // d of this step: frozen where it sits on a bound it is pushed against (its row then carries eta - zs as a multiplier),
          // frozen as well when no hinge row is on (nothing to eliminate it through; eta pushes it up: consistent only on d_max),
          // else eliminated as always (no weight from its rows: kappa = ro |rows on|)
          const double resd = (double)P.eta - zs, dcur = xd[t];
          aset_dtp = obs && dcur >= dmaxv && resd > 0.0; aset_dtm = obs && dcur <= dmin0 && resd < 0.0;
          aset_dfix = aset_dtp || aset_dtm || !(sg > 0.0);
          aset_resd = resd;
          if (aset_dfix) { ik = 0.0; r1d = 0.0; }
        }
      }
      double* S = St + t * QP_ST_LD;
      S[0] = s00 - v0 * v0 * ik; S[1] = s01 - v0 * v1 * ik; S[2] = s11 - v1 * v1 * ik;
      S[3] = v0; S[4] = v1; S[5] = sg; S[6] = ik; S[7] = r1d;
      // operand of Phi' for r1_u:  W .* (Phi x) + lin - [z; 0]
      q3[t * 3 + 0] = W0 * s3[t * 3 + 0] + lin[t * 3 + 0] - z0;
      q3[t * 3 + 1] = W1 * s3[t * 3 + 1] + lin[t * 3 + 1] - z1;
      q3[t * 3 + 2] = W2 * s3[t * 3 + 2] + lin[t * 3 + 2];
      r1dmax = fmax(r1dmax, fabs(r1d));
      S0r = S[0]; S1r = S[1]; S2r = S[2];
    }
    LSYNC();
    PROF_B(4);
    double r1u = phi_tmul(q3);                     // lane a < nu
    if (lane < nu) {
      const int a = lane;
      if (!(a & 1)) r1u += 2.0 * pu * pu * xu[a] + pub;
      r1u += ct_mul(lc, a);
    }
    PROF_B(5);
    // scaled dual and primal residuals in ONE max reduction (scaling is monotone: max of scaled = scaled max), side by side
    // with the sum of the complementarity products
    double gsum = gap, rmax = fmax(fmax(lane < nu ? fabs(r1u) : 0.0, r1dmax) * iscale_d, rpmax * iscale_p);
    wave_reduce2<OpSum, OpMax>(gsum, rmax);
    const double mu = gsum * inv_m;
    const double merit = fmax(rmax, mu);
    last_mu = mu;
#ifdef NPA_QP_DBGTRACE
    if (qp_info && lane == 0 && it < 4) { double* qq = qp_info + (size_t)b * QP_INFO_STRIDE; qq[5 + 2 * it] = merit; qq[6 + 2 * it] = mu; }
#endif
    if (!(merit == merit) || !(merit < 1e300)) { status = 2; break; }
    if constexpr (ASET) {
      if (aset) {
        if (aset_guess == 0) {
          aset_first = merit;
          if (merit > QP_ASET_FIRST_MAX) { aset_why = 5; break; }      // far from a KKT point of any guess: the interior-point launch takes it
        }
        // (a) multipliers of the tight rows of the guess in force, from the gradient at this point (first pass: the warm record's)
        double run_sum = 0.0;
        const bool is_rate = lane >= 2 * T && lane < npu;
        if (aset_guess > 0) {
          double vt, bb;
          aset::multipliers<NU>(lane < nu ? -r1u : 0.0, aset_tied, lane, AL, vt, bb, run_sum);
          // a speed pair sits in its variable's lane; rate pair p = 2T + q leads INTO variable q + 2 and reads that lane
          const double vt_r = aset::bperm_f64(vt, is_rate ? lane - 2 * T + 2 : lane);
          const bool tp_prev = ((ag_tp >> lane) & 1ull) != 0, tm_prev = ((ag_tm >> lane) & 1ull) != 0;
          if (lane < 2 * T) { aset_lam_p = tp_prev ? bb : 0.0; aset_lam_m = tm_prev ? -bb : 0.0; }
          else if (is_rate) { aset_lam_p = tp_prev ? vt_r : 0.0; aset_lam_m = tm_prev ? -vt_r : 0.0; }
        }
        // (b) the next guess: a linear row is tight iff its multiplier plus its violation is positive
        bool tp = false, tm = false;
        if (lane < npu && my_pair.actf != 0.0) {
          tp = aset_lam_p + (aset_cx - my_pair.bp) > 0.0;
          tm = aset_lam_m + (-aset_cx - my_pair.bm) > 0.0;
        }
        const unsigned long long n_on0 = __ballot(aset_onx && lane < mf / 2), n_on1 = __ballot(aset_ony && lane < mf / 2);
        const unsigned long long n_tp = __ballot(tp), n_tm = __ballot(tm);
        const unsigned long long n_d = __ballot(aset_dtp && lane < T) | (__ballot(aset_dtm && lane < T) << 16) | (__ballot(aset_dfix && lane < T) << 32);
        const bool same = aset_guess > 0 && n_on0 == ag_on0 && n_on1 == ag_on1 && n_tp == ag_tp && n_tm == ag_tm && n_d == ag_d;
        bool give_up = false;
        if (same) {
          // this point is the KKT point of its own guess.  What can be left of the dual residual: the sum over a run that no
          // speed row anchors (at its head), and eta - zs on a step whose d is not on a bound it is pushed against
          double left = 0.0, viol = 0.0;
          if (lane < nu && AL.head == lane && !AL.anchored) left = fabs(run_sum);
          if (lane < T && obs && !(aset_dtp || aset_dtm)) left = fmax(left, fabs(aset_resd));
          if (lane < npu && my_pair.actf != 0.0) viol = fmax(0.0, fmax(tp ? 0.0 : aset_cx - my_pair.bp, tm ? 0.0 : -aset_cx - my_pair.bm));
          const double merit_a = wave_reduce<OpMax>(fmax(left * iscale_d, viol * iscale_p));
          aset_left = merit_a;
          if (merit_a <= QP_ASET_TOL) {
            aset_why = 1;
            best_merit = merit_a; best_it = it; last_mu = 0.0; stall = 0;
            for (int a = lane; a < nu; a += QP_THREADS) xbest[a] = xu[a];
            for (int t = lane; t < T; t += QP_THREADS) xbest[nu + t] = xd[t];
            // the record the next solve starts from: the hinge rows carry theirs already; the linear rows get them here
            if (lane < npu) st2(lc + 2 * lane, tp ? fmax(aset_lam_p, 0.0) : 0.0, tm ? fmax(aset_lam_m, 0.0) : 0.0);
            if (lane < T && obs) st2(ld_ + 2 * lane, aset_dtp ? aset_resd : 0.0, aset_dtm ? -aset_resd : 0.0);
            LSYNC();
            aset_done = true;
            break;
          }
          give_up = true;
        }
        if (give_up || aset_guess >= QP_ASET_MAX_GUESS) {     // not this time: the interior-point launch solves this scene
          aset_why = give_up ? 2 : 3;
          break;
        }
        ag_on0 = n_on0; ag_on1 = n_on1; ag_tp = n_tp; ag_tm = n_tm; ag_d = n_d;
        ++aset_guess;
      }
    }
    if constexpr (!ASET) {
    // a warm start that is not paying off is dropped at once: a good one starts at merit <= 1.2e-2 and needs 3 - 5
    // iterations with the adaptive step; one that starts far from feasibility is dropped before its first iteration, one
    // that is not below 3e-3 after three or has not reached 1e-4 after six is stuck (the one case in 1360 QPs went on for
    // 27).  The checkpoint at 3 is round 4's: since the warm start is attempted after EVERY converged solve (see the flag at
    // the end of this kernel) the dropped attempts are the launch's slowest scenes -- 6 wasted iterations + a cold solve; the
    // replay over the benchmark QPs (tests/tools/qp_warm_share.py rules) puts the mean of the per-launch maximum at 17.4
    // instead of 18.8 iterations on configs[1], 19.9 instead of 21.7 on the car, at the same mean
    if constexpr (WARM) {
      if (warm_now && ((it == 0 && merit > 0.05) || (it == 3 && merit > 3e-3) || (it == 6 && merit > 1e-4))) {
        warm_code = it == 0 ? 2 : 3;
        it_total += it; warm_now = false; need_cold = true;
        it = -1;
        continue;
      }
    }
    if (merit < best_merit) {
      best_merit = merit; best_it = it; stall = 0;
      for (int a = lane; a < nu; a += QP_THREADS) xbest[a] = xu[a];
      for (int t = lane; t < T; t += QP_THREADS) xbest[nu + t] = xd[t];
    } else {
      ++stall;
    }
    // 1e-14, not 1e-12: along the QP's flat (steering) directions an iterate at 1e-12 is still up to 1e-4 from the
    // limit point; the next Newton step of the quadratic phase (+0.9 iterations on average, the slowest scene of a
    // launch is unchanged) brings it to <= 1e-8, so that two solvers of the same problem agree to the 1e-5 a
    // comparison of controls needs (oracle/nrmp_qp.py uses the same tolerance).  The adjoint solve of BWD is taken
    // at a 1e-12 iterate: the barrier Newton matrix it reuses is better conditioned there
    if (merit <= (BWD ? 1e-12 : 1e-14) || stall >= 3 || it == QP_MAX_IT || mu < 1e-17) {
      if constexpr (WARM) {
        if (warm_now && !(best_merit <= 1e-10)) {      // a warm-started solve that did not converge: once more, cold
          warm_code = 4;
          it_total += it; warm_now = false; need_cold = true;
          it = -1;
          continue;
        }
      }
      // A cold solve that JAMMED (three non-improving iterations far from convergence: a step that landed on the boundary
      // too early; one scene in 96 of the acker workload with the centred start, none with round 2's) gets one more
      // attempt from the other starting point.  qp_info[15] = 5 records it.
      if constexpr (WARM) {
        if (!cold_alt && !(best_merit <= QP_RETRY_MERIT)) {
          cold_alt = true; warm_code = 5;
          it_total += it; need_cold = true;
          it = -1;
          continue;
        }
      }
      if constexpr (BWD) {
        if (!bw.grad_theta) break;
        adj = true;                      // factor K' of this final iterate once more, then solve K' v = dL/dx
      } else {
        break;
      }
    }
    }   // !ASET
    PROF(1); PROF_B(6);

    // ================= reduced KKT matrix, Cholesky =================
    double arow[NU];                    // fast path: row `lane` of K' -> L (lives through the factorisation only)
    double myinv = 1.0;
    bool chol_ok = true;
    if constexpr (TT > 0) {
      // K' = H + band + sum_t Phi_xy(t)' S'_t Phi_xy(t) without touching all T terms per entry:
      // with P_t = S'_t + A(t+1)' P_{t+1} A(t+1) (3x3, backward in t; A = I + a e_2') the entries of
      // block row i are  K'[a][c] += (P_i B_i[:,a&1]) . Phi_i[:,c]  for c <= a   (3 FMAs each).
      {
        double* Pst = Yt;                              // [T][6] staging of P_t
        if constexpr (SCAN) {
          // A(t+1)...A(s) = I + (c_s - c_t) e_2' with c the prefix sum of a, so the blocks of P_t are suffix sums over
          // s >= t of S_s, S_s c_s and c_s'S_s c_s (lane = t; no serial recursion, no broadcast of S'):
          //   P_xy,xy = sS ;  P_xy,2 = sSc - sS c_t ;  P_22 = scSc - 2 c_t . sSc + c_t' sS c_t
          // The state cost's weights W = diag(W0, W1, W2) enter here as well (S_s -> S_s + diag(W0, W1); W2 sits on the
          // theta entry only and reaches P_22 as W2 x the number of steps s >= t): K' = band + sum_t Phi_t' (S'_t + W) Phi_t
          // with no precomputed Phi' W Phi.
          const bool on = lane < TT;
          const double2 cp = ld2(cpre + 2 * (on ? lane : 0));
          const double c0 = on ? cp.x : 0.0, c1 = on ? cp.y : 0.0;
          const double s00 = on ? S0r + W0 : 0.0, s01 = on ? S1r : 0.0, s11 = on ? S2r + W1 : 0.0;
          const double sc0 = s00 * c0 + s01 * c1, sc1 = s01 * c0 + s11 * c1;
          double p00 = s00, p01 = s01, p11 = s11, t0 = sc0, t1 = sc1, t2 = c0 * sc0 + c1 * sc1;
          scan_suffix3<WIDE>(p00, p01, p11, row0);
          scan_suffix3<WIDE>(t0, t1, t2, row0);
          const double u0 = p00 * c0 + p01 * c1, u1 = p01 * c0 + p11 * c1;
          if (on) {
            st2(Pst + lane * 6, p00, p01);
            st2(Pst + lane * 6 + 2, t0 - u0, p11);
            st2(Pst + lane * 6 + 4, t1 - u1, t2 - 2.0 * (c0 * t0 + c1 * t1) + (c0 * u0 + c1 * u1) + W2 * (double)(TT - lane));
          }
        } else {
        double p00 = 0, p01 = 0, p02 = 0, p11 = 0, p12 = 0, p22 = 0;
        // A_t = I + (a0, a1, 0)' e_2': lane t fetches its pair once, the chain below broadcasts them with v_readlane
        // (an LDS load per step would sit on the serial path)
        const double2 a01 = ld2(Abc + (lane < TT ? lane : 0) * QP_ABC_LD);
#pragma unroll
        for (int t = TT - 1; t >= 0; --t) {
          if (t < TT - 1) {
            const double a0 = readlane_f64(a01.x, t + 1), a1 = readlane_f64(a01.y, t + 1);
            const double pa0 = p00 * a0 + p01 * a1, pa1 = p01 * a0 + p11 * a1, pa2 = p02 * a0 + p12 * a1;
            p22 += 2.0 * pa2 + (a0 * pa0 + a1 * pa1);
            p02 += pa0; p12 += pa1;
          }
          p00 += readlane_f64(S0r, t); p01 += readlane_f64(S1r, t); p11 += readlane_f64(S2r, t);
          Pst[t * 6 + 0] = p00; Pst[t * 6 + 1] = p01; Pst[t * 6 + 2] = p02;
          Pst[t * 6 + 3] = p11; Pst[t * 6 + 4] = p12; Pst[t * 6 + 5] = p22;
        }
        }
      }
      const int ar = lane < NU ? lane : 0;
      // band terms of C_u' D C_u for this row go through the LDS copy of H (entries [a][a], [a][a-2]):
      // no per-column lane masks are needed to place them in the register row
      if (lane < NU) {
        const int a = lane, t = a >> 1;
        double dsum = lc[2 * a] * iwc[2 * a] + lc[2 * a + 1] * iwc[2 * a + 1], doff = 0;
        if (t >= 1) { int q = 4 * T + 2 * (a - 2); double v = lc[q] * iwc[q] + lc[q + 1] * iwc[q + 1]; dsum += v; doff = v; }
        if (t <= T - 2) { int q = 4 * T + 2 * a; dsum += lc[q] * iwc[q] + lc[q + 1] * iwc[q + 1]; }
        Hm[a * (a + 1) / 2 + a] = hdiag + dsum;
        if (t >= 1) Hm[a * (a + 1) / 2 + a - 2] = hoff - doff;
      }
      LSYNC();
      PROF(2);
      {
        const int i = ar >> 1, k2 = ar & 1;
        const double* Pi = Yt + i * 6;
        const double* Bi = Abc + i * QP_ABC_LD + 2 + k2;      // B_i[:, k2] = o[2+k2], o[4+k2], o[6+k2]
        const double b0 = Bi[0], b1 = Bi[2], b2 = Bi[4];
        const double g0 = Pi[0] * b0 + Pi[1] * b1 + Pi[2] * b2;
        const double g1 = Pi[1] * b0 + Pi[3] * b1 + Pi[4] * b2;
        const double g2 = Pi[2] * b0 + Pi[4] * b1 + Pi[5] * b2;
        if constexpr (SCAN) {
          // Phi_i[:, 2r + k] = (I + (c_i - c_r) e_2') B_r[:, k] for r <= i, so with G2 = g2 + g . c_i the entry is
          //   g0 B_r[0,k] + g1 B_r[1,k] + (G2 - g . c_r) B_r[2,k]
          // from wave-uniform loads of step r's B and c (two columns per step); columns beyond the row's own are garbage
          // that nobody reads, as before
          const double2 ci = ld2(cpre + 2 * i);
          const double G2 = g2 + g0 * ci.x + g1 * ci.y;
          // (software pipeline of depth one: step r + 1's five loads go out before step r's arithmetic -- with the loads
          // and their use in the same step every step waited a full LDS latency, ~1 000 of this phase's 1 450 cycles; a
          // deeper prefetch does not fit the register file here, where the demand of the kernel peaks)
          const double* hrow = Hm + ar * (ar + 1) / 2;
          double2 nb0 = ld2(Abc + 2), nb1 = ld2(Abc + 4), nb2 = ld2(Abc + 6), ncr = ld2(cpre);
          double nh0 = hrow[0], nh1 = hrow[1];
#pragma unroll
          for (int r = 0; r < TT; ++r) {
            const double2 bb0 = nb0, bb1 = nb1, bb2 = nb2, cr = ncr;
            const double h0 = nh0, h1 = nh1;
            if (r + 1 < TT) {
              const double* o = Abc + (r + 1) * QP_ABC_LD;
              nb0 = ld2(o + 2); nb1 = ld2(o + 4); nb2 = ld2(o + 6); ncr = ld2(cpre + 2 * (r + 1));
              nh0 = hrow[2 * r + 2]; nh1 = hrow[2 * r + 3];
            }
            __builtin_amdgcn_sched_barrier(0);
            const double e = G2 - (g0 * cr.x + g1 * cr.y);
            arow[2 * r] = fma(g0, bb0.x, fma(g1, bb1.x, fma(e, bb2.x, h0)));
            arow[2 * r + 1] = fma(g0, bb0.y, fma(g1, bb1.y, fma(e, bb2.y, h1)));
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
        const double* Ph = Phi + (size_t)i * 3 * ldp;  // Phi_i rows (zero beyond column 2i+1)
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          // (entries c > ar read other rows of the packed triangle: finite, and never used)
          arow[c] = fma(g0, Ph[c], fma(g1, Ph[ldp + c], fma(g2, Ph[2 * ldp + c], Hm[ar * (ar + 1) / 2 + c])));
          // (keep the scheduler from issuing all 4 NU loads ahead of the arithmetic: that is where the register
          // demand of this kernel peaked, above the 256 a wave may hold at two waves per SIMD)
          if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        }
      }
      PROF(3);
      // right-looking Cholesky, row i in lane i: after step k, arow[k] = L[i][k] for i > k (the entries a lane
      // computes on and above its diagonal are unused garbage and are not stored)
      // (written so that the next pivot's reciprocal square root -- the serial chain -- starts before the trailing
      // update of the current column, which is independent of it)
      if constexpr (ASET) {
        if (aset) {
          // this lane's variable a = lane: the tight rate row into it (pair 2T + a - 2: u_a - u_(a-2) <= bp, -(..) <= bm) and its tight
          // speed row (pair a: u_a <= bp, -u_a <= bm), from the guess in force
          const bool isv = lane < nu, has_r = isv && lane >= 2;
          const int pr = 2 * T + lane - 2;
          const bool t_p = has_r && ((ag_tp >> pr) & 1ull) != 0, t_m = has_r && ((ag_tm >> pr) & 1ull) != 0;
          const double rbp = aset::bperm_f64(my_pair.bp, has_r ? pr : lane), rbm = aset::bperm_f64(my_pair.bm, has_r ? pr : lane);
          const double tieoff = t_p ? rbp : (t_m ? -rbm : 0.0);
          const bool b_p = isv && ((ag_tp >> lane) & 1ull) != 0, b_m = isv && ((ag_tm >> lane) & 1ull) != 0;
          aset_tied = t_p || t_m;
          // the row build only fills the lower triangle (the factorisation reads nothing else); the reduction folds whole rows
          // AND columns: complete the row from the other lanes' lower parts through the LDS matrix (free until L is parked)
          if (lane < NU) {
#pragma unroll
            for (int c = 0; c < NU; ++c) Km[lane * ldk + c] = arow[c];
          }
          LSYNC();
          {
            const int a = lane < NU ? lane : 0;
            int lo = lane;
            asm volatile("" : "+v"(lo));
#pragma unroll
            // (both operands from LDS: with `c > lo ? Km[..] : arow[c]` the compiler selects between the two ADDRESSES, and an
            // address of the register row escaping sends the whole row to scratch memory)
            for (int c = 0; c < NU; ++c) { const double up = Km[c * ldk + a], low = Km[a * ldk + c]; arow[c] = c > lo ? up : low; }
          }
          LSYNC();
          double adj;
          aset::reduce_matrix<NU>(arow, tieoff, b_p || b_m, b_p ? my_pair.bp : -my_pair.bm, isv ? xu[lane] : 0.0, lane, AS, AL, adj);
          r1u += adj;                        // the pass forms its right-hand side as (...) - r1u: K' offd leaves it through here
          LSYNC();
          if (lane < NU) Km[lane * ldk + NU - 1] = 0.0;       // (Km served as the transposition buffer: its last column is never parked)
          LSYNC();
        }
      }
      double piv = readlane_f64(arow[0], 0);
      if (!(piv > 0.0)) chol_ok = false;
      double rinv = fast_rsqrt(piv);
      // The trailing update of pivot k splits three ways.  Column k + 1 is on the pivot chain (two v_readlane).  Columns up
      // to k + QP_CHOL_LOOK take the v_readlane route as well: they are read by the chains of the next pivots, and an LDS
      // round trip (store of the column, uniform-address loads: > 100 cycles) would sit on every one of them.  The rest --
      // most of the instructions -- is broadcast through LDS with 128-bit loads of two l_j each (a third of the VALU
      // instructions two v_readlane per value would cost), and APPLIED ONE PIVOT LATE: pivot k loads the column pivot
      // k - 1 stored, so that neither the store nor the loads wait on this pivot's chain.  (LDS operations of a wave
      // execute in order: pivot k's loads come before its own store into the same slot.)
      constexpr int LOOK = NU > 20 ? NU : QP_CHOL_LOOK;      // (T = 20: the loaded values do not fit beside the 40-entry row)
      double lprev = 0.0;
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        constexpr int JN = NU <= 20 ? NU / 2 : 1;
        double2 lj[JN];
        const int jf = k + LOOK;                 // first column of pivot k - 1's deferred update
        if (k >= 1 && jf < NU) {
#pragma unroll
          for (int j0 = jf & ~1; j0 < NU; j0 += 2) lj[(j0 >> 1) % JN] = ld2(dxu + j0);
        }
        if constexpr (NU <= 20) __builtin_amdgcn_sched_barrier(0);     // the loads go out before the chain, their use comes after it
        const double l = arow[k] * rinv;
        invd[k] = rinv;                      // uniform value, every lane stores it
        arow[k] = l;
        if (k + 1 < NU) {
          arow[k + 1] = fma(-l, readlane_f64(l, k + 1), arow[k + 1]);
          piv = readlane_f64(arow[k + 1], k + 1);
          if (!(piv > 0.0)) chol_ok = false;
          rinv = fast_rsqrt(piv);
        }
#pragma unroll
        for (int j = k + 2; j < NU && j <= k + LOOK; ++j) arow[j] = fma(-l, readlane_f64(l, j), arow[j]);
        // (dxu and dxd are dead between `update` and the substitution; lanes >= NU dump their copy into dxd[0]: no
        // predicate, the factorisation stays one basic block)
        if (k + 1 + LOOK < NU) dxu[lane < NU ? lane : NU] = l;
        if constexpr (NU <= 20) __builtin_amdgcn_sched_barrier(0);
        if (k >= 1 && jf < NU) {
#pragma unroll
          for (int j0 = jf & ~1; j0 < NU; j0 += 2) {
            const double2 v = lj[(j0 >> 1) % JN];
            if (j0 >= jf) arow[j0] = fma(-lprev, v.x, arow[j0]);
            if (j0 + 1 < NU) arow[j0 + 1] = fma(-lprev, v.y, arow[j0 + 1]);
          }
        }
        lprev = l;
      }
      // Park the strictly lower part of row `lane`, SCALED BY 1/L_ii, in the zero-padded LDS matrix: Lf[i][k] = L[i][k]/L[i][i].
      // Both substitutions then run without a multiplication on their serial chain:
      //   L y = b :  y_i = b_i/L_ii - sum_{k<i} Lf[i][k] y_k                          (row i of Lf)
      //   L'x = y :  z_i = y_i      - sum_{k>i} Lf[k][i] z_k,  z = diag(L) x,  x_i = z_i/L_ii   (column i of Lf)
      LSYNC();
      myinv = invd[ar];
      // (every lane of the matrix stores its whole row, zeros on and above the diagonal: one exec region and selects.  The
      // predicated form -- `if (c < lane) store` -- compiled to a branch per column with a wait for the previous store
      // ahead of each: 19 serialised LDS round trips, ~1 300 cycles per factorisation)
      if (lane < NU) {
        double* Lr = Km + lane * ldk;
        int lo = lane;
        asm volatile("" : "+v"(lo));       // (opaque: else the 19 lane masks are hoisted out of the solver loop, spilled, and reloaded)
#pragma unroll
        for (int c = 0; c < NU - 1; ++c) {
          const double v = arow[c] * myinv;
          Lr[c] = c < lo ? v : 0.0;
        }
      }
      LSYNC();
    } else {
    for (int q = lane; q < 2 * T * nu; q += QP_THREADS) {        // Y[t][k][c] = S'_t[k][:] Phi_xy[t][:, c]
      int tk = q / nu, c = q - tk * nu, t = tk >> 1, k = tk & 1;
      const double* Pt = Phi + (size_t)t * 3 * ldp;
      const double* S = St + t * QP_ST_LD;
      Yt[(size_t)tk * ldp + c] = (c <= 2 * t + 1) ? S[k] * Pt[c] + S[k + 1] * Pt[ldp + c] : 0.0;
    }
    LSYNC();
    for (int p = lane; p < npair; p += QP_THREADS) {
      int a = pa[p], c = pc[p];
      double acc0 = Hm[a * ldk + c], acc1 = 0;
#pragma unroll 4
      for (int t = a >> 1; t < T; ++t) {
        const double* Pt = Phi + (size_t)t * 3 * ldp;
        const double* Y = Yt + (size_t)t * 2 * ldp;
        acc0 = fma(Pt[a], Y[c], acc0);
        acc1 = fma(Pt[ldp + a], Y[ldp + c], acc1);
      }
      double acc = acc0 + acc1;
      if (a == c) {
        int t = a >> 1;
        double dsum = lc[2 * a] * iwc[2 * a] + lc[2 * a + 1] * iwc[2 * a + 1];
        if (t >= 1) { int q = 4 * T + 2 * (a - 2); dsum += lc[q] * iwc[q] + lc[q + 1] * iwc[q + 1]; }
        if (t <= T - 2) { int q = 4 * T + 2 * a; dsum += lc[q] * iwc[q] + lc[q + 1] * iwc[q + 1]; }
        acc += dsum;
      } else if (a == c + 2) {                       // rate rows couple u_k(t+1), u_k(t)
        int q = 4 * T + 2 * c;
        acc -= lc[q] * iwc[q] + lc[q + 1] * iwc[q + 1];
      }
      Km[a * ldk + c] = acc;
    }
    LSYNC();
    // Cholesky K' = L L' (left-looking, lane = row, matrix in LDS)
    for (int k = 0; k < nu; ++k) {
      double v = 0;
      if (lane >= k && lane < nu) {
        const double* Li = Km + (size_t)lane * ldk;
        const double* Lk = Km + (size_t)k * ldk;
        double a0 = Li[k], a1 = 0;
        int p = 0;
#pragma unroll 4
        for (; p + 1 < k; p += 2) { a0 = fma(-Li[p], Lk[p], a0); a1 = fma(-Li[p + 1], Lk[p + 1], a1); }
        if (p < k) a0 = fma(-Li[p], Lk[p], a0);
        v = a0 + a1;
      }
      double piv = readlane_f64(v, k);
      if (!(piv > 0.0)) { chol_ok = false; break; }
      double rinv = fast_rsqrt(piv);
      if (lane >= k && lane < nu) Km[(size_t)lane * ldk + k] = v * rinv;       // L[k][k] = sqrt(piv)
      if (lane == k) invd[k] = rinv;
      LSYNC();
    }
    if (chol_ok) myinv = invd[lane < nu ? lane : 0];
    }
    // (past 1e-11 the reduced matrix can lose positive definiteness in fp64: the best iterate stands, converged)
    if (!chol_ok) {
      if constexpr (ASET) { aset_why = 4; break; }        // (the reduced matrix of the guess does not factor: the interior-point launch)
      if constexpr (WARM) {
        if (warm_now && !(best_merit <= 1e-10)) { warm_code = 4; it_total += it; warm_now = false; need_cold = true; it = -1; continue; }
      }
      if constexpr (WARM) {
        if (!cold_alt && !(best_merit <= QP_RETRY_MERIT)) { cold_alt = true; warm_code = 5; it_total += it; need_cold = true; it = -1; continue; }
      }
      status = best_merit <= 1e-11 ? 0 : 3;
      break;
    }
    PROF(4);

    if constexpr (BWD) {
      if (adj) {
        // right-hand side: r1 := -dL/dx with dL/du collecting Phi' dL/ds (s_0 is pinned); every other
        // residual is zero.  s - ref of the final iterate is parked in `lin` for the q_s gradient.
        const float* gs = bw.grad_s + (size_t)b * 3 * (T + 1);
        if constexpr (TT > 0) {            // (the P_t staging of the K' build took over s3 | q3: Phi x once more)
          phi_mul(xu, s3);
          LSYNC();
        }
        for (int q = lane; q < 3 * T; q += QP_THREADS) {
          int t = q / 3, k = q - 3 * t;
          const float qsk = k == 0 ? P.q_s[0] : (k == 1 ? P.q_s[1] : P.q_s[2]);
          double refv = (double)__fmul_rn(qsk, rs[k * (T + 1) + t + 1]) / (qsk != 0.f ? (double)qsk : 1.0);
          lin[q] = (s3[q] + cv[q]) - refv;
          q3[q] = (double)gs[k * (T + 1) + t + 1];
        }
        LSYNC();
        r1u = phi_tmul(q3);
        if (lane < nu) r1u = -(r1u + (double)bw.grad_u[(size_t)b * 2 * T + (lane & 1) * T + (lane >> 1)]);
        if (lane < T) St[lane * QP_ST_LD + 7] = bw.grad_d ? -(double)bw.grad_d[(size_t)b * T + lane] : 0.0;
        LSYNC();
      }
    }
    // ================= predictor / corrector =================
    double sigma_mu = 0, alpha = 1.0;
#pragma nounroll
    for (int pass = 0; pass < ((adj || ASET) ? 1 : 2); ++pass) {
      PROF_C(7);
      if constexpr (!ASET) {
      // per-row weights of the rhs, staged in dwf/dwc/dwd (overwritten by the directions below)
      //   tfw = (r4f + lf r3)/(wf + lf/ro) ; tcw = (lc r2 - r4c)/wc ; r4 = lam w [+ dw dl - sigma mu]
      // (pass 0 must not read dw / dl: they hold the previous iteration's directions, nothing at all in the first one)
      if constexpr (HPAIR) {
        for (int h = lane; h < mf / 2; h += QP_THREADS) {
          const int i = 2 * h;
          const double2 l = ld2(lf + i), w = LD_WF(h), r = LD_R3(h), iw = ld2(iwf + i);
          double r4x = l.x * w.x, r4y = l.y * w.y;
          if (pass) { const double2 pw = ld2(dwf + i), pl = LD_DLF(h); r4x += pw.x * pl.x - sigma_mu; r4y += pw.y * pl.y - sigma_mu; }
          st2(dwf + i, adj ? 0.0 : (r4x + l.x * r.x) * iw.x, adj ? 0.0 : (r4y + l.y * r.y) * iw.y);
        }
      } else {
        for (int i = lane; i < mf; i += QP_THREADS) {
          double r4 = lf[i] * wf[i] + (pass ? dwf[i] * dlf[i] - sigma_mu : 0.0);
          dwf[i] = adj ? 0.0 : (r4 + lf[i] * r3[i]) * iwf[i];
        }
      }
      for (int p = lane; p < npc; p += QP_THREADS) {
        const int i = 2 * p;
        const double af = adj ? 0.0 : PAIR_C(p).actf;
        const double2 l = ld2(lc + i), w = LD_WC(p), r = LD_R2(p), iw = ld2(iwc + i);
        double r4x = l.x * w.x, r4y = l.y * w.y;
        if (pass) { const double2 pw = ld2(dwc + i), pl = LD_DLC(p); r4x += pw.x * pl.x - sigma_mu; r4y += pw.y * pl.y - sigma_mu; }
        st2(dwc + i, af * ((l.x * r.x - r4x) * iw.x), af * ((l.y * r.y - r4y) * iw.y));
      }
      LSYNC();
      }   // !ASET
      PROF_C(1);
      double pq0 = 0, pq1 = 0, rdr = 0;
      for (int t = lane; t < T; t += QP_THREADS) {
        double z0 = 0, z1 = 0, zs = 0;
        if constexpr (ASET) {
          // (the seeded rows carry no weights: nothing to sum)
        } else if constexpr (HPAIR) {
          constexpr int HB = WV >= 3 ? 2 : MM / 2;
#pragma unroll
          for (int j0 = 0; j0 < MM / 2; j0 += HB) {
            double2 w2[HB], p0[HB], p1[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
              const int i = t * MM + 2 * (j0 + j < MM / 2 ? j0 + j : 0);
              w2[j] = ld2(dwf + i); p0[j] = ld2(fa0 + i); p1[j] = ld2(fa1 + i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < HB; ++j)
              if (j0 + j < MM / 2) {
                z0 += w2[j].x * p0[j].x; z1 += w2[j].x * p1[j].x; zs += w2[j].x;
                z0 += w2[j].y * p0[j].y; z1 += w2[j].y * p1[j].y; zs += w2[j].y;
              }
            if constexpr (HB < MM / 2) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll 5
          for (int j = 0; j < M; ++j) { int i = t * M + j; double w = dwf[i]; z0 += w * fa0[i]; z1 += w * fa1[i]; zs += w; }
        }
        double rd = 0;
        const double* S = St + t * QP_ST_LD;                                       // v0 v1 at [3] [4], 1/kappa [6], r1_d [7]
        if (obs) rd = -S[7] - (ASET ? 0.0 : dwd[2 * t] - dwd[2 * t + 1]) + zs;           // rhs of the d rows
        double e = rd * S[6];                                               // rhs_d / kappa
        rdr = rd;
        pq0 = -(z0 - S[3] * e);
        pq1 = -(z1 - S[4] * e);
        q3[t * 3 + 0] = pq0; q3[t * 3 + 1] = pq1; q3[t * 3 + 2] = 0.0;
      }
      double rr;
      PROF(5);
      LSYNC();
      PROF_C(2);
      rr = phi_tmul(q3);
      if (lane < nu) rr += -r1u - (ASET ? 0.0 : ct_mul(dwc, lane));
      if constexpr (ASET) {
        if (aset) rr = aset::reduce_rhs<NU>(lane < nu ? rr : 0.0, lane, AL);      // Z'(r - K' offd) (offd went in with r1u), zero on the members that left
      }
      PROF_C(3);
      if constexpr (TT > 0) {
        // forward substitution L y = rhs, backward L' dx = y; lane i owns entry i and reads row i / column i of the
        // zero-padded L (loads that do not depend on the chain: they are issued ahead of it)
        const int lr = lane < NU ? lane : 0;
        rr *= myinv;                         // b_i / L_ii
        if constexpr (NU <= 20) {
          if constexpr (WV >= 3) {
            // (three waves per SIMD: row and column of L one after the other through the same registers)
            {
              double Lrow[NU];
#pragma unroll
              for (int k = 0; k < NU; ++k) Lrow[k] = Km[lr * ldk + k];
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int k = 0; k < NU; ++k) rr = fma(-Lrow[k], readlane_f64(rr, k), rr);          // -> y
            }
            __builtin_amdgcn_sched_barrier(0);
            {
              double Lcol[NU];
#pragma unroll
              for (int k = 0; k < NU; ++k) Lcol[k] = Km[k * ldk + lr];
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int k = NU - 1; k >= 0; --k) rr = fma(-Lcol[k], readlane_f64(rr, k), rr);     // -> z
            }
          } else {
            double Lrow[NU], Lcol[NU];
#pragma unroll
            for (int k = 0; k < NU; ++k) Lrow[k] = Km[lr * ldk + k];
            __builtin_amdgcn_sched_barrier(0);         // all of row `lane` is on its way before the chain starts
#pragma unroll
            for (int k = 0; k < NU; ++k) Lcol[k] = Km[k * ldk + lr];   // (free to overlap the forward chain)
#pragma unroll
            for (int k = 0; k < NU; ++k) rr = fma(-Lrow[k], readlane_f64(rr, k), rr);          // -> y
#pragma unroll
            for (int k = NU - 1; k >= 0; --k) rr = fma(-Lcol[k], readlane_f64(rr, k), rr);     // -> z
          }
        } else {                             // (T = 20: 2 x 40 doubles ahead of the chains do not fit the register file)
#pragma unroll
          for (int k = 0; k < NU; ++k) rr = fma(-Km[lr * ldk + k], readlane_f64(rr, k), rr);
#pragma unroll
          for (int k = NU - 1; k >= 0; --k) rr = fma(-Km[k * ldk + lr], readlane_f64(rr, k), rr);
        }
        rr *= myinv;                         // dx_u
        if constexpr (ASET) {
          if (aset) rr = aset::expand<NU>(rr, lane, AS, AL);                        // du = Z z + offd (offd waits in s3[0 .. NU): nothing writes s3 before phi_mul(dxu) below)
        }
      } else {
        for (int k = 0; k < nu; ++k) {
          double yk = readlane_f64(rr * myinv, k);
          if (lane == k) rr = yk;
          if (lane > k && lane < nu) rr = fma(-Km[(size_t)lane * ldk + k], yk, rr);
        }
        for (int k = nu - 1; k >= 0; --k) {
          double xk = readlane_f64(rr * myinv, k);
          if (lane == k) rr = xk;
          if (lane < k) rr = fma(-Km[(size_t)k * ldk + lane], xk, rr);
        }
      }
      if (lane < nu) dxu[lane] = rr;
      LSYNC();
      PROF(6); PROF_C(4);
      phi_mul(dxu, s3);
      LSYNC();
      for (int t = lane; t < T && obs; t += QP_THREADS)
        dxd[t] = (rdr + St[t * QP_ST_LD + 3] * s3[t * 3] + St[t * QP_ST_LD + 4] * s3[t * 3 + 1]) * St[t * QP_ST_LD + 6];
      LSYNC();
      PROF_C(5);
      if constexpr (ASET) {
        if (aset) { alpha = 1.0; break; }    // the Newton step of the guess is taken in full; the rows are re-seeded, not moved
      }
      // directions of multipliers / slacks and the step to the boundary
      double amax = 1.0, gap_aff = 0;
      // step to the boundary of one row: dl, dw < 0 bound alpha by -l/dl, -w/dw
      auto ratio = [&](double l, double dl, double w, double dw) {
        if (dl < 0) amax = fmin(amax, -l * rough_rcp(dl));
        if (dw < 0) amax = fmin(amax, -w * rough_rcp(dw));
      };
      if constexpr (HPAIR) {
        for (int h = lane; h < mf / 2; h += QP_THREADS) {
          const int t = h / (MM / 2), i = 2 * h;
          const double2 l = ld2(lf + i), w = LD_WF(h), a0 = ld2(fa0 + i), a1 = ld2(fa1 + i), iw = ld2(iwf + i), tw = ld2(dwf + i), r = LD_R3(h);
          const double sx = s3[t * 3], sy = s3[t * 3 + 1], d = dxd[t];
          const double Fx = a0.x * sx + a1.x * sy - d, Fy = a0.y * sx + a1.y * sy - d;
          const double dlx = -(tw.x + l.x * Fx * iw.x), dly = -(tw.y + l.y * Fy * iw.y);       // -(r4 + l r3 + l Fdx)/(w + l/ro)
          const double dwx = Fx + dlx * iro + r.x, dwy = Fy + dly * iro + r.y;
          ratio(l.x, dlx, w.x, dwx); ratio(l.y, dly, w.y, dwy);
          ST_ROW(Rdlf, dlf, h, dlx, dly); st2(dwf + i, dwx, dwy);
        }
      } else {
        for (int i = lane; i < mf; i += QP_THREADS) {
          int t = i / M;
          double l = lf[i], w = wf[i];
          double Fdx = fa0[i] * s3[t * 3] + fa1[i] * s3[t * 3 + 1] - dxd[t];
          double dl = -(dwf[i] + l * Fdx * iwf[i]);             // -(r4 + l r3 + l Fdx)/(w + l/ro)
          double dw = Fdx + dl * iro + r3[i];
          ratio(l, dl, w, dw);
          dlf[i] = dl; dwf[i] = dw;
        }
      }
      for (int p = lane; p < npc; p += QP_THREADS) {
        const PairC c = PAIR_C(p);
        const int i = 2 * p;
        const double2 l = ld2(lc + i), w = LD_WC(p), iw = ld2(iwc + i), tw = ld2(dwc + i), r = LD_R2(p);
        const double Cdx = fma(-c.sb, dxu[c.ib], dxu[c.ia]);
        const double dlx = tw.x + l.x * Cdx * iw.x, dly = tw.y - l.y * Cdx * iw.y;           // (l r2 - r4 + l Cdx)/w
        const double dwx = (-r.x - Cdx) * c.actf, dwy = (Cdx - r.y) * c.actf;
        ratio(l.x, dlx, w.x, dwx); ratio(l.y, dly, w.y, dwy);
        ST_ROW(Rdlc, dlc, p, dlx, dly); st2(dwc + i, dwx, dwy);
      }
      PROF_C(6);
      amax = wave_reduce<OpMin>(amax);
      if (pass == 0) {
        if constexpr (HPAIR) {
          for (int h = lane; h < mf / 2; h += QP_THREADS) {
            const double2 l = ld2(lf + 2 * h), w = LD_WF(h), dl = LD_DLF(h), dw = ld2(dwf + 2 * h);
            gap_aff += (l.x + amax * dl.x) * (w.x + amax * dw.x) + (l.y + amax * dl.y) * (w.y + amax * dw.y);
          }
        } else {
          for (int i = lane; i < mf; i += QP_THREADS) gap_aff += (lf[i] + amax * dlf[i]) * (wf[i] + amax * dwf[i]);
        }
        for (int p = lane; p < npc; p += QP_THREADS) {          // (switched-off pairs: l = dl = 0)
          const double2 l = ld2(lc + 2 * p), w = LD_WC(p), dl = LD_DLC(p), dw = ld2(dwc + 2 * p);
          gap_aff += (l.x + amax * dl.x) * (w.x + amax * dw.x) + (l.y + amax * dl.y) * (w.y + amax * dw.y);
        }
        double mu_aff = wave_reduce<OpSum>(gap_aff) * inv_m;
        double sg = mu_aff * fast_rcp(mu);
        sigma_mu = fmax(sg * sg * sg * mu, QP_SIGMA_MU_MIN);
      } else {
        alpha = fmin(1.0, fmin(fmax(QP_STEP_ETA, 1.0 - mu), 1.0 - QP_STEP_CAP) * amax);
      }
      LSYNC();
      PROF(7); PROF_C(7);
    }
    if constexpr (BWD) {
      if (adj) {
        // v = (dxu, dxd), Phi v in s3, D C v of the d rows in dld:  dL/dtheta = -v' d(Hx+g)/dtheta, dL/dc = D C v
        double g0 = 0, g1 = 0, g2 = 0, gp = 0, ge = 0, gmx = 0, gmn = 0;
        for (int q = lane; q < 3 * T; q += QP_THREADS) {
          int k = q % 3;
          double v = s3[q] * lin[q];
          g0 += k == 0 ? v : 0.0; g1 += k == 1 ? v : 0.0; g2 += k == 2 ? v : 0.0;
        }
        for (int t = lane; t < T; t += QP_THREADS) {
          double refu = (double)__fmul_rn(P.p_u, rus[t]) / (P.p_u != 0.f ? (double)P.p_u : 1.0);
          gp += dxu[2 * t] * (xu[2 * t] - refu);
          if (obs) {
            ge += dxd[t];
            if constexpr (!REGROWS) { gmx += dld[2 * t]; gmn += dld[2 * t + 1]; }
          }
        }
        if constexpr (REGROWS) {           // the d rows' multiplier directions sit in their owner lanes' registers (pair npu + t)
          if (obs && lane >= npu && lane < npc) { gmx = Rdlc.x; gmn = Rdlc.y; }
        }
        if (bw.grad_nom_s) {
          // the only input of this solve that the reference keeps on its autograd graph besides theta:
          // para_s in 0.5 bk |s - para_s|^2 (robot.py:178); d(Hx+g)/dpara_s[:,t] = -bk Phi_t'
          float* gn = bw.grad_nom_s + (size_t)b * 3 * (T + 1);
          for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) {
            int k = q / (T + 1), t = q - k * (T + 1);
            gn[q] = (t == 0) ? 0.f : (float)((double)P.bk * s3[(t - 1) * 3 + k]);
          }
        }
        g0 = wave_reduce<OpSum>(g0); g1 = wave_reduce<OpSum>(g1); g2 = wave_reduce<OpSum>(g2);
        gp = wave_reduce<OpSum>(gp); ge = wave_reduce<OpSum>(ge);
        gmx = wave_reduce<OpSum>(gmx); gmn = wave_reduce<OpSum>(gmn);
        if (lane == 0) {
          float* gt = bw.grad_theta + (size_t)b * 8;
          gt[0] = (float)(-4.0 * (double)P.q_s[0] * g0);
          gt[1] = (float)(-4.0 * (double)P.q_s[1] * g1);
          gt[2] = (float)(-4.0 * m2 * (double)P.q_s[2] * g2);
          gt[3] = (float)(-4.0 * pu * gp);
          gt[4] = (float)ge;
          gt[5] = (float)gmx;
          gt[6] = (P.d_min > 0.f) ? (float)(-gmn) : 0.f;
          gt[7] = (float)status;
        }
        break;
      }
    }
    for (int a = lane; a < nu; a += QP_THREADS) xu[a] += alpha * dxu[a];
    for (int t = lane; t < T && obs; t += QP_THREADS) xd[t] += alpha * dxd[t];
    if constexpr (ASET) {                    // d is projected onto its bounds (a frozen d then never has to travel); the rows are re-seeded, not moved
      for (int t = lane; t < T && obs; t += QP_THREADS) xd[t] = fmin(fmax(xd[t], dmin0), dmaxv);
    } else {
    if constexpr (HPAIR) {
      for (int h = lane; h < mf / 2; h += QP_THREADS) {
        const double2 l = ld2(lf + 2 * h), w = LD_WF(h), dl = LD_DLF(h), dw = ld2(dwf + 2 * h);
        st2(lf + 2 * h, l.x + alpha * dl.x, l.y + alpha * dl.y); ST_ROW(Rwf, wf, h, w.x + alpha * dw.x, w.y + alpha * dw.y);
      }
    } else {
      for (int i = lane; i < mf; i += QP_THREADS) { lf[i] += alpha * dlf[i]; wf[i] += alpha * dwf[i]; }
    }
    for (int p = lane; p < npc; p += QP_THREADS) {
      const double2 l = ld2(lc + 2 * p), w = LD_WC(p), dl = LD_DLC(p), dw = ld2(dwc + 2 * p);
      st2(lc + 2 * p, l.x + alpha * dl.x, l.y + alpha * dl.y); ST_ROW(Rwc, wc, p, w.x + alpha * dw.x, w.y + alpha * dw.y);
    }
    }   // !ASET
    LSYNC();
    PROF(8);
  }
  LSYNC();
  it_total += it;
  if constexpr (ASET) {
    if (!aset_done) {                      // nothing was written: the interior-point launch behind this one solves the scene
#ifndef NPA_QP_PROF
      if (qp_info && lane == 0) {
        double* qi = qp_info + (size_t)b * QP_INFO_STRIDE;
        qi[5] = aset_guess; qi[6] = aset_left; qi[7] = status == 2 ? 6 : aset_why; qi[8] = aset_first;
      }
#endif
      return;
    }
  }
  if (warm_now) warm_code = 1;
  if (aset_done) warm_code = 6;          // the active-set iteration delivered this solve
  if (status == 0 && !(best_merit <= QP_RETRY_MERIT)) status = 4;      // both cold attempts ended short of convergence

  if (bw.dbg_x)
    for (int a = lane; a < nu + T; a += QP_THREADS) bw.dbg_x[(size_t)b * (nu + T) + a] = (a < nu || obs) ? xbest[a] : 0.0;
  // ---- write the solution (fp64 -> fp32, nrmp.py:145-148) ------------------------------------
  float* so = cur_s_out + (size_t)b * 3 * (T + 1);
  float* uo = cur_u_out + (size_t)b * 2 * T;
  phi_mul(xbest, s3);
  LSYNC();
  float* stage = reinterpret_cast<float*>(Km);       // cur_s_out may alias cur_s_in
  for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) {
    int k = q / (T + 1), t = q - k * (T + 1);
    stage[q] = (t == 0) ? s_in[k * (T + 1)] : (float)(s3[(t - 1) * 3 + k] + cv[(t - 1) * 3 + k]);
  }
  LSYNC();
  for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) {
    float fv = stage[q];
    so[q] = fv;
    if (out_s) out_s[(size_t)b * 3 * (T + 1) + q] = fv;
  }
  if (trig_out)                         // the next iteration's DUNE launches read the rotation from this table
    for (int t = lane; t <= T; t += QP_THREADS) {
      float c, sn;
      npa_trig(stage[2 * (T + 1) + t], c, sn);
      trig_out[((size_t)b * (T + 1) + t) * 2] = c;
      trig_out[((size_t)b * (T + 1) + t) * 2 + 1] = sn;
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
  if (wrm) {
    for (int a = lane; a < nu + T; a += QP_THREADS) wrm[a] = xbest[a];
    for (int i = lane; i < mf; i += QP_THREADS) wrm[nu + T + i] = lf[i];
    for (int i = lane; i < mcu; i += QP_THREADS) wrm[nu + T + mf + i] = lc[i];
    for (int i = lane; i < 2 * T && obs; i += QP_THREADS) wrm[nu + T + mf + mcu + i] = ld_[i];
    // the next solve of this scene may start from this one when it converged.  (Rounds 2 and 3 also meant to require that
    // the solve moved the controls by < 0.1 from the nominal it was linearised around -- but the forward call's working
    // nominal is updated in place (cur_u_out aliases cur_u_in), so by the time that distance was taken it compared the
    // solution with itself and the condition always held.  The CPU replay of both rules over the benchmark QPs
    // (tests/tools/qp_warm_share.py) says the accident is the better rule: a start that is far off is refused at iteration 0
    // by its merit anyway (> 0.05), and the moderately far ones that get through save more iterations than the few that
    // are dropped at iteration 6 cost -- 7.7 vs 8.1 iterations per solve on configs[1], 7.4 vs 7.5 on the car.  So the
    // condition is gone, on purpose.)
    if (flags && lane == 0) {
      flags[b * 4 + 2] = (status == 0 && best_merit <= 1e-12) ? 1 : 0;
      if constexpr (ASET) flags[b * 4 + 3] = 1;          // (the interior-point launch behind this one skips the scene)
    }
  }
  if (qp_info && lane == 0) {
    double* qi = qp_info + (size_t)b * QP_INFO_STRIDE;
    qi[0] = best_it; qi[1] = best_merit; qi[2] = last_mu; qi[3] = status; qi[4] = it;
    qi[14] = it_total; qi[15] = warm_code;
#ifndef NPA_QP_PROF
    if constexpr (ASET) { qi[5] = aset_guess; qi[6] = aset_left; qi[7] = aset_why; qi[8] = aset_first; }
    else if (P.qp_aset && !(can_warm)) { qi[5] = 0; qi[6] = 0; qi[7] = 0; qi[8] = 0; }      // (no attempt was made on this scene)
#endif
#ifdef NPA_QP_PROF
    PROF(9);
    for (int i = 0; i < 10; ++i) qi[5 + i] = (double)pacc_[i];
#endif
  }

  // ---- per-forward outputs of the last executed iteration -------------------------------------
  const int cnt0 = (obs && count) ? count[(size_t)b * (T + 1)] : 0;
  if (out_min_distance && lane == 0) {
    // DUNE.min_distance is only assigned by a forward WITH points (dune.py:97-98) and keeps its value otherwise
    // (pan.py:246-252 reads the attribute): the last value lives in the scene's state record, next to the stop
    // criterion's memory (ints 2, 3 of its tail), so it carries over whether or not anybody read it in between
    float mdv = __builtin_inff();
    const bool have_md = obs && cnt0 > 0;
    if (have_md) mdv = dist_sorted[(size_t)b * (T + 1) * M];
    if (state) {
      const int Ms_ = M > 0 ? M : 1;
      int* tail = reinterpret_cast<int*>(state + (size_t)(b + 1) * npa_state_floats(T, Ms_, E)) - 4;
      if (have_md) { tail[2] = 1; tail[3] = __float_as_int(mdv); }
      else if (tail[2]) mdv = __int_as_float(tail[3]);
    }
    out_min_distance[b] = mdv;
  }
  if (out_nrmp_points && obs)
    for (int q = lane; q < 2 * M; q += QP_THREADS) {
      int k = q / M, j = q - k * M;
      out_nrmp_points[(size_t)b * 2 * M + q] = cnt0 > 0 ? pts_sorted[((size_t)b * (T + 1) * M + j) * 2 + k] : 0.f;
    }

  // ---- stop criterion (pan.py:215-243); state persists across forward calls ---------------------
  if (state && flags) {
    const int Ms = M > 0 ? M : 1;
    const size_t nsf = npa_state_floats(T, Ms, E);
    float* st = state + (size_t)b * nsf;
    float* ps = st;
    float* pu_ = ps + 3 * (T + 1);
    float* pmu = pu_ + 2 * T;
    float* plam = pmu + (size_t)(T + 1) * Ms * E;
    int* pint = reinterpret_cast<int*>(plam + (size_t)(T + 1) * Ms * 2);
    const int valid = pint[0], prev_n = pint[1];
    const bool have = obs && cnt0 > 0;
    double acc_s = 0, acc_u = 0, acc_mu = 0, acc_lam = 0;
    int eff = 0;
    if (valid) {
      if (!have || prev_n == 0) {
        for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) { double d = (double)stage[q] - (double)ps[q]; acc_s += d * d; }
        for (int q = lane; q < 2 * T; q += QP_THREADS) {
          int k = q / T, t = q - k * T;
          double d = (double)(float)xbest[2 * t + k] - (double)pu_[q];
          acc_u += d * d;
        }
      } else {
        eff = cnt0 < prev_n ? cnt0 : prev_n;
        for (int q = lane; q < (T + 1) * eff; q += QP_THREADS) {
          int t = q / eff, j = q - t * eff;
          size_t row = ((size_t)b * (T + 1) + t) * M + j;
          size_t prow = (size_t)t * M + j;
          // (loads first, unconditional: a run-time trip count made every one of them a round trip of its own)
          float mn[NPA_MAX_E], mo[NPA_MAX_E], ln[2], lo[2];
#pragma unroll
          for (int e = 0; e < NPA_MAX_E; ++e) { const int ee = e < E ? e : 0; mn[e] = mu_sorted[row * E + ee]; mo[e] = pmu[prow * E + ee]; }
#pragma unroll
          for (int k = 0; k < 2; ++k) { ln[k] = lam_sorted[row * 2 + k]; lo[k] = plam[prow * 2 + k]; }
#pragma unroll
          for (int e = 0; e < NPA_MAX_E; ++e) { const double d = (double)mn[e] - (double)mo[e]; acc_mu = e < E ? fma(d, d, acc_mu) : acc_mu; }
#pragma unroll
          for (int k = 0; k < 2; ++k) { const double d = (double)ln[k] - (double)lo[k]; acc_lam = fma(d, d, acc_lam); }      // (fused, as the compiler contracted the loop form)
        }
      }
    }
    acc_s = wave_reduce<OpSum>(acc_s); acc_u = wave_reduce<OpSum>(acc_u);
    acc_mu = wave_reduce<OpSum>(acc_mu); acc_lam = wave_reduce<OpSum>(acc_lam);
    // remember the current iterate
    for (int q = lane; q < 3 * (T + 1); q += QP_THREADS) ps[q] = stage[q];
    for (int q = lane; q < 2 * T; q += QP_THREADS) { int k = q / T, t = q - k * T; pu_[q] = (float)xbest[2 * t + k]; }
    if (have)
      for (int q = lane; q < (T + 1) * M; q += QP_THREADS) {
        size_t row = (size_t)b * (T + 1) * M + q;
        float mn[NPA_MAX_E];
#pragma unroll
        for (int e = 0; e < NPA_MAX_E; ++e) mn[e] = mu_sorted[row * E + (e < E ? e : 0)];
        const float ln0 = lam_sorted[row * 2], ln1 = lam_sorted[row * 2 + 1];
#pragma unroll
        for (int e = 0; e < NPA_MAX_E; ++e) if (e < E) pmu[(size_t)q * E + e] = mn[e];
        plam[(size_t)q * 2] = ln0; plam[(size_t)q * 2 + 1] = ln1;
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

static bool qp_fast_path(int T, int M) { return (T == 10 || T == 20) && M == 10; }

// LDS bytes of one scene; `fast` = the register-resident instantiation (packed H, L and P staging)
static bool qp_scan_wide() {
  static const bool v = getenv("NPA_QP_NOSCAN_WIDE") == nullptr;
  return v;
}
// LDS bytes of one scene; fast = the register-resident instantiation (packed H, L and P staging); scan = its form
// without a stored Phi (T <= 16, or T <= 32 with the wide scans)
extern "C" size_t npa_qp_shmem_bytes_path2(int T, int M, int fast, int scan) {
  const bool obs = M > 0;
  size_t nu = 2 * T, ldp = nu + 1, mcd = 8 * T - 4 + 2 * T, mf = obs ? (size_t)T * M : 0, npair = nu * (nu + 1) / 2;
  const size_t mfe = (mf + 1) & ~(size_t)1;
  // (mirrors the kernel's carve: REGROWS keeps 3 of the 6 u / d row arrays and 5 of the 9 hinge row arrays in LDS)
  const bool regrows = fast && M > 0 && M % 2 == 0 && T * M / 2 <= QP_THREADS && 5 * T - 2 <= QP_THREADS;
  const size_t rows = (regrows ? 3 : 6) * mcd + (regrows ? 5 : 9) * mfe;
  const size_t mats = fast ? nu * (nu + 1) / 2 + nu * ldp : (size_t)T * 2 * ldp + 2 * nu * ldp;
  const size_t phi = (fast && scan) ? 0 : (size_t)T * 3 * ldp;
  size_t d = rows + phi + mats + 4 * (T * 3) + T * QP_ABC_LD + (((size_t)T * QP_ST_LD + 1) & ~(size_t)1) + nu + T + (nu + T) + nu + T + nu +
             ((fast && scan) ? 2 * (size_t)T : 0);
  size_t bytes = d * sizeof(double) + (fast ? 0 : 2 * ((npair + 7) & ~(size_t)7));
  return (bytes + 15) & ~(size_t)15;
}
extern "C" size_t npa_qp_shmem_bytes_path(int T, int M, int fast) {
  const bool scan = fast && (T <= 16 || (T <= 32 && qp_scan_wide()));
  return npa_qp_shmem_bytes_path2(T, M, fast, scan ? 1 : 0);
}

extern "C" size_t npa_qp_shmem_bytes(int T, int M) { return npa_qp_shmem_bytes_path(T, M, 0); }

extern "C" hipError_t npa_launch_qp(const DevParams& P, int batch, int scene0, const float* cur_s_in,
                                    const float* cur_u_in, const float* ref_s, const float* ref_us,
                                    const float* mu_sorted, const float* lam_sorted, const float* pts_sorted,
                                    const float* dist_sorted, const int* count, float* cur_s_out, float* cur_u_out,
                                    float* cur_d_out, float* out_s, float* out_u, float* out_d,
                                    float* out_min_distance, int* out_iters, float* out_nrmp_points, int* flags,
                                    float* state, double* qp_info, double* warm, float* trig_out, float* dbg_abc,
                                    float* dbg_f, double* dbg_x,
                                    hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, int aset_launch) {
  static const bool force_generic = getenv("NPA_QP_GENERIC") != nullptr;     // tests: the generic (LDS) instantiation for every (T, M)
  const bool fast = qp_fast_path(P.T, P.M) && !force_generic;
  const size_t shmem = npa_qp_shmem_bytes_path(P.T, P.M, fast ? 1 : 0);      // one scene (wave) per workgroup, see the kernel
  // (T = 20 without the wide scans keeps Phi: npa_qp_shmem_bytes_path follows the same switch)
  const int nblocks = batch;
  static NpaDeviceOnce attr_set;       // (per device: the attribute is, and a process may hold handles on several GPUs)
  int dev_ = 0;
  if (attr_set.need(&dev_)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<10, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<10, 10, false, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<20, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<20, 10, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.done(dev_);
  }
#define QP_LAUNCH(...)                                                                                           \
  hipExtLaunchKernelGGL((nrmp_qp_kernel<__VA_ARGS__>), dim3(nblocks), dim3(QP_THREADS), shmem, stream, ev_start, ev_stop, 0, \
                        P, cur_s_in, cur_u_in, ref_s, ref_us, mu_sorted, lam_sorted, pts_sorted, dist_sorted, count,  \
                        cur_s_out, cur_u_out, cur_d_out, out_s, out_u, out_d, out_min_distance, out_iters,            \
                        out_nrmp_points, flags, state, qp_info, warm, scene0, batch,                                 \
                        QpBackward{nullptr, nullptr, nullptr, nullptr, nullptr, dbg_abc, dbg_f, dbg_x}, trig_out)
  const bool scan_wide = qp_scan_wide();
  if (aset_launch) {
    // the active-set launch that precedes the interior-point launch of the same PAN iteration (see the kernel's top)
    if (!(P.T == 10 && P.M == 10 && !force_generic && P.qp_aset && warm && flags)) return hipErrorInvalidValue;
    QP_LAUNCH(10, 10, false, false, 2, true);
  }
  else if (P.T == 10 && P.M == 10 && !force_generic) QP_LAUNCH(10, 10);
  else if (P.T == 20 && P.M == 10 && !force_generic && scan_wide) QP_LAUNCH(20, 10, false, true);
  else if (P.T == 20 && P.M == 10 && !force_generic) QP_LAUNCH(20, 10);
  else QP_LAUNCH(0, 0);
#undef QP_LAUNCH
  return hipGetLastError();
}

// forward solve + gradient w.r.t. the adjust parameters (generic kernel, one scene per workgroup)
extern "C" hipError_t npa_launch_qp_backward(const DevParams& P, int batch, const float* nom_s, const float* nom_u,
                                             const float* ref_s, const float* ref_us, const float* mu_sorted,
                                             const float* lam_sorted, const float* pts_sorted, const int* count,
                                             float* out_s, float* out_u, float* out_d, const float* grad_s,
                                             const float* grad_u, const float* grad_d, float* grad_theta,
                                             float* grad_nom_s, double* qp_info, hipStream_t stream) {
  static const bool force_generic = getenv("NPA_QP_GENERIC") != nullptr;
  const bool fast = qp_fast_path(P.T, P.M) && !force_generic && (P.T <= 16 || qp_scan_wide());
  const size_t wave_bytes = npa_qp_shmem_bytes_path(P.T, P.M, fast ? 1 : 0);
  static NpaDeviceOnce attr_set;
  int dev_ = 0;
  if (attr_set.need(&dev_)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<10, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<20, 10, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.done(dev_);
  }
  // (the register-resident instantiations since round 3: the gradient pass reads the multiplier directions of the d rows
  // from their owner lanes; NPA_QP_GENERIC=1 keeps the LDS kernel)
#define QPB_LAUNCH(...)                                                                                                \
  hipLaunchKernelGGL((nrmp_qp_kernel<__VA_ARGS__>), dim3(batch), dim3(QP_THREADS), wave_bytes, stream, P, nom_s, nom_u, ref_s, \
                     ref_us, mu_sorted, lam_sorted, pts_sorted, (const float*)nullptr, count, out_s, out_u, out_d,        \
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int*)nullptr, (float*)nullptr,  \
                     (int*)nullptr, (float*)nullptr, qp_info, (double*)nullptr, 0, batch,                                 \
                     QpBackward{grad_s, grad_u, grad_d, grad_theta, grad_nom_s, nullptr, nullptr, nullptr}, (float*)nullptr)
  if (fast && P.T == 10) QPB_LAUNCH(10, 10, true);
  else if (fast && P.T == 20) QPB_LAUNCH(20, 10, true, true);
  else QPB_LAUNCH(0, 0, true);
#undef QPB_LAUNCH
  return hipGetLastError();
}
