// aset_reduce.h -- device functions of the active-set iteration for the warm QP solves (branch qp-active-set; DESIGN.md
// section 7).  One 64-lane wave per system, variable a = 2t + k in lane a.  tests/tools/qp_active_set_study.py states every
// step in numpy (lane_level_reduction, lane_level_multipliers); tests/test_aset_reduce.py compares them on the GPU through the
// debug kernel of aset_reduce.hip (green on an MI355X for the single-call form this file was split from).
//
// The system before the tight speed / rate rows of the controls are eliminated:  K du = r  (Newton form: du = u_new - u_cur),
// row a of K in lane a (NU registers).  Per variable lane:
//     tieoff:  0, or the signed offset of a tight rate row into a:  u_new(a) = u_new(a - 2) + tieoff
//     bound:   whether a speed row of a is tight, bndval its value:  u_new(a) = bndval
// Runs of tied controls share one unknown z (the head's new value); a run with a tight speed row is fixed altogether:
//     u_new = Z z + offvec,   du = Z z + offd,  offd = offvec - u_cur,   (Z'KZ) z = Z'(r - K offd),   unit rows for the rest.
#pragma once
#include <hip/hip_runtime.h>

namespace aset {

__device__ __forceinline__ int bperm_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ double bperm_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
#define ASET_LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// LDS scratch, caller-provided: ov / slotv [NU] doubles, Mt [NU][ldm] doubles (ldm >= NU), winner [NU] ints
struct Scratch { double *ov, *slotv, *Mt; int* winner; int ldm; };
// what a lane keeps between the three parts
struct Lane { int head, mwin; bool anchored, gone, fold; };

// Part 1, when the rows of K are in registers (before the factorisation): structure of the runs, K -> Z'KZ with unit rows for
// the members that left.  adj: this lane's entry of K offd -- the caller takes it off its right-hand side before part 2 (the QP
// kernel adds it to the dual residual it keeps anyway).  offd stays in S.ov[lane] for part 3.
template <int NU>
__device__ __forceinline__ void reduce_matrix(double (&arow)[NU], double tieoff, bool bound, double bndval, double ucur, int lane, const Scratch& S, Lane& L,
                              double& adj) {
  const bool live = lane < NU;
  constexpr int NSTEP = NU > 32 ? 5 : 4;
  // runs: segmented inclusive scan along the stride-2 chains (flag = head of a run)
  double v = live ? tieoff : 0.0;
  bool flag = !live || tieoff == 0.0;
  int h = flag ? lane : -1;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int dist = 2 << s, src = lane - dist;
    const bool ok = src >= 0;
    const double vp = bperm_f64(v, ok ? src : lane);
    const int fp = bperm_i(flag ? 1 : 0, ok ? src : lane), hp = bperm_i(h, ok ? src : lane);
    if (!flag) { v += ok ? vp : 0.0; h = ok ? hp : -1; }
    flag = flag || !ok || fp != 0;
  }
  const double off = v;
  const int head = live ? h : lane;
  if (live) S.winner[lane] = 0x7fffffff;
  ASET_LSYNC();
  if (live && bound) atomicMin(&S.winner[head], lane);           // the lowest member with a tight speed row anchors the run
  ASET_LSYNC();
  if (live && bound && S.winner[head] == lane) S.slotv[head] = bndval - off;
  ASET_LSYNC();
  const bool anchored = live && S.winner[head] != 0x7fffffff;
  const double offvec = off + (anchored ? S.slotv[head] : 0.0);
  L.head = head; L.anchored = anchored; L.mwin = anchored ? S.winner[head] : 0x7fffffff;
  L.fold = live && tieoff != 0.0 && !anchored;                     // column / row `lane` folds into lane - 2
  L.gone = live && (tieoff != 0.0 || anchored);                    // variable `lane` leaves the system
  if (live) S.ov[lane] = offvec - ucur;
  // which columns fold into their predecessor / leave the system: wave-uniform per column, so two ballots in scalar registers
  // (variable a sits in lane a: bit c of a ballot is column c) instead of 2 NU values through LDS and vector registers
  const unsigned long long fold_mask = __ballot(L.fold), gone_mask = __ballot(L.gone);
  ASET_LSYNC();
  {
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int c = 0; c < NU; c += 2) { acc0 = fma(arow[c], S.ov[c], acc0); acc1 = fma(arow[c + 1], S.ov[c + 1], acc1); }
    adj = live ? acc0 + acc1 : 0.0;
  }
#define ASET_MERGE()                                                                                      \
  do {                                                                                                    \
    _Pragma("unroll") for (int c = NU - 1; c >= 2; --c) arow[c - 2] += ((fold_mask >> c) & 1ull) ? arow[c] : 0.0; \
    _Pragma("unroll") for (int c = 0; c < NU; ++c) arow[c] = ((gone_mask >> c) & 1ull) ? 0.0 : arow[c];   \
  } while (0)
  ASET_MERGE();
  if (live) {
#pragma unroll
    for (int c = 0; c < NU; ++c) S.Mt[lane * S.ldm + c] = arow[c];
  }
  ASET_LSYNC();
  {
    const int a = live ? lane : 0;
#pragma unroll
    for (int c = 0; c < NU; ++c) arow[c] = S.Mt[c * S.ldm + a];     // K is symmetric: Z'KZ = ((KZ)'Z)'
  }
  ASET_LSYNC();
  ASET_MERGE();
#undef ASET_MERGE
  {
    int lo = lane;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int c = 0; c < NU; ++c) arow[c] = L.gone ? (c == lo ? 1.0 : 0.0) : arow[c];
  }
}

// Part 2: the right-hand side of the reduced system, Z'r with r = (right-hand side) - K offd, zero on the lanes that left
template <int NU>
__device__ __forceinline__ double reduce_rhs(double r, int lane, const Lane& L) {
  const bool live = lane < NU;
  constexpr int NSTEP = NU > 32 ? 5 : 4;
  double val = live ? r : 0.0;
  const bool okn = lane + 2 < 64;
  const int nxt = bperm_i(L.fold ? 1 : 0, okn ? lane + 2 : lane);       // (every lane executes every gather)
  int link = (live && okn) ? nxt : 0;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int dist = 2 << s, src = lane + dist;
    const bool ok = src < 64;
    const double vn = bperm_f64(val, ok ? src : lane);
    const int ln = bperm_i(link, ok ? src : lane);
    val += (link && ok) ? vn : 0.0;
    link = (link && ok) ? ln : 0;
  }
  return L.gone ? 0.0 : val;
}

// Part 3: du of every variable from the reduced solution z (z in the head lanes); offd is read back from S.ov
template <int NU>
__device__ __forceinline__ double expand(double z, int lane, const Scratch& S, const Lane& L) {
  const bool live = lane < NU;
  const double zh = bperm_f64(z, live ? L.head : lane);
  return live ? ((L.anchored ? 0.0 : zh) + S.ov[lane]) : 0.0;
}

// Multipliers of the tight rows from the stationarity residual res (per variable lane) at the new point.  Along a run the tie
// row INTO member j carries S[j] = sum of res over the members from j to the run's end, less beta = S[head] when the member whose
// speed row anchors the run sits at or behind j; that member's speed row carries beta.  `tied`: a rate row into this lane is tight
// (whatever the run's anchoring).  Returns the raw values: the caller applies the row's sign (a correct guess gives >= 0 then).
// run_sum: the sum of res over the run from this lane on -- at the head of a run that no speed row anchors it is what the reduced
// system left unsolved (the dual residual of the free unknown).
template <int NU>
__device__ __forceinline__ void multipliers(double res, bool tied, int lane, const Lane& L, double& v_tie, double& beta_bnd, double& run_sum) {
  const bool live = lane < NU;
  constexpr int NSTEP = NU > 32 ? 5 : 4;
  double val = live ? res : 0.0;
  const bool okn = lane + 2 < 64;
  const int nxt = bperm_i((live && tied) ? 1 : 0, okn ? lane + 2 : lane);
  int link = (live && okn) ? nxt : 0;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int dist = 2 << s, src = lane + dist;
    const bool ok = src < 64;
    const double vn = bperm_f64(val, ok ? src : lane);
    const int ln = bperm_i(link, ok ? src : lane);
    val += (link && ok) ? vn : 0.0;
    link = (link && ok) ? ln : 0;
  }
  const double beta_h = bperm_f64(val, live ? L.head : lane);
  const double beta = L.anchored ? beta_h : 0.0;
  const int mwin = L.mwin;
  v_tie = (live && tied) ? val - ((L.anchored && mwin >= lane) ? beta : 0.0) : 0.0;
  beta_bnd = (L.anchored && mwin == lane) ? beta : 0.0;
  run_sum = live ? val : 0.0;
}

}  // namespace aset
