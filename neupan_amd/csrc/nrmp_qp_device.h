// nrmp_qp_device.h -- device side of nrmp_qp.hip (constants, helpers, the kernel); the host launchers stay in nrmp_qp.hip.
// Included by nrmp_qp.hip.  (No include guard games: each translation unit includes it once.)
#pragma once
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
#ifndef NPA_QP_WIDE_LDS
#define NPA_QP_WIDE_LDS 1      // T = 20: the factorisation's deferred trailing updates through LDS in chunks (nrmp_qp_body.inc, WLDS); 0: all through v_readlane
#endif
#ifndef QP_WLDS_CH
#define QP_WLDS_CH 4           // ... pairs of columns per chunk
#endif
#ifndef QP_CHOL_LOOK
#define QP_CHOL_LOOK 3         // columns behind the pivot whose trailing update is broadcast with v_readlane (the rest: LDS, one pivot late)
#endif
#define QP_ASET_FIRST_MAX 0.05 // the attempt is not made from a warm point whose seeded merit is above this (most of those cycle: 85 % of the failures)
#define QP_ASET_MAX_GUESS 2     // factorisations the active-set iteration may spend before the interior-point warm start takes over
#define QP_ASET_TOL 1e-13      // what may be left of the scaled dual residual at a guess that repeated
#define QP_RETRY_MERIT 1e-9    // a cold solve that ends above this is repeated once from round 2's start (unit multipliers)
#define QP_SIGMA_MU_MIN 1e-15  // floor of the centring target: a gap driven to 1e-20 leaves the Newton matrix too ill
                               //   conditioned for the residuals to follow (solves that ended at 1e-11: 8 -> 1 of 640)
#define QP_SIGMA_MU_RES 0.01   // ... and never below this x the largest scaled residual of the iterate: complementarity may not run
                               //   more than 100x ahead of feasibility.  The solves that "stalled" (mu at 1e-15, dual residual stuck at
                               //   1e-10 .. 1e-12, best iterate after three non-improving iterations: 13 of 3 360 benchmark QPs, one of
                               //   them 1.2e-4 from the oracle in the controls) end at <= 1e-13 with it (CPU replay of the kernel's
                               //   method, profiles/r05_qp_stall_study.txt: +0.2 .. +1.0 iterations per solve, the maximum unchanged)
#define QP_STALL_NEAR_MERIT 1e-6 // the rule of three non-improving iterations holds below this best merit; above it the patience is
#define QP_STALL_FAR 8           //   this (profiles/r06_qp_stall_patience.txt: no effect on the solves that converged before)
#define QP_CENTRAL_GAMMA 1e-3    // centrality safeguard of a blocked step: every product l w stays above this x their mean,
#define QP_CENTRAL_SHRINK 0.7    //   the step shortened by this factor,
#define QP_CENTRAL_TRIES 6       //   at most this often,
#define QP_CENTRAL_ALPHA 0.9     //   checked for steps shorter than this only (profiles/r06_qp_centrality.txt)
#define QP_CORRECTOR_MIN_AFF 0.1 // an affine direction that can be followed for less than this takes no second-order corrector: dw_aff dl_aff
                                 //   describes a point the iterate never gets near, and correcting for it can send the iterate round a
                                 //   cycle (one solve in 10 240 of the polygon robot's: mu 2e-3 -> 6e-3 -> 4e-3 -> 8e-3 -> 2e-3 ..., every
                                 //   residual at 1e-9, both cold starts).  The other solves are unchanged (profiles/r06_qp_corrector.txt)
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
// ---- fp64 broadcast inside a 16-lane row, FUSED with the multiply-add (round 6) ------------------------------------------------
// gfx90a+ give the double-precision ALU one DPP control, row_newbcast:N -- every lane reads lane N of its own 16-lane row -- and
// v_fmac_f64 takes it:   acc += src[lane N of my row] * (-mul)   is ONE instruction.  The route through two v_readlane_b32 into a
// scalar pair and a v_fma_f64 with a scalar operand is three, with wait states between them (s_nop 0 after the write, s_nop 1 in
// front of the use).  The compiler neither selects the DPP form of v_fmac_f64 (update_dpp on a double becomes v_mov_b64_dpp +
// v_fma_f64) nor sees the hazards of an asm statement, so the wait states are part of the statements:
//   VALU write of a VGPR -> DPP read of it: 2 wait states  (LEAD: s_nop 1 in front; `src` has just been written)
//   VALU write of a VGPR -> v_readlane of it: 1 wait state (TAIL: s_nop 0 behind; the compiler's v_readlane follows)
// LEAD marks `src` read-write so that every later statement that reads it is ordered behind this one.
template <int N, bool LEAD, bool TAIL>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double& src, double mul) {
  static_assert(N >= 0 && N < 16, "row_newbcast lane");
  if constexpr (LEAD && TAIL)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(acc), "+v"(src) : "v"(mul), "n"(N));
  else if constexpr (LEAD)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(src) : "v"(mul), "n"(N));
  else if constexpr (TAIL)
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
  else
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
// one step of a substitution chain:  x -= x[lane N of my row] * mul   (x is written by the step before: always LEAD)
template <int N, bool TAIL>
__device__ __forceinline__ void subst_rowbcast(double& x, double mul) {
  static_assert(N >= 0 && N < 16, "row_newbcast lane");
  if constexpr (TAIL)
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(x) : "v"(mul), "n"(N));
  else
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(mul), "n"(N));
}
// One column of the factor's LDS image, written by the lanes of MASK only: the EXEC mask is set and restored inside the statement,
// both from constants of the instruction stream (the call sits under `if (lane < 2T)` of wave-uniform code in a 64-lane workgroup:
// EXEC there is lanes 0 .. 2T-1 = RESTORE; the compiler never sees another EXEC).  LDS operations of a wave complete in order, so
// the counts the compiler keeps for its own s_waitcnt stay on the safe side of these stores.  LAST: five wait states behind the
// restore, what a DPP instruction wants after a scalar write of EXEC.  (With the statements outside any `if` -- EXEC all ones -- the
// stand-alone kernel came out with a 36-byte private segment that no instruction touches: a register-allocation artefact of hipcc
// 7.2 around 106 scalar registers; tests/test_abi.py asserts there is none.)
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(size_t)((const lds_f64*)p); }
template <unsigned MASK, int OFF, bool LAST, unsigned RESTORE>
__device__ __forceinline__ void park_store(unsigned addr, double v) {
  // (restored from a constant, not from a saved copy: the kernel is at its limit of scalar registers)
  if constexpr (LAST)
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1 offset:%3\n\ts_mov_b64 exec, %4\n\ts_nop 4" : : "v"(addr), "v"(v), "n"(MASK), "n"(OFF), "n"(RESTORE) : "memory");
  else
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1 offset:%3\n\ts_mov_b64 exec, %4" : : "v"(addr), "v"(v), "n"(MASK), "n"(OFF), "n"(RESTORE) : "memory");
}
// columns C .. N-2 of the row a lane holds, scaled by 1/L_ii, four columns per statement
template <int C, int N>
__device__ __forceinline__ void park_rows(unsigned lrow, const double (&arow)[N], double myinv) {
  constexpr unsigned ALL = (1u << N) - 1u;
  if constexpr (C + 4 <= N - 1) {
    const double v0 = arow[C] * myinv, v1 = arow[C + 1] * myinv, v2 = arow[C + 2] * myinv, v3 = arow[C + 3] * myinv;
    asm volatile("s_mov_b64 exec, %5\n\tds_write_b64 %0, %1 offset:%9\n\ts_mov_b64 exec, %6\n\tds_write_b64 %0, %2 offset:%10\n\t"
                 "s_mov_b64 exec, %7\n\tds_write_b64 %0, %3 offset:%11\n\ts_mov_b64 exec, %8\n\tds_write_b64 %0, %4 offset:%12\n\ts_mov_b64 exec, %13"
                 : : "v"(lrow), "v"(v0), "v"(v1), "v"(v2), "v"(v3),
                     "n"(ALL & ~((2u << C) - 1u)), "n"(ALL & ~((2u << (C + 1)) - 1u)), "n"(ALL & ~((2u << (C + 2)) - 1u)), "n"(ALL & ~((2u << (C + 3)) - 1u)),
                     "n"(C * 8), "n"(C * 8 + 8), "n"(C * 8 + 16), "n"(C * 8 + 24), "n"(ALL) : "memory");
    park_rows<C + 4, N>(lrow, arow, myinv);
  } else if constexpr (C < N - 1) {
    park_store<ALL & ~((2u << C) - 1u), C * 8, C == N - 2, ALL>(lrow, arow[C] * myinv);
    park_rows<C + 1, N>(lrow, arow, myinv);
  }
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I = I0 .. N-1 (the row_newbcast lane is an immediate of the instruction)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
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
#include "nrmp_qp_body.inc"
}

// nrmp_qp_kernel over a GROUP of forward calls (pan_common.h: merged launches): blockIdx.y = the call, blockIdx.x = the scene
// of that call; the per-call pointers come out of the kernel arguments, the body is the same statements.  Forward solves only.
template <int TT, int MM, bool SCANW = false>
__global__ __attribute__((amdgpu_flat_work_group_size(QP_THREADS, QP_THREADS), amdgpu_waves_per_eu(NPA_QP_WAVES, 3)))
void nrmp_qp_group_kernel(DevParams P, QpGroup G, int nscene) {
  extern __shared__ __attribute__((aligned(16))) double sm_all[];
  constexpr bool BWD = false, ASET_T = false;
  constexpr int WV = NPA_QP_WAVES;
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= nscene) return;
  const int b = blockIdx.x;
  const QpCall& q = G.c[blockIdx.y];
  const float* cur_s_in = q.cur_s_in; const float* cur_u_in = q.cur_u_in;
  const float* __restrict__ ref_s = q.ref_s; const float* __restrict__ ref_us = q.ref_us;
  const float* __restrict__ mu_sorted = q.mu_sorted; const float* __restrict__ lam_sorted = q.lam_sorted;
  const float* __restrict__ pts_sorted = q.pts_sorted; const float* __restrict__ dist_sorted = q.dist_sorted;
  const int* __restrict__ count = q.count;
  float* cur_s_out = q.cur_s_out; float* cur_u_out = q.cur_u_out; float* __restrict__ cur_d_out = q.cur_d_out;
  float* __restrict__ out_s = q.out_s; float* __restrict__ out_u = q.out_u; float* __restrict__ out_d = q.out_d;
  float* __restrict__ out_min_distance = q.out_min_distance; int* __restrict__ out_iters = q.out_iters;
  float* __restrict__ out_nrmp_points = q.out_nrmp_points; int* __restrict__ flags = q.flags; float* __restrict__ state = q.state;
  double* __restrict__ qp_info = q.qp_info; double* __restrict__ warm = q.warm; float* __restrict__ trig_out = q.trig_out;
  const QpBackward bw{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#include "nrmp_qp_body.inc"
}
