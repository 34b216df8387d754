// pan_scene.hip -- the whole K-iteration PAN loop of a forward call as ONE launch: a wave owns its scene from the first
// selection to the last stop test.  gfx950 (MI355X) only.  OPT-IN (NPA_SCENE_KERNEL=1), measured, not the default path.
//
// Replaces, for the batches it is used on, the 2 K launches of npa_forward_iter (c_api.hip):
//   PAN.forward's loop                                         neupan/blocks/pan.py:128-145
//     generate_point_flow + DUNE.forward (top-M rows)          pan.py:150-212, dune.py:58-127      -> select_geo_body.inc
//     NRMP.forward + stop_criteria                             nrmp.py:114-150, pan.py:215-243     -> nrmp_qp_body.inc
//
// Why: in the two-launch form every launch is a barrier over the batch -- a chain cannot start its next launch before the
// slowest of its 256 scenes is done, so at 20 x 256 scenes in flight most wave slots idle behind tails, and the chains are
// capped by the hardware-queue budget.  Here nothing waits for another scene: a wave runs
//     for k < K:  for t in slices: select(b, t);   solve(b);   stop?
// on its own scene and retires; the dispatcher refills its slot from the launches queued behind.  The bodies are the very
// statements of select_geo_kernel and nrmp_qp_kernel (textual includes): rows, distances and iteration counts equal those of
// the two-launch path, the controls agree to rounding (another kernel around the same statements is other machine code: a
// solve may end a last bit away) -- tests/test_gpu_parity.py::test_scene_kernel_agrees_with_the_two_launch_path.
//
// What it costs, measured (profiles/r04_scene_kernel.txt): the T slices of an iteration run one after the other on ONE wave
// (the two-launch form spreads them over T waves), and the two bodies together want 310 registers -- one wave per SIMD:
// 387 k plans/s against 763 k.  A stepping stone: its pay-off needs the selection of select_scene.h made fast and a build
// that fits two waves per SIMD.  This file also holds the selection stage with one wave per scene as a
// launch of its own (select_scene_kernel, NPA_SELECT_SCENE=1).
#include "pan_common.h"
#include "aset_reduce.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "dune_device.h"
#include "nrmp_qp_device.h"

// one slice of the selection on the calling wave (the body of select_geo_kernel; `return` leaves the slice).  Every input is
// passed through an opaque asm first: inlined next to the solve inside the PAN loop, the slice's loop-invariant parts (weight
// fragments and band tables per lane, addresses, frame constants) were hoisted out of both loops and stayed live across the
// solve -- 212 registers spilled to scratch.  Laundered, nothing of a slice outlives it.
#define SCENE_UNI(x) asm volatile("" : "+s"(x))
template <int E, bool BF16>
__device__ __forceinline__ void select_geo_scene(
    const DevParams& P, const float* wpack_, int n_stride, const float* cur_s_, const float* points_, const float* vel_,
    const int* n_points_, const int* flags_, float* mu_sorted_, float* lam_sorted_, float* pts_sorted_, float* dist_sorted_,
    int* count_, int debug, unsigned* stats_, const float* trig_, unsigned* audit_, unsigned audit_thresh, unsigned audit_seed,
    float margin_scale, int b_, int t_) {
  SCENE_UNI(wpack_); SCENE_UNI(cur_s_); SCENE_UNI(points_); SCENE_UNI(vel_); SCENE_UNI(n_points_); SCENE_UNI(flags_);
  SCENE_UNI(mu_sorted_); SCENE_UNI(lam_sorted_); SCENE_UNI(pts_sorted_); SCENE_UNI(dist_sorted_); SCENE_UNI(count_);
  SCENE_UNI(stats_); SCENE_UNI(trig_); SCENE_UNI(audit_);
  const float* __restrict__ wpack = wpack_; const float* __restrict__ cur_s = cur_s_; const float* __restrict__ points = points_;
  const float* __restrict__ vel = vel_; const int* __restrict__ n_points = n_points_; const int* __restrict__ flags = flags_;
  float* __restrict__ mu_sorted = mu_sorted_; float* __restrict__ lam_sorted = lam_sorted_; float* __restrict__ pts_sorted = pts_sorted_;
  float* __restrict__ dist_sorted = dist_sorted_; int* __restrict__ count = count_; unsigned* __restrict__ stats = stats_;
  const float* __restrict__ trig = trig_; unsigned* __restrict__ audit = audit_;
  const int b = b_, t = t_;
  constexpr bool KEYS16 = false;
#include "select_geo_carve.inc"
  int lane_ = threadIdx.x;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_, j = lane & 31, hf = lane >> 5;
  SELP_DECL;
#include "select_geo_body.inc"
}

#include "select_scene.h"

// All slices t_first .. TT of scene b on the calling wave: the shared-pass fast path (select_scene.h) for the slices it can
// take, the per-slice body for the rest (more candidates than the ranking holds, fewer points than rows, a distrusted margin,
// the slices that owe an audit tile, debug statistics).
template <int E, int TT>
__device__ __forceinline__ void select_scene(
    const DevParams& P, const float* wpack, int n_stride, const float* cur_s, const float* points, const float* vel,
    const int* n_points, const int* flags, float* mu_sorted, float* lam_sorted, float* pts_sorted, float* dist_sorted, int* count,
    int debug, unsigned* stats, const float* trig, unsigned* audit, unsigned audit_thresh, unsigned audit_seed, float margin_scale,
    const int b, const int t_first) {
  if (flags && __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + b * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
  const int lane = threadIdx.x;
  const unsigned todo = ((1u << (TT + 1)) - 1u) & ~((1u << t_first) - 1u);
  unsigned slow = 0;
  if (debug) slow = todo;
  else {
    if (audit && audit_thresh) {             // the slices whose wave would run an audit tile (select_geo_body.inc's hash): per-slice body
      const unsigned seed = audit_seed + (unsigned)__builtin_amdgcn_readfirstlane(
                                             (int)__hip_atomic_load(audit + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll 1
      for (int t = t_first; t <= TT; ++t) {
        unsigned hsh = (seed * 0x9E3779B1u) ^ ((unsigned)b * 0x85EBCA77u) ^ ((unsigned)t * 0xC2B2AE3Du);
        hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12; hsh *= 0x297A2D39u; hsh ^= hsh >> 15;
        if (hsh < audit_thresh || audit_thresh == 0xFFFFFFFFu) slow |= 1u << t;
      }
    }
    unsigned left;
    if (P.geo_rect && E == 4)
      left = select_scene_fast<E, TT, true>(P, wpack, n_stride, cur_s, points, vel, n_points, mu_sorted, lam_sorted, pts_sorted,
                                            dist_sorted, count, trig, audit, margin_scale, b, t_first, slow, lane);
    else
      left = select_scene_fast<E, TT, false>(P, wpack, n_stride, cur_s, points, vel, n_points, mu_sorted, lam_sorted, pts_sorted,
                                             dist_sorted, count, trig, audit, margin_scale, b, t_first, slow, lane);
    slow |= left;
    if (stats && lane == 0) {                 // (words 1, 2 of the statistics: slices for the per-slice body / finished here)
      if (slow & todo) atomicAdd(stats + 1, (unsigned)__popc(slow & todo));
      atomicAdd(stats + 2, (unsigned)__popc(todo & ~slow));
    }
  }
  slow &= todo;
#pragma unroll 1
  for (int t = t_first; t <= TT; ++t)
    if (slow >> t & 1u)
      select_geo_scene<E, false>(P, wpack, n_stride, cur_s, points, vel, n_points, flags, mu_sorted, lam_sorted, pts_sorted,
                                 dist_sorted, count, debug, stats, trig, audit, audit_thresh, audit_seed, margin_scale, b, t);
}

// the selection stage with one wave per SCENE (npa_launch_select_scene: NPA_SELECT_SCENE=1 instead of select_geo_kernel)
template <int E, int TT>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(3, 4)))
void select_scene_kernel(
    DevParams P, const float* wpack, int n_stride, const float* cur_s, const float* points, const float* vel, const int* n_points,
    const int* flags, float* mu_sorted, float* lam_sorted, float* pts_sorted, float* dist_sorted, int* count, int scene0, int t0,
    int nscene, int debug, unsigned* stats, const float* trig, unsigned* audit, unsigned audit_thresh, unsigned audit_seed,
    float margin_scale) {
  if ((int)blockIdx.x >= nscene) return;
  select_scene<E, TT>(P, wpack, n_stride, cur_s, points, vel, n_points, flags, mu_sorted, lam_sorted, pts_sorted, dist_sorted, count,
                      debug, stats, trig, audit, audit_thresh, audit_seed, margin_scale, (int)blockIdx.x + scene0, t0);
}

// the NRMP step of one scene on the calling wave (the body of nrmp_qp_kernel; `return` leaves the step)
template <int TT, int MM, bool BWD, bool SCANW, int WV, bool ASET_T>
__device__ __forceinline__ void nrmp_qp_scene(
    const DevParams& P, const float* cur_s_in, const float* cur_u_in, const float* __restrict__ ref_s,
    const float* __restrict__ ref_us, const float* __restrict__ mu_sorted, const float* __restrict__ lam_sorted,
    const float* __restrict__ pts_sorted, const float* __restrict__ dist_sorted, const int* __restrict__ count,
    float* cur_s_out, float* cur_u_out, float* __restrict__ cur_d_out, float* __restrict__ out_s,
    float* __restrict__ out_u, float* __restrict__ out_d, float* __restrict__ out_min_distance,
    int* __restrict__ out_iters, float* __restrict__ out_nrmp_points, int* __restrict__ flags,
    float* __restrict__ state, double* __restrict__ qp_info, double* __restrict__ warm, const int b,
    const QpBackward& bw, float* __restrict__ trig_out) {
  extern __shared__ __attribute__((aligned(16))) double sm_all[];
  // (laundered like the selection's inputs: per-lane constants of the set-up hoisted out of the PAN loop would be live through
  // every interior-point iteration of every solve)
  int lane_ = threadIdx.x;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
#include "nrmp_qp_body.inc"
}

// What one phase wrote to global memory (rows / counts, or the working nominal, the cos / sin table, the flags) is read by
// the next phase of the SAME wave: the vector stores must have landed, and neither the scalar cache (wave-uniform loads of
// flags / counts / the frame may be scalar loads) nor the vector L1 may answer with an older line.
__device__ __forceinline__ void scene_phase_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_s_dcache_inv();
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // (s_dcache_inv is not a memory operation to the scheduler: nothing may move above it)
}

template <int E, int TT, int MM, bool SCANW>
// ONE wave per SIMD: the two bodies in one kernel need 310 registers.  Built for two waves per SIMD (256 registers, 54 spilled
// around the phases, 220 B of scratch per lane) the kernel FAULTS on the GPU from the second solve of a scene on (round 4,
// gpurun_out/r04_scene2: memory access fault with K >= 2, clean with K = 1 and in this build) -- the same family of symptoms
// as the register-starved QP builds of DESIGN.md 3.3; not root-caused.  -DNPA_SCENE_WAVES=2 rebuilds that variant.
#ifndef NPA_SCENE_WAVES
#define NPA_SCENE_WAVES 1
#endif
__global__ __attribute__((amdgpu_flat_work_group_size(QP_THREADS, QP_THREADS), amdgpu_waves_per_eu(NPA_SCENE_WAVES, 3)))
void pan_scene_kernel(
    DevParams P, const float* wpack, int n_stride, const float* points, const float* vel, const int* n_points,
    float* cur_s, float* cur_u, float* cur_d, const float* ref_s, const float* ref_us, float* mu_sorted, float* lam_sorted,
    float* pts_sorted, float* dist_sorted, int* count, float* out_s, float* out_u, float* out_d, float* out_min_distance,
    int* out_iters, float* out_nrmp_points, int* flags, float* state, double* qp_info, double* warm, float* trig,
    int scene0, int nscene, int iters, int debug, unsigned* stats, unsigned* audit, unsigned audit_thresh, unsigned audit_seed,
    float margin_scale) {
  if ((int)blockIdx.x >= nscene) return;
  const int b = blockIdx.x + scene0;
  const int K = iters;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    // a scene whose stop test fired does nothing more (the two-launch form: both kernels return at their top)
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + b * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
    // slice 0 does not depend on the iterate (s(0) is pinned, robot.py:234): after the first iteration only slices 1..T
#pragma unroll 1
    for (int t = (k == 0 ? 0 : 1); t <= TT; ++t)
      select_geo_scene<E, false>(P, wpack, n_stride, cur_s, points, vel, n_points, flags, mu_sorted, lam_sorted, pts_sorted,
                                 dist_sorted, count, debug, stats, trig, audit, audit_thresh, audit_seed + (unsigned)k,
                                 margin_scale, b, t);
    scene_phase_fence();
    nrmp_qp_scene<TT, MM, false, SCANW, 2, false>(P, cur_s, cur_u, ref_s, ref_us, mu_sorted, lam_sorted, pts_sorted, dist_sorted,
                                                  count, cur_s, cur_u, cur_d, out_s, out_u, out_d, out_min_distance, out_iters,
                                                  out_nrmp_points, flags, state, qp_info, warm, b,
                                                  QpBackward{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
                                                  trig);
    scene_phase_fence();
  }
}

extern "C" size_t npa_qp_shmem_bytes_path(int T, int M, int fast);

// which (E, T, M) the scene kernel is instantiated for: the benchmark configurations with a register-resident QP
extern "C" int npa_pan_scene_supported(int E, int T, int M) {
  return ((E == 4 || E == 8) && T == 10 && M == 10) || (E == 4 && T == 20 && M == 10) ? 1 : 0;
}

extern "C" hipError_t npa_launch_pan_scene(const DevParams& P, const float* wpack, int batch, int n_stride, const float* points,
                                           const float* vel, const int* n_points, float* cur_s, float* cur_u, float* cur_d,
                                           const float* ref_s, const float* ref_us, float* mu, float* lam, float* pts,
                                           float* dist, int* count, float* out_s, float* out_u, float* out_d, float* out_md,
                                           int* out_iters, float* out_np, int* flags, float* state, double* qp_info,
                                           double* warm, float* trig, int iters, int debug, unsigned* stats, unsigned* audit,
                                           unsigned audit_thresh, unsigned audit_seed, float margin_scale, hipStream_t stream,
                                           hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!npa_pan_scene_supported(P.E, P.T, P.M) || iters < 1 || iters > P.K) return hipErrorInvalidValue;
  // dynamic LDS: the two bodies use the same block one after the other
  int n_use_max = n_stride < P.dune_max_num ? n_stride : P.dune_max_num;
  if (n_use_max < 1) n_use_max = 1;
  const size_t n_pad = ((size_t)n_use_max + SEL2_TRIP - 1) / SEL2_TRIP * SEL2_TRIP;
  const size_t key_area = std::max<size_t>(n_pad * sizeof(unsigned), SEL_CAP * (NPA_MAX_E + 5 + 2) * sizeof(float));
  const size_t sel_bytes = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + (2 * SEL_CAP + NPA_MAX_M) * sizeof(int) +
                           (key_area + 15) / 16 * 16;
  const size_t shmem = std::max(sel_bytes, npa_qp_shmem_bytes_path(P.T, P.M, 1));
#define SCENE_LAUNCH(EE, TT_, MM_, SW_)                                                                               \
  do {                                                                                                                 \
    static NpaDeviceOnce big_lds;                                                                                      \
    int dev_ = 0;                                                                                                      \
    if (shmem > 60 * 1024 && big_lds.need(&dev_)) {                                                                    \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(pan_scene_kernel<EE, TT_, MM_, SW_>),          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
      if (e_ != hipSuccess) return e_;                                                                                 \
      big_lds.done(dev_);                                                                                              \
    }                                                                                                                  \
    hipExtLaunchKernelGGL((pan_scene_kernel<EE, TT_, MM_, SW_>), dim3(batch), dim3(QP_THREADS), shmem, stream, ev_start, \
                          ev_stop, 0, P, wpack, n_stride, points, vel, n_points, cur_s, cur_u, cur_d, ref_s, ref_us, mu, lam,  \
                          pts, dist, count, out_s, out_u, out_d, out_md, out_iters, out_np, flags, state, qp_info, warm, trig, \
                          0, batch, iters, debug, stats, audit, audit_thresh, audit_seed, margin_scale);              \
  } while (0)
  if (P.E == 4 && P.T == 10) SCENE_LAUNCH(4, 10, 10, false);
  else if (P.E == 8 && P.T == 10) SCENE_LAUNCH(8, 10, 10, false);
  else SCENE_LAUNCH(4, 20, 10, true);
#undef SCENE_LAUNCH
  return hipGetLastError();
}

// ---- the selection stage alone, one wave per scene (same contract and arguments as npa_launch_select_geo) ----------------
static size_t scene_select_lds(const DevParams& P, int n_stride) {
  int n_use_max = n_stride < P.dune_max_num ? n_stride : P.dune_max_num;
  if (n_use_max < 1) n_use_max = 1;
  const size_t n_pad = ((size_t)n_use_max + SEL2_TRIP - 1) / SEL2_TRIP * SEL2_TRIP;
  const size_t key_area = std::max<size_t>(n_pad * sizeof(unsigned), SEL_CAP * (NPA_MAX_E + 5 + 2) * sizeof(float));
  const size_t slice_body = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + (2 * SEL_CAP + NPA_MAX_M) * sizeof(int) +
                            (key_area + 15) / 16 * 16;
  const size_t ns = (size_t)P.T + 1;
  const size_t fast = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + ns * 16 + ns * SCN_CAP * 2 +
                      SCN_CHUNK * (NPA_MAX_E + 5) * sizeof(float) + SCN_CHUNK * 2 * sizeof(unsigned) + 2 * ns * sizeof(int);
  return (std::max(slice_body, fast) + 15) / 16 * 16;
}
extern "C" int npa_select_scene_supported(int E, int T) { return ((E == 4 || E == 8) && T == 10) || (E == 4 && T == 20) ? 1 : 0; }
extern "C" hipError_t npa_launch_select_scene(const DevParams& P, const float* wpack, int batch, int scene0, int t0, int n_stride,
                                              const float* cur_s, const float* points, const float* vel, const int* n_points,
                                              const int* flags, const float* trig, float* mu_sorted, float* lam_sorted,
                                              float* pts_sorted, float* dist_sorted, int* count, unsigned* stats, int debug,
                                              unsigned* audit, unsigned audit_thresh, unsigned audit_seed, float margin_scale,
                                              hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!npa_select_scene_supported(P.E, P.T)) return hipErrorInvalidValue;
  const size_t shmem = scene_select_lds(P, n_stride);
#define SSEL_LAUNCH(EE, TT_)                                                                                           \
  do {                                                                                                                 \
    static NpaDeviceOnce big_lds;                                                                                      \
    int dev_ = 0;                                                                                                      \
    if (shmem > 60 * 1024 && big_lds.need(&dev_)) {                                                                    \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(select_scene_kernel<EE, TT_>),                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
      if (e_ != hipSuccess) return e_;                                                                                 \
      big_lds.done(dev_);                                                                                              \
    }                                                                                                                  \
    hipExtLaunchKernelGGL((select_scene_kernel<EE, TT_>), dim3(batch), dim3(64), shmem, stream, ev_start, ev_stop, 0, P, wpack, \
                          n_stride, cur_s, points, vel, n_points, flags, mu_sorted, lam_sorted, pts_sorted, dist_sorted, count, \
                          scene0, t0, batch, debug, stats, trig, audit, audit_thresh, audit_seed, margin_scale);       \
  } while (0)
  if (P.E == 4 && P.T == 10) SSEL_LAUNCH(4, 10);
  else if (P.E == 8 && P.T == 10) SSEL_LAUNCH(8, 10);
  else SSEL_LAUNCH(4, 20);
#undef SSEL_LAUNCH
  return hipGetLastError();
}
