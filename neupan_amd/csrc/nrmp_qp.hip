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
#include <type_traits>

#include "nrmp_qp_device.h"

static bool qp_fast_path(int T, int M) { return (T == 10 || T == 20) && M == 10; }

// LDS bytes of one scene; `fast` = the register-resident instantiation (packed H, L and P staging)
static bool qp_scan_wide() {
#ifdef NPA_EXPERIMENTS       // (NPA_QP_NOSCAN_WIDE=1: the T = 20 instantiation with a stored Phi, for A/B measurements)
  static const bool v = getenv("NPA_QP_NOSCAN_WIDE") == nullptr;
  return v;
#else
  return true;
#endif
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
  // (SLIM, nrmp_qp_body.inc: the wide-scan instantiation folds the packed H and the packed factor L into one block)
  const bool slim = fast && scan && T > 16;
  const size_t mats = slim ? nu * (nu + 1) / 2 : (fast ? nu * (nu + 1) / 2 + nu * ldp : (size_t)T * 2 * ldp + 2 * nu * ldp);
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
#ifdef NPA_EXPERIMENTS
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<10, 10, false, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
#ifdef NPA_EXPERIMENTS
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_kernel<20, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
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
#ifdef NPA_EXPERIMENTS
    if (!(P.T == 10 && P.M == 10 && !force_generic && P.qp_aset && warm && flags)) return hipErrorInvalidValue;
    QP_LAUNCH(10, 10, false, false, 2, true);
#else
    return hipErrorInvalidValue;          // (the active-set instantiation exists in the experiments build only)
#endif
  }
  else if (P.T == 10 && P.M == 10 && !force_generic) QP_LAUNCH(10, 10);
  else if (P.T == 20 && P.M == 10 && !force_generic && scan_wide) QP_LAUNCH(20, 10, false, true);
#ifdef NPA_EXPERIMENTS
  else if (P.T == 20 && P.M == 10 && !force_generic) QP_LAUNCH(20, 10);
#endif
  else QP_LAUNCH(0, 0);
#undef QP_LAUNCH
  return hipGetLastError();
}

// nrmp_qp_kernel for a group of n forward calls that share the configuration, the batch size and the stream: ONE launch,
// grid (batch, n).  The register-resident instantiations only (c_api.hip keeps everything else call by call).
extern "C" int npa_qp_group_supported(int T, int M) {
  static const bool force_generic = getenv("NPA_QP_GENERIC") != nullptr;
  return !force_generic && M == 10 && (T == 10 || (T == 20 && qp_scan_wide())) ? 1 : 0;
}
extern "C" hipError_t npa_launch_qp_group(const DevParams& P, const QpGroup& G, int n, int batch, hipStream_t stream,
                                          hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n < 1 || n > NPA_GROUP_MAX || !npa_qp_group_supported(P.T, P.M)) return hipErrorInvalidValue;
  const size_t shmem = npa_qp_shmem_bytes_path(P.T, P.M, 1);
  static NpaDeviceOnce attr_set;
  int dev_ = 0;
  if (attr_set.need(&dev_)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_group_kernel<10, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nrmp_qp_group_kernel<20, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.done(dev_);
  }
  if (P.T == 10)
    hipExtLaunchKernelGGL((nrmp_qp_group_kernel<10, 10>), dim3(batch, n), dim3(QP_THREADS), shmem, stream, ev_start, ev_stop, 0, P, G, batch);
  else
    hipExtLaunchKernelGGL((nrmp_qp_group_kernel<20, 10, true>), dim3(batch, n), dim3(QP_THREADS), shmem, stream, ev_start, ev_stop, 0, P, G, batch);
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
