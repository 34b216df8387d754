/* neupan_amd.h -- C ABI of the MI355X-native PAN inner solver (libneupan_amd.so).
 *
 * The reference (hanruihua/NeuPAN, pure Python) has NO FFI/plugin interface; its seam for
 * this path is the torch.nn.Module contract of `PAN` (neupan/blocks/pan.py:28-147), built
 * at neupan/neupan.py:84 and called at neupan/neupan.py:129-131.  The entry points below
 * are what a ctypes binding behind that class needs; each cites the reference code it
 * replaces.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = NPA_E_* (never throws);
 *   - all array arguments of npa_forward_batch are DEVICE pointers owned by the caller
 *     (torch-ROCm tensors' data_ptr()); the library allocates nothing per call;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises;
 *   - layouts are the reference's tensors with a leading scene (batch) axis, fp32,
 *     C-contiguous: coordinates on the slow axis, time/points on the fast axis.
 */
#ifndef NEUPAN_AMD_H
#define NEUPAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPA_OK 0
#define NPA_E_ARG (-1)      /* bad argument (null pointer, size out of range)            */
#define NPA_E_HIP (-2)      /* a HIP runtime call failed; see npa_last_error()           */
#define NPA_E_UNSUPPORTED (-3) /* configuration outside the compiled limits (NPA_MAX_*)  */

#define NPA_MAX_T 21        /* receding horizon (3T <= 64: one KKT row per lane)          */
#define NPA_MAX_M 32        /* nrmp_max_num (points kept per horizon step)                */
#define NPA_MAX_E 8         /* polytope edges of the robot (rows of G)                    */

#define NPA_KIN_DIFF 0
#define NPA_KIN_ACKER 1
#define NPA_KIN_OMNI 2

/* Constructor arguments of PAN (pan.py:43-56), of its adjust_kwargs (pan.py:75-82) and the
 * robot attributes the path reads (robot.py:53-71: G, h, kinematics, L, speed_bound,
 * acce_bound = max_acce*dt).  Unbounded speed/acceleration = INFINITY (robot.py:37-38). */
typedef struct npa_config {
  int32_t receding;        /* T   */
  int32_t iter_num;        /* K: PAN iterations per forward (pan.py:128)                  */
  int32_t dune_max_num;    /* points fed to DUNE per scene after decimation (pan.py:171)  */
  int32_t nrmp_max_num;    /* M: nearest points kept per step (nrmp.py:254)               */
  int32_t edge_num;        /* E = G.shape[0]                                              */
  int32_t kinematics;      /* NPA_KIN_*                                                   */
  float iter_threshold;    /* pan.py:243; <= 0 disables the early exit                    */
  /* python floats in the reference (fp64): */
  double step_time;        /* dt  */
  double wheelbase;        /* L (acker only)                                              */
  double speed_bound[2];
  double acce_bound[2];
  double ro_obs, bk;
  /* fp32 tensors in the reference (nrmp.py:79-95, configuration/__init__.py:27): */
  float q_s[3];            /* scalar q_s is passed replicated                             */
  float p_u, eta, d_max, d_min;
  float G[NPA_MAX_E][2];
  float h[NPA_MAX_E];
} npa_config;

/* The 18 tensors of a DUNE checkpoint in state_dict order (obs_point_net.py:31-46;
 * written by dune_train.py:266-274, loaded by dune.py:131-144), HOST pointers, fp32,
 * torch layout: Linear weight is (out,in) row-major. */
typedef struct npa_dune_weights {
  const float *lin_w[6];   /* MLP.{0,3,5,8,10,13}.weight : (32,2) (32,32)x4 (E,32)        */
  const float *lin_b[6];   /* MLP.{0,3,5,8,10,13}.bias                                    */
  const float *ln_w[3];    /* MLP.{1,6,11}.weight (32,)                                   */
  const float *ln_b[3];    /* MLP.{1,6,11}.bias   (32,)                                   */
} npa_dune_weights;

typedef struct npa_handle npa_handle;

/* Replaces PAN.__init__ + DUNE.load_model (pan.py:43-107, dune.py:131-144).  Uploads the
 * repacked weights (a few KiB) to the current device.  One handle per (device, config). */
int npa_create(const npa_config *cfg, const npa_dune_weights *w, npa_handle **out);
int npa_destroy(npa_handle *h);

/* How this handle computes the distance KEYS that nominate the nrmp_max_num nearest points of a slice (the rows it
 * emits are always re-encoded with the exact fp32 encoder and ranked on the exact result; dune.py:100 argsort is
 * reproduced on those):
 *   key_terms 4 = geometric: closed-form distance to the robot polygon, computed inside the selection kernel (no
 *               encoder pass over all points); measured_error = largest |network distance - geometric distance| [m]
 *               over the distance bands 0.25 .. 8 m, margin_e0 = the largest candidate margin over those bands
 *               (1.5 x (error + grid slack), NPA_KEY_SAFETY); both measured by npa_create for THIS checkpoint on nested
 *               4096 x 4096 grids out to +-128 m; points further out are always candidates;
 *   key_terms 1 = network keys with single fp16 products, 3 = fp16x2 split products, 0 = exact fp32 encoder:
 *               measured_error = max |key - exact| / (1 + |exact|) over a 1024 x 1024 grid of the training square
 *               (|x|, |y| <= 25 m), margin_e0 = 5 x that (NPA_KEY_SAFETY). */
int npa_key_mode(const npa_handle *h, int *key_terms, float *measured_error, float *margin_e0);

/* What npa_create measured about the geometric keys of THIS checkpoint (key_terms 4 or not), out[0..n), n <= 6:
 *   [0] 1 when the polygon's rows are consecutive counter-clockwise edges (else geometric keys are never used),
 *   [1] largest |network distance - geometric distance| [m] and [2] largest candidate margin over the bands 0.25 .. 8 m,
 *   [3] refinement ratio: |f| measured at the CELL CENTRES of the calibration grids over what the grid nodes predicted for
 *       the space between them (node maximum + neighbour difference), largest over the distance bands; <= 1 when the
 *       grids resolve f, and a checkpoint above 1.25 keeps network keys (a ridge narrower than a grid cell),
 *   [4] steepest neighbour difference per metre on the finest grid (an estimate of the Lipschitz constant of f next to
 *       the robot), [5] g_far: points at or beyond this geometric distance are always candidates,
 *   [6], [7] (handles created with NPA_KEYS_PRECISION=bf16, else 0): largest measured |bf16-encoder distance - exact distance|
 *       and the largest margin built from it over the bands below 8 m (the bf16 KEY tier: a slice whose candidate list
 *       overflows is filtered with the bf16-MFMA encoder, the survivors re-encoded exactly -- rows bitwise the default path's).
 *   [8], [9] (0 when the handle has no key table: NPA_GEO_TABLE=0, network keys): largest measured |g + f_table - exact
 *       distance| and the largest margin built from it over the bands below 8 m -- the TABLE-corrected geometric key, the
 *       second-stage filter of candidate lists longer than one encoder tile (f = network - geometric distance tabulated at
 *       creation on four nested 512 x 512-cell squares (half extents 2 .. 128 m), bilinear; survivors re-encoded exactly and audited: rows bitwise).
 * The margin is MEASURED, not proven (LayerNorm leaves no usable analytic Lipschitz bound); npa_audit_read is the
 * run-time check. */
int npa_geo_report(const npa_handle *h, float *out, int n);

/* Run-time audit of the geometric-key margin (key_terms 4).  Every launch of the selection kernel checks the bound
 * |exact network distance - geometric distance| <= margin on every point it encodes exactly (the candidates), and a
 * fraction of its slice waves (default 1 in 64; NPA_AUDIT_RATE) on 32 further points spread over the slice.  Counters
 * since creation (or the last reset): audit tiles run, points they checked, violations seen (candidates included),
 * largest excess over the margin [m].  While violations != 0 the kernel treats EVERY point as a candidate (exact keys for
 * the whole slice: slow and right), so a wrong margin cannot keep producing wrong plans; the owner should then rebuild
 * the handle with network keys (NPA_KEY_TERMS=1).  Synchronises the device.  reset != 0 zeroes the counters (and with them
 * the distrust). */
int npa_audit_read(npa_handle *h, uint64_t *tiles, uint64_t *points, uint64_t *violations, float *worst_excess, int reset);

/* The violation count of npa_audit_read WITHOUT synchronising the device: the selection kernel mirrors every violation
 * it counts into pinned host memory, and this reads that word.  Cheap enough to poll after every step of a serving loop
 * (neupan_amd.PAN does, and warns).  A non-zero value means: at least one launch may have emitted rows from a selection
 * that missed a true member (the plans of THAT launch are suspect), every later launch ran on exact keys (slow, right).
 * What to do then: npa_use_network_keys (or rebuild the handle with NPA_KEY_TERMS=1), re-plan the affected step. */
int npa_audit_peek(const npa_handle *h, uint64_t *violations);

/* Switch a handle from geometric to network keys for good (calibrates them, resets the audit).  The workspace grows by the
 * key buffer: call npa_workspace_bytes again and re-allocate (a forward call with the old size fails with NPA_E_ARG).  No
 * forward call may be in progress; synchronises the device.  A no-op on a handle that already uses network keys. */
int npa_use_network_keys(npa_handle *h);

/* What the create-time self-test (npa_create runs every kernel of the handle once on a fixed synthetic problem sized to
 * the robot) changed about this handle.  Hard failures of npa_create are only: two runs that differ bitwise, a control
 * that is not finite or outside its speed bound.  Soft outcomes are reported here:
 *   NPA_SELFTEST_WARM_OFF      warm- and cold-started solves of the test problem differed by more than 1e-4 (a QP that is
 *                              flat along steering directions can do that legitimately): the interior-point warm start
 *                              across PAN iterations is switched off for this handle;
 *   NPA_SELFTEST_GEO_REJECTED  the geometric-key selection missed a member of the exact selection on the test cloud: the
 *                              handle uses network keys (npa_key_mode tells which). */
#define NPA_SELFTEST_WARM_OFF 1
#define NPA_SELFTEST_GEO_REJECTED 2
int npa_selftest_flags(const npa_handle *h, int *flags);

/* Handles of the same checkpoint, polygon, calibration knobs and device SHARE what npa_create derives from the checkpoint: the
 * repacked weights, the margins of the geometric key, the 8.4 MB key table and its margins (the reference loads one model per
 * planner, dune.py:131-144; a serving process makes one handle per batch in flight).  The first npa_create of a key measures
 * (ten calibration launches, ~35 ms), the later ones take the device buffer and the figures and run only their self-test.  The
 * buffer lives as long as a handle uses it.  Everything a handle can change stays its own: the key mode in force
 * (npa_use_network_keys), audit counters, statistics, self-test outcomes.  This reports, for the process: packs calibrated,
 * creations that shared one, packs alive now.  NPA_PACK_CACHE=0 in the environment gives every handle a private pack. */
int npa_pack_cache_stats(int64_t *calibrations, int64_t *shared_creates, int64_t *alive);

/* Replaces NRMP.update_adjust_parameters_value (nrmp.py:170-217). */
int npa_set_adjust(npa_handle *h, const float q_s[3], float p_u, float eta, float d_max, float d_min);

/* Bytes of caller-owned device memory npa_forward_batch needs for `batch` scenes:
 * scratch (no meaning between calls) and state (the stop criterion's memory of the
 * previous iterate, pan.py:100-105 / 215-243; zero it to reset a scene). */
size_t npa_workspace_bytes(const npa_handle *h, int batch);
size_t npa_state_bytes(const npa_handle *h, int batch);
/* Byte offsets inside the workspace of the arrays a caller may look at BETWEEN npa_forward_iter calls (stream-ordered),
 * out[0..n), n <= 8: the working nominal cur_s [B][3][T+1], cur_u [B][2][T], cur_d [B][T], and the sorted rows the last
 * selection launch emitted -- mu [B][T+1][M][E], lam [B][T+1][M][2], pts [B][T+1][M][2], dist [B][T+1][M], count [B][T+1]
 * (int32): the layout of npa_dune_stage's outputs, so a gradient pass can reuse them instead of re-running the stage. */
int npa_workspace_layout(const npa_handle *h, int batch, size_t *out, int n);
/* Byte offset inside the workspace of the per-scene QP diagnostics written by the last NRMP launch
 * of npa_forward_batch: [B][16] doubles per scene:
 *   [0] iteration of the iterate that was kept   [1] its merit (max of the scaled KKT residuals and the gap)   [2] last mu
 *   [3] solver status: 0 converged, 2 non-finite data, 3 factorisation lost positive definiteness above 1e-11,
 *       4 ended above 1e-9 after both cold attempts
 *   [4] interior-point iterations of the last attempt   [14] iterations over all attempts of the solve
 *   [15] how the solve started: 0 cold, 1 from the previous PAN iteration's solution (warm), 2 / 3 a warm attempt refused at
 *       iteration 0 / dropped at iteration 6 and restarted cold, 4 a warm attempt that ended above 1e-10 and was repeated
 *       cold, 5 a cold attempt that jammed and was repeated from unit multipliers
 *   [5..13] per-phase cycle counters of the -DNPA_QP_PROF builds (0 otherwise). */
size_t npa_workspace_qp_info_offset(const npa_handle *h, int batch);

/* Replaces PAN.forward (pan.py:109-147) for `batch` independent scenes:
 *   K x { generate_point_flow (pan.py:150-212) -> DUNE.forward (dune.py:58-127) ->
 *         NRMP.forward (nrmp.py:114-150, robot.py:239-316, the QP of nrmp.py:263-383) ->
 *         stop_criteria (pan.py:215-243) }.
 * Inputs  nom_s [B][3][T+1], nom_u [B][2][T], ref_s [B][3][T+1], ref_us [B][T],
 *         points [B][2][n_stride] (global frame), velocities same shape or NULL,
 *         n_points [B] int32 (0 = no obstacle points for that scene; values outside [0, n_stride] are clamped
 *         into it by the kernels) or NULL meaning every scene has n_stride points.  Point sets larger than dune_max_num
 *         are decimated in-kernel exactly like util.downsample_decimation (util:285-305).
 * Outputs out_s [B][3][T+1], out_u [B][2][T], out_d [B][T] (undefined when nrmp_max_num=0),
 *         out_min_distance [B] (DUNE.min_distance, dune.py:97-98; +inf without points),
 *         out_iters [B] int32 iterations executed, out_nrmp_points [B][2][M] or NULL
 *         (NRMP.obstacle_points, nrmp.py:135-138).
 * Inputs are not modified.  Everything is enqueued on `stream`. */
int npa_forward_batch(npa_handle *h, int batch, int n_stride,
                      const float *nom_s, const float *nom_u, const float *ref_s, const float *ref_us,
                      const float *points, const float *velocities, const int32_t *n_points,
                      float *out_s, float *out_u, float *out_d, float *out_min_distance,
                      int32_t *out_iters, float *out_nrmp_points,
                      void *workspace, size_t workspace_bytes, void *state, size_t state_bytes,
                      void *stream);

/* npa_forward_batch with the flags of npa_forward_begin (NPA_FWD_RESET_STATE): one call per step of a serving loop. */
int npa_forward_batch_flags(npa_handle *h, int batch, int n_stride,
                            const float *nom_s, const float *nom_u, const float *ref_s, const float *ref_us,
                            const float *points, const float *velocities, const int32_t *n_points,
                            float *out_s, float *out_u, float *out_d, float *out_min_distance,
                            int32_t *out_iters, float *out_nrmp_points,
                            void *workspace, size_t workspace_bytes, void *state, size_t state_bytes,
                            void *stream, int flags);

/* npa_forward_batch == npa_forward_begin + iter_num x npa_forward_iter(k) + npa_forward_end, all enqueued on `stream`.
 * The split lets a caller look at the working nominal between PAN iterations (it sits at the head of the workspace:
 * cur_s [B][3][T+1], then cur_u [B][2][T] at the next 16-byte boundary).  Same arguments as npa_forward_batch; buffers
 * must stay valid until the enqueued work has completed.  Independent batches overlap by running on different
 * streams, one handle each.
 * flags: NPA_FWD_RESET_STATE zeroes the stop criterion's state buffer first (a fresh planner), inside the staging launch. */
#define NPA_FWD_RESET_STATE 2
int npa_forward_begin(npa_handle *h, int batch, int n_stride,
                      const float *nom_s, const float *nom_u, const float *ref_s, const float *ref_us,
                      const float *points, const float *velocities, const int32_t *n_points,
                      float *out_s, float *out_u, float *out_d, float *out_min_distance,
                      int32_t *out_iters, float *out_nrmp_points,
                      void *workspace, size_t workspace_bytes, void *state, size_t state_bytes,
                      void *stream, int flags);
int npa_forward_iter(npa_handle *h, int k);
int npa_forward_end(npa_handle *h);
/* A burst of n INDEPENDENT forward calls -- n planners of the reference (one PAN.forward each, pan.py:109-147), one handle, one
 * stream and one argument set per call -- enqueued BREADTH-FIRST: the staging launch of every call, then PAN iteration 0 of
 * every call, then iteration 1, ...  The launches and the results are those of n npa_forward_batch_flags calls in a row;
 * what changes is the order in which the host enqueues them: issued call by call, the last of 20 chains starts ~1.8 ms
 * after the first (420 launches later) and a short burst is mostly that ramp; issued breadth-first every chain is running
 * after the first 2 n launches.  The handles must be distinct (a handle plans one batch at a time).  iter_num = PAN iterations
 * of that call, 1 .. the handle's iter_num (the reference's PAN.iter_num is an attribute its callers may lower between
 * calls).  On an error the calls already begun are ended and the error is returned; work enqueued before it stays enqueued. */
typedef struct npa_forward_call {
  npa_handle *h;
  int32_t batch, n_stride, iter_num;
  const float *nom_s, *nom_u, *ref_s, *ref_us, *points, *velocities;
  const int32_t *n_points;
  float *out_s, *out_u, *out_d, *out_min_distance;
  int32_t *out_iters;
  float *out_nrmp_points;
  void *workspace;
  size_t workspace_bytes;
  void *state;
  size_t state_bytes;
  void *stream;
} npa_forward_call;
int npa_forward_batch_group(int n, const npa_forward_call *calls, int flags);
/* MERGED LAUNCHES.  Calls of a group that share ONE stream, a batch size and a configuration (same npa_config, geometric
 * keys, T = 10 or 20 with M = 10, E = 4 or 8) run every stage of theirs as ONE launch over all their scenes, in runs of
 * <= 8 calls: blockIdx.y = the call, the kernels' statements and every result bitwise those of the call-by-call form.  A
 * launch is a barrier over its scenes (the chain goes on when its slowest scene is done): merged, the wave slots a
 * straggler leaves idle are refilled from the same launch and the run occupies one hardware queue instead of one per
 * call.  Same ownership rules: every call keeps its own handle, tensors, workspace and planner state.
 * npa_forward_group_merged: 1 when npa_forward_batch_group(n, calls, .) would run calls[0..n) as ONE merged run, else 0
 * (different streams / batch sizes / configurations, network keys, n < 2 or n > 8, NPA_GROUP_MERGE=0 in the environment). */
int npa_forward_group_merged(int n, const npa_forward_call *calls);

/* Stage entry points (used by the parity tests and for profiling one stage alone).
 * npa_dune_stage  = generate_point_flow + DUNE.forward + the top-M gather:
 *   mu_sorted [B][T+1][M][E], lam_sorted [B][T+1][M][2], pts_sorted [B][T+1][M][2],
 *   dist_sorted [B][T+1][M], count [B][T+1] (= min(N,M); rows >= count replicate row 0).
 * npa_nrmp_stage  = generate_state_parameter_value + generate_coefficient_parameter_value
 *   + the QP solve, from those arrays; writes s,u,d plus qp_info [B][16] doubles
 *   (best iteration, final merit, mu, status, iterations run; rest reserved for profiling builds), and, when x64 is
 *   not NULL, the fp64 solution before the cast to fp32 (nrmp.py:145-148): x64 [B][3T] = u_0x, u_0y, ..., u_(T-1)y,
 *   d_0..d_(T-1) -- what an optimality certificate should be computed on. */
int npa_dune_stage(npa_handle *h, int batch, int n_stride, const float *nom_s,
                   const float *points, const float *velocities, const int32_t *n_points,
                   float *mu_sorted, float *lam_sorted, float *pts_sorted, float *dist_sorted,
                   int32_t *count, void *stream);
int npa_nrmp_stage(npa_handle *h, int batch, const float *nom_s, const float *nom_u,
                   const float *ref_s, const float *ref_us, const float *mu_sorted,
                   const float *lam_sorted, const float *pts_sorted, const int32_t *count,
                   float *out_s, float *out_u, float *out_d, double *qp_info, double *x64, void *stream);

/* npa_nrmp_params = the parameter build of npa_nrmp_stage alone, for parity tests: what generate_state_parameter_value
 * (robot.py:239-316: A_t, B_t, C_t of the linearised model) and generate_coefficient_parameter_value (nrmp.py:220-261:
 * fa, fb of the hinge rows, slice t+1 of the sorted DUNE output, padding rule nrmp.py:258-259) hand to the solver, in
 * fp32 exactly as the kernel built them (they never leave LDS otherwise):
 *   out_abc [B][T][11]: A[0][2] A[1][2] B[0][0] B[0][1] B[1][0] B[1][1] B[2][0] B[2][1] C[0] C[1] C[2]
 *                       (A's other entries are those of the identity);
 *   out_f   [B][T][M][3]: fa[.,0], fa[.,1], fb (NULL allowed when nrmp_max_num == 0). */
int npa_nrmp_params(npa_handle *h, int batch, const float *nom_s, const float *nom_u, const float *mu_sorted,
                    const float *lam_sorted, const float *pts_sorted, const int32_t *count, float *out_abc,
                    float *out_f, void *stream);

/* npa_nrmp_backward = npa_nrmp_stage + the gradient of a scalar loss L(opt_s, opt_u, opt_d) w.r.t. the
 * adjust parameters.  Replaces what cvxpylayers provides in the reference (the adjust parameters are
 * created with requires_grad=True, neupan/blocks/nrmp.py:79-95; the layer is differentiated at
 * nrmp.py:144; example/LON/LON_corridor.py:94-127 trains p_u, eta, d_max through it): one extra
 * solve with the Newton matrix of the converged interior-point iterate (implicit differentiation of
 * the KKT system).  Covers the direct dependence of THIS solve on (q_s[3], p_u, eta, d_max, d_min),
 * including their appearance in gamma_a = q_s*ref_s and gamma_b = p_u*ref_us (nrmp.py:158-160).
 *   grad_s [B][3][T+1], grad_u [B][2][T], grad_d [B][T] (may be NULL): dL/d(opt_s, opt_u, opt_d);
 *   grad_theta [B][8]: dL/d q_s[0..2], p_u, eta, d_max, d_min, and the solver status (0 = converged);
 *   grad_nom_s [B][3][T+1] (may be NULL): dL/d nom_s as the PROXIMAL CENTRE of this solve (robot.py:178), the one
 *   input besides theta that the reference keeps on its autograd graph between PAN iterations (A/B/C are rebuilt
 *   from python floats, robot.py:272-316; mu under no_grad, dune.py:81; R from python floats, pan.py:207).
 *   Feeding it back as grad_s of the previous iteration's solve (grad_u = 0) chains the gradient through the
 *   whole PAN loop, as PAN.forward_batch_grad does. */
int npa_nrmp_backward(npa_handle *h, int batch, const float *nom_s, const float *nom_u,
                      const float *ref_s, const float *ref_us, const float *mu_sorted,
                      const float *lam_sorted, const float *pts_sorted, const int32_t *count,
                      float *out_s, float *out_u, float *out_d, const float *grad_s,
                      const float *grad_u, const float *grad_d, float *grad_theta,
                      float *grad_nom_s, double *qp_info, void *stream);

/* Timing hook for bench.py: HIP events around every kernel launch of subsequent forward calls (on the stream each
 * launch goes to) and the average per-launch durations in ms: dune_kernel (0 when the handle uses geometric keys:
 * there is no such launch), select_kernel, nrmp_qp_kernel; launches = QP launches timed.  enable=0 turns it off. */
int npa_profile_enable(npa_handle *h, int enable);
int npa_profile_read(npa_handle *h, double *dune_ms_avg, double *select_ms_avg, double *nrmp_ms_avg, int64_t *launches);
/* The active-set launches (NPA_QP_ASET=1: an extra launch of the QP kernel's active-set instantiation in front of the
 * interior-point launch of every PAN iteration) seen by the LAST npa_profile_read: their average duration and count. */
int npa_profile_read_aset(npa_handle *h, double *aset_ms_avg, int64_t *launches);

/* ---- the two steps in front of PAN.forward (handle-free, stream-ordered) --------------------------
 *
 * npa_nominal_ref_states replaces InitialPath.generate_nom_ref_state (+ motion_predict_model)
 *   neupan/blocks/initial_path.py:68-126, :388-444, called at neupan/neupan.py:117-119:
 *   for each scene the T-step rollout of the previous control (nom_s, nom_u) and the reference
 *   states / gears sampled along its current path curve (ref_s, ref_us).  float64 arithmetic in
 *   the reference's order, float32 outputs in the layout npa_forward_batch consumes
 *   (the reference casts at neupan.py:121).
 *   state [B][3] f64; cur_vel [B][2][T] f32 (PAN's previous opt_u; NULL = zeros, the reference's
 *   first call, neupan.py:73); ref_speed [B] f64; path [rows][4] f64 rows (x, y, theta, gear) of
 *   every scene's CURRENT curve (initial_path.py:446-448), scene b owns rows
 *   curve_off[b] .. curve_off[b]+curve_len[b]-1; point_index [B] (closest_point's result,
 *   :160-181); interval [B] f64 (:56, :139).  The path is not modified (the reference writes
 *   2*pi-equivalent headings back into it, :111-112, :190-192).
 *
 * npa_scan_to_points replaces neupan.scan_to_point (mode 0, neupan/neupan.py:173-222) and
 *   neupan.scan_to_point_velocity (mode 1, :224-281): range/angle filter, polar -> sensor frame ->
 *   robot frame -> world frame, ordered compaction, down-sampling of the kept list.
 *   ranges [B][beam_stride] f64, beam_vel [B][2][beam_stride] f64 or NULL, n_beams [B] or NULL
 *   (= beam_stride); points / velocities [B][2][out_stride] f32 (velocities may be NULL), count [B]
 *   (0 where the reference returns None).  Beams beyond out_stride kept points are dropped. */
typedef struct npa_scan_params {
  double angle_min, angle_max;   /* scan["angle_min"], scan["angle_max"]                         */
  double range_min, range_max;   /* scan["range_min"], scan["range_max"]                         */
  double state[3];               /* robot pose x, y, theta                                        */
  double offset[3];              /* scan_offset: sensor pose in the robot frame                   */
  double angle_range[2];         /* beams outside (lo, hi) are dropped                            */
  int32_t down_sample;           /* keep every down_sample-th of the surviving points             */
  int32_t reserved;
} npa_scan_params;

int npa_nominal_ref_states(int batch, int receding, int kinematics, double step_time, double wheelbase,
                           const double *state, const float *cur_vel, const double *ref_speed,
                           const double *path, const int32_t *curve_off, const int32_t *curve_len,
                           const int32_t *point_index, const double *interval, float *nom_s,
                           float *nom_u, float *ref_s, float *ref_us, void *stream);
/* npa_path_progress replaces InitialPath.closest_point + check_curve_arrive as check_arrive runs them
 *   (neupan/blocks/initial_path.py:160-181, :279-287, :247-252; called at neupan/neupan.py:113):
 *   point_index [B] is advanced to the closest of the next `ind_range` path points (first one closer than
 *   close_threshold wins), arrived [B] = 1 when the pose is within arrive_threshold of the curve's last
 *   point and point_index >= len - arrive_index_threshold - 2.  min_dis [B] f32 may be NULL.  Switching
 *   to the next curve / gear stays with the host. */
int npa_path_progress(int batch, const double *state, const double *path, const int32_t *curve_off,
                      const int32_t *curve_len, int32_t *point_index, double close_threshold, int ind_range,
                      double arrive_threshold, int arrive_index_threshold, float *min_dis, int32_t *arrived,
                      void *stream);
int npa_scan_to_points(int batch, int beam_stride, const double *ranges, const double *beam_vel,
                       const int32_t *n_beams, const npa_scan_params *params, int mode,
                       int out_stride, float *points, float *velocities, int32_t *count,
                       void *stream);

/* ---- DUNE training labels (offline) ---------------------------------------------------------------
 * npa_dune_labels replaces DUNETrain.prob_solve / generate_data_set
 *   (neupan/blocks/dune_train.py:82-99, :109-140): for every point p the maximiser mu of
 *   mu^T (G p - h) s.t. ||G^T mu|| <= 1, mu >= 0 and the optimal value (the distance of p to the
 *   robot polygon), by closed form instead of one ECOS call per point.
 *   G [E][2], h [E]: HOST arrays (float64), consecutive counter-clockwise edges as
 *   gen_inequal_from_vertex produces them (util/__init__.py:161-206);
 *   points [n][2] f64 (device); mu [n][E] f32, dist [n] f32 (device; the reference stores float32
 *   tensors, dune_train.py:101-107). */
int npa_dune_labels(int edge_num, const double *G, const double *h, int64_t n, const double *points,
                    float *mu, float *dist, void *stream);

const char *npa_last_error(void);
const char *npa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NEUPAN_AMD_H */
