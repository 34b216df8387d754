// Shared device/host definitions for the PAN kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define NPA_MAX_T 21
#define NPA_MAX_M 32
#define NPA_MAX_E 8
#define NPA_MAX_POINTS 32768       // points per scene after decimation (select_kernel keeps a slice's keys in LDS)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute, and launches come from several host threads
// (serve.StepLoop): one bit per device ordinal instead of a process-wide `static bool` -- a second handle on another GPU of
// the same process gets its attribute too, and concurrent first launches at worst set it twice.
struct NpaDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  // true when the attribute still has to be set on the current device (then call done() after setting it)
  bool need(int* dev_out) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    *dev_out = d;
    return !(mask.load(std::memory_order_acquire) & (1ull << (d & 63)));
  }
  void done(int d) { mask.fetch_or(1ull << (d & 63), std::memory_order_release); }
};

// Kernel-argument copy of npa_config (include/neupan_amd.h), passed by value.
struct DevParams {
  int T, M, E, kin, K, dune_max_num;
  int key_stride;             // per-slice stride of the distance-key buffer: min(dune_max_num, NPA_MAX_POINTS) up to 32
  float iter_threshold;
  float dt32;                 // (float)dt, the value fp32 tensor*python-float products see
  double dt, L;
  double speed_bound[2], acce_bound[2];
  double ro_obs, bk;
  float q_s[3], p_u, eta, d_max, d_min;
  float G[NPA_MAX_E][2];
  float h[NPA_MAX_E];
  // robot polygon as vertices (fp64 intersection of consecutive edges, rounded once): edge e runs V[e] -> V[e] + D[e];
  // used by the GEOMETRIC distance keys (select_kernel<E, true>), never by the rows that are emitted
  float pvx[NPA_MAX_E], pvy[NPA_MAX_E], pdx[NPA_MAX_E], pdy[NPA_MAX_E], pil[NPA_MAX_E];   // pil = 1 / |D|^2
  float geo_rcal;             // half extent of the square (robot frame) over which the geometric key's error was measured
  float geo_far;              // geo_rcal - (largest vertex radius): a point with a smaller geometric distance lies inside that square
  // the common case -- an axis-aligned rectangle in the robot frame (robot.py:342-375 builds length x width boxes) --
  // has a cheaper closed form: centre, half extents; geo_rect != 0 selects it.  Any other polygon: rc / rh describe its
  // bounding box (the key pass of select_geo_kernel ranks on the distance to it, pan_common.h WP_TABH[4])
  int geo_rect;
  float rcx, rcy, rhx, rhy;
  int qp_aset;                // (experiments build only: the QP's warm solves try the active-set iteration first, NPA_QP_ASET)
  int geo_tab;                // the weight pack carries the correction table of the geometric key (WP_TAB) and its margins (WP_KTAB)
};

// ---- merged launches of a GROUP of forward calls (npa_forward_batch_group, c_api.hip) --------------------------------
// The chains of a burst that share a stream, a batch size and a configuration run each stage as ONE launch: blockIdx.y
// selects the call, blockIdx.x is what it is in the single-call kernels.  A launch is a barrier over its scenes (the chain
// cannot go on before its slowest scene is done): with G x 256 scenes behind one barrier the wave slots a tail leaves idle
// are refilled from the same launch, and G chains need ONE hardware queue instead of G.  The per-call pointers travel by value
// in the kernel arguments (an array indexed with blockIdx.y: scalar loads from the kernarg segment, nothing to upload).
#define NPA_GROUP_MAX 8
struct SelGeoCall {
  const float* wpack; const float* cur_s; const float* points; const float* vel; const int* n_points; const int* flags;
  float* mu_sorted; float* lam_sorted; float* pts_sorted; float* dist_sorted; int* count; unsigned* stats; const float* trig;
  unsigned* audit; int n_stride; unsigned audit_seed;
};
struct SelGeoGroup { SelGeoCall c[NPA_GROUP_MAX]; };
struct QpCall {
  const float* cur_s_in; const float* cur_u_in; const float* ref_s; const float* ref_us; const float* mu_sorted;
  const float* lam_sorted; const float* pts_sorted; const float* dist_sorted; const int* count; float* cur_s_out;
  float* cur_u_out; float* cur_d_out; float* out_s; float* out_u; float* out_d; float* out_min_distance; int* out_iters;
  float* out_nrmp_points; int* flags; float* state; double* qp_info; double* warm; float* trig_out;
};
struct QpGroup { QpCall c[NPA_GROUP_MAX]; };
struct StageCall {
  float* cur_s; const float* nom_s; float* cur_u; const float* nom_u; int* flags; int* count; int* state; float* trig;
};
struct StageGroup { StageCall c[NPA_GROUP_MAX]; };

// ---- geometric distance keys ------------------------------------------------------------------------
// The DUNE network approximates the distance from a point to the robot polygon; the closed-form distance is ~12
// VALU instructions per edge, so select_kernel<E, true> uses IT to nominate the candidates (the kept rows are still
// re-encoded with the exact network).  What the margin must cover is |network distance - geometric distance|, a
// smooth function of the robot-frame position and a property of the checkpoint: npa_create measures its maximum
// per DISTANCE BAND on nested grids (geo_calib_kernel) and stores margin[band] in the weight pack (WP_GEO).
// band(g): 8 bands per octave of g + 0.25 m (32 mm wide next to the robot, ~12 % of g far away).
#define NPA_GEO_BANDS 88                          // g + 0.25 in [2^-2, 2^9)
#define NPA_GEO_KEY_FAR 0x7F800000u               // key of a point outside the calibrated square: always a candidate
__host__ __device__ inline int npa_geo_band(float g) {
  union { float f; unsigned u; } v;
  v.f = g + 0.25f;
  // (signed keys reach this -- the table-corrected key, bf16 / exact distances: anything at or below -0.25 m has its sign bit
  // set and belongs to band 0 like every other negative key, the interval the threshold folds of select_geo_body.inc give band 0)
  if (v.u & 0x80000000u) return 0;
  const int b = (int)(v.u >> 20) - (int)(0x3E800000u >> 20);
  return b < 0 ? 0 : (b >= NPA_GEO_BANDS ? NPA_GEO_BANDS - 1 : b);
}

// ---- packed DUNE weights (device buffer, floats) -----------------------------------------
// MFMA A-operand fragments of v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5].
// For K-step r of a 32x32 layer the two K entries are the features
//   feat(r, hf) = (r&3) + 8*(r>>2) + 4*hf          (hf = lane>>5)
// i.e. exactly the feature the C/D layout leaves in accumulator register r of that lane, so
// a layer's activated output registers ARE the next layer's B operands with no data movement.
#define WP_W1 0                                   // [64]            Linear(2,32)
#define WP_WL (WP_W1 + 64)                        // [4][16][64]     Linear(32,32) x4
#define WP_VEC (WP_WL + 4 * 16 * 64)              // [11][32]        per-feature vectors
#define WP_W6 (WP_VEC + 11 * 32)                  // [8][32]         Linear(32,E) rows (zero padded)
#define WP_B6 (WP_W6 + 8 * 32)                    // [8]
// ---- key path (dune_kernel's distance keys) ----------------------------------------------------
// The four 32x32 layers as fp16x2 split products on v_mfma_f32_32x32x16_f16: w = w1 + w2, x = x1 + x2
// with fp16 terms (RNE of the running residual: 22-23 significant bits), product terms (2,1) (1,2)
// (1,1) accumulated in fp32.  Weights and activations carry exact power-of-two scales chosen on the
// host (c_api.hip) so that every fp16 term stays in the normal range and cannot overflow; LayerNorm
// absorbs the scale (its eps is scaled alike), ReLU passes it on.
// A-operand layout: lane l holds A[i = l&31][k = 8*(l>>5) + q], q = 0..7; K-step s (0,1) of a layer
// contracts over the features feat(8s+q, hf).   [layer 4][term 2][step 2][lane 64][8 fp16]
// The keys only decide WHICH points are kept (the kept rows are re-encoded exactly), so the key path
// may reassociate: the LayerNorm mean is removed through the weights,
//   W_c = (I - 11'/32) W,  b_c = b - mean(b)     (Linear 1, 3, 5: the layers followed by LayerNorm).
#define WP_BF (WP_B6 + 8)
#define WP_BF_FLOATS (4 * 2 * 2 * 64 * 4)
#define WP_KVEC (WP_BF + WP_BF_FLOATS)            // [5][32]  key-path biases of Linear 1..5 (centred / scaled)
#define WP_KSC (WP_KVEC + 5 * 32)                 // [8]      LayerNorm eps x3 (scaled), tanh output scale x3
#define WP_KW1 (WP_KSC + 8)                       // [64]     centred Linear(2,32) A-fragment (read per lane)
#define WP_KEY_LDS_FLOATS (WP_BF_FLOATS + 5 * 32 + 8)
#define WP_GEO (WP_KW1 + 64)                      // [NPA_GEO_BANDS] margin of the geometric key per distance band (+inf = uncalibrated)
// the four 32x32 layers once more, lane-major: [layer 4][lane 64][K-step 16] -- select_geo_kernel streams a layer's sixteen
// A-fragments with four 16-byte loads per lane right before the layer that uses them (16 + 16 registers for two layers in
// flight instead of 64 for all four: that is what lets the kernel fit 128 registers without spills)
#define WP_WLS (((WP_GEO + NPA_GEO_BANDS) + 3) & ~3)
// the four 32x32 layers as bf16 A-fragments of v_mfma_f32_32x32x16_bf16 (the labelled reduced-precision tier of the ROWS,
// NPA_ROWS_PRECISION=bf16): [layer 4][K-step 2][lane 64][8 bf16], lane l holds W[i = l&31][feat(8 s + q, l >> 5)] -- the
// exact network's weights rounded to bf16 (RNE), no scaling, no centring
#define WP_WB16 (WP_WLS + 4 * 64 * 16)
#define WP_WB16_FLOATS (4 * 2 * 64 * 4)
// the bf16 KEY tier (NPA_KEYS_PRECISION=bf16): margin per band of the EXACT distance for |bf16-encoder distance - exact distance|,
// measured at creation (k16_calib_kernel); +inf = uncalibrated
#define WP_K16 (WP_WB16 + WP_WB16_FLOATS)
// the TABLE-corrected geometric key (select_geo_kernel's second-stage filter of a long candidate list): f(p) = network distance
// - geometric distance is a smooth function of the robot-frame position p, so it is TABULATED at creation (geo_table_kernel:
// the exact encoder at the nodes of four nested 512 x 512-cell squares, half extents 2 / 8 / 32 / 128 m, cells of 8 / 31 / 125 / 500 mm;
// a cell = its four corner values as fp16, ONE 8-byte gather per point) and g(p) + bilinear f(p) is a key whose error is what
// the interpolation leaves (the network has creases -- ReLU, LayerNorm -- and a bilinear cell across one is off by ~ cell x
// slope jump / 4: the finest cells sit where the M nearest points of a dense cloud are) -- millimetres, not the centimetres of g.  WP_KTAB: margin per band of the KEY for
// |corrected key - exact distance| (ktab_calib_kernel on 4096 x 4096 nodes per square, not aligned with the cells; +inf =
// uncalibrated).  The table itself follows the pack in the same device allocation (WP_TAB: not part of the host image).
#define WP_KTAB (WP_K16 + ((NPA_GEO_BANDS + 3) & ~3))
// header of the table: centre (x, y) of its squares = the centre of the polygon's bounding box, half extent h0 of level 0
// (>= NPA_TAB_HALF0, and 1.25 x the box: a 4.6 m car is covered by the finest cells nose to tail), cells per metre of level 0
// [4]: S, the slack of the key pass's BOX key for a polygon that is not an axis-aligned box (0 for one that is): the key pass
// ranks every point on the distance to the polygon's bounding box -- 8 instructions instead of 12 per edge -- which is a lower
// bound of the distance g to the polygon with g <= box distance + S, S = the largest g over the box's corners (g is convex)
#define WP_TABH (WP_KTAB + ((NPA_GEO_BANDS + 3) & ~3))
// ---- the 16-point tile of the exact encoder (v_mfma_f32_16x16x4_f32; dune_device.h: encode_tile16_stream) ---------------------
// A 16x16x4 MFMA holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = 4 (l >> 4) + v][j = l & 15] (v = 0..3).  A 32-wide
// layer is TWO row blocks mb = 0, 1 (two independent accumulators of 4 registers) and EIGHT K-steps s; lane group kq = l >> 4 holds,
// in accumulator register v of block mb, the output feature npa_feat16(4 mb + v, kq) -- chosen so that
//   * register s = 4 mb + v of a lane IS its B operand of K-step s of the next layer (no data movement between the layers), and
//   * the K order of a point's fma chain, (s, kq) lexicographic, is feature 0,4,1,5,2,6,3,7,8,12,... : the order of the 32-point
//     tile's chain (step r: npa_feat(r, 0), npa_feat(r, 1)).  The f32 MFMA is an fmaf chain in K order, so both tile shapes give
//     bitwise the same accumulators; LayerNorm sums and the output layer reduce in one canonical order in both (dune_device.h).
// WP_W116: [mb 2][lane 64] A-fragments of Linear(2,32) (K = 2 of 4 used: lanes kq >= 2 hold 0)
// WP_WL16: [layer 4][lane 64][2 s + mb] A-fragments of the four 32x32 layers, lane-major (streamed with four 16-byte loads per layer)
// WP_VEC16: the per-feature vectors, the rows of Linear(32,E) and its bias as WP_VEC / WP_W6 / WP_B6 hold them, every 32-vector
//           permuted to [kq 4][s 8] (a lane reads its eight entries with two 16-byte LDS loads)
#define WP_W116 (WP_TABH + 8)
#define WP_WL16 (WP_W116 + 2 * 64)
#define WP_VEC16 (WP_WL16 + 4 * 64 * 16)
#define WP_VEC16_FLOATS (11 * 32 + 8 * 32 + 8)
#define WP_TOTAL (WP_VEC16 + WP_VEC16_FLOATS)
#define NPA_TAB_KEY_FAR 0xFFFFFFFDu                // filter key of a point beyond the calibrated square (above every ordered_key of a number)
#define NPA_TAB_N 512                             // cells per side of one level
#define NPA_TAB_LEVELS 4
#define NPA_TAB_HALF0 2.0f                        // smallest half extent of level 0; level l: x 4^l (2 / 8 / 32 / 128 m: cells of 8 / 31 / 125 / 500 mm)
#define WP_TAB ((WP_TOTAL + 3) & ~3)              // [levels][N][N] cells of 4 x fp16 (2 floats each)
#define WP_TAB_FLOATS (NPA_TAB_LEVELS * NPA_TAB_N * NPA_TAB_N * 2)
// order of the per-feature vectors
enum { V_B1 = 0, V_G1, V_BE1, V_B2, V_B3, V_G2, V_BE2, V_B4, V_B5, V_G3, V_BE3 };

__host__ __device__ inline int npa_feat(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
// feature in accumulator register s (= 4 mb + v) of lane group kq (= lane >> 4) of the 16-point tile (see WP_W116 above)
__host__ __device__ inline int npa_feat16(int s, int kq) { return 8 * (s >> 1) + 2 * (s & 1) + (kq >> 1) + 4 * (kq & 1); }

// ---- per-scene persistent state (stop criterion memory, pan.py:100-105) -------------------
// floats: prev_s[3(T+1)] prev_u[2T] prev_mu[(T+1) M E] prev_lam[(T+1) M 2]; ints: valid, prev_n, min-distance valid,
// min-distance bits (DUNE.min_distance of the last forward WITH points, dune.py:97-98)
__host__ __device__ inline size_t npa_state_floats(int T, int M, int E) {
  return (size_t)3 * (T + 1) + 2 * T + (size_t)(T + 1) * M * E + (size_t)(T + 1) * M * 2 + 4;
}

// ---- scratch: struct-of-arrays over the batch (offsets in 4-byte words) ---------------------
// cur_s [B][3][T+1]  cur_u [B][2][T]  cur_d [B][T]  mu [B][T+1][M][E]  lam [B][T+1][M][2]
// pts [B][T+1][M][2]  dist [B][T+1][M]  count [B][T+1] (int)
// flags [B][4] (int: done, iters, warm start usable)   warm [B][nwarm] (double: x and multipliers of the last QP)
// trig [B][T+1][2] (float: cos, sin of the nominal heading of every horizon step, written by whoever writes cur_s)
// keys [B][T+1][key_stride] (uint: order-preserving distance key of every point of every slice)
// pan.py:207 builds R from torch.cos / torch.sin of the fp32 heading (fp32 libm: within 1 ulp of the correctly rounded
// value); here: fp64 libm on the fp32 angle, rounded once to fp32.  ONE definition so that every producer of the table
// (stage_kernel, the QP's write-out, trig_kernel) agrees bitwise
__device__ inline void npa_trig(float th, float& c, float& s) {
  c = (float)cos((double)th);
  s = (float)sin((double)th);
}

struct ScratchLayout {
  size_t cur_s, cur_u, cur_d, mu, lam, pts, dist, count, flags, warm, qp_info, trig, keys, total;
};
__host__ __device__ inline size_t npa_warm_doubles(int T, int M) {   // per scene: x (2T + T), lf (T M), lc (8T - 4), ld (2T)
  return (size_t)2 * T + T + (size_t)T * M + (8 * T - 4) + 2 * T;
}
__host__ __device__ inline ScratchLayout npa_scratch_layout(int B, int T, int M, int E, int key_stride) {
  ScratchLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
  L.cur_s = take((size_t)B * 3 * (T + 1));
  L.cur_u = take((size_t)B * 2 * T);
  L.cur_d = take((size_t)B * T);
  L.mu = take((size_t)B * (T + 1) * M * E);
  L.lam = take((size_t)B * (T + 1) * M * 2);
  L.pts = take((size_t)B * (T + 1) * M * 2);
  L.dist = take((size_t)B * (T + 1) * M);
  L.count = take((size_t)B * (T + 1));
  L.flags = take((size_t)B * 4);
  o = (o + 3) & ~(size_t)3;                       // 16-byte alignment for the doubles
  L.warm = take((size_t)B * npa_warm_doubles(T, M) * 2);
  L.qp_info = take((size_t)B * 16 * 2);          // per-scene solver diagnostics of the last QP (16 doubles)
  L.trig = take((size_t)B * (T + 1) * 2);        // (cos, sin) of every nominal heading, see npa_trig()
  L.keys = take((size_t)B * (T + 1) * key_stride);
  L.total = o;
  return L;
}
