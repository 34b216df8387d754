// dune.hip -- point flow + DUNE encoder + nearest-M, gfx950 (MI355X) only.
//
// Replaces, for every horizon slice t of every scene (reference file:line):
//   PAN.generate_point_flow / point_state_transform   neupan/blocks/pan.py:150-212
//   util.downsample_decimation                        neupan/util/__init__.py:285-305
//   ObsPointNet forward                               neupan/blocks/obs_point_net.py:31-49
//   DUNE.forward (lam, distance, sort, gathers)       neupan/blocks/dune.py:58-127
// Only the first M sorted columns are ever consumed downstream (nrmp.py:254-255,
// pan.py:234-237), so no full argsort is done and nothing per-point goes to HBM.
//
// Work decomposition (two launches per PAN iteration)
//   dune_kernel    a flat stream of 32-point tiles over all (scene, slice) pairs of the launch, cut
//                  into equal contiguous chunks, one chunk per resident wave: every wave does the
//                  same number of tiles (+-1), never synchronises with another wave and writes one
//                  order-preserving 32-bit distance key per point (4 B/point, L2/MALL resident).
//   select_kernel  one wave per slice: reads the slice's keys, extracts the M smallest
//                  (distance, index) pairs in ascending order (stable ties), re-encodes just those
//                  M points (one tile) and writes the sorted mu / lam / point / distance rows the
//                  QP consumes.  Rows >= min(N,M) replicate row 0 (padding rule of nrmp.py:258-259).
//
// Mapping to the hardware
//   * the four 32x32 layers run on v_mfma_f32_32x32x2_f32 (exact fp32, 16 K-steps per layer) with
//     the points on the N axis: lane (j = lane&31, hf = lane>>5) ends a layer holding features
//     feat(r,hf) of point j in accumulator register r, which is precisely the B operand layout
//     of the next layer's K-step r -- activations never leave their registers.  Weight
//     A-fragments (65 VGPRs) are loaded once per wave; 128 VGPRs total -> 4 waves per SIMD.
//   * bias / LayerNorm affine vectors and the 32xE output layer sit in LDS (broadcast
//     ds_read_b128); the tanh scale 2*log2(e) is folded into gamma/beta on the host.
//   * LayerNorm reductions: 16 in-lane adds + one v_permlane32_swap (features of a point live in
//     lanes j and j+32).  The 2->32 input layer is one MFMA (K=2); the 32->E output layer is VALU.
//   * on gfx950 the fp32-input MFMA and fp32 VALU work serialise on the SIMD (PMC:
//     SQ_VALU_MFMA_COEXEC_CYCLES = 0), so the VALU instruction count per tile matters as much as
//     the 65 MFMAs.
#include "pan_common.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "dune_device.h"

// ---- host-side launchers (called from c_api.hip) --------------------------------------------------
// t0 = first horizon slice to evaluate: slice 0 does not depend on the iterate (s(0) is pinned,
// robot.py:234), so after the first PAN iteration of a forward call only slices 1..T are redone.
// The two launches may go to different streams (select only needs the keys of its own scenes).
static int tiles_per_slice(const DevParams& P, int n_stride) {
  int n_use_max = n_stride < P.dune_max_num ? n_stride : P.dune_max_num;
  if (n_use_max < 1) n_use_max = 1;
  return (n_use_max + 31) / 32;
}

extern "C" hipError_t npa_launch_encode(const DevParams& P, const float* wpack, int batch, int scene0, int t0,
                                        int n_stride, const float* cur_s, const float* points, const float* vel,
                                        const int* n_points, const int* flags, unsigned* gkeys, const float* trig,
                                        int n_cu, int blocks_per_cu, int key_terms, hipStream_t stream,
                                        hipEvent_t ev_start, hipEvent_t ev_stop) {
  // key_terms: 0 = exact fp32 encoder for the keys, 3 = fp16x2 split products, 1 = single fp16 products
  // ev_start / ev_stop (may be null) are attached to the dispatch itself (hipExtLaunchKernelGGL): no
  // separate marker packets on the stream, which cost ~5 us each between back-to-back launches
  const int nsl = P.T + 1 - t0;
  const int tps = tiles_per_slice(P, n_stride);
  if (tps * 32 > P.key_stride) return hipErrorInvalidValue;
  const long long tiles = (long long)batch * nsl * tps;
  if (tiles >= (1ll << 31)) return hipErrorInvalidValue;
  // resident workgroups (4-wave form): blocks_per_cu per CU; the exact-fp32 variant needs 117 VGPRs
  const bool split = key_terms != 0, single = key_terms == 1;
  if (!split && blocks_per_cu > 4) blocks_per_cu = 4;
  // one 16-wave workgroup per CU once every wave has a few tiles to stream; small launches keep
  // 4-wave workgroups (more CUs busy).  NPA_ENC_WAVES=4 forces the small form.
  static const bool small_only = getenv("NPA_ENC_WAVES") && atoi(getenv("NPA_ENC_WAVES")) == 4;
  // (the 16-wave form is instantiated for E = 4, the polygon of every shipped robot but one; other sizes keep 4-wave workgroups)
  int waves = (split && !small_only && P.E == 4 && tiles >= (long long)n_cu * 16 * 4) ? 16 : DUNE_WAVES;
  const int slots = waves >= 8 ? n_cu : n_cu * blocks_per_cu;
  int blocks = (int)((tiles + waves - 1) / waves);
  if (blocks > slots) blocks = slots;
  if (blocks < 1) blocks = 1;
  const size_t shmem = (11 * 32 + 8 * 32 + 8) * sizeof(float) + (split ? WP_KEY_LDS_FLOATS * sizeof(float) : 0);
  static const int chunk_env = getenv("NPA_ENC_CHUNK") ? atoi(getenv("NPA_ENC_CHUNK")) : 2;
  const int chunk = chunk_env < 1 ? 1 : chunk_env;
#define LAUNCH1(EE, SP, WV)                                                                                         \
  hipExtLaunchKernelGGL((dune_kernel<EE, SP, WV>), dim3(blocks), dim3(64 * WV), shmem, stream, ev_start, ev_stop, 0, \
                        P, wpack, n_stride, cur_s, points, vel, n_points, flags, gkeys, tps * 32, scene0, batch, t0, \
                        chunk, trig)
#define LAUNCH(EE)                                                                                                  \
  do {                                                                                                              \
    if (single) LAUNCH1(EE, 1, DUNE_WAVES);                                                                         \
    else if (split) LAUNCH1(EE, 3, DUNE_WAVES);                                                                     \
    else LAUNCH1(EE, 0, DUNE_WAVES);                                                                                \
  } while (0)
#define LAUNCH16(EE)                                                                                                \
  do {                                                                                                              \
    if (single && waves == 16) LAUNCH1(EE, 1, 16);                                                                  \
    else if (split && waves == 16) LAUNCH1(EE, 3, 16);                                                              \
    else LAUNCH(EE);                                                                                                \
  } while (0)
  switch (P.E) {
    case 3: LAUNCH(3); break;
    case 4: LAUNCH16(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCH(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH16
#undef LAUNCH
#undef LAUNCH1
  return hipGetLastError();
}

extern "C" hipError_t npa_launch_select(const DevParams& P, const float* wpack, int batch, int scene0, int t0,
                                        int n_stride, const float* cur_s, const float* points, const float* vel,
                                        const int* n_points, const int* flags, const unsigned* gkeys,
                                        const float* trig, float* mu_sorted, float* lam_sorted, float* pts_sorted,
                                        float* dist_sorted, int* count, int key_terms, float e0, unsigned* stats,
                                        int debug, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
  // ev_start / ev_stop (may be null) ride on the dispatch (hipExtLaunchKernelGGL): no marker packets on the stream
  // debug != 0: count[] also carries the number of candidates and the overflow flag (tests/tools/geo_check.py)
  // key_terms: 0 = exact network keys in gkeys, 1 / 3 = reduced-precision network keys in gkeys (candidates within
  // the margin e0 (1 + |d|) are re-ranked), 4 = geometric keys computed by select_kernel itself (gkeys unused)
  const int nsl = P.T + 1 - t0;
  const int tps = tiles_per_slice(P, n_stride);
  const int dbg = debug ? 2 : 1;
  const bool geo = key_terms == 4;
  const int approx = key_terms == 0 ? 0 : dbg;
  const size_t key_area = std::max<size_t>((size_t)tps * 32 * sizeof(unsigned), SEL_CAP * (NPA_MAX_E + 5 + 2) * sizeof(float));
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + (SEL_CAP + NPA_MAX_M + 64) * sizeof(int) +
                       (key_area + 15) / 16 * 16;
  // slices of more than ~15 000 points keep more than the default 64 KB of dynamic LDS (4 B per key)
#define LAUNCH1(EE, GG)                                                                                             \
  do {                                                                                                              \
    static NpaDeviceOnce big_lds;                                                                                   \
    int dev_ = 0;                                                                                                   \
    if (shmem > 60 * 1024 && big_lds.need(&dev_)) {                                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel<EE, GG>),                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (e_ != hipSuccess) return e_;                                                                              \
      big_lds.done(dev_);                                                                                           \
    }                                                                                                               \
    hipExtLaunchKernelGGL((select_kernel<EE, GG>), dim3(nsl, batch), dim3(64), shmem, stream, ev_start, ev_stop, 0, P,    \
                          wpack, n_stride, cur_s, points, vel, n_points, flags, gkeys, tps * 32, mu_sorted, lam_sorted, \
                          pts_sorted, dist_sorted, count, scene0, t0, approx, e0, stats, trig);                      \
  } while (0)
#ifdef NPA_EXPERIMENTS
#define LAUNCH(EE)                                                                                                  \
  do {                                                                                                              \
    if (geo) LAUNCH1(EE, true);                                                                                     \
    else LAUNCH1(EE, false);                                                                                        \
  } while (0)
#else               // (the first form of the geometric selection, NPA_SELECT_V1: experiments build only)
#define LAUNCH(EE)                                                                                                  \
  do {                                                                                                              \
    if (geo) return hipErrorInvalidValue;                                                                           \
    LAUNCH1(EE, false);                                                                                             \
  } while (0)
#endif
  switch (P.E) {
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCH(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH
#undef LAUNCH1
  return hipGetLastError();
}

extern "C" hipError_t npa_launch_select_geo(const DevParams& P, const float* wpack, int batch, int scene0, int t0,
                                            int n_stride, const float* cur_s, const float* points, const float* vel,
                                            const int* n_points, const int* flags, const float* trig, float* mu_sorted,
                                            float* lam_sorted, float* pts_sorted, float* dist_sorted, int* count,
                                            unsigned* stats, int debug, unsigned* audit, unsigned audit_thresh,
                                            unsigned audit_seed, float margin_scale, int rows_bf16, hipStream_t stream,
                                            hipEvent_t ev_start, hipEvent_t ev_stop) {
  // select_geo_kernel: one wave per (scene, slice); the grid is padded to a multiple of 8 scenes so that the XCD-aware
  // block -> (scene, slice) map covers every scene (the surplus workgroups return at once)
  const int nsl = P.T + 1 - t0;
  int n_use_max = n_stride < P.dune_max_num ? n_stride : P.dune_max_num;
  if (n_use_max < 1) n_use_max = 1;
  const size_t n_pad = ((size_t)n_use_max + SEL2_TRIP - 1) / SEL2_TRIP * SEL2_TRIP;
  const size_t key_area = std::max<size_t>(n_pad * sizeof(unsigned), SEL_CAP * (NPA_MAX_E + 5 + 2) * sizeof(float));
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + (2 * SEL_CAP + NPA_MAX_M) * sizeof(int) +
                       (key_area + 15) / 16 * 16;
  const int blocks = (batch + 7) / 8 * 8 * nsl;
#define LAUNCHG(EE, BB, KK)                                                                                         \
  do {                                                                                                              \
    static NpaDeviceOnce big_lds;                                                                                   \
    int dev_ = 0;                                                                                                   \
    if (shmem > 60 * 1024 && big_lds.need(&dev_)) {                                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(select_geo_kernel<EE, BB, KK>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (e_ != hipSuccess) return e_;                                                                              \
      big_lds.done(dev_);                                                                                           \
    }                                                                                                               \
    hipExtLaunchKernelGGL((select_geo_kernel<EE, BB, KK>), dim3(blocks), dim3(64), shmem, stream, ev_start, ev_stop, 0, P, wpack,  \
                          n_stride, cur_s, points, vel, n_points, flags, mu_sorted, lam_sorted, pts_sorted, dist_sorted, \
                          count, scene0, t0, nsl, batch, debug, stats, trig, audit, audit_thresh, audit_seed, margin_scale); \
  } while (0)
  // rows_bf16: 0 = exact; 1 = the reduced-precision tier of the ROWS; 2 = the bf16 tier of the KEYS (rows exact).  Both tiers are
  // instantiated for the two polygon sizes the benchmark configurations use.
#define LAUNCH(EE)                                                                                                  \
  do {                                                                                                              \
    if (rows_bf16) return hipErrorInvalidValue;                                                                     \
    LAUNCHG(EE, false, false);                                                                                      \
  } while (0)
#define LAUNCHB(EE)                                                                                                 \
  do {                                                                                                              \
    if (rows_bf16 == 1) LAUNCHG(EE, true, false);                                                                   \
    else if (rows_bf16 == 2) LAUNCHG(EE, false, true);                                                              \
    else LAUNCHG(EE, false, false);                                                                                 \
  } while (0)
  switch (P.E) {
    case 3: LAUNCH(3); break;
    case 4: LAUNCHB(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCHB(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH
#undef LAUNCHB
#undef LAUNCHG
  return hipGetLastError();
}

// select_geo_kernel for a group of n forward calls that share the configuration, the batch size and the stream (c_api.hip
// checks that): ONE launch, grid (blocks of one call, n).  n_stride_max sizes the LDS key area.
extern "C" hipError_t npa_launch_select_geo_group(const DevParams& P, const SelGeoGroup& G, int n, int batch, int t0, int n_stride_max,
                                                  int debug, unsigned audit_thresh, float margin_scale, int rows_bf16,
                                                  hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n < 1 || n > NPA_GROUP_MAX) return hipErrorInvalidValue;
  const int nsl = P.T + 1 - t0;
  int n_use_max = n_stride_max < P.dune_max_num ? n_stride_max : P.dune_max_num;
  if (n_use_max < 1) n_use_max = 1;
  const size_t n_pad = ((size_t)n_use_max + SEL2_TRIP - 1) / SEL2_TRIP * SEL2_TRIP;
  const size_t key_area = std::max<size_t>(n_pad * sizeof(unsigned), SEL_CAP * (NPA_MAX_E + 5 + 2) * sizeof(float));
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float) + (2 * SEL_CAP + NPA_MAX_M) * sizeof(int) +
                       (key_area + 15) / 16 * 16;
  const int blocks = (batch + 7) / 8 * 8 * nsl;
#define LAUNCHG(EE, BB, KK)                                                                                         \
  do {                                                                                                              \
    static NpaDeviceOnce big_lds;                                                                                   \
    int dev_ = 0;                                                                                                   \
    if (shmem > 60 * 1024 && big_lds.need(&dev_)) {                                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(select_geo_group_kernel<EE, BB, KK>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (e_ != hipSuccess) return e_;                                                                              \
      big_lds.done(dev_);                                                                                           \
    }                                                                                                               \
    hipExtLaunchKernelGGL((select_geo_group_kernel<EE, BB, KK>), dim3(blocks, n), dim3(64), shmem, stream, ev_start, ev_stop, 0, P, G, \
                          t0, nsl, batch, debug, audit_thresh, margin_scale);                                       \
  } while (0)
  // (instantiated for the polygon sizes of the benchmark configurations, exact and with bf16 KEYS; the lossy bf16 ROWS tier and
  // other polygon sizes stay call by call -- c_api.hip asks npa_select_geo_group_supported)
  if (rows_bf16 == 1) return hipErrorInvalidValue;
  if (P.E == 4) { if (rows_bf16 == 2) LAUNCHG(4, false, true); else LAUNCHG(4, false, false); }
  else if (P.E == 8) { if (rows_bf16 == 2) LAUNCHG(8, false, true); else LAUNCHG(8, false, false); }
  else return hipErrorInvalidValue;
#undef LAUNCHG
  return hipGetLastError();
}
extern "C" int npa_select_geo_group_supported(int E) { return E == 4 || E == 8; }

// ---- geometric-key calibration (npa_create) ----------------------------------------------------------
// f(p) = network distance - geometric distance is a smooth function of the robot-frame position and a property of
// the checkpoint.  Per distance band (npa_geo_band) this records, over a square grid of spacing d, max |f| and the
// largest difference of f between grid neighbours in x and y (how much f can exceed its grid maximum between grid
// points).  The
// host runs it on nested grids (fine next to the robot, coarser outwards; a grid skips the inner square a finer one
// covers) and builds margin[band] = safety x (max |f| + max |delta f|), taken over the band and its two neighbours.
template <int E>
__global__ __launch_bounds__(256) void geo_calib_kernel(DevParams P, const float* __restrict__ wpack, float half, int nside,
                                                        float inner, float shift, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* vec = smem;
  float* w6 = vec + 11 * 32;
  float* b6 = w6 + 8 * 32;
  unsigned* tab = reinterpret_cast<unsigned*>(b6 + 8);          // [2][NPA_GEO_BANDS]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 11 * 32 + 8 * 32 + 8; i += blockDim.x) smem[i] = wpack[WP_VEC + i];
  for (int i = tid; i < 2 * NPA_GEO_BANDS; i += blockDim.x) tab[i] = 0u;
  __syncthreads();
  WaveWeights W;
  load_weights(wpack, lane, W);
  // 8 x 4 points per tile so that a lane's x- and y-neighbours (lanes j + 1, j + 8) sit in the same tile; nside % 8 == 0
  const float step = 2.0f * half / (float)(nside - 1);
  const int tiles_x = nside >> 3, tiles_y = nside >> 2;
  const long long tiles = (long long)tiles_x * tiles_y;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (tid >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long tile = wave; tile < tiles; tile += nwave) {
    const int ix = (int)(tile % tiles_x) * 8 + (j & 7), iy = (int)(tile / tiles_x) * 4 + (j >> 3);
    // shift = 0.5: the nodes sit at the cell centres of the unshifted grid -- the points FARTHEST from its nodes, where a
    // bound "node maximum + neighbour difference" is most at risk (npa_create compares the two grids)
    const float p0x = -half + step * ((float)ix + shift), p0y = -half + step * ((float)iy + shift);
    float me[E];
    encode_tile<E>(W, vec, w6, b6, p0x, p0y, lane, me);
    float de = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) de = fmaf(me[e], __fsub_rn(fmaf(P.G[e][0], p0x, __fmul_rn(P.G[e][1], p0y)), P.h[e]), de);
    const float g = geo_dist<E>(P, p0x, p0y);
    float f = de - g;
    if (!(f == f)) f = 3.0e38f;                                 // NaN anywhere disqualifies the band
    const float fx = __shfl_down(f, 1, 64), fy = __shfl_down(f, 8, 64);
    if (hf == 0 && fmaxf(fabsf(p0x), fabsf(p0y)) >= inner) {
      const int band = npa_geo_band(g);
      atomicMax(&tab[band], __float_as_uint(fabsf(f)));
      float df = 0.f;
      if ((j & 7) < 7) df = fabsf(fx - f);
      if (j < 24) df = fmaxf(df, fabsf(fy - f));
      atomicMax(&tab[NPA_GEO_BANDS + band], __float_as_uint(df));
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * NPA_GEO_BANDS; i += blockDim.x)
    if (tab[i]) atomicMax(&out[i], tab[i]);
}

extern "C" hipError_t npa_launch_geo_calib(const DevParams& P, const float* wpack, int nside, float half, float inner,
                                           float shift, unsigned* out, int n_cu, hipStream_t stream) {
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + 2 * NPA_GEO_BANDS) * sizeof(float);
  const int blocks = n_cu * 4;
#define LAUNCH(EE) hipLaunchKernelGGL((geo_calib_kernel<EE>), dim3(blocks), dim3(256), shmem, stream, P, wpack, half, nside, inner, shift, out)
  switch (P.E) {
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCH(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH
  return hipGetLastError();
}

// ---- bf16 KEY tier: |bf16-encoder distance - exact distance| per band of the exact distance (npa_create) -----------------
// Rounding noise, not a smooth function: measured on the same nested squares as the geometric margin (1024 x 1024 nodes each),
// the host applies a safety factor and the kernel audits every survivor at run time (select_geo_body.inc).
template <int E>
__global__ __launch_bounds__(256) void k16_calib_kernel(DevParams P, const float* __restrict__ wpack, float half, int nside, float inner,
                                                        unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* vec = smem;
  float* w6 = vec + 11 * 32;
  float* b6 = w6 + 8 * 32;
  unsigned* tab = reinterpret_cast<unsigned*>(b6 + 8);          // [NPA_GEO_BANDS]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 11 * 32 + 8 * 32 + 8; i += blockDim.x) smem[i] = wpack[WP_VEC + i];
  for (int i = tid; i < NPA_GEO_BANDS; i += blockDim.x) tab[i] = 0u;
  __syncthreads();
  const float w1 = wpack[WP_W1 + lane];
  const float step = 2.0f * half / (float)(nside - 1);
  const int tiles_x = nside >> 3, tiles_y = nside >> 2;
  const long long tiles = (long long)tiles_x * tiles_y;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (tid >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long tile = wave; tile < tiles; tile += nwave) {
    const int ix = (int)(tile % tiles_x) * 8 + (j & 7), iy = (int)(tile / tiles_x) * 4 + (j >> 3);
    const float p0x = -half + step * (float)ix, p0y = -half + step * (float)iy;
    float me[E], mb[E];
    encode_tile_stream<E>(w1, wpack + WP_WLS, vec, w6, b6, p0x, p0y, lane, me);
    encode_tile_bf16<E>(w1, wpack + WP_WB16, vec, w6, b6, p0x, p0y, lane, mb);
    float de = 0.f, db = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float tmp = __fsub_rn(fmaf(P.G[e][0], p0x, __fmul_rn(P.G[e][1], p0y)), P.h[e]);
      de = fmaf(me[e], tmp, de);
      db = fmaf(mb[e], tmp, db);
    }
    float f = fabsf(db - de);
    if (!(f == f)) f = 3.0e38f;
    if (hf == 0 && fmaxf(fabsf(p0x), fabsf(p0y)) >= inner) atomicMax(&tab[npa_geo_band(de)], __float_as_uint(f));
  }
  __syncthreads();
  for (int i = tid; i < NPA_GEO_BANDS; i += blockDim.x)
    if (tab[i]) atomicMax(&out[i], tab[i]);
}
extern "C" hipError_t npa_launch_k16_calib(const DevParams& P, const float* wpack, int nside, float half, float inner, unsigned* out,
                                           int n_cu, hipStream_t stream) {
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float);
  const int blocks = n_cu * 4;
  if (P.E == 4) hipLaunchKernelGGL((k16_calib_kernel<4>), dim3(blocks), dim3(256), shmem, stream, P, wpack, half, nside, inner, out);
  else if (P.E == 8) hipLaunchKernelGGL((k16_calib_kernel<8>), dim3(blocks), dim3(256), shmem, stream, P, wpack, half, nside, inner, out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- table-corrected geometric key (npa_create): the table, then its residual ----------------------------------------------
// f = network distance - geometric key at the (N + 1)^2 nodes of every level (exact encoder; the key is the run-time one,
// geo_key: whatever it rounds, the table absorbs), then packed cell by cell (four corners as fp16: one gather per lookup).
template <int E>
__global__ __launch_bounds__(256) void geo_table_nodes_kernel(DevParams P, const float* __restrict__ wpack, float* __restrict__ nodes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* vec = smem;
  float* w6 = vec + 11 * 32;
  float* b6 = w6 + 8 * 32;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 11 * 32 + 8 * 32 + 8; i += blockDim.x) smem[i] = wpack[WP_VEC + i];
  __syncthreads();
  const float w1 = wpack[WP_W1 + lane];
  constexpr int nside = NPA_TAB_N + 1, per = nside * nside, total = NPA_TAB_LEVELS * per;
  const int tiles = (total + 31) / 32;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (tid >> 6), nwave = gridDim.x * (blockDim.x >> 6);
  for (int tile = wave; tile < tiles; tile += nwave) {
    const int n = tile * 32 + j, nc = n < total ? n : total - 1;
    const int lvl = nc / per, r = nc - lvl * per, iy = r / nside, ix = r - iy * nside;
    const float half = wpack[WP_TABH + 2] * (lvl == 0 ? 1.0f : (lvl == 1 ? 4.0f : (lvl == 2 ? 16.0f : 64.0f)));
    const float step = 2.0f * half / (float)NPA_TAB_N;
    const float p0x = wpack[WP_TABH] + fmaf(step, (float)ix, -half), p0y = wpack[WP_TABH + 1] + fmaf(step, (float)iy, -half);
    float me[E];
    encode_tile_stream<E>(w1, wpack + WP_WLS, vec, w6, b6, p0x, p0y, lane, me);
    float de = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) de = fmaf(me[e], __fsub_rn(fmaf(P.G[e][0], p0x, __fmul_rn(P.G[e][1], p0y)), P.h[e]), de);
    if (hf == 0 && n < total) nodes[n] = de - geo_key<E>(P, p0x, p0y);
  }
}
__global__ void geo_table_pack_kernel(const float* __restrict__ nodes, float* __restrict__ wpack) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int N = NPA_TAB_N, nside = N + 1;
  if (c >= NPA_TAB_LEVELS * N * N) return;
  const int lvl = c / (N * N), r = c - lvl * N * N, iy = r / N, ix = r - iy * N;
  const float* nl = nodes + (size_t)lvl * nside * nside;
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  union { unsigned u; h2_t h; } lo, hi;
  lo.h[0] = (_Float16)nl[iy * nside + ix]; lo.h[1] = (_Float16)nl[iy * nside + ix + 1];
  hi.h[0] = (_Float16)nl[(iy + 1) * nside + ix]; hi.h[1] = (_Float16)nl[(iy + 1) * nside + ix + 1];
  reinterpret_cast<uint2*>(wpack + WP_TAB)[c] = make_uint2(lo.u, hi.u);
}
extern "C" hipError_t npa_launch_geo_table(const DevParams& P, float* wpack, float* nodes, int n_cu, hipStream_t stream) {
  const size_t shmem = (11 * 32 + 8 * 32 + 8) * sizeof(float);
  const int blocks = n_cu * 4;
  // (built for the polygon sizes the filter is built for: select_geo_body.inc, TABF)
#define LAUNCH(EE) hipLaunchKernelGGL((geo_table_nodes_kernel<EE>), dim3(blocks), dim3(256), shmem, stream, P, wpack, nodes)
  switch (P.E) {
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int cells = NPA_TAB_LEVELS * NPA_TAB_N * NPA_TAB_N;
  hipLaunchKernelGGL(geo_table_pack_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, nodes, wpack);
  return hipGetLastError();
}
// |corrected key - exact distance| per band of the KEY, on nside x nside nodes of one calibration square (not
// aligned with the table's cells: 4095 steps against 512 cells, the offset inside a cell drifts through every value)
template <int E>
__global__ __launch_bounds__(256) void ktab_calib_kernel(DevParams P, const float* __restrict__ wpack, float half, int nside, float inner,
                                                         float cx, float cy, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* vec = smem;
  float* w6 = vec + 11 * 32;
  float* b6 = w6 + 8 * 32;
  unsigned* tab = reinterpret_cast<unsigned*>(b6 + 8);          // [NPA_GEO_BANDS]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 11 * 32 + 8 * 32 + 8; i += blockDim.x) smem[i] = wpack[WP_VEC + i];
  for (int i = tid; i < NPA_GEO_BANDS; i += blockDim.x) tab[i] = 0u;
  __syncthreads();
  const float w1 = wpack[WP_W1 + lane];
  const float step = 2.0f * half / (float)(nside - 1);
  const int tiles_x = nside >> 3, tiles_y = nside >> 2;
  const long long tiles = (long long)tiles_x * tiles_y;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (tid >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long tile = wave; tile < tiles; tile += nwave) {
    const int ix = (int)(tile % tiles_x) * 8 + (j & 7), iy = (int)(tile / tiles_x) * 4 + (j >> 3);
    const float qx = -half + step * (float)ix, qy = -half + step * (float)iy;        // (relative to the table's centre)
    const float p0x = cx + qx, p0y = cy + qy;
    float me[E];
    encode_tile_stream<E>(w1, wpack + WP_WLS, vec, w6, b6, p0x, p0y, lane, me);
    float de = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) de = fmaf(me[e], __fsub_rn(fmaf(P.G[e][0], p0x, __fmul_rn(P.G[e][1], p0y)), P.h[e]), de);
    const float kc = geo_key<E>(P, p0x, p0y) + geo_tab_corr(wpack, p0x, p0y);
    float f = fabsf(kc - de);
    if (!(f == f)) f = 3.0e38f;
    // (per band of the KEY: that is what a wave knows of a point when it looks the margin up)
    if (hf == 0 && fmaxf(fabsf(qx), fabsf(qy)) >= inner) atomicMax(&tab[npa_geo_band(kc)], __float_as_uint(f));
  }
  __syncthreads();
  for (int i = tid; i < NPA_GEO_BANDS; i += blockDim.x)
    if (tab[i]) atomicMax(&out[i], tab[i]);
}
extern "C" hipError_t npa_launch_ktab_calib(const DevParams& P, const float* wpack, int nside, float half, float inner, float cx, float cy,
                                            unsigned* out, int n_cu, hipStream_t stream) {
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + NPA_GEO_BANDS) * sizeof(float);
  const int blocks = n_cu * 4;
#define LAUNCH(EE) hipLaunchKernelGGL((ktab_calib_kernel<EE>), dim3(blocks), dim3(256), shmem, stream, P, wpack, half, nside, inner, cx, cy, out)
  switch (P.E) {
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH
  return hipGetLastError();
}

// ---- key-error calibration (npa_create) ------------------------------------------------------------
// The distance of a point depends on its robot-frame position only, so the error of the reduced-precision key
// path is a property of the checkpoint: evaluate both encoders on a grid over the square the DUNE models are
// trained on (|x|, |y| <= 25 m, dune_train.py data_range) and return max |key - exact| / (1 + |exact|).
template <int E, int TERMS>
__global__ __launch_bounds__(256) void key_calib_kernel(DevParams P, const float* __restrict__ wpack, float lo, float step,
                                                        int nside, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* vec = smem;
  float* w6 = vec + 11 * 32;
  float* b6 = w6 + 8 * 32;
  float* wb = b6 + 8;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 11 * 32 + 8 * 32 + 8; i += blockDim.x) smem[i] = wpack[WP_VEC + i];
  for (int i = tid; i < WP_KEY_LDS_FLOATS; i += blockDim.x) wb[i] = wpack[WP_BF + i];
  __syncthreads();
  const f16x8* wbf = reinterpret_cast<const f16x8*>(wb);
  WaveWeights W;
  load_weights(wpack, lane, W);
  const float kw1 = wpack[WP_KW1 + lane];
  const int tile = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
  const int n = tile * 32 + j, total = nside * nside;
  const int nc = n < total ? n : total - 1;
  const float p0x = lo + step * (float)(nc % nside), p0y = lo + step * (float)(nc / nside);
  float mk[E], me[E];
  encode_tile_keys<E, TERMS>(kw1, wbf, vec, w6, b6, p0x, p0y, lane, mk);
  encode_tile<E>(W, vec, w6, b6, p0x, p0y, lane, me);
  float dk = 0.f, de = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    // (ONE instantiation, E = NPA_MAX_E, serves every polygon: the rows of the output layer beyond P.E are zero in the pack, so
    // mk / me are 0 there and the sums below are those of the E-sized loop, term by term)
    const float tmp = __fsub_rn(fmaf(P.G[e][0], p0x, __fmul_rn(P.G[e][1], p0y)), P.h[e]);
    dk = e < P.E ? fmaf(mk[e], tmp, dk) : dk;
    de = e < P.E ? fmaf(me[e], tmp, de) : de;
  }
  float rel = fabsf(dk - de) / (1.0f + fabsf(de));
  if (!(rel == rel)) rel = 1e30f;                            // NaN anywhere disqualifies the mode
  if (hf == 0 && n < total) atomicMax(out, __float_as_uint(rel));
}

extern "C" hipError_t npa_launch_key_calib(const DevParams& P, const float* wpack, int key_terms, int nside, float half,
                                           unsigned* out, hipStream_t stream) {
  const int tiles = (nside * nside + 31) / 32, blocks = (tiles + 3) / 4;
  const float lo = -half, step = 2.0f * half / (float)(nside - 1);
  const size_t shmem = (11 * 32 + 8 * 32 + 8 + WP_KEY_LDS_FLOATS) * sizeof(float);
#define LAUNCH(EE)                                                                                                   \
  do {                                                                                                               \
    if (key_terms == 1) hipLaunchKernelGGL((key_calib_kernel<EE, 1>), dim3(blocks), dim3(256), shmem, stream, P, wpack, lo, step, nside, out); \
    else hipLaunchKernelGGL((key_calib_kernel<EE, 3>), dim3(blocks), dim3(256), shmem, stream, P, wpack, lo, step, nside, out); \
  } while (0)
  if (P.E < 3 || P.E > NPA_MAX_E) return hipErrorInvalidValue;
  LAUNCH(NPA_MAX_E);
#undef LAUNCH
  return hipGetLastError();
}

// (cos, sin) table for a nominal trajectory that no kernel of ours wrote (npa_dune_stage)
__global__ void trig_kernel(const float* __restrict__ cur_s, int nscene, int T, float* __restrict__ trig) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nscene * (T + 1)) return;
  const int b = i / (T + 1), t = i - b * (T + 1);
  float c, s;
  npa_trig(cur_s[(size_t)b * 3 * (T + 1) + 2 * (T + 1) + t], c, s);
  trig[2 * (size_t)i] = c;
  trig[2 * (size_t)i + 1] = s;
}
extern "C" hipError_t npa_launch_trig(const float* cur_s, int batch, int T, float* trig, hipStream_t stream) {
  const int n = batch * (T + 1);
  hipLaunchKernelGGL(trig_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cur_s, batch, T, trig);
  return hipGetLastError();
}
