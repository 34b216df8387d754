// The two steps in front of the PAN loop, batched over scenes (gfx950 only):
//   nominal_kernel  InitialPath.generate_nom_ref_state   neupan/blocks/initial_path.py:68-126
//   scan_kernel     neupan.scan_to_point / scan_to_point_velocity   neupan/neupan.py:173-281
// Both are float64 in the reference (numpy) and cast to float32 at the PAN boundary
// (neupan.py:121); the kernels keep the reference's operation order in float64 and emit float32
// in the layout PAN.forward_batch consumes.  No FMA contraction in this file: a*b+c must round
// twice as it does in numpy.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neupan_amd.h"

namespace {

constexpr double kPi = 3.141592653589793;

// util/__init__.py:114-117 (subtraction loops: the low bits depend on it)
__device__ inline double wrap_to_pi(double r) {
  while (r > kPi) r = r - 2 * kPi;
  while (r < -kPi) r = r + 2 * kPi;
  return r;
}

// ---- nominal / reference rollout: one thread per scene (a serial T-step chain) ----------------
__global__ void nominal_kernel(int batch, int T, int kin, double dt, double L, const double* __restrict__ state,
                               const float* __restrict__ vel, const double* __restrict__ ref_speed,
                               const double* __restrict__ path, const int* __restrict__ curve_off,
                               const int* __restrict__ curve_len, const int* __restrict__ point_index,
                               const double* __restrict__ interval, float* __restrict__ nom_s,
                               float* __restrict__ nom_u, float* __restrict__ ref_s, float* __restrict__ ref_us) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double* cv = path + (size_t)curve_off[b] * 4;         // rows (x, y, theta, gear)
  const int n = curve_len[b];
  const int pidx = point_index[b];
  const double speed = ref_speed[b], itv = interval[b];
  const float* v_row = vel ? vel + (size_t)b * 2 * T : nullptr;
  const float* w_row = vel ? v_row + T : nullptr;
  float* ns = nom_s + (size_t)b * 3 * (T + 1);
  float* rs = ref_s + (size_t)b * 3 * (T + 1);
  float* nu = nom_u + (size_t)b * 2 * T;
  float* ru = ref_us + (size_t)b * T;

  double px = state[b * 3 + 0], py = state[b * 3 + 1], pth = state[b * 3 + 2];
  double rx = cv[pidx * 4 + 0], ry = cv[pidx * 4 + 1], rth = cv[pidx * 4 + 2];      // :76
  const double gear0 = cv[pidx * 4 + 3];                                            // :84
  int ref_index = pidx;
  ns[0] = (float)px; ns[(T + 1)] = (float)py; ns[2 * (T + 1)] = (float)pth;
  rs[0] = (float)rx; rs[(T + 1)] = (float)ry; rs[2 * (T + 1)] = (float)rth;
  const double fwd = speed * dt;                                                    // :86
  const float dt32 = (float)dt, L32 = (float)L;

  for (int t = 0; t < T; ++t) {
    // motion_predict_model (:388-444).  The velocity array is PAN's float32 output: numpy keeps
    // float32 x Python-float products in float32, the state update itself is float64.
    const float v = v_row ? v_row[t] : 0.f, w = w_row ? w_row[t] : 0.f;
    nu[t] = v; nu[T + t] = w;
    if (kin == NPA_KIN_OMNI) {                 // :434-444 (the literal 0 makes that array float64)
      const float vx = v * (float)cos((double)w), vy = v * (float)sin((double)w);
      px = px + dt * (double)vx;
      py = py + dt * (double)vy;
    } else {
      const float c32 = (float)cos(pth), s32 = (float)sin(pth);
      const float d0 = (v * c32) * dt32, d1 = (v * s32) * dt32;
      float d2;
      if (kin == NPA_KIN_ACKER) d2 = ((v * (float)tan((double)w)) / L32) * dt32;    // :398-414
      else d2 = w * dt32;                                                            // :416-432
      px = px + (double)d0; py = py + (double)d1; pth = pth + (double)d2;
    }
    ns[t + 1] = (float)px; ns[(T + 1) + t + 1] = (float)py; ns[2 * (T + 1) + t + 1] = (float)pth;

    double gear = gear0;
    if (fwd >= itv) {                                                               // :93-101
      ref_index = ref_index + (int)(fwd / itv);
      if (ref_index > n - 1) { ref_index = n - 1; gear = 0.0; }
      rx = cv[ref_index * 4 + 0]; ry = cv[ref_index * 4 + 1]; rth = cv[ref_index * 4 + 2];
    } else {                                                                        // :103-109, :183-207
      const double cx = rx, cy = ry;                 // circle centre: the previous reference point
      for (;;) {
        if (ref_index > n - 2) {
          rx = cv[(n - 1) * 4 + 0]; ry = cv[(n - 1) * 4 + 1]; rth = wrap_to_pi(cv[(n - 1) * 4 + 2]);
          break;
        }
        const double* p0 = cv + ref_index * 4;
        const double* p1 = p0 + 4;
        // range_cir_seg (:209-245): far intersection of |p - c| = fwd with the segment p0 -> p1
        const double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
        bool hit = false;
        if (!(dx == 0.0 && dy == 0.0)) {
          const double fx = p0[0] - cx, fy = p0[1] - cy;
          const double a = dx * dx + dy * dy;
          const double bq = (2 * fx) * dx + (2 * fy) * dy;
          const double c = (fx * fx + fy * fy) - fwd * fwd;
          const double disc = bq * bq - (4 * a) * c;
          if (!(disc < 0)) {
            const double t2 = (-bq + sqrt(disc)) / (2 * a);
            if (t2 >= 0 && t2 <= 1) {
              rx = p0[0] + t2 * dx; ry = p0[1] + t2 * dy;
              const double diff = wrap_to_pi(p1[2] - p0[2]);
              rth = wrap_to_pi(p0[2] + diff / 2);
              hit = true;
            }
          }
        }
        if (hit) break;
        ref_index += 1;
      }
      if (ref_index > n - 1) gear = 0.0;
    }
    rth = pth + wrap_to_pi(rth - pth);                                              // :111-112
    rs[t + 1] = (float)rx; rs[(T + 1) + t + 1] = (float)ry; rs[2 * (T + 1) + t + 1] = (float)rth;
    ru[t] = (float)(gear * speed);                                                  // :124
  }
}

// ---- progress along the path: closest_point + check_curve_arrive, one thread per scene --------------
__global__ void progress_kernel(int batch, const double* __restrict__ state, const double* __restrict__ path,
                                const int* __restrict__ curve_off, const int* __restrict__ curve_len,
                                int* __restrict__ point_index, double close_threshold, int ind_range,
                                double arrive_threshold, int arrive_index_threshold, float* __restrict__ min_dis,
                                int* __restrict__ arrived) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double* cv = path + (size_t)curve_off[b] * 4;
  const int n = curve_len[b];
  const double sx = state[b * 3 + 0], sy = state[b * 3 + 1];
  int pidx = point_index[b];
  const int start = pidx > 0 ? pidx : 0;                                   // initial_path.py:165-166
  const int end = pidx + ind_range < n ? pidx + ind_range : n;
  double best = __builtin_inf();
  for (int i = start; i < end; ++i) {
    const double dx = sx - cv[i * 4 + 0], dy = sy - cv[i * 4 + 1];
    const double dis = sqrt(dx * dx + dy * dy);                            // util distance, __init__.py:133
    if (dis < best) {
      best = dis;
      pidx = i;
      if (dis < close_threshold) break;                                    // :176-177
    }
  }
  point_index[b] = pidx;
  if (min_dis) min_dis[b] = (float)best;
  const double ex = sx - cv[(n - 1) * 4 + 0], ey = sy - cv[(n - 1) * 4 + 1];
  const double ad = sqrt(ex * ex + ey * ey);                               // :281-282
  arrived[b] = (ad < arrive_threshold && pidx >= n - arrive_index_threshold - 2) ? 1 : 0;   // :284-287
}

// ---- lidar scan -> global-frame point cloud: one workgroup per scan, ordered compaction -------
constexpr int SCAN_THREADS = 256;

__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(
    int beam_stride, const double* __restrict__ ranges, const double* __restrict__ beam_vel,
    const int* __restrict__ n_beams, const npa_scan_params* __restrict__ params, int mode, int out_stride,
    float* __restrict__ points, float* __restrict__ velocities, int* __restrict__ count) {
  __shared__ int wave_tot[SCAN_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const npa_scan_params P = params[b];
  const int n = n_beams ? n_beams[b] : beam_stride;
  const double* rg = ranges + (size_t)b * beam_stride;
  const double* bvx = beam_vel ? beam_vel + (size_t)b * 2 * beam_stride : nullptr;
  const double* bvy = bvx ? bvx + beam_stride : nullptr;
  float* ox = points + (size_t)b * 2 * out_stride;
  float* oy = ox + out_stride;
  float* ovx = velocities ? velocities + (size_t)b * 2 * out_stride : nullptr;
  float* ovy = ovx ? ovx + out_stride : nullptr;
  const int ds = P.down_sample < 1 ? 1 : P.down_sample;
  // numpy.linspace(angle_min, angle_max, n): i*step + start, last element = stop exactly
  const double step = n > 1 ? (P.angle_max - P.angle_min) / (double)(n - 1) : 0.0;
  const double rmax = P.range_max - 0.02;                                   // neupan.py:205, :256
  const double sc = cos(P.offset[2]), ss = sin(P.offset[2]);                // get_transform(scan_offset)
  const double rc = cos(P.state[2]), rsn = sin(P.state[2]);                 // get_transform(state)
  int base = 0;                                                             // kept beams before this chunk
  for (int i0 = 0; i0 < n; i0 += SCAN_THREADS) {
    const int i = i0 + tid;
    bool keep = false;
    double r = 0, ang = 0;
    if (i < n) {
      r = rg[i];
      ang = (n > 1 && i == n - 1) ? P.angle_max : (double)i * step + P.angle_min;
      const bool in_range = mode == 0 ? (r < rmax && r > P.range_min) : (r < rmax && r >= P.range_min);
      keep = in_range && ang > P.angle_range[0] && ang < P.angle_range[1];
    }
    const unsigned long long m = __ballot(keep);
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    if (lane == 0) wave_tot[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int k = 0; k < wv; ++k) off += wave_tot[k];
    int tot = 0;
    for (int k = 0; k < SCAN_THREADS / 64; ++k) tot += wave_tot[k];
    __syncthreads();
    if (keep) {
      const int k = off + before;                   // index in the reference's point_cloud list
      if (k % ds == 0 && k / ds < out_stride) {     // [:, ::down_sample]
        const double lx = r * cos(ang), ly = r * sin(ang);                  // :208-210
        double tx, ty;
        if (mode == 0) {                            // s_R @ p + s_trans                 :214-215
          tx = (sc * lx + (-ss) * ly) + P.offset[0];
          ty = (ss * lx + sc * ly) + P.offset[1];
        } else {                                    // s_R.T @ (p - s_trans)             :268-270
          const double qx = lx - P.offset[0], qy = ly - P.offset[1];
          tx = sc * qx + ss * qy;
          ty = (-ss) * qx + sc * qy;
        }
        ox[k / ds] = (float)((rc * tx + (-rsn) * ty) + P.state[0]);        // R @ temp + trans  :217-218
        oy[k / ds] = (float)((rsn * tx + rc * ty) + P.state[1]);
        if (ovx) {
          ovx[k / ds] = bvx ? (float)bvx[i] : 0.f;
          ovy[k / ds] = bvy ? (float)bvy[i] : 0.f;
        }
      }
    }
    base += tot;
  }
  if (tid == 0) {
    int c = (base + ds - 1) / ds;
    count[b] = c < out_stride ? c : out_stride;
  }
}

}  // namespace

extern "C" hipError_t npa_launch_nominal(int batch, int T, int kin, double dt, double L, const double* state,
                                         const float* vel, const double* ref_speed, const double* path,
                                         const int* curve_off, const int* curve_len, const int* point_index,
                                         const double* interval, float* nom_s, float* nom_u, float* ref_s,
                                         float* ref_us, hipStream_t stream) {
  const int threads = 64;
  hipLaunchKernelGGL(nominal_kernel, dim3((batch + threads - 1) / threads), dim3(threads), 0, stream, batch, T, kin, dt,
                     L, state, vel, ref_speed, path, curve_off, curve_len, point_index, interval, nom_s, nom_u, ref_s,
                     ref_us);
  return hipGetLastError();
}

extern "C" hipError_t npa_launch_progress(int batch, const double* state, const double* path, const int* curve_off,
                                          const int* curve_len, int* point_index, double close_threshold, int ind_range,
                                          double arrive_threshold, int arrive_index_threshold, float* min_dis,
                                          int* arrived, hipStream_t stream) {
  const int threads = 64;
  hipLaunchKernelGGL(progress_kernel, dim3((batch + threads - 1) / threads), dim3(threads), 0, stream, batch, state, path,
                     curve_off, curve_len, point_index, close_threshold, ind_range, arrive_threshold,
                     arrive_index_threshold, min_dis, arrived);
  return hipGetLastError();
}

extern "C" hipError_t npa_launch_scan(int batch, int beam_stride, const double* ranges, const double* beam_vel,
                                      const int* n_beams, const npa_scan_params* params, int mode, int out_stride,
                                      float* points, float* velocities, int* count, hipStream_t stream) {
  hipLaunchKernelGGL(scan_kernel, dim3(batch), dim3(SCAN_THREADS), 0, stream, beam_stride, ranges, beam_vel, n_beams,
                     params, mode, out_stride, points, velocities, count);
  return hipGetLastError();
}
