// DUNE training labels, batched (gfx950 only).  Replaces the per-point SOCP of
// neupan/blocks/dune_train.py:82-99, :137-140 (cvxpy -> ECOS, "1-2 h on CPU" for 100 k points):
//        max_mu  mu^T (G p - h)   s.t.  || G^T mu ||_2 <= 1,  mu >= 0
// by its closed form: the dual certificate of the distance from p to the polygon {x : G x <= h}
// (oracle/dune_label_oracle.py states the derivation).  One thread per point, float64, float32
// labels out (the reference stores float32 tensors, dune_train.py:101-107).
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neupan_amd.h"

namespace {

struct Polygon {
  int E;
  double G[NPA_MAX_E][2], h[NPA_MAX_E];
  double V[NPA_MAX_E][2];        // vertex e = edges e-1 and e
  double inv_norm[NPA_MAX_E];    // 1 / |G_e|
};

__global__ void label_kernel(Polygon Q, long long n, const double* __restrict__ points, float* __restrict__ mu_out,
                             float* __restrict__ dist_out) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int E = Q.E;
  const double px = points[2 * k], py = points[2 * k + 1];
  double mu[NPA_MAX_E];
  double smax = -1e300;
#pragma unroll
  for (int e = 0; e < NPA_MAX_E; ++e) {
    mu[e] = 0.0;
    if (e < E) {
      const double s = (Q.G[e][0] * px + Q.G[e][1] * py) - Q.h[e];
      smax = s > smax ? s : smax;
    }
  }
  double dist = 0.0;
  if (smax > 0) {
    double best = 1e300, bt = 0, bqx = 0, bqy = 0;
    int be = 0;
    for (int e = 0; e < E; ++e) {                          // edge e: V[e] -> V[e+1]
      const int e1 = e + 1 == E ? 0 : e + 1;
      const double ax = Q.V[e][0], ay = Q.V[e][1];
      const double dx = Q.V[e1][0] - ax, dy = Q.V[e1][1] - ay;
      double t = ((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy);
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      const double qx = ax + t * dx, qy = ay + t * dy;
      const double dd = (px - qx) * (px - qx) + (py - qy) * (py - qy);
      if (dd < best) { best = dd; be = e; bt = t; bqx = qx; bqy = qy; }
    }
    dist = sqrt(best);
    if (bt > 0.0 && bt < 1.0) {                            // nearest point interior to edge be
      mu[be] = Q.inv_norm[be];
      dist = ((Q.G[be][0] * px + Q.G[be][1] * py) - Q.h[be]) * Q.inv_norm[be];
    } else {                                               // a vertex: the two edges meeting there
      const int i = bt == 0.0 ? (be == 0 ? E - 1 : be - 1) : be;
      const int j = bt == 0.0 ? be : (be + 1 == E ? 0 : be + 1);
      const double nx = (px - bqx) / dist, ny = (py - bqy) / dist;
      // [G_i^T G_j^T] (mu_i, mu_j)^T = n
      const double a = Q.G[i][0], b = Q.G[j][0], c = Q.G[i][1], d = Q.G[j][1];
      const double det = a * d - b * c;
      const double mi = (nx * d - b * ny) / det, mj = (a * ny - nx * c) / det;
      mu[i] = mi > 0.0 ? mi : 0.0;
      mu[j] = mj > 0.0 ? mj : 0.0;
    }
  }
#pragma unroll
  for (int e = 0; e < NPA_MAX_E; ++e)
    if (e < E) mu_out[k * E + e] = (float)mu[e];
  dist_out[k] = (float)dist;
}

}  // namespace

// G [E][2], h [E]: host arrays, consecutive counter-clockwise edges (gen_inequal_from_vertex order)
extern "C" hipError_t npa_launch_labels(int E, const double* G, const double* h, long long n, const double* points,
                                        float* mu, float* dist, hipStream_t stream) {
  Polygon Q;
  Q.E = E;
  for (int e = 0; e < NPA_MAX_E; ++e) {
    Q.G[e][0] = e < E ? G[2 * e] : 0.0; Q.G[e][1] = e < E ? G[2 * e + 1] : 0.0; Q.h[e] = e < E ? h[e] : 0.0;
    Q.V[e][0] = Q.V[e][1] = 0.0; Q.inv_norm[e] = 0.0;
  }
  for (int e = 0; e < E; ++e) {
    const int p = e == 0 ? E - 1 : e - 1;
    const double a = Q.G[p][0], b = Q.G[p][1], c = Q.G[e][0], d = Q.G[e][1];
    const double det = a * d - b * c;
    if (det == 0.0) return hipErrorInvalidValue;           // parallel consecutive edges
    Q.V[e][0] = (Q.h[p] * d - b * Q.h[e]) / det;
    Q.V[e][1] = (a * Q.h[e] - Q.h[p] * c) / det;
    const double nn = sqrt(c * c + d * d);
    if (nn == 0.0) return hipErrorInvalidValue;
    Q.inv_norm[e] = 1.0 / nn;
  }
  if (n < 1) return hipSuccess;
  const int threads = 256;
  hipLaunchKernelGGL(label_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, stream, Q, n, points,
                     mu, dist);
  return hipGetLastError();
}
