// Hardware probe (gfx950): v_pk_fma_f32 whose destination pair overlaps a source pair that is read
// with an op_sel half-swap.  hipcc --offload-arch=gfx950 -O2 tests/tools/hw/pk_opsel_overlap.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void probe(const float* in, float* out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f2 acc = {in[tid * 4 + 0], in[tid * 4 + 1]};
  const f2 a = {in[tid * 4 + 2], in[tid * 4 + 3]};
  const f2 c = {0.25f, -0.5f};
  f16v m = {0};
  h8 x = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)        // dest == src1, lo result reads src1.hi
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(c));
    else if (MODE == 1) { // same arithmetic, separate destination
      f2 t;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(t) : "v"(a), "v"(acc), "v"(c));
      acc = t;
    } else {              // dest == src1, lo AND hi read src1.lo (op_sel_hi src1 = 0)
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(c));
    }
    if ((i & 3) == 0) m = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, m, 0, 0, 0);   // matrix pipe busy beside it
  }
  out[tid * 2 + 0] = acc.x + m[0] * 0.f;
  out[tid * 2 + 1] = acc.y;
}

static void ref(int mode, const float* in, float* o, int iters) {
  float lo = in[0], hi = in[1], a0 = in[2], a1 = in[3];
  for (int i = 0; i < iters; ++i) {
    float nlo, nhi;
    if (mode == 2) { nlo = fmaf(a0, lo, 0.25f); nhi = fmaf(a1, lo, -0.5f); }
    else { nlo = fmaf(a0, hi, 0.25f); nhi = fmaf(a1, hi, -0.5f); }
    lo = nlo; hi = nhi;
  }
  o[0] = lo; o[1] = hi;
}

int main() {
  const int N = 256 * 1024, iters = 7;
  std::vector<float> h(N * 4), o(N * 2);
  for (int i = 0; i < N * 4; ++i) h[i] = 0.3f + 0.001f * (float)((i * 2654435761u) % 997) / 997.f;
  float *d_in, *d_out;
  hipMalloc(&d_in, N * 16); hipMalloc(&d_out, N * 8);
  hipMemcpy(d_in, h.data(), N * 16, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    long bad = 0, bad_hi_lanes = 0;
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(N / 256), dim3(256), 0, 0, d_in, d_out, iters);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(N / 256), dim3(256), 0, 0, d_in, d_out, iters);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(N / 256), dim3(256), 0, 0, d_in, d_out, iters);
      hipMemcpy(o.data(), d_out, N * 8, hipMemcpyDeviceToHost);
      for (int t = 0; t < N; ++t) {
        float r[2]; ref(mode, &h[t * 4], r, iters);
        if (r[0] != o[t * 2] || r[1] != o[t * 2 + 1]) { ++bad; if ((t & 63) >= 16 && (t & 31) >= 16) ++bad_hi_lanes; }
      }
    }
    printf("mode %d: mismatching lanes %ld of %ld (in lanes 16-31/48-63: %ld)\n", mode, bad, (long)N * 20, bad_hi_lanes);
  }
  return 0;
}
