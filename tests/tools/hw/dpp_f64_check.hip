// dpp_f64_check.hip -- what v_fmac_f64_dpp ... row_newbcast:N does on the device the kernels run on (gfx950), checked against the
// semantics nrmp_qp_device.h relies on:  D[lane] = fma(S0[lane N of lane's own 16-lane row], S1[lane], D[lane])  in every lane,
// the source read before the destination is written when they are the same register (the substitution chains), the neg modifier
// on src1 exact.  Build + run:  hipcc --offload-arch=gfx950 -O2 tests/tools/hw/dpp_f64_check.hip -o tests/tools/hw/_dpp_f64_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

template <int N>
__device__ void one(double* out, const double* in, int lane) {
  double acc = in[lane], src = in[64 + lane], mul = in[128 + lane];
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
  out[N * 64 + lane] = acc;
  double x = in[64 + lane];
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x) : "v"(mul), "n"(N));
  out[(16 + N) * 64 + lane] = x;
}
__global__ void k(double* out, const double* in) {
  const int lane = threadIdx.x;
  one<0>(out, in, lane); one<1>(out, in, lane); one<2>(out, in, lane); one<3>(out, in, lane);
  one<4>(out, in, lane); one<5>(out, in, lane); one<6>(out, in, lane); one<7>(out, in, lane);
  one<8>(out, in, lane); one<9>(out, in, lane); one<10>(out, in, lane); one<11>(out, in, lane);
  one<12>(out, in, lane); one<13>(out, in, lane); one<14>(out, in, lane); one<15>(out, in, lane);
}
int main() {
  double h[192], *din, *dout, ho[32 * 64];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < 192; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 11) / 9007199254740992.0 * 4.0 - 2.0; }
  hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof ho);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din);
  if (hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: launch\n"); return 2; }
  int bad = 0;
  for (int n = 0; n < 16; ++n)
    for (int l = 0; l < 64; ++l) {
      const int sl = (l & ~15) + n;
      const double want = fma(h[64 + sl], -h[128 + l], h[l]), wantx = fma(h[64 + sl], -h[128 + l], h[64 + l]);
      if (memcmp(&want, &ho[n * 64 + l], 8) || memcmp(&wantx, &ho[(16 + n) * 64 + l], 8)) {
        if (bad++ < 8) printf("N=%d lane %d: got %.17g / %.17g want %.17g / %.17g\n", n, l, ho[n * 64 + l], ho[(16 + n) * 64 + l], want, wantx);
      }
    }
  printf(bad ? "FAIL: %d mismatches\n" : "OK: v_fmac_f64_dpp row_newbcast:0..15, 64 lanes, separate and in-place source: bitwise fma(src[row lane N], -mul, acc)%.0d\n", bad);
  return bad ? 1 : 0;
}
