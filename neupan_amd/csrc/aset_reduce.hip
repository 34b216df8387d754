// aset_reduce.hip -- debug kernel around aset_reduce.h (branch qp-active-set): NOT part of the product path, driven by
// tests/test_aset_reduce.py against the numpy statements of tests/tools/qp_active_set_study.py.
#include "aset_reduce.h"
#include <stdint.h>

namespace {

// one wave per system.  K [B][NU][NU], r / tie / bnd [B][NU] (tie, bnd in {0, +1, -1}), acc / spd [2]; outputs alike + head
// (int), offvec, anchored (int); with res [B][NU]: the signed multipliers of the tie row into each variable and of the anchoring
// member's speed row.  (u_cur = 0: the absolute form the test states.)
template <int NU>
__global__ __launch_bounds__(64) void aset_reduce_test_kernel(const double* K, const double* r, const double* tie, const double* bnd,
                                                               const double* acc, const double* spd, double* outK, double* outr,
                                                               int* outhead, double* outoff, int* outanch, const double* res,
                                                               double* outlt, double* outlb) {
  __shared__ double ov[NU], slotv[NU], Mt[NU * (NU + 1)];
  __shared__ int winner[NU];
  const aset::Scratch S{ov, slotv, Mt, winner, NU + 1};
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool live = lane < NU;
  double arow[NU];
#pragma unroll
  for (int c = 0; c < NU; ++c) arow[c] = live ? K[((size_t)b * NU + lane) * NU + c] : 0.0;
  const double rr = live ? r[(size_t)b * NU + lane] : 0.0;
  const double t = live ? tie[(size_t)b * NU + lane] : 0.0, bd = live ? bnd[(size_t)b * NU + lane] : 0.0;
  aset::Lane L;
  double adj;
  aset::reduce_matrix<NU>(arow, t * acc[lane & 1], bd != 0.0, bd * spd[lane & 1], 0.0, lane, S, L, adj);
  const double rred = aset::reduce_rhs<NU>(rr - adj, lane, L);
  if (res) {
    double vt, bb, rs;
    aset::multipliers<NU>(live ? res[(size_t)b * NU + lane] : 0.0, t != 0.0, lane, L, vt, bb, rs);
    if (live) { outlt[(size_t)b * NU + lane] = t * vt; outlb[(size_t)b * NU + lane] = bd * bb; }
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < NU; ++c) outK[((size_t)b * NU + lane) * NU + c] = arow[c];
    outr[(size_t)b * NU + lane] = rred;
    outhead[(size_t)b * NU + lane] = L.head;
    outoff[(size_t)b * NU + lane] = ov[lane];
    outanch[(size_t)b * NU + lane] = L.anchored ? 1 : 0;
  }
}

}  // namespace

extern "C" int npa_dbg_aset_reduce(int batch, int nu, const double* K, const double* r, const double* tie, const double* bnd,
                                   const double* acc, const double* spd, double* outK, double* outr, int* outhead, double* outoff,
                                   int* outanch, const double* res, double* outlt, double* outlb, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nu == 20) aset_reduce_test_kernel<20><<<batch, 64, 0, st>>>(K, r, tie, bnd, acc, spd, outK, outr, outhead, outoff, outanch, res, outlt, outlb);
  else if (nu == 40) aset_reduce_test_kernel<40><<<batch, 64, 0, st>>>(K, r, tie, bnd, acc, spd, outK, outr, outhead, outoff, outanch, res, outlt, outlb);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
