// serve_group.hip -- host side only: a burst of independent forward calls enqueued breadth-first (include/neupan_amd.h,
// npa_forward_batch_group).  Built on the public begin / iter / end entry points of c_api.hip: no kernel, no device state.
//
// Why: a forward call is a chain of 1 + 2K dependent launches on its stream.  Issued call by call, chain j of a burst starts
// when the host has enqueued the 21 j launches in front of it (~5 us each; ~20 us with four threads contending for the
// runtime's locks): the last of 20 chains starts ~1.8 ms after the first, and a burst of 20 batches of 256 scenes -- 6.5 ms of
// GPU work -- spends a quarter of its wall time filling up.  Breadth-first, every chain is running after 2 n launches.
//
// Merged launches (round 5).  Calls of a group that share ONE stream, a batch size and a configuration run every stage as one
// launch over all their scenes (c_api.hip: npa_group_mergeable / _begin_merged / _iter_merged; kernels select_geo_group_kernel,
// nrmp_qp_group_kernel): the same kernels' statements, the same results bitwise, 1 + 2K launches for the whole group.  A launch
// is a barrier over its scenes; with G x 256 scenes behind it the wave slots its stragglers leave idle are refilled from the
// same launch, and the group needs one hardware queue instead of G.  Calls on different streams keep the breadth-first order.
#include "../../include/neupan_amd.h"

extern "C" int npa_group_mergeable(int n, const npa_forward_call* calls);
extern "C" int npa_group_begin_merged(int n, const npa_forward_call* calls, int flags, int* begun);
extern "C" int npa_group_iter_merged(int n, const npa_forward_call* calls, int k);

extern "C" int npa_forward_batch_group(int n, const npa_forward_call* calls, int flags) {
  if (n < 1 || !calls) return NPA_E_ARG;
  int kmax = 0;
  for (int c = 0; c < n; ++c) {
    if (!calls[c].h || calls[c].iter_num < 1) return NPA_E_ARG;
    for (int d = 0; d < c; ++d)
      if (calls[d].h == calls[c].h) return NPA_E_ARG;                 // (a handle plans one batch at a time)
    if (calls[c].iter_num > kmax) kmax = calls[c].iter_num;
  }
  // merged: runs of up to NPA_GROUP_MAX (8) consecutive calls, each run one chain of merged launches; the runs themselves are
  // interleaved breadth-first like single calls
  {
    int lo = 0, nrun = 0, all = 1;
    int run_lo[64], run_n[64];
    // balanced runs first (ten calls on one stream: 5 + 5, not 8 + 2) ...
    {
      const int want = (n + 7) / 8, base = n / want, extra = n % want;
      for (int r = 0; r < want && want <= 64 && all; ++r) {
        const int len = base + (r < extra ? 1 : 0);
        if (len < 2 || !npa_group_mergeable(len, calls + lo)) { all = 0; break; }
        run_lo[nrun] = lo; run_n[nrun] = len; ++nrun;
        lo += len;
      }
    }
    // ... else the longest mergeable prefixes (calls grouped by stream)
    if (!all) {
      lo = 0; nrun = 0; all = 1;
      while (lo < n && nrun < 64) {
        int len = n - lo < 8 ? n - lo : 8;
        while (len >= 2 && !npa_group_mergeable(len, calls + lo)) --len;
        if (len < 2) { all = 0; break; }
        run_lo[nrun] = lo; run_n[nrun] = len; ++nrun;
        lo += len;
        if (n - lo == 1) { all = 0; break; }                          // (a single call left over: keep the whole group call by call)
      }
    }
    if (all && lo == n && nrun >= 1) {
      int rc = NPA_OK, begun_total = 0;
      int begun_run[64];
      for (int r = 0; r < nrun; ++r) begun_run[r] = 0;
      for (int r = 0; r < nrun && rc == NPA_OK; ++r) {
        rc = npa_group_begin_merged(run_n[r], calls + run_lo[r], flags, &begun_run[r]);
        begun_total += begun_run[r];
      }
      for (int k = 0; k < kmax && rc == NPA_OK; ++k)
        for (int r = 0; r < nrun && rc == NPA_OK; ++r)
          if (k < calls[run_lo[r]].iter_num) rc = npa_group_iter_merged(run_n[r], calls + run_lo[r], k);
      for (int r = 0; r < nrun; ++r)
        for (int c = 0; c < begun_run[r]; ++c) {
          const int e = npa_forward_end(calls[run_lo[r] + c].h);
          if (rc == NPA_OK) rc = e;
        }
      (void)begun_total;
      return rc;
    }
  }
  int begun = 0, rc = NPA_OK;
  for (; begun < n && rc == NPA_OK; ++begun) {
    const npa_forward_call& a = calls[begun];
    rc = npa_forward_begin(a.h, a.batch, a.n_stride, a.nom_s, a.nom_u, a.ref_s, a.ref_us, a.points, a.velocities, a.n_points,
                           a.out_s, a.out_u, a.out_d, a.out_min_distance, a.out_iters, a.out_nrmp_points, a.workspace,
                           a.workspace_bytes, a.state, a.state_bytes, a.stream, flags);
    if (rc != NPA_OK) break;                                          // (calls[begun] itself did not begin)
  }
  for (int k = 0; k < kmax && rc == NPA_OK; ++k)
    for (int c = 0; c < n && rc == NPA_OK; ++c)
      if (k < calls[c].iter_num) rc = npa_forward_iter(calls[c].h, k);
  for (int c = 0; c < begun; ++c) {
    const int e = npa_forward_end(calls[c].h);
    if (rc == NPA_OK) rc = e;
  }
  return rc;
}
