// serve_group.hip -- host side only: a burst of independent forward calls enqueued breadth-first (include/neupan_amd.h,
// npa_forward_batch_group).  Built on the public begin / iter / end entry points of c_api.hip: no kernel, no device state.
//
// Why: a forward call is a chain of 1 + 2K dependent launches on its stream.  Issued call by call, chain j of a burst starts
// when the host has enqueued the 21 j launches in front of it (~5 us each; ~20 us with four threads contending for the
// runtime's locks): the last of 20 chains starts ~1.8 ms after the first, and a burst of 20 batches of 256 scenes -- 6.5 ms of
// GPU work -- spends a quarter of its wall time filling up.  Breadth-first, every chain is running after 2 n launches.
#include "../../include/neupan_amd.h"

extern "C" int npa_forward_batch_group(int n, const npa_forward_call* calls, int flags) {
  if (n < 1 || !calls) return NPA_E_ARG;
  int kmax = 0;
  for (int c = 0; c < n; ++c) {
    if (!calls[c].h || calls[c].iter_num < 1) return NPA_E_ARG;
    for (int d = 0; d < c; ++d)
      if (calls[d].h == calls[c].h) return NPA_E_ARG;                 // (a handle plans one batch at a time)
    if (calls[c].iter_num > kmax) kmax = calls[c].iter_num;
  }
  int begun = 0, rc = NPA_OK;
  for (; begun < n && rc == NPA_OK; ++begun) {
    const npa_forward_call& a = calls[begun];
    rc = npa_forward_begin(a.h, a.batch, a.n_stride, a.nom_s, a.nom_u, a.ref_s, a.ref_us, a.points, a.velocities, a.n_points,
                           a.out_s, a.out_u, a.out_d, a.out_min_distance, a.out_iters, a.out_nrmp_points, a.workspace,
                           a.workspace_bytes, a.state, a.state_bytes, a.stream, flags);
    if (rc != NPA_OK) break;                                          // (calls[begun] itself did not begin)
  }
  // a call that runs as ONE launch (npa_forward_scene: opt-in scene kernel, all of the handle's iterations) needs no interleaving
  unsigned long long whole = 0;                                       // (bit c: call c went out as one launch; n <= 64 checked below)
  for (int c = 0; c < n && rc == NPA_OK; ++c) {
    if (c >= 64) break;
    const int r = npa_forward_scene(calls[c].h, calls[c].iter_num);
    if (r < 0) rc = r;
    else if (r == 1) whole |= 1ull << c;
  }
  for (int k = 0; k < kmax && rc == NPA_OK; ++k)
    for (int c = 0; c < n && rc == NPA_OK; ++c)
      if (k < calls[c].iter_num && !(c < 64 && (whole >> c & 1ull))) rc = npa_forward_iter(calls[c].h, k);
  for (int c = 0; c < begun; ++c) {
    const int e = npa_forward_end(calls[c].h);
    if (rc == NPA_OK) rc = e;
  }
  return rc;
}
