// select_scene.h -- the geometric selection for ALL slices of one scene on ONE wave, with shared passes.  gfx950 only.
// Included by pan_scene.hip (after dune_device.h).  OPT-IN (NPA_SELECT_SCENE=1), see DESIGN.md 3.4b / 7.
//
// Same contract as select_geo_kernel (dune_device.h; replaces generate_point_flow + DUNE.forward + the top-M gather,
// pan.py:150-212, dune.py:58-127, nrmp.py:254-259): for every slice t the M rows (mu, lam, point, distance) of the M points
// with the smallest exact network distance, ties by index.  The rows come from the same exact fp32 encoder
// (point_features_stream) on the same points and are ranked on the same exact (distance, index) keys, so they are BITWISE
// those of select_geo_kernel whenever the nominated candidates contain the true nearest M -- which the margin guarantees
// for both kernels alike; how the candidates are nominated is free, and differs:
//
//   select_geo_kernel: one wave per slice; keys of the slice into LDS, bound, window over the stored keys, encoder on one tile.
//   here:              one wave per scene; pass A reads the points ONCE and computes the keys of all slices per point (lane
//                      minima per slice, no key array), the bounds of all slices, pass B recomputes the keys and compacts the
//                      candidates per slice, the encoder runs over the candidates of several slices packed into full tiles
//                      (the frame is a per-lane operand), rank + emit per slice.
// A slice the fast path cannot take -- more candidates than the ranking holds, fewer points than M, a distrusted margin, a
// wave that owes an audit tile, debug statistics -- is handed to the per-slice body (select_geo_body.inc) afterwards.
#pragma once

#define SCN_CAP SEL_CAP        // candidates of a slice the fast path ranks (lane q speaks for candidate q)
#define SCN_CHUNK 96           // entries encoded per chunk: three 32-point tiles

// keys of four points (n0 + lane + 64 u) for one slice frame; RECT / polygon as in key_pass
template <int E, bool RECT>
__device__ __forceinline__ void scene_keys4(const DevParams& P, const float4 fr, const float (&gx)[4], const float (&gy)[4],
                                            const float (&vx)[4], const float (&vy)[4], bool has_vel, float tdt, int n0, int lane,
                                            int n_use, unsigned (&k)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float x = gx[u], y = gy[u];
    if (has_vel) {      // pan.py:182 in the reference's rounding order (the kept rows recompute it the same way)
      x = __fadd_rn(x, __fmul_rn(tdt, __fmul_rn(vx[u], P.dt32)));
      y = __fadd_rn(y, __fmul_rn(tdt, __fmul_rn(vy[u], P.dt32)));
    }
    const float dx = x - fr.z, dy = y - fr.w;
    const float p0x = fmaf(fr.x, dx, fr.y * dy), p0y = fmaf(fr.x, dy, -(fr.y * dx));
    const unsigned kk = __float_as_uint(geo_key_t<E, RECT>(P, p0x, p0y));
    k[u] = (n0 + lane + 64 * u) < n_use ? kk : 0xFFFFFFFFu;
  }
}

// Returns a bit mask (bit t) of the slices it did NOT finish: the caller runs the per-slice body on those.
template <int E, int TT, bool RECT>
__device__ __forceinline__ unsigned select_scene_fast(
    const DevParams& P, const float* __restrict__ wpack, int n_stride, const float* __restrict__ cur_s,
    const float* __restrict__ points, const float* __restrict__ vel, const int* __restrict__ n_points,
    float* __restrict__ mu_sorted, float* __restrict__ lam_sorted, float* __restrict__ pts_sorted,
    float* __restrict__ dist_sorted, int* __restrict__ count, const float* __restrict__ trig, unsigned* __restrict__ audit,
    float margin_scale, const int b, const int t_first, const unsigned skip_mask, const int lane) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NS = TT + 1;                          // slices 0 .. TT
  constexpr int ROW_W = E + 5;                        // mu[E], lam[2], point[2], distance
  // LDS: the head is the per-slice body's (vectors of the encoder, margins per band: it rewrites the same values when it runs
  // afterwards), the rest is this function's
  float* vec = smem;                                  // [11][32]
  float* w6 = vec + 11 * 32;                          // [8][32]
  float* b6 = w6 + 8 * 32;                            // [8]
  float* etab = b6 + 8;                               // [NPA_GEO_BANDS]
  float4* frames = reinterpret_cast<float4*>(etab + NPA_GEO_BANDS);           // [NS] (c, s, tx, ty)   (16-byte aligned: 704 floats in front)
  unsigned short* cand = reinterpret_cast<unsigned short*>(frames + NS);       // [NS][SCN_CAP] candidate indices
  float* rows = reinterpret_cast<float*>(cand + NS * SCN_CAP);                 // [SCN_CHUNK][ROW_W]
  unsigned* rkey = reinterpret_cast<unsigned*>(rows + SCN_CHUNK * ROW_W);      // [SCN_CHUNK][2]: (index, exact key)
  int* effc = reinterpret_cast<int*>(rkey + 2 * SCN_CHUNK);                    // [NS] candidates of a slice the fast path keeps (0: not its slice)
  int* offs = effc + NS;                                                       // [NS] first entry of a slice in the current chunk
  const int j = lane & 31, hf = lane >> 5;
  const int T = TT, M = P.M;
  int n_raw = n_points ? n_points[b] : n_stride;
  n_raw = n_raw < 0 ? 0 : (n_raw > n_stride ? n_stride : n_raw);
  const int n_use = n_raw < P.dune_max_num ? n_raw : P.dune_max_num;
  const unsigned all_mask = ((1u << NS) - 1u) & ~((1u << t_first) - 1u) & ~skip_mask;
  // what the fast path does not do: slices with fewer points than rows (the padding rule), a margin that was violated
  // (everything is a candidate from then on)
  unsigned aud_viol = 0;
  if (audit) aud_viol = __hip_atomic_load(audit + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_use < M || M > SCN_CAP || __builtin_amdgcn_readfirstlane((int)aud_viol) != 0) return all_mask;

  // ---- preamble, once per scene: encoder vectors and margins to LDS, the frames of all slices ---------------------------
  constexpr int NVEC = 11 * 32 + 8 * 32 + 8, NVT = (NVEC + 63) / 64, NGT = (NPA_GEO_BANDS + 63) / 64;
  float pre_v[NVT], pre_g[NGT];
#pragma unroll
  for (int i = 0; i < NVT; ++i) { const int k = lane + 64 * i; pre_v[i] = wpack[WP_VEC + (k < NVEC ? k : NVEC - 1)]; }
#pragma unroll
  for (int i = 0; i < NGT; ++i) { const int k = lane + 64 * i; pre_g[i] = wpack[WP_GEO + (k < NPA_GEO_BANDS ? k : NPA_GEO_BANDS - 1)]; }
  const float w1 = wpack[WP_W1 + lane];
  {
    const int tt = lane < NS ? lane : 0;
    const float* s = cur_s + (size_t)b * 3 * (T + 1);
    const float4 fr = make_float4(trig[((size_t)b * (T + 1) + tt) * 2], trig[((size_t)b * (T + 1) + tt) * 2 + 1], s[tt], s[(T + 1) + tt]);
    if (lane < NS) frames[lane] = fr;
  }
#pragma unroll
  for (int i = 0; i < NVT; ++i) { const int k = lane + 64 * i; if (k < NVEC) smem[k] = pre_v[i]; }
#pragma unroll
  for (int i = 0; i < NGT; ++i) { const int k = lane + 64 * i; if (k < NPA_GEO_BANDS) etab[k] = pre_g[i] * margin_scale; }
  const float* px_row = points + (size_t)b * 2 * n_stride;
  const float* py_row = px_row + n_stride;
  const float* vx_row = vel ? vel + (size_t)b * 2 * n_stride : nullptr;
  const float* vy_row = vel ? vx_row + n_stride : nullptr;
  const bool has_vel = vel != nullptr, decim = n_use < n_raw;
  const unsigned far_thr = __float_as_uint(P.geo_far);
  const float* wls = wpack + WP_WLS;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px_row), 0, n_raw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(py_row), 0, n_raw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_vel ? vx_row : px_row), 0, n_raw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_vel ? vy_row : py_row), 0, n_raw * 4, 0x00020000);
  WSYNC();
  const int n_pad = (n_use + SEL2_TRIP - 1) & ~(SEL2_TRIP - 1);
  // four points per lane and trip: n = n0 + lane + 64 u (reads behind n_raw return 0: their keys are discarded)
  auto load4 = [&](int n0, float (&gx)[4], float (&gy)[4], float (&vx)[4], float (&vy)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + lane + 64 * u;
      unsigned off = (unsigned)n * 4u;
      if (decim) off = (unsigned)src_index(n < n_use ? n : n_use - 1, n_raw, n_use) * 4u;
      gx[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      gy[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
      vx[u] = has_vel ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvx, off, 0, 0)) : 0.f;
      vy[u] = has_vel ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvy, off, 0, 0)) : 0.f;
    }
  };

  // ---- pass A: the smallest key of every lane, per slice --------------------------------------------------------------
  unsigned lmin[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) lmin[t] = 0xFFFFFFFFu;
  for (int n0 = 0; n0 < n_pad; n0 += SEL2_TRIP) {
    float gx[4], gy[4], vx[4], vy[4];
    load4(n0, gx, gy, vx, vy);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      if (!(all_mask >> t & 1u)) continue;
      unsigned k[4];
      scene_keys4<E, RECT>(P, frames[t], gx, gy, vx, vy, has_vel, (float)t, n0, lane, n_use, k);
      lmin[t] = min(min(lmin[t], k[0]), min(k[1], min(k[2], k[3])));
    }
  }

  // ---- the threshold of every slice (select_geo_body.inc's rule: the M-th smallest lane minimum bounds the M-th smallest key
  // from above; a point of band b is a candidate iff g <= U + margin[b], one threshold g* for the window) -----------------
  unsigned thr[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    thr[t] = 0xFFFFFFFEu;
    if (!(all_mask >> t & 1u)) continue;
    unsigned bound = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned trial = bound | ((1u << bit) - 1u);
      if (__popcll(__ballot(lmin[t] <= trial)) < M) bound |= 1u << bit;
    }
    float hi = 0.f;
    if (lmin[t] <= bound) {
      const float g = __uint_as_float(lmin[t]);
      hi = (lmin[t] >= far_thr) ? __builtin_inff() : g + etab[npa_geo_band(g)];
    }
    const float U = wave_max_f32(hi);
    if (U < 3.0e38f) {
      float gs = 0.f;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int bnd = lane + 64 * rep;
        if (bnd < NPA_GEO_BANDS) {
          const float lo_b = __uint_as_float((unsigned)(bnd + (0x3E800000u >> 20)) << 20) - 0.25f;
          const float hi_b = bnd == NPA_GEO_BANDS - 1 ? P.geo_far : __uint_as_float((unsigned)(bnd + 1 + (0x3E800000u >> 20)) << 20) - 0.25f;
          const float reach = U + etab[bnd];
          if (lo_b < P.geo_far && lo_b <= reach) gs = fmaxf(gs, fminf(hi_b, reach));
        }
      }
      gs = wave_max_f32(gs);
      thr[t] = gs < 3.0e38f ? __float_as_uint(gs) : 0xFFFFFFFEu;
    }
  }

  // ---- pass B: the candidates of every slice, compacted into cand[t][] -------------------------------------------------
  int cnt[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) cnt[t] = 0;
  for (int n0 = 0; n0 < n_pad; n0 += SEL2_TRIP) {
    float gx[4], gy[4], vx[4], vy[4];
    load4(n0, gx, gy, vx, vy);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      if (!(all_mask >> t & 1u)) continue;
      unsigned k[4];
      scene_keys4<E, RECT>(P, frames[t], gx, gy, vx, vy, has_vel, (float)t, n0, lane, n_use, k);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool hit = k[u] <= thr[t] || (k[u] - far_thr) < (0xFFFFFFFFu - far_thr);
        const unsigned long long bal = __ballot(hit);
        const int pos = cnt[t] + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
        if (hit && pos < SCN_CAP) cand[t * SCN_CAP + pos] = (unsigned short)(n0 + lane + 64 * u);
        cnt[t] += __popcll(bal);
      }
    }
  }
  // lane t takes slice t's count; what the fast path keeps (effc) and the offsets of a chunk live in LDS from here on
  int my_cnt = 0;
#pragma unroll
  for (int t = 0; t < NS; ++t) my_cnt = lane == t ? cnt[t] : my_cnt;
  const bool in_mask = lane < NS && (all_mask >> (lane & 31) & 1u);
  const bool too = in_mask && (my_cnt > SCN_CAP || my_cnt < M);
  const unsigned left = (unsigned)__ballot(too);       // slices for the per-slice body (NS <= 32)
  if (lane < NS) effc[lane] = (in_mask && !too) ? my_cnt : 0;
  WSYNC();

  // ---- encoder over the candidates, chunk by chunk: consecutive slices whose candidates fill up to three tiles ----------
  int viol = 0;
  float worst = 0.f;
  int ta = t_first;
  while (ta < NS) {
    int tb = ta, tot = 0;                               // the chunk [ta, tb) and its entries (wave-uniform)
    while (tb < NS) {
      const int c = __builtin_amdgcn_readfirstlane(effc[tb]);
      if (tot + c > SCN_CHUNK) break;
      if (lane == 0) offs[tb] = tot;
      tot += c;
      ++tb;
    }
    WSYNC();
    for (int q0 = 0; q0 < tot; q0 += 32) {
      const int e = q0 + j, ec = e < tot ? e : tot - 1;
      int te = ta, off_te = 0;                          // the slice of entry ec: the last one with entries whose offset is <= ec
      for (int t = ta; t < tb; ++t) {
        const int c = effc[t], o = offs[t];
        const bool hitt = c > 0 && ec >= o;
        te = hitt ? t : te;
        off_te = hitt ? o : off_te;
      }
      const int idx = (int)cand[te * SCN_CAP + (ec - off_te)];
      // the frame of that slice, per lane (load_frame's arithmetic)
      const float4 fr = frames[te];
      SliceFrame F;
      F.c = fr.x; F.s = fr.y; F.tx = fr.z; F.ty = fr.w; F.tstep = (float)te;
#pragma unroll
      for (int ee = 0; ee < E; ++ee) {
        F.rg[0][ee] = fmaf(-F.c, P.G[ee][0], __fmul_rn(F.s, P.G[ee][1]));
        F.rg[1][ee] = fmaf(-F.s, P.G[ee][0], -__fmul_rn(F.c, P.G[ee][1]));
      }
      float mu[E], gxx, gyy, lx, ly, dist, p0x, p0y;
      point_features_stream<E, false>(P, F, w1, wls, vec, w6, b6, px_row, py_row, vx_row, vy_row, src_index(idx, n_raw, n_use),
                                      lane, mu, gxx, gyy, lx, ly, dist, p0x, p0y);
      if (audit) {
        // the bound the candidates rest on, checked on every exactly encoded point: |exact - g| <= margin[band(g)]
        const float g = geo_key_t<E, RECT>(P, p0x, p0y);
        const float ex = fabsf(dist - g) - etab[npa_geo_band(g)];
        const bool bad = hf == 0 && e < tot && g < P.geo_far && ex > 0.f;
        viol += bad ? 1 : 0;
        worst = bad ? fmaxf(worst, ex) : worst;
      }
      if (hf == 0 && e < tot) {
        float* r = rows + e * ROW_W;
#pragma unroll
        for (int ee = 0; ee < E; ++ee) r[ee] = mu[ee];
        r[E] = lx; r[E + 1] = ly; r[E + 2] = gxx; r[E + 3] = gyy; r[E + 4] = dist;
        rkey[2 * e] = (unsigned)idx; rkey[2 * e + 1] = ordered_key(dist);
      }
    }
    WSYNC();
    // rank + emit, slice by slice: lane q speaks for candidate q of the slice
    for (int t = ta; t < tb; ++t) {
      const int nc = __builtin_amdgcn_readfirstlane(effc[t]);
      if (nc == 0) continue;
      const int o0 = __builtin_amdgcn_readfirstlane(offs[t]);
      const bool have = lane < nc;
      const int eq = o0 + (have ? lane : 0);
      const unsigned long long kx = have ? (((unsigned long long)rkey[2 * eq + 1] << 32) | rkey[2 * eq]) : ~0ull;
      int rank = 0;
      for (int i = 0; i < nc; ++i) rank += readlane_u64(kx, i) < kx ? 1 : 0;
      if (have && rank < M) {
        const float* r = rows + eq * ROW_W;
        const size_t o = ((size_t)b * (T + 1) + t) * M + rank;
#pragma unroll
        for (int ee = 0; ee < E; ++ee) mu_sorted[o * E + ee] = r[ee];
        lam_sorted[o * 2 + 0] = r[E]; lam_sorted[o * 2 + 1] = r[E + 1];
        pts_sorted[o * 2 + 0] = r[E + 2]; pts_sorted[o * 2 + 1] = r[E + 3];
        dist_sorted[o] = r[E + 4];
      }
      if (lane == 0) count[(size_t)b * (T + 1) + t] = M;
    }
    WSYNC();
    ta = tb > ta ? tb : ta + 1;                         // (tb == ta cannot happen: one slice holds at most SCN_CAP <= SCN_CHUNK entries)
  }
  if (audit) {
    const unsigned long long vb = __ballot(viol > 0);
    if (vb != 0ull) {                          // rare: the counters are touched only then
      int v = viol;
      float wv = worst;
      for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); wv = fmaxf(wv, __shfl_xor(wv, o, 64)); }
      if (lane == 0) {
        atomicAdd(audit + 2, (unsigned)v); atomicMax(audit + 3, __float_as_uint(wv));
        unsigned* hp = *reinterpret_cast<unsigned* const*>(audit + 6);
        if (hp) __hip_atomic_fetch_add(hp, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  return left & all_mask;
}
