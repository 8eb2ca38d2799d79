// Grouping ops for gfx950: ball query, group (+grad), selection-sort top-k, exact batched kNN.
// Behaviour contract: reference tf_ops/grouping/tf_grouping_g.cu (:3-123) and
// utils/nearest_neighbors/knn_.cxx:72-135; restated in oracle/.
//
// Shared shape of the two search kernels (ball query, kNN): a workgroup of 4 waves works on QPB
// queries of ONE cloud; the cloud streams through LDS in tiles (coalesced flat loads of the AoS
// (n,3) array, stored as x|y|z planes so that lane l reads point base+l conflict-free); ONE WAVE
// OWNS ONE QUERY at a time and looks at 64 points per step, so "first nsample in index order" and
// "sorted by (distance, index)" both fall out of lane order + ballot/popcount, never from atomics.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "common.hpp"

namespace pasnl {

constexpr int SEARCH_TILE = 2048;  // points per LDS tile (24 KiB)
constexpr int SEARCH_WAVES = 4;

__device__ __forceinline__ void stage_tile(const float* __restrict__ cloud, int n, int base, int cnt, float* sx, float* sy,
                                           float* sz) {
  // flat coalesced copy of `cnt` points starting at point `base`
  const float* src = cloud + (size_t)base * 3;
  for (int f = threadIdx.x; f < cnt * 3; f += SEARCH_WAVES * 64) {
    float v = src[f];
    int p = f / 3, c = f - p * 3;
    (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Ball query.  `thr2` is the smallest float t with sqrtf(t) >= radius (computed on the host), so
// "max(sqrtf(d2),1e-20f) < radius" == "d2 < thr2" bit-for-bit without a per-pair square root.
// QW queries per wave, processed one after another inside each tile; per-query state that must
// survive a tile boundary (count, first hit) lives in registers, the hit list in LDS.
// ---------------------------------------------------------------------------------------------
template <int QW>
__global__ __launch_bounds__(SEARCH_WAVES * 64) void ball_query_kernel(int n, int m, float thr2, int nsample,
                                                                      const float* __restrict__ xyz1,
                                                                      const float* __restrict__ xyz2,
                                                                      int* __restrict__ idx, int* __restrict__ pts_cnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);
  float* sy = sx + SEARCH_TILE;
  float* sz = sy + SEARCH_TILE;
  int* hits = reinterpret_cast<int*>(sz + SEARCH_TILE);  // [SEARCH_WAVES*QW][nsample]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.y;
  const float* cloud = xyz1 + (size_t)bi * n * 3;
  const int q0 = (blockIdx.x * SEARCH_WAVES + wave) * QW;

  float qx[QW], qy[QW], qz[QW];
  int cnt[QW], first[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    int j = q0 + q;
    bool ok = j < m;
    const float* p = xyz2 + ((size_t)bi * m + (ok ? j : 0)) * 3;
    qx[q] = p[0]; qy[q] = p[1]; qz[q] = p[2];
    cnt[q] = ok ? 0 : nsample;  // out-of-range queries are born finished
    first[q] = 0;
  }

  for (int base = 0; base < n; base += SEARCH_TILE) {
    int tcnt = min(SEARCH_TILE, n - base);
    __syncthreads();
    stage_tile(cloud, n, base, tcnt, sx, sy, sz);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      int c = cnt[q];
      if (c >= nsample) continue;  // wave-uniform
      int* hq = hits + (size_t)(wave * QW + q) * nsample;
      for (int it = 0; it < tcnt; it += 64) {
        int p = it + lane;
        bool in = p < tcnt;
        float d2 = dist2(qx[q], qy[q], qz[q], in ? sx[p] : 0.f, in ? sy[p] : 0.f, in ? sz[p] : 0.f);
        bool hit = in && (d2 < thr2);
        unsigned long long mask = __ballot(hit);
        if (mask) {
          if (c == 0) first[q] = base + it + (int)__builtin_ctzll(mask);
          int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          int slot = c + rank;
          if (hit && slot < nsample) hq[slot] = base + it + lane;
          c += (int)__builtin_popcountll(mask);
          if (c >= nsample) { c = nsample; break; }
        }
      }
      cnt[q] = c;
    }
  }
  // emit: hits in index order, then the first hit as padding; zero-hit rows -> 0
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    int j = q0 + q;
    if (j >= m) continue;
    const int* hq = hits + (size_t)(wave * QW + q) * nsample;
    int* o = idx + ((size_t)bi * m + j) * nsample;
    int c = cnt[q], pad = c > 0 ? first[q] : 0;
    for (int s = lane; s < nsample; s += 64) o[s] = s < c ? hq[s] : pad;
    if (lane == 0) pts_cnt[(size_t)bi * m + j] = c;
  }
}

// The grid-pruned kernel for LDS-sized clouds (every in-model use, the north-star shape) is ball_grid.hip.
int ball_grid_launch(int b, int n, int m, float radius, float thr2, int nsample, const float* xyz1, const float* xyz2, int* idx,
                     int* pts_cnt, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Exact kNN, K <= 64*SLOTS.  The running result of a query is a list sorted ascending by
// (d, index) spread over the wave: lane l, slot s holds rank s*64 + l.  Points are visited in
// ascending index, so a candidate enters iff d < tau (tau = current K-th distance): an equal
// distance with a larger index never displaces.  Insertion = ballot-count of list entries <= d
// (its rank), DPP shift of the tail by one lane, write.  Output order is therefore exactly
// (distance, index) ascending.
// ---------------------------------------------------------------------------------------------
template <int SLOTS>
struct KnnList {
  float d[SLOTS];
  int i[SLOTS];
};

template <int SLOTS>
__device__ __forceinline__ void knn_insert(KnnList<SLOTS>& L, float cd, int ck, int lane) {
  // rank of the candidate = number of entries with d <= cd (all of those have smaller indices)
  int pos = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) pos += (int)__builtin_popcountll(__ballot(L.d[s] <= cd));
  // shift entries at rank >= pos up by one; entry 63 of slot s carries into entry 0 of slot s+1
  float carry_d = 0.f;
  int carry_i = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    float top_d = readlane_f(L.d[s], 63);
    int top_i = __builtin_amdgcn_readlane(L.i[s], 63);
    float sd = wave_shr1_f(L.d[s]);
    int si = wave_shr1_i(L.i[s]);
    if (lane == 0) { sd = carry_d; si = carry_i; }
    int r = s * 64 + lane;
    if (r > pos) { L.d[s] = sd; L.i[s] = si; }
    if (r == pos) { L.d[s] = cd; L.i[s] = ck; }
    carry_d = top_d;
    carry_i = top_i;
  }
}

template <int SLOTS, int QW, typename IdxT>
__global__ __launch_bounds__(SEARCH_WAVES * 64) void knn_kernel(int n, int m, int k, const float* __restrict__ support,
                                                               const float* __restrict__ queries, IdxT* __restrict__ idx,
                                                               float* __restrict__ dist_out, const KnnTieFlags flags) {
  __shared__ float sx[SEARCH_TILE], sy[SEARCH_TILE], sz[SEARCH_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.y;
  const float* cloud = support + (size_t)bi * n * 3;
  const int q0 = (blockIdx.x * SEARCH_WAVES + wave) * QW;

  float qx[QW], qy[QW], qz[QW], tau[QW];
  // (flagging for pasnl_knn_batch_ref) the value of tau at which a point that is NOT in the list was last seen at exactly the
  // K-th distance -- a candidate refused with d == tau, or the old K-th entry pushed out by an insertion that left an equal
  // distance in its place.  tau only falls: the K-list ends on a tie iff this equals the final tau.
  float tie_tau[QW];
  const bool flagging = flags.nflag != nullptr;
  KnnList<SLOTS> L[QW];
  const int ks = (k - 1) >> 6, kl = (k - 1) & 63;  // slot / lane of the K-th entry
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    int j = min(q0 + q, m - 1);
    const float* p = queries + ((size_t)bi * m + j) * 3;
    qx[q] = p[0]; qy[q] = p[1]; qz[q] = p[2];
    tau[q] = INFINITY;
    tie_tau[q] = -1.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) { L[q].d[s] = INFINITY; L[q].i[s] = 0; }
  }

  for (int base = 0; base < n; base += SEARCH_TILE) {
    int tcnt = min(SEARCH_TILE, n - base);
    __syncthreads();
    stage_tile(cloud, n, base, tcnt, sx, sy, sz);
    __syncthreads();
    for (int it = 0; it < tcnt; it += 64) {
      int p = it + lane;
      bool in = p < tcnt;
      float x = in ? sx[p] : 0.f, y = in ? sy[p] : 0.f, z = in ? sz[p] : 0.f;
      float dq[QW];
      dist2_multi<QW>(qx, qy, qz, x, y, z, dq);
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        float d = dq[q];
        unsigned long long mask = __ballot(in && d < tau[q]);
        if (flagging && __ballot(in && d == tau[q]) != 0ull) tie_tau[q] = tau[q];
        while (mask) {
          int src = (int)__builtin_ctzll(mask);
          mask &= mask - 1;
          float cd = readlane_f(d, src);
          if (cd < tau[q]) {  // tau may have dropped since the ballot
            knn_insert<SLOTS>(L[q], cd, base + it + src, lane);
            float t = INFINITY;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
              if (s == ks) t = readlane_f(L[q].d[s], kl);
            if (t == tau[q]) tie_tau[q] = t;  // the entry that fell off the end had the distance the new K-th has
            tau[q] = t;
          } else if (cd == tau[q]) {
            tie_tau[q] = cd;
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    int j = q0 + q;
    if (j >= m) continue;
    size_t o = ((size_t)bi * m + j) * k;
    bool tie = flagging && tie_tau[q] == tau[q];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      int r = s * 64 + lane;
      if (r < k) {
        idx[o + r] = (IdxT)L[q].i[s];
        if (dist_out) dist_out[o + r] = L[q].d[s];
      }
      if (flagging) {  // two equal distances next to each other inside the list (rank r - 1 sits in the lane below, or in lane 63 of the slot below)
        float prev = wave_shr1_f(L[q].d[s]);
        if (lane == 0) prev = s > 0 ? readlane_f(L[q].d[s > 0 ? s - 1 : 0], 63) : -1.f;
        tie = tie || __ballot(r >= 1 && r < k && L[q].d[s] == prev) != 0ull;
      }
    }
    if (tie) knn_flag_query(flags, bi, m, j, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// Exact kNN, two-pass selection (K <= 64).  Serial insertion costs ~K(1+ln(N/K)) dependent steps per query;
// this kernel has none on the common path:
//   pass 1  every lane keeps the R smallest distances it sees (R = 1 for K <= 32, 2 for K <= 64).  The
//           64*R lane minima are distinct points, so U = their K-th smallest value is an UPPER BOUND of the
//           K-th nearest distance (bitonic sort of the minima, 32-bit keys);
//   pass 2  the cloud streams through LDS again; points with d <= U (expected ~1.4 K of them) are appended,
//           in index order, to a per-query LDS buffer by ballot + prefix popcount;
//   sort    the candidates are sorted by the 64-bit key (distance bits << 32 | index) with an in-wave bitonic
//           network; the first K are the answer, already in canonical (distance, index) order.
// More than 128 candidates (heavy ties / duplicates) falls back to the insertion kernel's method in a third
// pass, so the result is exact for every input.  Distances are compared as their bit patterns (non-negative
// floats order like unsigned ints).
// ---------------------------------------------------------------------------------------------
constexpr int KNN2_CAP = 128;  // candidate buffer per query (keys)

// Diagnostic build only (-DPASNL_KNN_PROBE, tools/knn_probe.py): s_memtime marks between the phases, summed over waves
#ifdef PASNL_KNN_PROBE
__device__ unsigned long long knn_probe[8];
#define KNN_MARK(t) do { __builtin_amdgcn_sched_barrier(0); t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KNN_PROBE(...) __VA_ARGS__
#else
#define KNN_MARK(t)
#define KNN_PROBE(...)
#endif

template <int R, int QW, typename IdxT>
__global__ __launch_bounds__(SEARCH_WAVES * 64) void knn2_kernel(int n, int m, int k, const float* __restrict__ support,
                                                                const float* __restrict__ queries, IdxT* __restrict__ idx,
                                                                float* __restrict__ dist_out, const KnnTieFlags flags) {
  __shared__ float sx[SEARCH_TILE], sy[SEARCH_TILE], sz[SEARCH_TILE];
  __shared__ unsigned long long cand[SEARCH_WAVES * QW][KNN2_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.y;
  const float* cloud = support + (size_t)bi * n * 3;
  const int q0 = (blockIdx.x * SEARCH_WAVES + wave) * QW;
  constexpr uint32_t INF_BITS = 0x7f800000u;

  float qx[QW], qy[QW], qz[QW];
  uint32_t m1[QW], m2[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    int j = min(q0 + q, m - 1);
    const float* p = queries + ((size_t)bi * m + j) * 3;
    qx[q] = p[0]; qy[q] = p[1]; qz[q] = p[2];
    m1[q] = INF_BITS; m2[q] = INF_BITS;
  }
  KNN_PROBE(unsigned long long k0, k1, k2, k3, k4; KNN_MARK(k0);)
  // ---- pass 1: per-lane R smallest distances
  for (int base = 0; base < n; base += SEARCH_TILE) {
    int tcnt = min(SEARCH_TILE, n - base);
    __syncthreads();
    stage_tile(cloud, n, base, tcnt, sx, sy, sz);
    __syncthreads();
    for (int it = 0; it < tcnt; it += 64) {
      int p = it + lane;
      bool in = p < tcnt;
      const float x = sx[p], y = sy[p], z = sz[p];  // unconditional (p < SEARCH_TILE; stale words past tcnt are masked by `in` below):
                                                   // behind `in ? sx[p] : 0` every read was its own exec-masked branch
      float dq[QW];
      dist2_multi<QW>(qx, qy, qz, x, y, z, dq);
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        uint32_t di = in ? __float_as_uint(dq[q]) : INF_BITS;
        if (R == 2) m2[q] = min(m2[q], max(m1[q], di));
        m1[q] = min(m1[q], di);
      }
    }
  }
  KNN_MARK(k1);
  // ---- bound U = K-th smallest of the lane minima
  uint32_t U[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    uint32_t mv[R];
    mv[0] = m1[q];
    if (R == 2) mv[1] = m2[q];
    wave_bitonic_sort<R, uint32_t>(mv, lane);
    uint32_t u = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r == ((k - 1) >> 6)) u = (uint32_t)__builtin_amdgcn_readlane((int)mv[r], (k - 1) & 63);
    U[q] = u;
  }
  KNN_MARK(k2);
  // ---- pass 2: collect candidates d <= U in index order
  int cnt[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) cnt[q] = 0;
  for (int base = 0; base < n; base += SEARCH_TILE) {
    int tcnt = min(SEARCH_TILE, n - base);
    if (n > SEARCH_TILE) {  // single-tile clouds are still resident from pass 1
      __syncthreads();
      stage_tile(cloud, n, base, tcnt, sx, sy, sz);
      __syncthreads();
    }
    for (int it = 0; it < tcnt; it += 64) {
      int p = it + lane;
      bool in = p < tcnt;
      const float x = sx[p], y = sy[p], z = sz[p];  // unconditional (p < SEARCH_TILE; stale words past tcnt are masked by `in` below):
                                                   // behind `in ? sx[p] : 0` every read was its own exec-masked branch
      float dq[QW];
      dist2_multi<QW>(qx, qy, qz, x, y, z, dq);
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        uint32_t di = __float_as_uint(dq[q]);
        bool c = in && di <= U[q];
        unsigned long long mask = __ballot(c);
        if (mask) {
          int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          int slot = cnt[q] + rank;
          if (c && slot < KNN2_CAP) cand[wave * QW + q][slot] = ((unsigned long long)di << 32) | (uint32_t)(base + p);
          cnt[q] += (int)__builtin_popcountll(mask);
        }
      }
    }
  }
  KNN_MARK(k3);
  // ---- sort + emit; overflowing queries are flagged for the fallback pass
  bool overflow = false;
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int j = q0 + q;
    const int c = cnt[q];
    if (c > KNN2_CAP) { overflow = overflow || (j < m); continue; }
    if (j >= m) continue;
    const unsigned long long* cb = cand[wave * QW + q];
    const size_t o = ((size_t)bi * m + j) * k;
    if (c <= 64) {
      unsigned long long key[1];
      key[0] = lane < c ? cb[lane] : ~0ull;
      wave_bitonic_sort<1, unsigned long long>(key, lane);
      if (lane < k) {
        idx[o + lane] = (IdxT)(uint32_t)key[0];
        if (dist_out) dist_out[o + lane] = __uint_as_float((uint32_t)(key[0] >> 32));
      }
      // (every point at the K-th distance is among the candidates: they were collected with d <= U and U bounds it from above)
      if (flags.nflag && knn_sorted_has_tie(key[0], ~0ull, k, lane)) knn_flag_query(flags, bi, m, j, lane);
    } else {
      unsigned long long key[2];
      key[0] = cb[lane];
      key[1] = 64 + lane < c ? cb[64 + lane] : ~0ull;
      wave_bitonic_sort<2, unsigned long long>(key, lane);
      if (lane < k) {
        idx[o + lane] = (IdxT)(uint32_t)key[0];
        if (dist_out) dist_out[o + lane] = __uint_as_float((uint32_t)(key[0] >> 32));
      }
      if (flags.nflag && knn_sorted_has_tie(key[0], key[1], k, lane)) knn_flag_query(flags, bi, m, j, lane);
    }
  }
#ifdef PASNL_KNN_PROBE
  KNN_MARK(k4);
  if (lane == 0) {
    atomicAdd(&knn_probe[0], k1 - k0); atomicAdd(&knn_probe[1], k2 - k1); atomicAdd(&knn_probe[2], k3 - k2);
    atomicAdd(&knn_probe[3], k4 - k3); atomicAdd(&knn_probe[4], 1ull);
  }
#endif
  // ---- fallback (rare): sorted-list insertion for the flagged queries; every wave helps staging the tiles
  if (!__syncthreads_or(overflow ? 1 : 0)) return;
  float tau[QW], tie_tau[QW];  // (tie_tau: knn_kernel's note)
  KnnList<1> L[QW];
  bool todo[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    todo[q] = cnt[q] > KNN2_CAP && (q0 + q) < m;
    tau[q] = INFINITY;
    tie_tau[q] = -1.f;
    L[q].d[0] = INFINITY;
    L[q].i[0] = 0;
  }
  for (int base = 0; base < n; base += SEARCH_TILE) {
    int tcnt = min(SEARCH_TILE, n - base);
    __syncthreads();
    stage_tile(cloud, n, base, tcnt, sx, sy, sz);
    __syncthreads();
    for (int it = 0; it < tcnt; it += 64) {
      int p = it + lane;
      bool in = p < tcnt;
      float x = in ? sx[p] : 0.f, y = in ? sy[p] : 0.f, z = in ? sz[p] : 0.f;
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        if (!todo[q]) continue;  // wave-uniform
        float d = dist2(qx[q], qy[q], qz[q], x, y, z);
        unsigned long long mask = __ballot(in && d < tau[q]);
        if (__ballot(in && d == tau[q]) != 0ull) tie_tau[q] = tau[q];
        while (mask) {
          int src = (int)__builtin_ctzll(mask);
          mask &= mask - 1;
          float cd = readlane_f(d, src);
          if (cd < tau[q]) {
            knn_insert<1>(L[q], cd, base + it + src, lane);
            const float t = readlane_f(L[q].d[0], k - 1);
            if (t == tau[q]) tie_tau[q] = t;
            tau[q] = t;
          } else if (cd == tau[q]) {
            tie_tau[q] = cd;
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    if (!todo[q]) continue;
    size_t o = ((size_t)bi * m + q0 + q) * k;
    if (lane < k) {
      idx[o + lane] = (IdxT)L[q].i[0];
      if (dist_out) dist_out[o + lane] = L[q].d[0];
    }
    if (flags.nflag) {  // (more than 128 candidates under the bound is not yet a tie: the same test as knn_kernel's)
      const float prev = wave_shr1_f(L[q].d[0]);
      if (tie_tau[q] == tau[q] || __ballot(lane >= 1 && lane < k && L[q].d[0] == prev) != 0ull) knn_flag_query(flags, bi, m, q0 + q, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// group_point: out row r=(b,j,k) <- points[b, idx[r], :].  VEC floats per thread.
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void group_point_kernel(int n, int c, long rows_per_batch, long total_chunks,
                                                         const float* __restrict__ points, const int* __restrict__ idx,
                                                         float* __restrict__ out) {
  const int cpr = c / VEC;  // chunks per row
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total_chunks; g += (long)gridDim.x * 256) {
    long row = g / cpr;
    int ch = (int)(g - row * cpr);
    long bi = row / rows_per_batch;
    int a = idx[row];
    const float* src = points + ((size_t)bi * n + a) * c + (size_t)ch * VEC;
    float* dst = out + (size_t)row * c + (size_t)ch * VEC;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
      *dst = *src;
    }
  }
}

__global__ __launch_bounds__(256) void group_point_grad_kernel(int n, int c, long rows_per_batch, long total,
                                                              const float* __restrict__ grad_out,
                                                              const int* __restrict__ idx, float* __restrict__ grad_points) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long row = e / c;
    int l = (int)(e - row * c);
    long bi = row / rows_per_batch;
    int a = idx[row];
    atomicAdd(&grad_points[((size_t)bi * n + a) * c + l], grad_out[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// sa_group: one workgroup per (cloud, query).  Thread t owns output column c = t % W of rows s = t / W,
// t / W + R, ... (R = T / W rows per pass), so a pass writes R*W consecutive floats (coalesced) and the
// running column maximum stays in a register; R partial maxima per column meet in LDS.
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void sa_group_kernel(int n, int c, int m, int k, const float* __restrict__ xyz,
                                                    const float* __restrict__ feature, const int* __restrict__ idx,
                                                    const float* __restrict__ new_xyz, float* __restrict__ new_point,
                                                    float* __restrict__ skip_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sidx = reinterpret_cast<int*>(smem);        // [k]
  float* part = reinterpret_cast<float*>(sidx + k);  // [R][W]
  const int W = 6 + c;
  const int R = T / W;
  const long g = blockIdx.x;  // (b, j) flattened
  const long bi = g / m;
  const int t = threadIdx.x;
  for (int s = t; s < k; s += T) sidx[s] = idx[g * k + s];
  __syncthreads();
  const float* cx = xyz + (size_t)bi * n * 3;
  const float* cf = feature + (size_t)bi * n * c;
  float* out = new_point + (size_t)g * k * W;
  if (W > T) {
    // wide rows (C > T-6): a thread owns columns t, t+T, ... and walks all k rows of each
    for (int col = t; col < W; col += T) {
      const float centre = col < 3 ? new_xyz[g * 3 + col] : 0.f;
      float mx = -INFINITY;
      for (int s = 0; s < k; ++s) {
        const int i = sidx[s];
        float v;
        if (col < 3) v = cx[(size_t)i * 3 + col] - centre;
        else if (col < 6) v = cx[(size_t)i * 3 + (col - 3)];
        else v = cf[(size_t)i * c + (col - 6)];
        out[(size_t)s * W + col] = v;
        mx = fmaxf(mx, v);
      }
      skip_max[g * W + col] = mx;
    }
    return;
  }
  const int col = t % W, r0 = t / W;
  const bool active = r0 < R;
  float centre = 0.f;
  if (active && col < 3) centre = new_xyz[g * 3 + col];
  float mx = -INFINITY;
  if (active) {
    for (int s = r0; s < k; s += R) {
      const int i = sidx[s];
      float v;
      if (col < 3) v = cx[(size_t)i * 3 + col] - centre;
      else if (col < 6) v = cx[(size_t)i * 3 + (col - 3)];
      else v = cf[(size_t)i * c + (col - 6)];
      out[(size_t)s * W + col] = v;
      mx = fmaxf(mx, v);
    }
    part[r0 * W + col] = mx;
  }
  __syncthreads();
  if (t < W) {
    float v = part[t];
    for (int r = 1; r < R; ++r) v = fmaxf(v, part[r * W + t]);
    skip_max[g * W + t] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Deterministic backward of the three gather-type ops (gather_point, group_point, three_interpolate).
// The reference scatters with atomicAdd (tf_sampling_g.cu:183-192, tf_grouping_g.cu:61-78: run-to-run different
// fp32 sums) or with a sequential CPU loop (tf_interpolate.cpp:131-153).  Here the gradient row of source point t
// is the sum of its contributions in ASCENDING ORDER OF THE FORWARD OUTPUT ELEMENT -- exactly the sequential loop,
// so the result is bit-reproducible and equals the oracle bit for bit.  No floating-point atomics:
//   1. grad_lists_kernel (one workgroup per cloud): histogram of the targets in LDS (integer atomics), exclusive
//      scan -> start[t], then every contribution e takes a slot of its target's list (order inside a list arbitrary);
//   2. grad_segsum_kernel (one wave per target row): sorts the row's list (wave bitonic network; lists longer than
//      64 are consumed 64 smallest at a time) and adds the source rows in that order, channels across the lanes.
// ---------------------------------------------------------------------------------------------
constexpr int GL_THREADS = 1024;

__global__ __launch_bounds__(GL_THREADS) void grad_lists_kernel(int n, long entries, const int* __restrict__ idx,
                                                               int* __restrict__ start, int* __restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* cnt = reinterpret_cast<int*>(smem);  // [n] counts, then cursors
  __shared__ int wsum[GL_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long bi = blockIdx.x;
  const int* ix = idx + bi * entries;
  for (int t = tid; t < n; t += GL_THREADS) cnt[t] = 0;
  __syncthreads();
  for (long e = tid; e < entries; e += GL_THREADS) {
    const int t = ix[e];
    if (t >= 0 && t < n) atomicAdd(&cnt[t], 1);
  }
  __syncthreads();
  // exclusive scan: thread t owns targets [t*per, t*per+per)
  const int per = (n + GL_THREADS - 1) / GL_THREADS;
  int sum = 0;
  for (int j = 0; j < per; ++j) {
    const int t = tid * per + j;
    if (t < n) sum += cnt[t];
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int* st = start + bi * (n + 1);
  for (int j = 0; j < per; ++j) {
    const int t = tid * per + j;
    if (t < n) {
      const int c = cnt[t];
      cnt[t] = base;
      st[t] = base;
      base += c;
    }
  }
  if (tid == GL_THREADS - 1) st[n] = base;  // the last thread's running total = number of valid contributions
  __syncthreads();
  int* ls = list + bi * entries;
  for (long e = tid; e < entries; e += GL_THREADS) {
    const int t = ix[e];
    if (t >= 0 && t < n) ls[atomicAdd(&cnt[t], 1)] = (int)e;
  }
}

// src row of contribution e of cloud bi: src + ((bi*entries + e) / rep) * c ; scale weight[bi*entries + e] (or none)
__global__ __launch_bounds__(256) void grad_segsum_kernel(int n, int c, long entries, int rep, const float* __restrict__ src,
                                                         const float* __restrict__ weight, const int* __restrict__ start,
                                                         const int* __restrict__ list, float* __restrict__ dst, long rows) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const long bi = r / n;
  const int t = (int)(r - bi * n);
  const int lo = start[bi * (n + 1) + t], hi = start[bi * (n + 1) + t + 1];
  const int* ls = list + bi * entries;
  float* out = dst + r * c;
  constexpr int CH = 8;  // 64-channel slabs accumulated per walk over the list
  for (int c0 = 0; c0 < c; c0 += CH * 64) {
    float acc[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = 0.f;
    uint32_t prev = 0;
    bool first = true;
    for (int done = lo; done < hi;) {
      // the 64 smallest not yet consumed contributions, ascending
      uint32_t key[2];
      key[0] = 0xffffffffu;
      for (int base = lo; base < hi; base += 64) {
        uint32_t cand = 0xffffffffu;
        if (base + lane < hi) {
          const uint32_t e = (uint32_t)ls[base + lane];
          if (first || e > prev) cand = e;
        }
        key[1] = cand;
        wave_bitonic_sort<2, uint32_t>(key, lane);
      }
      const int take = min(64, hi - done);
      for (int j = 0; j < take; ++j) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)key[0], j);
        const long ge = bi * entries + e;
        const float* row = src + (ge / rep) * c + c0 + lane;
        const float w = weight ? weight[ge] : 1.f;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          if (c0 + q * 64 + lane < c) {
            const float v = row[q * 64];
            acc[q] = acc[q] + (weight ? v * w : v);
          }
        }
      }
      prev = (uint32_t)__builtin_amdgcn_readlane((int)key[0], take - 1);
      first = false;
      done += take;
    }
#pragma unroll
    for (int q = 0; q < CH; ++q)
      if (c0 + q * 64 + lane < c) out[c0 + q * 64 + lane] = acc[q];
  }
}

// ---------------------------------------------------------------------------------------------
// AdaptiveSampling with as_neighbor == 0 (pointasnl_util.py:161-164): the sampled point is replaced by its nearest
// neighbour (index idx[b,j,0] -- normally itself, a lower-indexed duplicate otherwise):
//     new_xyz[b,j] = xyz[b,i],   new_feature[b,j] = [xyz[b,i] | feature[b,i]],   i = idx[b,j,0]
// One launch instead of the reference's slice + two gathers + concat + two slices.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void take_neighbor0_kernel(int n, int c, int m, int k, long total, const float* __restrict__ xyz,
                                                            const float* __restrict__ feature, const int* __restrict__ idx,
                                                            float* __restrict__ new_xyz, float* __restrict__ new_feature) {
  const int w = 3 + c;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / w;  // (b, j)
    const int col = (int)(e - row * w);
    const long bi = row / m;
    const int i = idx[row * k];
    const float v = col < 3 ? xyz[((size_t)bi * n + i) * 3 + col] : feature[((size_t)bi * n + i) * c + (col - 3)];
    new_feature[e] = v;
    if (col < 3) new_xyz[row * 3 + col] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// max_pool_rows: out[b, c] = max_s x[b, s, c]  -- the PointNet set-abstraction pooling over the points of a
// region (pointnet_util.py:137, tf.reduce_max(new_points, axis=[2])).  One workgroup per (cloud, 64-channel
// slab): lanes over channels (coalesced 256-byte rows), the 4 waves split the rows, partial maxima meet in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_pool_rows_kernel(int n, int c, const float* __restrict__ x, float* __restrict__ out,
                                                            long out_stride) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const bool ok = col < c;
  const float* p = x + (size_t)blockIdx.y * n * c + (ok ? col : 0);
  float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  int s = wave;
  for (; s + 12 < n; s += 16) {  // four independent loads in flight per lane
    float a = p[(size_t)s * c], b = p[(size_t)(s + 4) * c], d = p[(size_t)(s + 8) * c], e = p[(size_t)(s + 12) * c];
    m0 = fmaxf(m0, a); m1 = fmaxf(m1, b); m2 = fmaxf(m2, d); m3 = fmaxf(m3, e);
  }
  for (; s < n; s += 4) m0 = fmaxf(m0, p[(size_t)s * c]);
  part[wave][lane] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  __syncthreads();
  if (wave == 0 && ok)
    out[(size_t)blockIdx.y * out_stride + col] = fmaxf(fmaxf(part[0][lane], part[1][lane]), fmaxf(part[2][lane], part[3][lane]));
}

// ---------------------------------------------------------------------------------------------
// select_top_k: one wave per (b,m) row; k rounds of "first strict minimum of the current row in
// [s,n), swap with position s" (tf_grouping_g.cu:100-121), on the output arrays in global memory.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_top_k_kernel(int n, int k, long rows, const float* __restrict__ dist,
                                                          int* __restrict__ outi, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = dist + (size_t)row * n;
  float* pd = out + (size_t)row * n;
  int* pi = outi + (size_t)row * n;
  for (int s = lane; s < n; s += 64) { pd[s] = src[s]; pi[s] = s; }
  // Lane 0 rewrites two elements per round and other lanes must see them in the next round: stores are
  // drained (write-through to this XCD's L2) and the row is re-read with agent-scope loads that bypass the
  // CU's L1.  Only this wave ever touches the row.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int s = 0; s < k && s < n; ++s) {
    // (value, position) lexicographic minimum over [s, n): strict '<' scanning upward == lowest position on ties
    float bv = INFINITY;
    int bp = 0x7fffffff;
    for (int t = s + lane; t < n; t += 64) {
      float v = __hip_atomic_load(&pd[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v < bv) { bv = v; bp = t; }  // t ascending inside a lane
    }
    // NaN-free total order on (bv, bp): smaller value, then smaller position
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float ov = __shfl_xor(bv, off);
      int op = __shfl_xor(bp, off);
      if (ov < bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
    }
    // the reference starts with min = s and only moves on a strict '<': if nothing in (s,n) is smaller
    // than pd[s], position s stays.  The lexicographic minimum above returns s in that case as well
    // unless every value is +inf/NaN (bp stays 0x7fffffff) -> no swap.
    int mn = (bp == 0x7fffffff) ? s : bp;
    if (mn != s && lane == 0) {
      float vs = __hip_atomic_load(&pd[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int is = __hip_atomic_load(&pi[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int im = __hip_atomic_load(&pi[mn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&pd[mn], vs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&pd[s], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&pi[mn], is, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&pi[s], im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

}  // namespace pasnl

using namespace pasnl;

static int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// smallest float t such that sqrtf(t) >= r  (r > 0): then sqrtf(d) < r  <=>  d < t for every float d >= 0
static float ball_threshold(float r) {
  float t = r * r;
  if (!(t < INFINITY)) return INFINITY;
  while (sqrtf(t) >= r && t > 0.f) t = nextafterf(t, -INFINITY);  // now sqrtf(t) < r (or t == 0)
  while (sqrtf(t) < r) t = nextafterf(t, INFINITY);
  return t;
}

extern "C" int pasnl_query_ball_point(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                                      int* idx, int* pts_cnt, pasnl_stream_t stream) {
  PASNL_REQUIRE(radius > 0.f, PASNL_EINVAL);  // "QueryBallPoint expects positive radius"
  PASNL_REQUIRE(nsample > 0, PASNL_EINVAL);   // "QueryBallPoint expects positive nsample"
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0, PASNL_EINVAL);
  if (b == 0 || m == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz1 && xyz2 && idx && pts_cnt, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  constexpr int QW = 4;
  // max(sqrtf(d2),1e-20f) < radius: for radius <= 1e-20f nothing can hit -> threshold 0 (d2 < 0 never true)
  float thr2 = (radius > 1e-20f) ? ball_threshold(radius) : 0.f;
  // Grid-pruned kernel for LDS-sized clouds (every in-model use and the north-star shape); PASNL_BALL_BRUTE=1 (tuning
  // build only) forces the brute-force kernel, which also serves larger clouds.
  if (!tune_env("PASNL_BALL_BRUTE")) {
    const int rc = ball_grid_launch(b, n, m, radius, thr2, nsample, xyz1, xyz2, idx, pts_cnt, pasnl_hip_stream(stream));
    if (rc != PASNL_EUNSUPPORTED) return rc;
  }
  size_t lds = (size_t)SEARCH_TILE * 12 + (size_t)SEARCH_WAVES * QW * nsample * sizeof(int);
  PASNL_REQUIRE(lds <= 160 * 1024, PASNL_EUNSUPPORTED);
  auto kern = ball_query_kernel<QW>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  int qpb = SEARCH_WAVES * QW;
  hipLaunchKernelGGL(kern, dim3((m + qpb - 1) / qpb, b), dim3(SEARCH_WAVES * 64), lds, pasnl_hip_stream(stream), n, m, thr2,
                     nsample, xyz1, xyz2, idx, pts_cnt);
  return pasnl_launch_status();
}

template <int SLOTS, int QW>
static int knn_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                      float* dist2, pasnl::KnnTieFlags flags, hipStream_t st) {
  int qpb = SEARCH_WAVES * QW;
  dim3 grid((m + qpb - 1) / qpb, b), block(SEARCH_WAVES * 64);
  if (idx_is_i64)
    hipLaunchKernelGGL((knn_kernel<SLOTS, QW, long long>), grid, block, 0, st, n, m, k, support, queries,
                       static_cast<long long*>(idx), dist2, flags);
  else
    hipLaunchKernelGGL((knn_kernel<SLOTS, QW, int>), grid, block, 0, st, n, m, k, support, queries, static_cast<int*>(idx),
                       dist2, flags);
  return pasnl_launch_status();
}

#ifdef PASNL_KNN_PROBE
// [pass 1 (incl. staging), bound, pass 2, emit, waves] cycles summed over waves
extern "C" int pasnl_knn_probe_read(unsigned long long* host8) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(pasnl::knn_probe), sizeof(pasnl::knn_probe)) != hipSuccess) return -1;
  unsigned long long zero[8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pasnl::knn_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int pasnl_knn_batch(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                               int idx_is_i64, float* dist2, pasnl_stream_t stream) {
  return pasnl::knn_brute_launch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, pasnl::KnnTieFlags{nullptr, nullptr},
                                 pasnl_hip_stream(stream));
}

int pasnl::knn_brute_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                            float* dist2, pasnl::KnnTieFlags flags, hipStream_t st) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k <= n, PASNL_EINVAL);  // nanoflann leaves slots uninitialised when K > npts; refuse instead
  PASNL_REQUIRE(k <= PASNL_KNN_MAX_K, PASNL_EUNSUPPORTED);
  if (b == 0 || m == 0) return PASNL_OK;
  PASNL_REQUIRE(support && queries && idx, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  // Two-pass selection wins wherever selection dominates (measured: 2.6x at N=1024,K=32; 3.7x at N=512,K=64); for
  // small K over large clouds both kernels are bound by the distance loop and the single pass is ahead
  // (N=8192,K=16: 795 vs 982 us).  PASNL_KNN_INSERTION=1 forces the insertion kernel (A/B measurements).
  if (k <= 64 && (!(k <= 16 && n > 2048) || tune_env("PASNL_KNN_TWO_PASS")) && !tune_env("PASNL_KNN_INSERTION")) {
    // queries per wave: a wave works through its queries one after the other (two in-wave sorts each), so a launch with few
    // queries is ONE round of long chains (cls layer 2, 8 192 queries at four per wave: 2 048 waves on 1 024 SIMDs, 136 us of
    // which 2/3 are the sorts); fewer queries per wave until the chip holds ~8 waves per SIMD
    const long nq = (long)b * m;
#ifndef PASNL_KNN2_MIN_WAVES
#define PASNL_KNN2_MIN_WAVES 8192L
#endif
    const int qw = nq >= 4L * PASNL_KNN2_MIN_WAVES ? 4 : (nq >= 2L * PASNL_KNN2_MIN_WAVES ? 2 : 1);
    dim3 grid((m + SEARCH_WAVES * qw - 1) / (SEARCH_WAVES * qw), b), block(SEARCH_WAVES * 64);
#define PASNL_KNN2Q(RR, Q, T) hipLaunchKernelGGL((knn2_kernel<RR, Q, T>), grid, block, 0, st, n, m, k, support, queries, static_cast<T*>(idx), dist2, flags)
#define PASNL_KNN2(RR, T) { if (qw == 4) PASNL_KNN2Q(RR, 4, T); else if (qw == 2) PASNL_KNN2Q(RR, 2, T); else PASNL_KNN2Q(RR, 1, T); }
    if (k <= 32) { if (idx_is_i64) PASNL_KNN2(1, long long) else PASNL_KNN2(1, int) }
    else { if (idx_is_i64) PASNL_KNN2(2, long long) else PASNL_KNN2(2, int) }
#undef PASNL_KNN2Q
#undef PASNL_KNN2
    return pasnl_launch_status();
  }
  if (k <= 64) return knn_launch<1, 4>(b, n, m, k, support, queries, idx, idx_is_i64, dist2, flags, st);
  if (k <= 128) return knn_launch<2, 2>(b, n, m, k, support, queries, idx, idx_is_i64, dist2, flags, st);
  return knn_launch<4, 1>(b, n, m, k, support, queries, idx, idx_is_i64, dist2, flags, st);
}

extern "C" int pasnl_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out,
                                 pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && nsample >= 0, PASNL_EINVAL);
  long rows_per_batch = (long)m * nsample;
  long rows = (long)b * rows_per_batch;
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(points && idx && out, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  bool vec4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  if (vec4) {
    long chunks = rows * (c / 4);
    hipLaunchKernelGGL(group_point_kernel<4>, dim3(grid_for(chunks)), dim3(256), 0, st, n, c, rows_per_batch, chunks, points,
                       idx, out);
  } else {
    long chunks = rows * c;
    hipLaunchKernelGGL(group_point_kernel<1>, dim3(grid_for(chunks)), dim3(256), 0, st, n, c, rows_per_batch, chunks, points,
                       idx, out);
  }
  return pasnl_launch_status();
}

extern "C" int pasnl_sa_group(int b, int n, int c, int m, int k, const float* xyz, const float* feature, const int* idx,
                              const float* new_xyz, float* new_point, float* skip_max, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  const int W = 6 + c;
  long groups = (long)b * m;
  if (groups == 0) return PASNL_OK;
  PASNL_REQUIRE(groups < (1L << 31), PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(xyz && feature && idx && new_xyz && new_point && skip_max, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  if (W <= 256) {
    size_t lds = (size_t)k * 4 + (size_t)(256 / W) * W * 4;
    hipLaunchKernelGGL(sa_group_kernel<256>, dim3((unsigned)groups), dim3(256), lds, st, n, c, m, k, xyz, feature, idx, new_xyz,
                       new_point, skip_max);
  } else {
    size_t lds = (size_t)k * 4 + (size_t)(W <= 512 ? (512 / W) * W : 0) * 4;
    hipLaunchKernelGGL(sa_group_kernel<512>, dim3((unsigned)groups), dim3(512), lds, st, n, c, m, k, xyz, feature, idx, new_xyz,
                       new_point, skip_max);
  }
  return pasnl_launch_status();
}

extern "C" int pasnl_group_point_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                                      float* grad_points, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && nsample >= 0, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(grad_points, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  if (hipMemsetAsync(grad_points, 0, (size_t)b * n * c * sizeof(float), st) != hipSuccess) return PASNL_ELAUNCH;
  long rows_per_batch = (long)m * nsample;
  long total = (long)b * rows_per_batch * c;
  if (total == 0) return PASNL_OK;
  PASNL_REQUIRE(grad_out && idx, PASNL_ENULL);
  hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, c, rows_per_batch, total, grad_out,
                     idx, grad_points);
  return pasnl_launch_status();
}

static size_t grad_ws_bytes(int b, int n, long entries) {
  return ((size_t)b * (n + 1) + (size_t)b * entries) * sizeof(int);
}

extern "C" size_t pasnl_grad_workspace_bytes(int b, int n, long entries) {
  if (b <= 0 || n <= 0 || entries < 0) return 0;
  return grad_ws_bytes(b, n, entries);
}

static int grad_det(int b, int n, int c, long entries, int rep, const float* src, const float* weight, const int* idx,
                    float* dst, void* ws, size_t ws_bytes, hipStream_t st) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && entries >= 0, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(dst, PASNL_ENULL);
  PASNL_REQUIRE(entries == 0 || (src && idx), PASNL_ENULL);
  PASNL_REQUIRE(ws && ws_bytes >= grad_ws_bytes(b, n, entries), PASNL_EWORKSPACE);
  PASNL_REQUIRE((size_t)n * sizeof(int) <= 150 * 1024 && entries < (1L << 31), PASNL_EUNSUPPORTED);
  int* start = static_cast<int*>(ws);
  int* list = start + (size_t)b * (n + 1);
  size_t lds = (size_t)n * sizeof(int);
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(grad_lists_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(grad_lists_kernel, dim3(b), dim3(GL_THREADS), lds, st, n, entries, idx, start, list);
  long rows = (long)b * n;
  hipLaunchKernelGGL(grad_segsum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, n, c, entries, rep, src, weight, start,
                     list, dst, rows);
  return pasnl_launch_status();
}

extern "C" int pasnl_gather_point_grad_det(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* ws,
                                           size_t ws_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(m >= 0, PASNL_EINVAL);
  return grad_det(b, n, 3, m, 1, out_g, nullptr, idx, inp_g, ws, ws_bytes, pasnl_hip_stream(stream));
}

extern "C" int pasnl_group_point_grad_det(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                                          float* grad_points, void* ws, size_t ws_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(m >= 0 && nsample >= 0, PASNL_EINVAL);
  return grad_det(b, n, c, (long)m * nsample, 1, grad_out, nullptr, idx, grad_points, ws, ws_bytes, pasnl_hip_stream(stream));
}

extern "C" int pasnl_three_interpolate_grad_det(int b, int n, int c, int m, const float* grad_out, const int* idx,
                                                const float* weight, float* grad_points, void* ws, size_t ws_bytes,
                                                pasnl_stream_t stream) {
  PASNL_REQUIRE(n >= 0, PASNL_EINVAL);
  PASNL_REQUIRE(n == 0 || weight, PASNL_ENULL);
  // targets are the m known points; contributions e = 3*i + k of unknown point i (tf_interpolate.cpp:137-151 order)
  return grad_det(b, m, c, 3L * n, 3, grad_out, weight, idx, grad_points, ws, ws_bytes, pasnl_hip_stream(stream));
}

extern "C" int pasnl_take_neighbor0(int b, int n, int c, int m, int k, const float* xyz, const float* feature, const int* idx,
                                    float* new_xyz, float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  long total = (long)b * m * (3 + c);
  if (total == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && feature && idx && new_xyz && new_feature, PASNL_ENULL);
  hipLaunchKernelGGL(take_neighbor0_kernel, dim3(grid_for(total)), dim3(256), 0, pasnl_hip_stream(stream), n, c, m, k, total, xyz,
                     feature, idx, new_xyz, new_feature);
  return pasnl_launch_status();
}

extern "C" int pasnl_max_pool_rows_strided(int b, int n, int c, const float* x, float* out, long out_stride,
                                           pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && out_stride >= c, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(x && out, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  hipLaunchKernelGGL(max_pool_rows_kernel, dim3((c + 63) / 64, b), dim3(256), 0, pasnl_hip_stream(stream), n, c, x, out,
                     out_stride);
  return pasnl_launch_status();
}

extern "C" int pasnl_max_pool_rows(int b, int n, int c, const float* x, float* out, pasnl_stream_t stream) {
  return pasnl_max_pool_rows_strided(b, n, c, x, out, c, stream);
}

extern "C" int pasnl_select_top_k(int b, int n, int m, int k, const float* dist, int* outi, float* out,
                                  pasnl_stream_t stream) {
  PASNL_REQUIRE(k > 0, PASNL_EINVAL);  // "SelectionSort expects positive k"
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0, PASNL_EINVAL);
  long rows = (long)b * m;
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(dist && outi && out, PASNL_ENULL);
  hipLaunchKernelGGL(select_top_k_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, pasnl_hip_stream(stream), n, k, rows,
                     dist, outi, out);
  return pasnl_launch_status();
}
