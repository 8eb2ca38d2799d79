// Grid-pruned EXACT kNN for large clouds (n >= PASNL_KNN_GRID_MIN_N; K <= 64).
// Behaviour contract: cpp_knn_batch, utils/nearest_neighbors/knn_.cxx:72-135 (nanoflann L2): the K nearest support points
// of every query in ascending (squared distance, index) order, distances in the canonical fp32 arithmetic
// ((dx*dx)+(dy*dy))+(dz*dz) -- bit-identical to pasnl_knn_batch's brute-force kernels (tests compare them).
//
// The brute-force kernels evaluate N distances per query twice (8192 x 8192 x 16 clouds: 1.1 ms).  Here:
//   build  (one workgroup per cloud)  bounding box -> cubic cells of edge h chosen for ~0.4 K points per cell (flat / thin
//          clouds: the fixed point of h^3 * cells = prod max(extent_i, h), so a ground plane gets a 2-D grid) -> counting sort
//          of the points by cell (x fastest) into 16-byte records {x, y, z, index} + the cell start offsets, in a
//          caller-provided workspace.
//   query  (one wave per query)  the (2r+1)^3 cells around the query's cell are (2r+1)^2 CONTIGUOUS runs of records
//          (x-adjacent cells are adjacent in the sorted array).  The selection is the two-pass scheme of knn2_kernel over
//          those runs only: per-lane minima -> the K-th smallest of them bounds the K-th neighbour -> the records under the
//          bound are collected and sorted by (distance bits << 32 | index) in one in-wave bitonic network.
//   exactness  every record outside the examined block of cells lies beyond one of the block's faces, i.e. at least
//          b = (distance from the query to the nearest face that is not a face of the whole grid) away.  The result is
//          accepted iff  d_K < (b - 1e-3 h)^2 (1 - 2^-20)  -- margins that dwarf the rounding of the cell assignment
//          (<= G 2^-23 h) and of the fp32 distance (3 ulp) -- so a rejected point can neither enter the list nor tie with
//          its last entry.  Otherwise the ring grows (r = 1, 2, 3, then the whole cloud = brute force over the sorted
//          records), so the answer is exact for ANY input: queries outside the box (AdaptiveSampling moves them), empty
//          cells, duplicates.  More candidates under the bound than the sort network holds (heavy ties) are resolved by K
//          rounds of "smallest key above the previous one" over the same runs.
#include "common.hpp"

namespace pasnl {

constexpr int KG_CMAX = 4096;        // cells per cloud
constexpr int KG_BUILD_T = 1024;     // build workgroup
constexpr int KG_PPT = 16;           // points per build thread -> n <= 16384
constexpr int KG_WAVES = 4;          // query workgroup: one query per wave at a time
constexpr int KG_CAP = 128;          // candidate keys per query the sort network takes
constexpr int KG_RMAX = 3;           // rings tried before the whole cloud

struct KgParams {
  float x0, y0, z0, h, inv_h;
  int gx, gy, gz;
};

constexpr size_t KG_REC_OFFSET = (sizeof(KgParams) + (size_t)(KG_CMAX + 1) * 4 + 15) & ~(size_t)15;  // records are 16-byte loads
__host__ __device__ inline size_t kg_stride(int n) { return (KG_REC_OFFSET + (size_t)n * 16 + 255) & ~(size_t)255; }
__device__ __forceinline__ const KgParams* kg_params(const char* ws) { return reinterpret_cast<const KgParams*>(ws); }
__device__ __forceinline__ const int* kg_cells(const char* ws) { return reinterpret_cast<const int*>(ws + sizeof(KgParams)); }
__device__ __forceinline__ const float4* kg_records(const char* ws) {
  return reinterpret_cast<const float4*>(ws + KG_REC_OFFSET);
}

__device__ __forceinline__ int kg_cell1(float v, float v0, float inv_h, int g) {
  int c = (int)((v - v0) * inv_h);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KG_BUILD_T) void knn_grid_build_kernel(int n, float rho, int refine, const float* __restrict__ support,
                                                                  char* ws_all, size_t stride) {
  __shared__ int cnt[KG_CMAX];
  __shared__ float red[6][KG_BUILD_T / 64];
  __shared__ int wsum[KG_BUILD_T / 64];
  __shared__ KgParams P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cloud = support + (size_t)blockIdx.x * n * 3;
  char* ws = ws_all + (size_t)blockIdx.x * stride;

  float px[KG_PPT], py[KG_PPT], pz[KG_PPT];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int i = 0; i < KG_PPT; ++i) {
    const int p = i * KG_BUILD_T + tid;
    if (p < n) {
      px[i] = cloud[p * 3]; py[i] = cloud[p * 3 + 1]; pz[i] = cloud[p * 3 + 2];
      mn[0] = fminf(mn[0], px[i]); mx[0] = fmaxf(mx[0], px[i]);
      mn[1] = fminf(mn[1], py[i]); mx[1] = fmaxf(mx[1], py[i]);
      mn[2] = fminf(mn[2], pz[i]); mx[2] = fmaxf(mx[2], pz[i]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], s));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], s));
    }
    if (lane == 0) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
  }
  __shared__ float occ2[KG_BUILD_T / 64];
  __shared__ float hscale;
  int cell[KG_PPT];
  KgParams p;
  int ncell = 1;
  // Two passes: the cell edge from the bounding-box volume assumes the points fill the box; the occupancy a point
  // actually sees (sum c^2 / n over the cells) then corrects it once (a ball fills 52 % of its box, a scan far less).
  for (int pass = 0; pass < 2; ++pass) {
    for (int c = tid; c < KG_CMAX; c += KG_BUILD_T) cnt[c] = 0;
    __syncthreads();
    if (tid == 0) {
      float lo[3], e[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float l = red[a][0], u = red[3 + a][0];
#pragma unroll
        for (int w = 1; w < KG_BUILD_T / 64; ++w) { l = fminf(l, red[a][w]); u = fmaxf(u, red[3 + a][w]); }
        lo[a] = l;
        e[a] = u - l;
      }
      const float emax = fmaxf(e[0], fmaxf(e[1], e[2]));
      float h = 1.f;
      if (pass == 1) {
        h = P.h * hscale;
      } else if (emax > 0.f) {
        const float cells = fmaxf(1.f, (float)n / rho);
        h = emax;
        for (int it = 0; it < 8; ++it)  // h^3 * cells = prod max(e_i, h): flat axes drop out of the volume
          h = cbrtf(fmaxf(e[0], h) * fmaxf(e[1], h) * fmaxf(e[2], h) / cells);
      }
      h = fmaxf(h, emax * (1.f / 1024.f));
      if (!(h > 0.f)) h = 1.f;
      int gx, gy, gz;
      for (;;) {
        gx = (int)(e[0] / h) + 1; gy = (int)(e[1] / h) + 1; gz = (int)(e[2] / h) + 1;
        if ((long)gx * gy * gz <= KG_CMAX) break;
        h *= 1.1f;
      }
      P.x0 = lo[0]; P.y0 = lo[1]; P.z0 = lo[2]; P.h = h; P.inv_h = 1.f / h;
      P.gx = gx; P.gy = gy; P.gz = gz;
      *reinterpret_cast<KgParams*>(ws) = P;
    }
    __syncthreads();
    p = P;
    ncell = p.gx * p.gy * p.gz;
#pragma unroll
    for (int i = 0; i < KG_PPT; ++i) {
      if (i * KG_BUILD_T + tid < n) {
        cell[i] = (kg_cell1(pz[i], p.z0, p.inv_h, p.gz) * p.gy + kg_cell1(py[i], p.y0, p.inv_h, p.gy)) * p.gx +
                  kg_cell1(px[i], p.x0, p.inv_h, p.gx);
        atomicAdd(&cnt[cell[i]], 1);
      }
    }
    __syncthreads();
    if (pass == 1) break;
    // occupancy seen by a point: sum over cells of c^2 / n
    float s2 = 0.f;
    for (int c = tid; c < ncell; c += KG_BUILD_T) { const float cc = (float)cnt[c]; s2 += cc * cc; }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) s2 += __shfl_xor(s2, sft);
    if (lane == 0) occ2[wave] = s2;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int w = 0; w < KG_BUILD_T / 64; ++w) tot += occ2[w];
      const float seen = tot / (float)n;  // >= 1
      // dimension the cloud fills at this scale: flat axes (one cell thick) do not rescale the count
      const int dims = (p.gx > 1) + (p.gy > 1) + (p.gz > 1);
      const float ratio = rho / fmaxf(seen, 1e-3f);
      // refine: 0 = keep the box estimate, 1 = correct fully, 2 = half way (in log scale: robust against density gradients)
      hscale = (dims == 0 || refine == 0) ? 1.f : powf(ratio, (refine == 2 ? 0.5f : 1.f) / (float)dims);
      if (hscale > 0.93f && hscale < 1.07f) hscale = 1.f;  // close enough: keep the grid (the second pass repeats it)
    }
    __syncthreads();
  }
  // exclusive scan of cnt[0 .. KG_CMAX): 4 consecutive entries per thread
  int v[KG_CMAX / KG_BUILD_T], tsum = 0;
#pragma unroll
  for (int i = 0; i < KG_CMAX / KG_BUILD_T; ++i) { v[i] = cnt[tid * (KG_CMAX / KG_BUILD_T) + i]; tsum += v[i]; }
  int incl = tsum;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    int o = __shfl_up(incl, s);
    if (lane >= s) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + incl - tsum;
  int* cells = reinterpret_cast<int*>(ws + sizeof(KgParams));
#pragma unroll
  for (int i = 0; i < KG_CMAX / KG_BUILD_T; ++i) {
    const int c = tid * (KG_CMAX / KG_BUILD_T) + i;
    cnt[c] = run;            // becomes the fill pointer of the cell
    if (c <= ncell) cells[c] = run;
    run += v[i];
  }
  __syncthreads();
  if (tid == 0) cells[ncell] = n;  // (also the loop's value when ncell < KG_CMAX)
  float4* rec = reinterpret_cast<float4*>(ws + KG_REC_OFFSET);
#pragma unroll
  for (int i = 0; i < KG_PPT; ++i) {
    const int pi = i * KG_BUILD_T + tid;
    if (pi < n) {
      const int pos = atomicAdd(&cnt[cell[i]], 1);
      rec[pos] = make_float4(px[i], py[i], pz[i], __int_as_float(pi));
    }
  }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned long long o = shfl_xor_any<unsigned long long>(v, s);
    v = o < v ? o : v;
  }
  return v;
}

// Selection over the candidates `scan` enumerates for this lane (scan(f) calls f(distance bits, key) once per candidate;
// it may be called several times and must enumerate the same candidates each time).  Returns the wave's K nearest as keys
// (distance bits << 32 | index), ascending, lane t < k holding the t-th.
// tie (out, wave-uniform; for pasnl_knn_batch_ref): two equal distances among the K, or a candidate beyond them at the K-th distance
template <int R, typename Scan>
__device__ __forceinline__ unsigned long long kg_select(Scan&& scan, int k, int lane, unsigned long long* cb, bool& tie) {
  constexpr uint32_t INF_BITS = 0x7f800000u;
  // ---- pass 1: per-lane smallest distance(s) -> U = the K-th smallest of the lane minima bounds the K-th neighbour
  uint32_t m1 = INF_BITS, m2 = INF_BITS;
  scan([&](uint32_t di, unsigned long long) {
    if (R == 2) m2 = min(m2, max(m1, di));
    m1 = min(m1, di);
  });
  uint32_t mv[R];
  mv[0] = m1;
  if (R == 2) mv[1] = m2;
  wave_bitonic_sort<R, uint32_t>(mv, lane);
  uint32_t U = 0;
#pragma unroll
  for (int rr = 0; rr < R; ++rr)
    if (rr == ((k - 1) >> 6)) U = (uint32_t)__builtin_amdgcn_readlane((int)mv[rr], (k - 1) & 63);
  // ---- pass 2: candidates with d <= U  (U = +inf when fewer than K lanes saw one: everything is collected)
  int cnt = 0;
  scan([&](uint32_t di, unsigned long long key) {
    const bool c = di <= U;
    const unsigned long long mask = __ballot(c);
    if (mask) {
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      const int slot = cnt + rank;
      if (c && slot < KG_CAP) cb[slot] = key;
      cnt += (int)__builtin_popcountll(mask);
    }
  });
  // ---- order them
  if (cnt <= 64) {
    unsigned long long key[1];
    key[0] = lane < cnt ? cb[lane] : ~0ull;
    wave_bitonic_sort<1, unsigned long long>(key, lane);
    tie = knn_sorted_has_tie(key[0], ~0ull, k, lane);
    return key[0];
  }
  if (cnt <= KG_CAP) {
    unsigned long long key[2];
    key[0] = cb[lane];
    key[1] = 64 + lane < cnt ? cb[64 + lane] : ~0ull;
    wave_bitonic_sort<2, unsigned long long>(key, lane);
    tie = knn_sorted_has_tie(key[0], key[1], k, lane);
    return key[0];
  }
  // more candidates under the bound than the network holds (heavy ties -- or lane minima that bound the K-th distance loosely, which
  // depends on the order of the records: flagging every such query as tied made ~0.15 % of the queries of tie-free lidar clouds go
  // through the KD-tree, in varying numbers from call to call): K + 1 rounds of "smallest key not below `lower`" -- the extra
  // round yields the candidate behind the list for the tie test
  unsigned long long mykey = ~0ull, nextkey = ~0ull, lower = 0;
  for (int t = 0; t <= k; ++t) {
    unsigned long long best = ~0ull;
    scan([&](uint32_t di, unsigned long long key) {
      if (di != 0xffffffffu && key >= lower && key < best) best = key;
    });
    best = wave_min_u64(best);
    if (t < 64) { if (lane == t) mykey = best; }
    else if (lane == 0) nextkey = best;  // (k == 64: rank 64)
    if (best == ~0ull) break;            // nothing left
    lower = best + 1;
  }
  tie = knn_sorted_has_tie(mykey, nextkey, k, lane);
  return mykey;
}

// LOOP: a capped grid (the background form) walks the cloud's queries with the grid's stride; without it a wave has ONE query
// (the loop costs ten registers = a wave per SIMD: kept out of the usual form)
template <int R, bool WIDE, typename IdxT, bool LOOP = false>  // WIDE: two slots per run in the ring-1 fast path (runs up to 128 records)
__global__ __launch_bounds__(KG_WAVES * 64) void knn_grid_query_kernel(int n, int m, int k, const float* __restrict__ queries,
                                                                     const char* __restrict__ ws_all, size_t stride,
                                                                     IdxT* __restrict__ idx, float* __restrict__ dist_out,
                                                                     const KnnTieFlags flags) {
  __shared__ unsigned long long cand[KG_WAVES][KG_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.y;
  const char* ws = ws_all + (size_t)bi * stride;
  const KgParams P = *kg_params(ws);
  const int* __restrict__ cells = kg_cells(ws);
  const float4* __restrict__ rec = kg_records(ws);
  // one query per wave (pasnl_knn_batch_ws_bg: per wave and trip).  Wave-uniform; no workgroup barrier below.
  auto one = [&](const int j) {
  const float* qp = queries + ((size_t)bi * m + j) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const int cx = kg_cell1(qx, P.x0, P.inv_h, P.gx), cy = kg_cell1(qy, P.y0, P.inv_h, P.gy), cz = kg_cell1(qz, P.z0, P.inv_h, P.gz);
  unsigned long long* cb = cand[wave];

  const float coord_mag = fmaxf(fmaxf(fmaxf(fabsf(P.x0) + (float)P.gx * P.h, fabsf(P.y0) + (float)P.gy * P.h),
                                      fabsf(P.z0) + (float)P.gz * P.h), fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz)));
  // squared distance below which nothing outside the block [xl..xh] x [yl..yh] x [zl..zh] of cells can lie (see the header)
  auto bound_of = [&](int xl, int xh, int yl, int yh, int zl, int zh) {
    float b = INFINITY;
    if (xl > 0) b = fminf(b, qx - (P.x0 + (float)xl * P.h));
    if (xh < P.gx - 1) b = fminf(b, (P.x0 + (float)(xh + 1) * P.h) - qx);
    if (yl > 0) b = fminf(b, qy - (P.y0 + (float)yl * P.h));
    if (yh < P.gy - 1) b = fminf(b, (P.y0 + (float)(yh + 1) * P.h) - qy);
    if (zl > 0) b = fminf(b, qz - (P.z0 + (float)zl * P.h));
    if (zh < P.gz - 1) b = fminf(b, (P.z0 + (float)(zh + 1) * P.h) - qz);
    if (b == INFINITY) return INFINITY;  // the block is the whole grid
    // margins: a thousandth of a cell for the cell arithmetic, and four ulps of the LARGEST coordinate in play for the
    // rounding of the face planes x0 + c*h and of b itself (clouds far from the origin: |x0| >> h)
    b = fmaxf(b - 1e-3f * P.h - coord_mag * 4.76837158203125e-07f, 0.f);
    return b * b * (1.f - 9.5367431640625e-07f);
  };

  unsigned long long mykey = ~0ull;  // lane t < k ends up with the t-th neighbour's key
  bool done = false, tie = false;    // tie: of the selection that was accepted (the last one)
  int r = 1;
  // ---- ring 1, the common case: the nine runs' bounds, then their records, are requested together (ONE memory latency
  // instead of one per run and pass), and both selection passes work from registers
  {
    const int xl = max(cx - 1, 0), xh = min(cx + 1, P.gx - 1);
    int rs[9], rl[9], total = 0, longest = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int y = cy + q % 3 - 1, z = cz + q / 3 - 1;
      const bool ok = y >= 0 && y < P.gy && z >= 0 && z < P.gz;
      const int base = ok ? (z * P.gy + y) * P.gx : 0;
      const int s0 = cells[base + xl], e0 = cells[base + xh + 1];
      rs[q] = s0;
      rl[q] = ok ? e0 - s0 : 0;
      total += rl[q];
      longest = max(longest, rl[q]);
    }
    constexpr int NS = WIDE ? 18 : 9;  // lanes-wide slots: one or two per run
    if (total >= k && longest <= 64 * (NS / 9)) {
      uint32_t dq[NS];
      unsigned long long kq[NS];
      const bool wide = longest > 64;  // wave-uniform: the second slot of every run is dead (and not even loaded) otherwise
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        dq[q] = 0xffffffffu;
        kq[q] = ~0ull;
        const int run_q = WIDE ? q >> 1 : q, half = WIDE ? (q & 1) : 0;
        if (!half || wide) {
          const int off = lane + 64 * half;
          const float4 c = rec[min(rs[run_q] + off, n - 1)];
          const uint32_t di = __float_as_uint(dist2(qx, qy, qz, c.x, c.y, c.z));
          const bool live = off < rl[run_q];
          dq[q] = live ? di : 0xffffffffu;  // (above every real distance and above +inf: never selected, never collected)
          kq[q] = ((unsigned long long)di << 32) | (uint32_t)__float_as_int(c.w);
        }
      }
      auto scan = [&](auto&& f) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
          if (!(WIDE && (q & 1)) || wide) f(dq[q], kq[q]);
      };
      // (dead slots carry 0xffffffff: kg_select's pass 1 sees them as > INF, pass 2 never collects them unless U is
      // 0xffffffff itself, which cannot happen: U is a lane minimum <= INF_BITS or INF_BITS)
      mykey = kg_select<R>(scan, k, lane, cb, tie);
      const uint32_t dk = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mykey >> 32), k - 1);
      const float bnd = bound_of(xl, xh, max(cy - 1, 0), min(cy + 1, P.gy - 1), max(cz - 1, 0), min(cz + 1, P.gz - 1));
      done = __uint_as_float(dk) < bnd || bnd == INFINITY;
    }
    r = done ? 1 : (total >= k && longest <= 64 * (NS / 9) ? 2 : 1);  // an overlong run: ring 1 again, through the generic path
  }
  for (; !done; ++r) {
    const bool whole_req = r > KG_RMAX;
    const int xl = whole_req ? 0 : max(cx - r, 0), xh = whole_req ? P.gx - 1 : min(cx + r, P.gx - 1);
    const int yl = whole_req ? 0 : max(cy - r, 0), yh = whole_req ? P.gy - 1 : min(cy + r, P.gy - 1);
    const int zl = whole_req ? 0 : max(cz - r, 0), zh = whole_req ? P.gz - 1 : min(cz + r, P.gz - 1);
    const float bnd = bound_of(xl, xh, yl, yh, zl, zh);
    const bool whole = bnd == INFINITY;
    // a block that is the whole grid is ONE run; otherwise one run per (y, z) row
    if (!whole) {
      int total = 0;
      for (int z = zl; z <= zh; ++z)
        for (int y = yl; y <= yh; ++y) {
          const int base = (z * P.gy + y) * P.gx;
          total += cells[base + xh + 1] - cells[base + xl];
        }
      if (total < k) continue;  // not even K records in the block: grow the ring
    }
    auto scan = [&](auto&& f) {
      auto run = [&](int s0, int e0) {
        for (int p0 = s0; p0 < e0; p0 += 64) {
          const int p = p0 + lane;
          const float4 c = rec[min(p, n - 1)];
          const uint32_t di = __float_as_uint(dist2(qx, qy, qz, c.x, c.y, c.z));
          f(p < e0 ? di : 0xffffffffu, ((unsigned long long)di << 32) | (uint32_t)__float_as_int(c.w));
        }
      };
      if (whole) { run(0, n); return; }
      for (int z = zl; z <= zh; ++z)
        for (int y = yl; y <= yh; ++y) {
          const int base = (z * P.gy + y) * P.gx;
          run(cells[base + xl], cells[base + xh + 1]);
        }
    };
    mykey = kg_select<R>(scan, k, lane, cb, tie);
    const uint32_t dk = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mykey >> 32), k - 1);
    done = whole || __uint_as_float(dk) < bnd;
  }
  const size_t o = ((size_t)bi * m + j) * k;
  if (lane < k) {
    idx[o + lane] = (IdxT)(uint32_t)mykey;
    if (dist_out) dist_out[o + lane] = __uint_as_float((uint32_t)(mykey >> 32));
  }
  // (a point outside the examined block is farther than the bound, which the K-th distance is strictly below: every point that
  // ties with the K-th was among the candidates)
  if (flags.nflag && tie) knn_flag_query(flags, bi, m, j, lane);
  };
  if constexpr (LOOP) {
    for (int j = blockIdx.x * KG_WAVES + wave; j < m; j += gridDim.x * KG_WAVES) one(j);
  } else {
    const int j = blockIdx.x * KG_WAVES + wave;
    if (j < m) one(j);
  }
}

// ---------------------------------------------------------------------------------------------
// knn_batch_distance_pick (utils/nearest_neighbors/knn_.cxx:136-200, binding knn.pyx:111-148): coverage-driven query
// selection.  Per cloud, nq times: among the points whose use count equals `current` (raised to the minimum count when none
// is left) take number (random % how many) in ascending index order; its K nearest neighbours (ascending (distance, index))
// get their count raised by one, the pick itself by 100.  Inherently sequential per cloud: one workgroup per cloud walks the
// nq iterations; a thread owns KD_PPT consecutive points (coordinates and keys in registers), the counts live in LDS, the K
// nearest are extracted by K rounds of "smallest key above the previous one".  The random stream is the caller's (the
// reference seeds ONE std::mt19937 with time(0) and walks the clouds in order: cloud b consumes outputs [b nq, (b+1) nq)).
// ---------------------------------------------------------------------------------------------
constexpr int KD_T = 1024, KD_PPT = 16;  // n <= 16384

__device__ __forceinline__ unsigned long long kd_block_min_u64(unsigned long long v, unsigned long long* red, int lane, int wave) {
  v = wave_min_u64(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  unsigned long long r = lane < KD_T / 64 ? red[lane] : ~0ull;
  r = wave_min_u64(r);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(KD_T) void knn_distance_pick_kernel(int n, int nq, int k, const float* __restrict__ pts,
                                                                const uint32_t* __restrict__ rnd, long long* __restrict__ idx,
                                                                float* __restrict__ queries) {
  extern __shared__ int used[];  // [n]
  __shared__ unsigned long long red[KD_T / 64];
  __shared__ int wsum[KD_T / 64];
  __shared__ int pick_s, cur_s, total_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.x;
  const float* cloud = pts + (size_t)bi * n * 3;
  const int c = (n + KD_T - 1) / KD_T;  // points per thread, consecutive: [tid*c, tid*c + c)
  float px[KD_PPT], py[KD_PPT], pz[KD_PPT];
#pragma unroll
  for (int j = 0; j < KD_PPT; ++j) {
    const int i = tid * c + j;
    const bool ok = j < c && i < n;
    px[j] = ok ? cloud[i * 3] : 0.f; py[j] = ok ? cloud[i * 3 + 1] : 0.f; pz[j] = ok ? cloud[i * 3 + 2] : 0.f;
  }
  for (int i = tid; i < n; i += KD_T) used[i] = 0;
  if (tid == 0) cur_s = 0;
  __syncthreads();
  for (int it = 0; it < nq; ++it) {
    // ---- how many points have the current count, and where
    int mine, incl;
    for (;;) {
      const int cur = cur_s;
      mine = 0;
#pragma unroll
      for (int j = 0; j < KD_PPT; ++j) {
        const int i = tid * c + j;
        if (j < c && i < n) mine += used[i] == cur;
      }
      incl = mine;
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) {
        const int o = __shfl_up(incl, sft);
        if (lane >= sft) incl += o;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int base = 0, tot = 0;
      for (int w = 0; w < KD_T / 64; ++w) { if (w < wave) base += wsum[w]; tot += wsum[w]; }
      incl += base;
      if (tid == 0) total_s = tot;
      __syncthreads();
      if (total_s > 0) break;
      // none left: raise `current` to the minimum count
      unsigned long long mn = ~0ull;
#pragma unroll
      for (int j = 0; j < KD_PPT; ++j) {
        const int i = tid * c + j;
        if (j < c && i < n) mn = min(mn, (unsigned long long)(unsigned)used[i]);
      }
      mn = kd_block_min_u64(mn, red, lane, wave);
      if (tid == 0) cur_s = (int)mn;
      __syncthreads();
    }
    // ---- the r-th of them in ascending index order
    const int r = (int)(rnd[(size_t)bi * nq + it] % (uint32_t)total_s);
    const int before = incl - mine;
    if (r >= before && r < incl) {
      int want = r - before;
      const int cur = cur_s;
#pragma unroll
      for (int j = 0; j < KD_PPT; ++j) {
        const int i = tid * c + j;
        if (j < c && i < n && used[i] == cur && want-- == 0) pick_s = i;
      }
    }
    __syncthreads();
    const int index = pick_s;
    const float qx = cloud[index * 3], qy = cloud[index * 3 + 1], qz = cloud[index * 3 + 2];
    // ---- its K nearest: keys (distance bits << 32 | index) in registers, K rounds of block-wide minimum
    unsigned long long key[KD_PPT];
#pragma unroll
    for (int j = 0; j < KD_PPT; ++j) {
      const int i = tid * c + j;
      key[j] = (j < c && i < n) ? ((unsigned long long)__float_as_uint(dist2(qx, qy, qz, px[j], py[j], pz[j])) << 32) | (unsigned)i : ~0ull;
    }
    unsigned long long lower = 0;
    long long* o = idx + ((size_t)bi * nq + it) * k;
    for (int t = 0; t < k; ++t) {
      unsigned long long best = ~0ull;
#pragma unroll
      for (int j = 0; j < KD_PPT; ++j)
        if (key[j] >= lower && key[j] < best) best = key[j];
      best = kd_block_min_u64(best, red, lane, wave);
      if (tid == 0) {
        const int nb = (int)(unsigned)best;
        o[t] = nb;
        used[nb] += 1;
      }
      lower = best + 1;
    }
    if (tid == 0) {
      used[index] += 100;
      float* qo = queries + ((size_t)bi * nq + it) * 3;
      qo[0] = qx; qo[1] = qy; qo[2] = qz;
    }
    __syncthreads();
  }
}

}  // namespace pasnl

using namespace pasnl;

extern "C" int pasnl_knn_batch(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                               int idx_is_i64, float* dist2, pasnl_stream_t stream);

static int kg_min_n() {
  if (const char* e = tune_env("PASNL_KNN_GRID_MIN_N")) return atoi(e);  // tuning build only
  return PASNL_KNN_GRID_MIN_N;
}

size_t pasnl::knn_grid_ws_bytes(int b, int n) {
  if (b <= 0 || n < kg_min_n() || n > KG_BUILD_T * KG_PPT) return 0;
  return (size_t)b * kg_stride(n);
}
extern "C" size_t pasnl_knn_workspace_bytes(int b, int n) { return pasnl::knn_grid_ws_bytes(b, n); }

#ifdef PASNL_TUNING
extern "C" void pasnl_tuning_stamp(int slot, hipStream_t st);
#endif
int pasnl::knn_grid_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                           float* dist2, void* workspace, size_t workspace_bytes, int max_workgroups, pasnl::KnnTieFlags flags,
                           hipStream_t st) {
  const size_t need = knn_grid_ws_bytes(b, n);
  if (need == 0 || k > 64 || k > n || m <= 0) {  // small clouds / wide lists: the brute-force kernels (same results)
#ifdef PASNL_TUNING
    const bool stamp = tune_env("PASNL_STAMP_N") && atoi(tune_env("PASNL_STAMP_N")) == n;  // (tools/step_stamps.py)
    if (stamp) pasnl_tuning_stamp(6, st);
    const int rc = knn_brute_launch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, flags, st);
    if (stamp) pasnl_tuning_stamp(7, st);
    return rc;
#else
    return knn_brute_launch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, flags, st);
#endif
  }
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(support && queries && idx, PASNL_ENULL);
  PASNL_REQUIRE(workspace != nullptr, PASNL_ENULL);
  PASNL_REQUIRE(workspace_bytes >= need, PASNL_EWORKSPACE);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  const size_t stride = kg_stride(n);
  // ~0.4 K records per cell: the sphere of radius h around a query (what ring 1 certifies) then holds ~1.7 K of them
  // target: ~0.7 K records in the cell of a point (measured optimum on uniform-box, ball and lidar-like clouds, K = 16 / 32)
  float rho = fmaxf(4.f, 0.7f * (float)k);
  if (const char* e = tune_env("PASNL_KNN_RHO")) rho = (float)atof(e) * (float)k;  // tuning build only
  int refine = 2;
  if (const char* e = tune_env("PASNL_KNN_REFINE")) refine = atoi(e);  // tuning build only
  hipLaunchKernelGGL(knn_grid_build_kernel, dim3(b), dim3(KG_BUILD_T), 0, st, n, rho, refine, support, static_cast<char*>(workspace),
                     stride);
  dim3 grid((m + KG_WAVES - 1) / KG_WAVES, b), block(KG_WAVES * 64);
  const bool bg = max_workgroups > 0;  // the background form: a capped grid that loops
  if (bg) grid.x = (unsigned)std::max(1, std::min((int)grid.x, max_workgroups / b));
#define PASNL_KG(RR, WW, T)                                                                                                   \
  {                                                                                                                            \
    if (bg) hipLaunchKernelGGL((knn_grid_query_kernel<RR, WW, T, true>), grid, block, 0, st, n, m, k, queries,                 \
                               static_cast<const char*>(workspace), stride, static_cast<T*>(idx), dist2, flags);               \
    else hipLaunchKernelGGL((knn_grid_query_kernel<RR, WW, T, false>), grid, block, 0, st, n, m, k, queries,                   \
                            static_cast<const char*>(workspace), stride, static_cast<T*>(idx), dist2, flags);                  \
  }
  if (k <= 16) { if (idx_is_i64) PASNL_KG(1, false, long long) else PASNL_KG(1, false, int) }
  else if (k <= 32) { if (idx_is_i64) PASNL_KG(1, true, long long) else PASNL_KG(1, true, int) }
  else { if (idx_is_i64) PASNL_KG(2, true, long long) else PASNL_KG(2, true, int) }
#undef PASNL_KG
  return pasnl_launch_status();
}

extern "C" int pasnl_knn_batch_ws(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                  int idx_is_i64, float* dist2, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  return pasnl::knn_grid_launch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, workspace, workspace_bytes, 0,
                                pasnl::KnnTieFlags{nullptr, nullptr}, pasnl_hip_stream(stream));
}

extern "C" int pasnl_knn_batch_ws_bg(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                     int idx_is_i64, float* dist2, void* workspace, size_t workspace_bytes, int max_workgroups,
                                     pasnl_stream_t stream) {
  PASNL_REQUIRE(max_workgroups > 0, PASNL_EINVAL);
  return pasnl::knn_grid_launch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, workspace, workspace_bytes, max_workgroups,
                                pasnl::KnnTieFlags{nullptr, nullptr}, pasnl_hip_stream(stream));
}

extern "C" int pasnl_knn_distance_pick(int b, int n, int nq, int k, const float* pts, const unsigned int* rnd, long long* idx,
                                       float* queries, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && nq >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k <= n, PASNL_EINVAL);
  PASNL_REQUIRE(n <= KD_T * KD_PPT, PASNL_EUNSUPPORTED);
  if (b == 0 || nq == 0) return PASNL_OK;
  PASNL_REQUIRE(pts && rnd && idx && queries, PASNL_ENULL);
  const size_t lds = (size_t)n * sizeof(int);
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(knn_distance_pick_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(knn_distance_pick_kernel, dim3(b), dim3(KD_T), lds, pasnl_hip_stream(stream), n, nq, k, pts, rnd, idx, queries);
  return pasnl_launch_status();
}
