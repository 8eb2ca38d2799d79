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
__global__ __launch_bounds__(KG_BUILD_T) void knn_grid_build_kernel(int n, float rho, const float* __restrict__ support, char* ws_all,
                                                                  size_t stride) {
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
  for (int c = tid; c < KG_CMAX; c += KG_BUILD_T) cnt[c] = 0;
  __syncthreads();
  if (tid == 0) {
    float lo[3], e[3];
    for (int a = 0; a < 3; ++a) {
      float l = red[a][0], u = red[3 + a][0];
      for (int w = 1; w < KG_BUILD_T / 64; ++w) { l = fminf(l, red[a][w]); u = fmaxf(u, red[3 + a][w]); }
      lo[a] = l;
      e[a] = u - l;
    }
    const float emax = fmaxf(e[0], fmaxf(e[1], e[2]));
    float h = 1.f;
    if (emax > 0.f) {
      const float cells = fmaxf(1.f, (float)n / rho);
      h = emax;
      for (int it = 0; it < 8; ++it)  // h^3 * cells = prod max(e_i, h): flat axes drop out of the volume
        h = cbrtf(fmaxf(e[0], h) * fmaxf(e[1], h) * fmaxf(e[2], h) / cells);
      h = fmaxf(h, emax * (1.f / 1024.f));
    }
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
  const KgParams p = P;
  const int ncell = p.gx * p.gy * p.gz;
  int cell[KG_PPT];
#pragma unroll
  for (int i = 0; i < KG_PPT; ++i) {
    if (i * KG_BUILD_T + tid < n) {
      cell[i] = (kg_cell1(pz[i], p.z0, p.inv_h, p.gz) * p.gy + kg_cell1(py[i], p.y0, p.inv_h, p.gy)) * p.gx +
                kg_cell1(px[i], p.x0, p.inv_h, p.gx);
      atomicAdd(&cnt[cell[i]], 1);
    }
  }
  __syncthreads();
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

template <int R, typename IdxT>
__global__ __launch_bounds__(KG_WAVES * 64) void knn_grid_query_kernel(int n, int m, int k, const float* __restrict__ queries,
                                                                     const char* __restrict__ ws_all, size_t stride,
                                                                     IdxT* __restrict__ idx, float* __restrict__ dist_out) {
  __shared__ unsigned long long cand[KG_WAVES][KG_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.y;
  const int j = blockIdx.x * KG_WAVES + wave;
  if (j >= m) return;  // wave-uniform; no workgroup barrier below
  const char* ws = ws_all + (size_t)bi * stride;
  const KgParams P = *kg_params(ws);
  const int* __restrict__ cells = kg_cells(ws);
  const float4* __restrict__ rec = kg_records(ws);
  const float* qp = queries + ((size_t)bi * m + j) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const int cx = kg_cell1(qx, P.x0, P.inv_h, P.gx), cy = kg_cell1(qy, P.y0, P.inv_h, P.gy), cz = kg_cell1(qz, P.z0, P.inv_h, P.gz);
  constexpr uint32_t INF_BITS = 0x7f800000u;
  unsigned long long* cb = cand[wave];

  unsigned long long mykey = ~0ull;  // lane t < k ends up with the t-th neighbour's key
  for (int r = 1;; ++r) {
    const bool whole_req = r > KG_RMAX;
    const int xl = whole_req ? 0 : max(cx - r, 0), xh = whole_req ? P.gx - 1 : min(cx + r, P.gx - 1);
    const int yl = whole_req ? 0 : max(cy - r, 0), yh = whole_req ? P.gy - 1 : min(cy + r, P.gy - 1);
    const int zl = whole_req ? 0 : max(cz - r, 0), zh = whole_req ? P.gz - 1 : min(cz + r, P.gz - 1);
    const bool whole = xl == 0 && yl == 0 && zl == 0 && xh == P.gx - 1 && yh == P.gy - 1 && zh == P.gz - 1;
    // squared distance below which nothing outside the block can lie (see the header)
    float bnd = INFINITY;
    if (!whole) {
      float b = INFINITY;
      if (xl > 0) b = fminf(b, qx - (P.x0 + (float)xl * P.h));
      if (xh < P.gx - 1) b = fminf(b, (P.x0 + (float)(xh + 1) * P.h) - qx);
      if (yl > 0) b = fminf(b, qy - (P.y0 + (float)yl * P.h));
      if (yh < P.gy - 1) b = fminf(b, (P.y0 + (float)(yh + 1) * P.h) - qy);
      if (zl > 0) b = fminf(b, qz - (P.z0 + (float)zl * P.h));
      if (zh < P.gz - 1) b = fminf(b, (P.z0 + (float)(zh + 1) * P.h) - qz);
      b = fmaxf(b - 1e-3f * P.h, 0.f);
      bnd = b * b * (1.f - 9.5367431640625e-07f);
    }
    const int nrows = (yh - yl + 1) * (zh - zl + 1), ny = yh - yl + 1;
    // a block that is the whole grid is ONE run; otherwise one run per (y, z) row
    auto run_of = [&](int row, int& s, int& e) {
      if (whole) { s = 0; e = n; return; }
      const int base = ((zl + row / ny) * P.gy + (yl + row % ny)) * P.gx;
      s = cells[base + xl];
      e = cells[base + xh + 1];
    };
    const int runs = whole ? 1 : nrows;
    if (!whole) {
      int total = 0;
      for (int row = 0; row < runs; ++row) { int s, e; run_of(row, s, e); total += e - s; }
      if (total < k) continue;  // not even K records in the block: grow the ring
    }
    // ---- pass 1: per-lane smallest distance(s)
    uint32_t m1 = INF_BITS, m2 = INF_BITS;
    for (int row = 0; row < runs; ++row) {
      int s, e;
      run_of(row, s, e);
      for (int p = s + lane; p < e; p += 64) {
        const float4 c = rec[p];
        const uint32_t di = __float_as_uint(dist2(qx, qy, qz, c.x, c.y, c.z));
        if (R == 2) m2 = min(m2, max(m1, di));
        m1 = min(m1, di);
      }
    }
    uint32_t mv[R];
    mv[0] = m1;
    if (R == 2) mv[1] = m2;
    wave_bitonic_sort<R, uint32_t>(mv, lane);
    uint32_t U = 0;
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
      if (rr == ((k - 1) >> 6)) U = (uint32_t)__builtin_amdgcn_readlane((int)mv[rr], (k - 1) & 63);
    // ---- pass 2: records with d <= U  (U = +inf when fewer than K lanes saw a record: everything is collected)
    int cnt = 0;
    for (int row = 0; row < runs; ++row) {
      int s, e;
      run_of(row, s, e);
      for (int p0 = s; p0 < e; p0 += 64) {
        const int p = p0 + lane;
        bool c = false;
        unsigned long long key = 0;
        if (p < e) {
          const float4 cr = rec[p];
          const uint32_t di = __float_as_uint(dist2(qx, qy, qz, cr.x, cr.y, cr.z));
          c = di <= U;
          key = ((unsigned long long)di << 32) | (uint32_t)__float_as_int(cr.w);
        }
        const unsigned long long mask = __ballot(c);
        if (mask) {
          const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          const int slot = cnt + rank;
          if (c && slot < KG_CAP) cb[slot] = key;
          cnt += (int)__builtin_popcountll(mask);
        }
      }
    }
    // ---- order them
    if (cnt <= 64) {
      unsigned long long key[1];
      key[0] = lane < cnt ? cb[lane] : ~0ull;
      wave_bitonic_sort<1, unsigned long long>(key, lane);
      mykey = key[0];
    } else if (cnt <= KG_CAP) {
      unsigned long long key[2];
      key[0] = cb[lane];
      key[1] = 64 + lane < cnt ? cb[64 + lane] : ~0ull;
      wave_bitonic_sort<2, unsigned long long>(key, lane);
      mykey = key[0];
    } else {
      // heavy ties: K rounds of "smallest key not below `lower`" over the same runs
      unsigned long long lower = 0;
      for (int t = 0; t < k; ++t) {
        unsigned long long best = ~0ull;
        for (int row = 0; row < runs; ++row) {
          int s, e;
          run_of(row, s, e);
          for (int p = s + lane; p < e; p += 64) {
            const float4 cr = rec[p];
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(dist2(qx, qy, qz, cr.x, cr.y, cr.z)) << 32) | (uint32_t)__float_as_int(cr.w);
            if (key >= lower && key < best) best = key;
          }
        }
        best = wave_min_u64(best);
        if (lane == t) mykey = best;
        lower = best + 1;
      }
    }
    // ---- accept iff the K-th distance is below what the unexamined cells can hold
    const uint32_t dk = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mykey >> 32), k - 1);
    if (whole || __uint_as_float(dk) < bnd) break;
  }
  const size_t o = ((size_t)bi * m + j) * k;
  if (lane < k) {
    idx[o + lane] = (IdxT)(uint32_t)mykey;
    if (dist_out) dist_out[o + lane] = __uint_as_float((uint32_t)(mykey >> 32));
  }
}

}  // namespace pasnl

using namespace pasnl;

extern "C" int pasnl_knn_batch(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                               int idx_is_i64, float* dist2, pasnl_stream_t stream);

extern "C" size_t pasnl_knn_workspace_bytes(int b, int n) {
  if (b <= 0 || n < PASNL_KNN_GRID_MIN_N || n > KG_BUILD_T * KG_PPT) return 0;
  return (size_t)b * kg_stride(n);
}

extern "C" int pasnl_knn_batch_ws(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                  int idx_is_i64, float* dist2, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  const size_t need = pasnl_knn_workspace_bytes(b, n);
  if (need == 0 || k > 64 || k > n || m <= 0)  // small clouds / wide lists: the brute-force kernels (same results)
    return pasnl_knn_batch(b, n, m, k, support, queries, idx, idx_is_i64, dist2, stream);
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(support && queries && idx, PASNL_ENULL);
  PASNL_REQUIRE(workspace != nullptr, PASNL_ENULL);
  PASNL_REQUIRE(workspace_bytes >= need, PASNL_EWORKSPACE);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  hipStream_t st = pasnl_hip_stream(stream);
  const size_t stride = kg_stride(n);
  // ~0.4 K records per cell: the sphere of radius h around a query (what ring 1 certifies) then holds ~1.7 K of them
  const float rho = fmaxf(4.f, 0.4f * (float)k);
  hipLaunchKernelGGL(knn_grid_build_kernel, dim3(b), dim3(KG_BUILD_T), 0, st, n, rho, support, static_cast<char*>(workspace), stride);
  dim3 grid((m + KG_WAVES - 1) / KG_WAVES, b), block(KG_WAVES * 64);
#define PASNL_KG(RR, T) hipLaunchKernelGGL((knn_grid_query_kernel<RR, T>), grid, block, 0, st, n, m, k, queries, \
                                           static_cast<const char*>(workspace), stride, static_cast<T*>(idx), dist2)
  if (k <= 32) { if (idx_is_i64) PASNL_KG(1, long long); else PASNL_KG(1, int); }
  else { if (idx_is_i64) PASNL_KG(2, long long); else PASNL_KG(2, int); }
#undef PASNL_KG
  return pasnl_launch_status();
}
