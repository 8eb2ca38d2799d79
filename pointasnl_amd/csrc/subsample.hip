// Voxel-grid subsampling for gfx950: the input stage of the ScanNet / SemanticKITTI "grid"
// pipelines.  Behaviour contract: reference utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:4-106
// (a sequential loop filling an unordered_map); restated in oracle/.
//
// What is kept bit for bit: the voxel key of every point (origin = floor(min * (1/dl)) * dl, i = floor((p - origin) / dl),
// key = iX + NX*iY + NX*NY*iZ, all in fp32 as written), the fp32 sums of a voxel's points / features IN INPUT ORDER,
// barycentre = sum * float(1.0 / count), feature mean = sum / float(count).
// What is defined here because the reference leaves it to the hash table: the OUTPUT ORDER (ascending voxel key; the
// reference emits unordered_map iteration order) and the label vote on ties (smallest label; the reference takes the
// first maximum in hash order).
//
// Pipeline (all on the caller's stream, no host synchronisation):
//   minmax partials -> keys -> stable radix sort of (key, point index) [rocPRIM] -> head flags -> inclusive scan
//   [rocPRIM] -> segment starts -> one thread per (voxel, output column) walks its members in input order.
#include <cstring>
#include <math.h>
#include <rocprim/rocprim.hpp>
#include "common.hpp"

namespace pasnl {

constexpr int GS_PARTS = 1024;  // min/max partials

__global__ __launch_bounds__(256) void gs_minmax_kernel(long n, const float* __restrict__ pts, float* __restrict__ part) {
  __shared__ float red[4][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[i * 3 + a];
      lo[a] = v < lo[a] ? v : lo[a];  // the reference's strict comparisons (cloud.cpp:30-67)
      hi[a] = v > hi[a] ? v : hi[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
    part[blockIdx.x * 6 + a] = v;
  }
}

// grid geometry from the partials: origin (3), NX, NY -- recomputed by every block (a few hundred floats)
struct GsGrid {
  float ox, oy, oz;
  unsigned long long nx, ny;
};
__device__ __forceinline__ GsGrid gs_grid(const float* __restrict__ part, int nparts, float dl, float* sh) {
  // sh: 6 floats of shared memory
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = part[a];
    for (int b = 1; b < nparts; ++b) v = a < 3 ? fminf(v, part[b * 6 + a]) : fmaxf(v, part[b * 6 + a]);
    sh[a] = v;
  }
  __syncthreads();
  GsGrid g;
  const float inv = 1 / dl;                                   // grid_subsampling.cpp:25: minCorner * (1/sampleDl)
  g.ox = floorf(sh[0] * inv) * dl;
  g.oy = floorf(sh[1] * inv) * dl;
  g.oz = floorf(sh[2] * inv) * dl;
  g.nx = (unsigned long long)floorf((sh[3] - g.ox) / dl) + 1;  // :28-29
  g.ny = (unsigned long long)floorf((sh[4] - g.oy) / dl) + 1;
  return g;
}

__global__ __launch_bounds__(256) void gs_keys_kernel(long n, float dl, const float* __restrict__ pts,
                                                     const float* __restrict__ part, int nparts,
                                                     unsigned long long* __restrict__ keys, unsigned int* __restrict__ vals) {
  __shared__ float sh[6];
  const GsGrid g = gs_grid(part, nparts, dl, sh);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const unsigned long long ix = (unsigned long long)floorf((pts[i * 3] - g.ox) / dl);      // :52-55
    const unsigned long long iy = (unsigned long long)floorf((pts[i * 3 + 1] - g.oy) / dl);
    const unsigned long long iz = (unsigned long long)floorf((pts[i * 3 + 2] - g.oz) / dl);
    keys[i] = ix + g.nx * iy + g.nx * g.ny * iz;
    vals[i] = (unsigned int)i;
  }
}

__global__ __launch_bounds__(256) void gs_heads_kernel(long n, const unsigned long long* __restrict__ keys,
                                                      unsigned int* __restrict__ head) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void gs_starts_kernel(long n, const unsigned int* __restrict__ head,
                                                       const unsigned int* __restrict__ rank, unsigned int* __restrict__ start,
                                                       int* __restrict__ out_count) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (head[i]) start[rank[i] - 1] = (unsigned int)i;
    if (i == n - 1) {
      start[rank[i]] = (unsigned int)n;
      out_count[0] = (int)rank[i];
    }
  }
}

// one thread per (voxel, output column): columns 0..2 barycentre, 3..3+fdim-1 feature means, then ldim label votes
__global__ __launch_bounds__(256) void gs_reduce_kernel(long n, int fdim, int ldim, const float* __restrict__ pts,
                                                       const float* __restrict__ feats, const int* __restrict__ cls,
                                                       const unsigned int* __restrict__ order,
                                                       const unsigned int* __restrict__ start,
                                                       const int* __restrict__ out_count, float* __restrict__ out_pts,
                                                       float* __restrict__ out_feats, int* __restrict__ out_cls) {
  const int ncol = 3 + fdim + ldim;
  const long total = (long)out_count[0] * ncol;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long v = e / ncol;
    const int col = (int)(e - v * ncol);
    const unsigned int lo = start[v], hi = start[v + 1];
    const int count = (int)(hi - lo);
    if (col < 3 + fdim) {
      float sum = 0.f;  // PointXYZ() / vector<float>(fdim) start at zero, members are added in input order
      for (unsigned int j = lo; j < hi; ++j) {
        const unsigned int i = order[j];
        sum += col < 3 ? pts[(size_t)i * 3 + col] : feats[(size_t)i * fdim + (col - 3)];
      }
      if (col < 3) out_pts[v * 3 + col] = sum * (float)(1.0 / count);              // :86: point * (1.0 / count)
      else out_feats[v * fdim + (col - 3)] = sum / (float)count;                    // :89-93
    } else {
      const int l = col - 3 - fdim;
      int best = 0, best_count = 0;
      for (unsigned int j = lo; j < hi; ++j) {
        const int lab = cls[(size_t)order[j] * ldim + l];
        int cnt = 0;
        for (unsigned int j2 = lo; j2 < hi; ++j2) cnt += cls[(size_t)order[j2] * ldim + l] == lab;
        if (cnt > best_count || (cnt == best_count && lab < best)) { best = lab; best_count = cnt; }
      }
      out_cls[v * ldim + l] = best;
    }
  }
}

static size_t gs_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct GsLayout {
  size_t part, keys_in, keys_out, vals_in, vals_out, head, rank, start, temp, temp_bytes, total;
};

static GsLayout gs_layout(long n) {
  GsLayout L{};
  size_t sort_bytes = 0, scan_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                  (unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)n,
                                rocprim::plus<unsigned int>(), (hipStream_t)0);
  size_t off = 0;
  L.part = off; off += gs_align((size_t)GS_PARTS * 6 * 4);
  L.keys_in = off; off += gs_align((size_t)n * 8);
  L.keys_out = off; off += gs_align((size_t)n * 8);
  L.vals_in = off; off += gs_align((size_t)n * 4);
  L.vals_out = off; off += gs_align((size_t)n * 4);
  L.head = off; off += gs_align((size_t)n * 4);
  L.rank = off; off += gs_align((size_t)n * 4);
  L.start = off; off += gs_align((size_t)(n + 1) * 4);
  L.temp = off;
  L.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  off += gs_align(L.temp_bytes);
  L.total = off;
  return L;
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_grid_subsample_workspace_bytes(long n) { return n > 0 ? gs_layout(n).total : 0; }

extern "C" int pasnl_grid_subsample(long n, int fdim, int ldim, const float* points, const float* features, const int* classes,
                                    float sample_dl, float* out_points, float* out_features, int* out_classes, int* out_count,
                                    void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(n >= 0 && fdim >= 0 && ldim >= 0 && sample_dl > 0.f, PASNL_EINVAL);
  PASNL_REQUIRE(n < (1L << 31), PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(out_count, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  if (n == 0) return hipMemsetAsync(out_count, 0, sizeof(int), st) == hipSuccess ? PASNL_OK : PASNL_ELAUNCH;
  PASNL_REQUIRE(points && out_points && (fdim == 0 || (features && out_features)) && (ldim == 0 || (classes && out_classes)),
                PASNL_ENULL);
  const GsLayout L = gs_layout(n);
  PASNL_REQUIRE(workspace && workspace_bytes >= L.total, PASNL_EWORKSPACE);
  char* ws = static_cast<char*>(workspace);
  float* part = reinterpret_cast<float*>(ws + L.part);
  auto* keys_in = reinterpret_cast<unsigned long long*>(ws + L.keys_in);
  auto* keys_out = reinterpret_cast<unsigned long long*>(ws + L.keys_out);
  auto* vals_in = reinterpret_cast<unsigned int*>(ws + L.vals_in);
  auto* vals_out = reinterpret_cast<unsigned int*>(ws + L.vals_out);
  auto* head = reinterpret_cast<unsigned int*>(ws + L.head);
  auto* rank = reinterpret_cast<unsigned int*>(ws + L.rank);
  auto* start = reinterpret_cast<unsigned int*>(ws + L.start);
  void* temp = ws + L.temp;
  size_t temp_bytes = L.temp_bytes;

  const int nparts = (int)std::min<long>(GS_PARTS, (n + 255) / 256);
  const unsigned grid = (unsigned)std::min<long>(4096, (n + 255) / 256);
  hipLaunchKernelGGL(gs_minmax_kernel, dim3(nparts), dim3(256), 0, st, n, points, part);
  hipLaunchKernelGGL(gs_keys_kernel, dim3(grid), dim3(256), 0, st, n, sample_dl, points, part, nparts, keys_in, vals_in);
  // rocPRIM's radix sort is stable: inside a voxel the point indices stay ascending = the reference's input order
  if (rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 64, st) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(gs_heads_kernel, dim3(grid), dim3(256), 0, st, n, keys_out, head);
  temp_bytes = L.temp_bytes;
  if (rocprim::inclusive_scan(temp, temp_bytes, head, rank, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(gs_starts_kernel, dim3(grid), dim3(256), 0, st, n, head, rank, start, out_count);
  const long cols = (long)n * (3 + fdim + ldim);  // upper bound: as many voxels as points
  hipLaunchKernelGGL(gs_reduce_kernel, dim3((unsigned)std::min<long>(8192, (cols + 255) / 256)), dim3(256), 0, st, n, fdim, ldim,
                     points, features, classes, vals_out, start, out_count, out_points, out_features, out_classes);
  return pasnl_launch_status();
}
