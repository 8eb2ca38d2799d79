// 3-NN search and inverse-distance interpolation for gfx950.
// Behaviour contract: reference tf_ops/3d_interpolation/tf_interpolate.cpp:60-153 (single-threaded CPU
// loops in the reference); restated in oracle/.
#include <math.h>
#include "common.hpp"

namespace pasnl {

// 64 unknown points per workgroup, one per lane; the workgroup's four waves each scan a contiguous quarter of the known cloud
// and the quarters' triples are merged in index order, so the result is the sequential scan's: the 3-deep cascade keeps
// strict '<' and equal distances keep the lower index (tf_interpolate.cpp:74-90).  The known points are wave-uniform: they
// arrive by scalar loads (12 dwords = 4 points per request) and feed the vector ALU as scalar operands -- no LDS, no barrier
// in the scan.  The cascade is branch-free (3 compares + 10 selects per pair): with 64 lanes some lane's top three changes at
// most steps (3/k per lane), so a skip test would rarely skip.  NaN distances fail every compare, as in the reference.
constexpr int NN_SPLIT = 4;

#define PASNL_NN_STEP(d_, k_)                                   \
  {                                                             \
    const float d = (d_);                                       \
    const int kk = (k_);                                        \
    const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;           \
    b3 = c2 ? b2 : (c3 ? d : b3);                               \
    i3 = c2 ? i2 : (c3 ? kk : i3);                              \
    b2 = c1 ? b1 : (c2 ? d : b2);                               \
    i2 = c1 ? i1 : (c2 ? kk : i2);                              \
    b1 = c1 ? d : b1;                                           \
    i1 = c1 ? kk : i1;                                          \
  }

__global__ __launch_bounds__(64 * NN_SPLIT) void three_nn_kernel(int n, int m, const float* __restrict__ xyz1,
                                                                const float* __restrict__ xyz2, float* __restrict__ dist,
                                                                int* __restrict__ idx) {
  __shared__ float mb[NN_SPLIT - 1][3][64];
  __shared__ int mi[NN_SPLIT - 1][3][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 64 + lane;
  const bool ok = j < n;
  const float* u = xyz1 + ((size_t)bi * n + (ok ? j : 0)) * 3;
  const float x1 = u[0], y1 = u[1], z1 = u[2];
  const float* __restrict__ kc = xyz2 + (size_t)bi * m * 3;
  const int per = ((m + NN_SPLIT - 1) / NN_SPLIT + 3) & ~3;  // a multiple of 4: only the cloud's end has a ragged tail
  const int lo = min(m, wave * per), hi = min(m, lo + per);

  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;  // the reference's 1e40 doubles == +inf as float
  int i1 = 0, i2 = 0, i3 = 0;
  int p = lo;
  if (p + 4 <= hi) {  // the next four points are requested before the current four are used
    float c[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) c[t] = kc[(size_t)p * 3 + t];
    for (; p + 4 <= hi; p += 4) {
      const int pn = p + 8 <= hi ? p + 4 : p;
      const float* __restrict__ s = kc + (size_t)pn * 3;
      float nx[12];
#pragma unroll
      for (int t = 0; t < 12; ++t) nx[t] = s[t];
#pragma unroll
      for (int t = 0; t < 4; ++t) PASNL_NN_STEP(dist2(c[3 * t], c[3 * t + 1], c[3 * t + 2], x1, y1, z1), p + t)
#pragma unroll
      for (int t = 0; t < 12; ++t) c[t] = nx[t];
    }
  }
  for (; p < hi; ++p) {
    const float* __restrict__ s = kc + (size_t)p * 3;
    PASNL_NN_STEP(dist2(s[0], s[1], s[2], x1, y1, z1), p)
  }
  if (wave > 0) {
    mb[wave - 1][0][lane] = b1; mb[wave - 1][1][lane] = b2; mb[wave - 1][2][lane] = b3;
    mi[wave - 1][0][lane] = i1; mi[wave - 1][1][lane] = i2; mi[wave - 1][2][lane] = i3;
  }
  __syncthreads();
  if (wave == 0) {
    // later quarters hold higher indices: inserting their (ascending) triples with strict '<' is the sequential scan
#pragma unroll
    for (int w = 0; w < NN_SPLIT - 1; ++w)
#pragma unroll
      for (int t = 0; t < 3; ++t) PASNL_NN_STEP(mb[w][t][lane], mi[w][t][lane])
    if (ok) {
      const size_t o = ((size_t)bi * n + j) * 3;
      dist[o] = b1; dist[o + 1] = b2; dist[o + 2] = b3;
      idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
    }
  }
}
#undef PASNL_NN_STEP

// out[row, l] = (p[i1,l]*w1 + p[i2,l]*w2) + p[i3,l]*w3 ; one thread per VEC channels of one output row.
template <int VEC>
__global__ __launch_bounds__(256) void three_interpolate_kernel(int m, int c, int n, long total_chunks,
                                                               const float* __restrict__ points,
                                                               const int* __restrict__ idx,
                                                               const float* __restrict__ weight, float* __restrict__ out) {
  const int cpr = c / VEC;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total_chunks; g += (long)gridDim.x * 256) {
    long row = g / cpr;
    int l = (int)(g - row * cpr) * VEC;
    long bi = row / n;
    const int* ip = idx + row * 3;
    const float* wp = weight + row * 3;
    const float w1 = wp[0], w2 = wp[1], w3 = wp[2];
    const float* base = points + (size_t)bi * m * c + l;
    const float* p1 = base + (size_t)ip[0] * c;
    const float* p2 = base + (size_t)ip[1] * c;
    const float* p3 = base + (size_t)ip[2] * c;
    float* o = out + (size_t)row * c + l;
    if constexpr (VEC == 4) {
      float4 a = *reinterpret_cast<const float4*>(p1), bq = *reinterpret_cast<const float4*>(p2),
             cq = *reinterpret_cast<const float4*>(p3), r;
      r.x = (a.x * w1 + bq.x * w2) + cq.x * w3;
      r.y = (a.y * w1 + bq.y * w2) + cq.y * w3;
      r.z = (a.z * w1 + bq.z * w2) + cq.z * w3;
      r.w = (a.w * w1 + bq.w * w2) + cq.w * w3;
      *reinterpret_cast<float4*>(o) = r;
    } else {
      *o = (*p1 * w1 + *p2 * w2) + *p3 * w3;
    }
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(int n, int c, int m, long total,
                                                                    const float* __restrict__ grad_out,
                                                                    const int* __restrict__ idx,
                                                                    const float* __restrict__ weight,
                                                                    float* __restrict__ grad_points) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long row = e / c;
    int l = (int)(e - row * c);
    long bi = row / n;
    float g = grad_out[e];
    float* base = grad_points + (size_t)bi * m * c + l;
    atomicAdd(base + (size_t)idx[row * 3 + 0] * c, g * weight[row * 3 + 0]);
    atomicAdd(base + (size_t)idx[row * 3 + 1] * c, g * weight[row * 3 + 1]);
    atomicAdd(base + (size_t)idx[row * 3 + 2] * c, g * weight[row * 3 + 2]);
  }
}

// d=max(d,1e-10); r=1/d; norm=(r0+r1)+r2; w=r/norm  (pointasnl_util.py:308-311; tf.reduce_sum over 3
// elements adds left to right)
__global__ __launch_bounds__(256) void three_weights_kernel(long rows, const float* __restrict__ dist,
                                                           float* __restrict__ weight) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float d0 = fmaxf(dist[r * 3], 1e-10f), d1 = fmaxf(dist[r * 3 + 1], 1e-10f), d2 = fmaxf(dist[r * 3 + 2], 1e-10f);
    float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
    float norm = (r0 + r1) + r2;
    weight[r * 3] = r0 / norm;
    weight[r * 3 + 1] = r1 / norm;
    weight[r * 3 + 2] = r2 / norm;
  }
}

// The head of a feature-propagation module in one pass (pointnet_util.py:212-219; pointasnl_util.py:308-313):
//   w = three_weights(dist);  out[row] = [ three_interpolate(points2, idx, w)[row] | points1[row] ]
// -- the weights (three_weights_kernel's arithmetic), the interpolation (three_interpolate_kernel's) and tf.concat's copy of
// the dense level's own features, written once at their place in the wider row: three launches and two passes over the
// interpolated features less.  A workgroup takes FP_ROWS rows: their weights and neighbours once into LDS, then one thread
// per 16 bytes of output.
constexpr int FP_ROWS = 16;
template <int VEC>
__global__ __launch_bounds__(256) void fp_interpolate_cat_kernel(int m, int c2, int n, int c1, long rows, uint32_t cpr_magic,
                                                                 const float* __restrict__ points2, const int* __restrict__ idx,
                                                                 const float* __restrict__ dist, const float* __restrict__ points1,
                                                                 float* __restrict__ out) {
  __shared__ float sw[FP_ROWS][3];
  __shared__ int si[FP_ROWS][3];
  const int ctot = c2 + c1, cpr = ctot / VEC, cpr2 = c2 / VEC;
  const int tid = threadIdx.x;
  for (long r0 = (long)blockIdx.x * FP_ROWS; r0 < rows; r0 += (long)gridDim.x * FP_ROWS) {
    __syncthreads();  // the previous tile's weights have been read
    if (tid < FP_ROWS && r0 + tid < rows) {
      const long r = r0 + tid;
      const float d0 = fmaxf(dist[r * 3], 1e-10f), d1 = fmaxf(dist[r * 3 + 1], 1e-10f), d2 = fmaxf(dist[r * 3 + 2], 1e-10f);
      const float q0 = 1.0f / d0, q1 = 1.0f / d1, q2 = 1.0f / d2;
      const float norm = (q0 + q1) + q2;
      sw[tid][0] = q0 / norm; sw[tid][1] = q1 / norm; sw[tid][2] = q2 / norm;
      si[tid][0] = idx[r * 3]; si[tid][1] = idx[r * 3 + 1]; si[tid][2] = idx[r * 3 + 2];
    }
    __syncthreads();
    const int nrow = (int)min((long)FP_ROWS, rows - r0);
    for (int e = tid; e < nrow * cpr; e += 256) {
      const int rr = cpr_magic ? (int)__umulhi((uint32_t)e, cpr_magic) : e;  // e / cpr (e < 2^16)
      const int ch = e - rr * cpr;
      const long row = r0 + rr;
      float* o = out + (size_t)row * ctot + (size_t)ch * VEC;
      if (ch < cpr2) {
        const long bi = row / n;
        const float w1 = sw[rr][0], w2 = sw[rr][1], w3 = sw[rr][2];
        const float* base = points2 + (size_t)bi * m * c2 + (size_t)ch * VEC;
        const float* p1 = base + (size_t)si[rr][0] * c2;
        const float* p2 = base + (size_t)si[rr][1] * c2;
        const float* p3 = base + (size_t)si[rr][2] * c2;
        if constexpr (VEC == 4) {
          const float4 a = *reinterpret_cast<const float4*>(p1), bq = *reinterpret_cast<const float4*>(p2),
                       cq = *reinterpret_cast<const float4*>(p3);
          float4 r;
          r.x = (a.x * w1 + bq.x * w2) + cq.x * w3;
          r.y = (a.y * w1 + bq.y * w2) + cq.y * w3;
          r.z = (a.z * w1 + bq.z * w2) + cq.z * w3;
          r.w = (a.w * w1 + bq.w * w2) + cq.w * w3;
          *reinterpret_cast<float4*>(o) = r;
        } else {
          *o = (*p1 * w1 + *p2 * w2) + *p3 * w3;
        }
      } else {
        const float* q = points1 + (size_t)row * c1 + (size_t)(ch - cpr2) * VEC;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(q);
        else *o = *q;
      }
    }
  }
}

}  // namespace pasnl

using namespace pasnl;

static int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int pasnl_three_nn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx,
                              pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n >= 0 && m >= 0, PASNL_EINVAL);
  if (b == 0 || n == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz1 && dist && idx && (m == 0 || xyz2), PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  hipLaunchKernelGGL(three_nn_kernel, dim3((n + 63) / 64, b), dim3(64 * NN_SPLIT), 0, pasnl_hip_stream(stream), n, m, xyz1, xyz2, dist,
                     idx);
  return pasnl_launch_status();
}

extern "C" int pasnl_three_interpolate(int b, int m, int c, int n, const float* points, const int* idx, const float* weight,
                                       float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && m > 0 && c > 0 && n >= 0, PASNL_EINVAL);
  long rows = (long)b * n;
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(points && idx && weight && out, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  bool vec4 = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  if (vec4) {
    long chunks = rows * (c / 4);
    hipLaunchKernelGGL(three_interpolate_kernel<4>, dim3(grid_for(chunks)), dim3(256), 0, st, m, c, n, chunks, points, idx,
                       weight, out);
  } else {
    long chunks = rows * c;
    hipLaunchKernelGGL(three_interpolate_kernel<1>, dim3(grid_for(chunks)), dim3(256), 0, st, m, c, n, chunks, points, idx,
                       weight, out);
  }
  return pasnl_launch_status();
}

extern "C" int pasnl_three_interpolate_grad(int b, int n, int c, int m, const float* grad_out, const int* idx,
                                            const float* weight, float* grad_points, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && m > 0 && c > 0 && n >= 0, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(grad_points, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  if (hipMemsetAsync(grad_points, 0, (size_t)b * m * c * sizeof(float), st) != hipSuccess) return PASNL_ELAUNCH;
  long total = (long)b * n * c;
  if (total == 0) return PASNL_OK;
  PASNL_REQUIRE(grad_out && idx && weight, PASNL_ENULL);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, c, m, total, grad_out, idx,
                     weight, grad_points);
  return pasnl_launch_status();
}

extern "C" int pasnl_three_weights(int rows, const float* dist, float* weight, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows >= 0, PASNL_EINVAL);
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(dist && weight, PASNL_ENULL);
  hipLaunchKernelGGL(three_weights_kernel, dim3(grid_for(rows)), dim3(256), 0, pasnl_hip_stream(stream), (long)rows, dist,
                     weight);
  return pasnl_launch_status();
}

extern "C" int pasnl_fp_interpolate_cat(int b, int m, int c2, int n, int c1, const float* points2, const int* idx, const float* dist,
                                        const float* points1, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && m > 0 && c2 > 0 && n >= 0 && c1 >= 0, PASNL_EINVAL);
  const long rows = (long)b * n;
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(points2 && idx && dist && out && (c1 == 0 || points1), PASNL_ENULL);
  PASNL_REQUIRE(c2 + c1 <= 4096, PASNL_EUNSUPPORTED);  // (FP_ROWS rows of 16-byte pieces stay below 2^16)
  hipStream_t st = pasnl_hip_stream(stream);
  const bool vec4 = c2 % 4 == 0 && c1 % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(points2) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(points1)) % 16 == 0);
  const unsigned cpr = (unsigned)(c2 + c1) / (vec4 ? 4u : 1u);
  const uint32_t magic = cpr == 1 ? 0u : (uint32_t)((0x100000000ull / cpr) + 1ull);
  const long tiles = (rows + FP_ROWS - 1) / FP_ROWS;
  const unsigned grid = (unsigned)(tiles > 16384 ? 16384 : tiles);
  if (vec4)
    hipLaunchKernelGGL(fp_interpolate_cat_kernel<4>, dim3(grid), dim3(256), 0, st, m, c2, n, c1, rows, magic, points2, idx, dist,
                       points1, out);
  else
    hipLaunchKernelGGL(fp_interpolate_cat_kernel<1>, dim3(grid), dim3(256), 0, st, m, c2, n, c1, rows, magic, points2, idx, dist,
                       points1, out);
  return pasnl_launch_status();
}
