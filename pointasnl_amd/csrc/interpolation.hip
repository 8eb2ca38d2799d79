// 3-NN search and inverse-distance interpolation for gfx950.
// Behaviour contract: reference tf_ops/3d_interpolation/tf_interpolate.cpp:60-153 (single-threaded CPU
// loops in the reference); restated in oracle/.
#include <math.h>
#include "common.hpp"

namespace pasnl {

constexpr int NN_TILE = 2048;  // known points per LDS tile, 16 B each (x,y,z,pad) = 32 KiB

// One unknown point per lane; the known cloud streams through LDS as float4 and every lane reads the
// same element (LDS broadcast, one b128 read per pair).  The 3-deep cascade keeps strict '<' so equal
// distances keep the lower index (tf_interpolate.cpp:74-90).
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float* __restrict__ xyz1,
                                                      const float* __restrict__ xyz2, float* __restrict__ dist,
                                                      int* __restrict__ idx) {
  __shared__ float4 known[NN_TILE];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool ok = j < n;
  const float* u = xyz1 + ((size_t)bi * n + (ok ? j : 0)) * 3;
  const float x1 = u[0], y1 = u[1], z1 = u[2];
  const float* kc = xyz2 + (size_t)bi * m * 3;

  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;  // the reference's 1e40 doubles == +inf as float
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += NN_TILE) {
    int tcnt = min(NN_TILE, m - base);
    __syncthreads();
    for (int p = threadIdx.x; p < tcnt; p += 256) {
      const float* s = kc + (size_t)(base + p) * 3;
      known[p] = make_float4(s[0], s[1], s[2], 0.f);
    }
    __syncthreads();
    for (int p = 0; p < tcnt; ++p) {
      float4 q = known[p];
      float d = dist2(q.x, q.y, q.z, x1, y1, z1);
      int k = base + p;
      if (d < b1) {
        b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
      } else if (d < b2) {
        b3 = b2; i3 = i2; b2 = d; i2 = k;
      } else if (d < b3) {
        b3 = d; i3 = k;
      }
    }
  }
  if (ok) {
    size_t o = ((size_t)bi * n + j) * 3;
    dist[o] = b1; dist[o + 1] = b2; dist[o + 2] = b3;
    idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
  }
}

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
  hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, b), dim3(256), 0, pasnl_hip_stream(stream), n, m, xyz1, xyz2, dist,
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
