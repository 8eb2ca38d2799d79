// Sampling ops for gfx950: farthest point sampling, 3-wide gather (+grad), probability sampling.
// Behaviour contract: reference tf_ops/sampling/tf_sampling_g.cu (kernels :7-192), restated in oracle/.
// The design is NOT the reference's: FPS keeps every point and its running min-distance in VGPRs,
// reduces with DPP inside a wave and one LDS hop across waves, one workgroup per cloud.
#include "common.hpp"

namespace pasnl {

// ---------------------------------------------------------------------------------------------
// Farthest point sampling.
//   One workgroup (WAVES x 64 lanes) owns one cloud; lane t holds points k = i*T + t, i < PPL.
//   Round j:  read the last pick's coordinates from the LDS copy of the cloud (broadcast read),
//             update PPL running distances in registers, form the 64-bit key
//                 (bits(d2) << 32) | ~tiekey(k),  tiekey(k) = ((k & 511) << 22) | k
//             whose maximum is exactly the reference's winner: largest d2, then lowest k mod 512,
//             then lowest k (tf_sampling_g.cu:142-164),
//             DPP-reduce inside the wave, exchange WAVES partial keys through a double-buffered LDS
//             slot (one barrier per round).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fps_tiekey(int k) { return (((uint32_t)k & 511u) << 22) | (uint32_t)k; }
__device__ __forceinline__ int fps_key_to_index(uint64_t key) { return (int)((~(uint32_t)key) & ((1u << 22) - 1u)); }

template <int WAVES, int PPL>
__global__ __launch_bounds__(WAVES * 64) void fps_kernel(int n, int m, const float* __restrict__ xyz,
                                                        int* __restrict__ idx) {
  constexpr int T = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [2][WAVES] u64 exchange slots (16-byte aligned block), then the cloud as x|y|z planes
  uint64_t* slots = reinterpret_cast<uint64_t*>(smem);
  constexpr int SLOT_BYTES = ((2 * WAVES * 8 + 15) / 16) * 16;
  float* sx = reinterpret_cast<float*>(smem + SLOT_BYTES);
  float* sy = sx + n;
  float* sz = sy + n;

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const float* cloud = xyz + (size_t)blockIdx.x * n * 3;
  int* out = idx + (size_t)blockIdx.x * m;

  // coalesced flat copy of the AoS cloud into SoA planes
  for (int f = tid; f < n * 3; f += T) {
    float v = cloud[f];
    int p = f / 3, c = f - p * 3;
    (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
  }
  __syncthreads();

  float px[PPL], py[PPL], pz[PPL], td[PPL];
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    int k = i * T + tid;
    bool ok = k < n;
    px[i] = ok ? sx[k] : 0.f;
    py[i] = ok ? sy[k] : 0.f;
    pz[i] = ok ? sz[k] : 0.f;
    td[i] = 1e38f;
  }

  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = sx[old], y1 = sy[old], z1 = sz[old];
    uint64_t best = 0;  // every real point has a key > 0 (tiekey < 2^31 => ~tiekey != 0)
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      int k = i * T + tid;
      float d = dist2(px[i], py[i], pz[i], x1, y1, z1);
      float d2 = fminf(d, td[i]);
      td[i] = d2;
      uint64_t key = ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)(~fps_tiekey(k));
      key = k < n ? key : 0;
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if constexpr (WAVES > 1) {
      uint64_t* slot = slots + (j & 1) * WAVES;
      if ((tid & 63) == 0) slot[wave] = best;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        uint64_t o = slot[w];
        best = o > best ? o : best;
      }
    }
    old = fps_key_to_index(best);
    if (tid == 0) out[j] = old;
  }
}

template <int WAVES, int PPL>
static int fps_launch(int b, int n, int m, const float* xyz, int* idx, hipStream_t st) {
  size_t lds = ((2 * WAVES * 8 + 15) / 16) * 16 + (size_t)n * 12;
  auto kern = fps_kernel<WAVES, PPL>;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return PASNL_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(b), dim3(WAVES * 64), lds, st, n, m, xyz, idx);
  return pasnl_launch_status();
}

// ---------------------------------------------------------------------------------------------
// gather_point / grad
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_point_kernel(int n, int m, long total, const float* __restrict__ inp,
                                                          const int* __restrict__ idx, float* __restrict__ out) {
  // one thread per output float: coalesced stores, 12-byte source rows stay in L1/L2
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long row = e / 3;
    int c = (int)(e - row * 3);
    long bi = row / m;
    int a = idx[row];
    out[e] = inp[(bi * n + a) * 3 + c];
  }
}

__global__ __launch_bounds__(256) void gather_point_grad_kernel(int n, int m, long total, const float* __restrict__ out_g,
                                                               const int* __restrict__ idx, float* __restrict__ inp_g) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long row = e / 3;
    int c = (int)(e - row * 3);
    long bi = row / m;
    int a = idx[row];
    atomicAdd(&inp_g[(bi * n + a) * 3 + c], out_g[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// prob_sample = per-row running sum + binary search (tf_sampling_g.cu:7-104).
// The reference's running sum is a blocked scan whose fp32 association is part of the contract
// (a different association changes cdf bits and therefore the searched index).  The kernel below
// reproduces that association: tiles of 8192 elements; inside a tile groups of 4 are summed
// left-to-right, group totals go through the up-sweep/down-sweep binary tree over n2 = ceil(len/4)
// leaves, element = in-group prefix + tree prefix of the previous group, + compensated carry of
// earlier tiles (runningsum / runningsum2, :81-84).
// ---------------------------------------------------------------------------------------------
constexpr int PS_TILE = 8192;   // BlockSize*4 in the reference
constexpr int PS_GROUPS = 2048;

__global__ __launch_bounds__(512) void cumsum_kernel(int n, const float* __restrict__ inp, float* __restrict__ out) {
  __shared__ float g4[PS_TILE];
  __shared__ float tree[PS_GROUPS];
  const float* row = inp + (size_t)blockIdx.x * n;
  float* orow = out + (size_t)blockIdx.x * n;
  float runningsum = 0.f, runningsum2 = 0.f;
  for (int j = 0; j < n; j += PS_TILE) {
    int len = min(n - j, PS_TILE);
    int len4 = (len + 3) & ~3;
    int n2 = len4 >> 2;
    for (int g = threadIdx.x; g < n2; g += 512) {
      int k = g * 4;
      if (k + 3 < len) {
        float v1 = row[j + k], v2 = row[j + k + 1], v3 = row[j + k + 2], v4 = row[j + k + 3];
        v2 += v1;
        v4 += v3;
        v3 += v2;
        v4 += v2;
        g4[k] = v1; g4[k + 1] = v2; g4[k + 2] = v3; g4[k + 3] = v4;
        tree[g] = v4;
      } else {
        float v = 0.f;
        for (int k2 = k; k2 < len; ++k2) { v += row[j + k2]; g4[k2] = v; }
        for (int k2 = len; k2 < len4; ++k2) g4[k2] = v;
        tree[g] = v;
      }
    }
    int u = 0;
    for (; (2 << u) <= n2; ++u) {
      __syncthreads();
      for (int k = threadIdx.x; k < (n2 >> (u + 1)); k += 512) {
        int i1 = (((k << 1) + 2) << u) - 1, i2 = (((k << 1) + 1) << u) - 1;
        tree[i1] += tree[i2];
      }
    }
    --u;
    for (; u >= 0; --u) {
      __syncthreads();
      for (int k = threadIdx.x; k < ((n2 - (1 << u)) >> (u + 1)); k += 512) {
        int i1 = (((k << 1) + 3) << u) - 1, i2 = (((k << 1) + 2) << u) - 1;
        tree[i1] += tree[i2];
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < len; k += 512) {
      float v = g4[k];
      if (k >= 4) v += tree[(k >> 2) - 1];
      orow[j + k] = v + runningsum;
    }
    float t = tree[n2 - 1] + runningsum2;
    float r2 = runningsum + t;
    runningsum2 = t - (r2 - runningsum);
    runningsum = r2;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void binary_search_kernel(int n, int m, const float* __restrict__ cdf,
                                                           const float* __restrict__ query, int* __restrict__ result) {
  int base = 1;
  while (base < n) base <<= 1;
  const float* c = cdf + (size_t)blockIdx.y * n;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
    float q = query[(size_t)blockIdx.y * m + j] * c[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
      if (r >= k && c[r - k] >= q) r -= k;
    result[(size_t)blockIdx.y * m + j] = r;
  }
}

}  // namespace pasnl

using namespace pasnl;

extern "C" int pasnl_farthest_point_sample(int b, int n, int m, const float* xyz, int* idx, pasnl_stream_t stream) {
  PASNL_REQUIRE(m > 0, PASNL_EINVAL);  // "FarthestPointSample expects positive npoint"
  PASNL_REQUIRE(b >= 0 && n > 0, PASNL_EINVAL);
  PASNL_REQUIRE(n < (1 << 22), PASNL_EUNSUPPORTED);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && idx, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  // lanes x points-per-lane must cover n; LDS holds the whole cloud (12 B/point, <= 160 KiB)
  if (n <= 64) return fps_launch<1, 1>(b, n, m, xyz, idx, st);
  if (n <= 256) return fps_launch<1, 4>(b, n, m, xyz, idx, st);
  if (n <= 512) return fps_launch<2, 4>(b, n, m, xyz, idx, st);
  if (n <= 1024) return fps_launch<4, 4>(b, n, m, xyz, idx, st);
  if (n <= 2048) return fps_launch<8, 4>(b, n, m, xyz, idx, st);
  if (n <= 4096) return fps_launch<16, 4>(b, n, m, xyz, idx, st);
  if (n <= 8192) return fps_launch<16, 8>(b, n, m, xyz, idx, st);
  if (n <= 12288) return fps_launch<16, 12>(b, n, m, xyz, idx, st);
  return PASNL_EUNSUPPORTED;  // > 12288 points/cloud: LDS-resident design limit (reference configs <= 10240)
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int pasnl_gather_point(int b, int n, int m, const float* inp, const int* idx, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0, PASNL_EINVAL);
  long total = (long)b * m * 3;
  if (total == 0) return PASNL_OK;
  PASNL_REQUIRE(inp && idx && out, PASNL_ENULL);
  hipLaunchKernelGGL(gather_point_kernel, dim3(grid_for(total)), dim3(256), 0, pasnl_hip_stream(stream), n, m, total, inp,
                     idx, out);
  return pasnl_launch_status();
}

extern "C" int pasnl_gather_point_grad(int b, int n, int m, const float* out_g, const int* idx, float* inp_g,
                                       pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(inp_g, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  if (hipMemsetAsync(inp_g, 0, (size_t)b * n * 3 * sizeof(float), st) != hipSuccess) return PASNL_ELAUNCH;
  long total = (long)b * m * 3;
  if (total == 0) return PASNL_OK;
  PASNL_REQUIRE(out_g && idx, PASNL_ENULL);
  hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, m, total, out_g, idx, inp_g);
  return pasnl_launch_status();
}

extern "C" int pasnl_prob_sample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out,
                                 pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(inp_p && temp, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  hipLaunchKernelGGL(cumsum_kernel, dim3(b), dim3(512), 0, st, n, inp_p, temp);
  if (m > 0) {
    PASNL_REQUIRE(inp_r && out, PASNL_ENULL);
    int gx = (m + 255) / 256;
    hipLaunchKernelGGL(binary_search_kernel, dim3(gx > 64 ? 64 : gx, b), dim3(256), 0, st, n, m, temp, inp_r, out);
  }
  return pasnl_launch_status();
}
