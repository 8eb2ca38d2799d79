// fp32-grade products on the bf16 matrix pipe -- an explicit, separately labelled MODE (tf_util.DENSE_BF16X3, OFF by default
// and off for the headline benchmark): out = act(x . w + bias) for the long, thin GEMMs of the segmentation decoders and the
// after_conv layers (reference utils/pointasnl_util.py:275, 337: tf_util.conv2d over the flattened [nsample x channel] window;
// tf_util.py:120-185), which run at 128-148 TF on the vendor's fp32 kernels = 0.9 of the fp32 matrix peak -- every step of the
// segmentation models is bound by them.
//
// Every fp32 operand is the sum of three bf16 terms (hi + mid + lo = 24 mantissa bits); of the nine cross products the six
// whose weight is >= 2^-16 of the leading one are kept:  a b ~= ah bh + (ah bm + am bh) + (ah bl + al bh + am bm).  A product
// of two bf16 values is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result differs from an fp32
// fmaf chain by about one more rounding per product (tools/bf16x3_probe: worst error / sum |a b| 4.8e-7 against 2.3e-7,
// K = 4192) -- inside the 1e-5 contract, but NOT the same bits as fp32: hence a mode of its own.
//   * the weights are split ONCE (pasnl_bf16x3_split_weights: three bf16 planes in the MFMA's operand order, so that a lane's
//     eight k values of a column are 16 contiguous bytes);
//   * the activations are split on the fly, once per element, while a 128 x 32 tile goes from global memory into LDS (three
//     bf16 planes, 80-byte row pitch: conflict-free 16-byte reads): v_cvt_pk_bf16_f32 for the rounding, the remainder in fp32;
//   * a workgroup of EIGHT waves (__launch_bounds__(512, 4)) owns a 128 x 128 tile of the output, a wave a 64 x 32 slab (two
//     blocks of 32 x 32: two accumulators); per 32 contraction indices a wave issues 24 matrix instructions (six products x two
//     k-halves x two row blocks) from 12 LDS reads (A: three planes x two row blocks x two k-halves) and 6 global reads (W:
//     L2-resident, requested one k-step ahead); the next tile of A is requested before the products of the current one and
//     split / stored behind them (double-buffered LDS, two tiles in flight).  104 VGPRs, 61 KB of LDS: two workgroups per CU.
#include "common.hpp"

namespace pasnl {

typedef float bx_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BX_BM = 128, BX_BN = 128, BX_BK = 32;
constexpr int BX_PITCH = 40;  // bf16 per LDS row (32 + 8 of padding = 80 bytes: 16-byte reads of 32 rows hit every bank group once)

__device__ __forceinline__ float bx_bf16_to_f32(__bf16 h) { return (float)h; }

// x -> (hi, mid, lo) with hi + mid + lo == x up to 2^-24 relative (round to nearest even at every step)
__device__ __forceinline__ void bx_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - bx_bf16_to_f32(h);
  m = (__bf16)r1;
  l = (__bf16)(r1 - bx_bf16_to_f32(m));
}

// Weights (K, N) fp32 -> three planes in operand order: plane p, k-group g = k / 8, column n, the 8 values k = 8 g .. 8 g + 7:
// out[((p * (K / 8) + g) * N + n) * 8 + j].  One thread per (g, n).
__global__ __launch_bounds__(256) void bx_split_weights_kernel(int K, int N, const float* __restrict__ w, __bf16* __restrict__ out) {
  const long total = (long)(K / 8) * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int g = (int)(e / N), n = (int)(e - (long)g * N);
    bx_bf16x8 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, l;
      bx_split3(w[(size_t)(8 * g + j) * N + n], h, m, l);
      ph[j] = h; pm[j] = m; pl[j] = l;
    }
    const size_t plane = (size_t)(K / 8) * N * 8;
    bx_bf16x8* o = reinterpret_cast<bx_bf16x8*>(out);
    o[e] = ph;
    o[plane / 8 + e] = pm;
    o[2 * (plane / 8) + e] = pl;
  }
}

// out (M, N) = act(x (M, K; row stride lda) . w + bias);  K % 32 == 0, N % 128 == 0, x and wsplit 16-byte aligned
// 8 waves: wave (wr, wc) owns rows 64 wr .. + 63, columns 32 wc .. + 31 of the tile (two accumulators); two workgroups per CU
// (61 KB of LDS each, <= 128 registers): four waves per SIMD, so that one wave's split / LDS / barrier phases lie under
// the others' products.
__global__ __launch_bounds__(512, 4) void bx_gemm_kernel(int M, int K, int N, int lda, const float* __restrict__ x,
                                                         const __bf16* __restrict__ ws, const float* __restrict__ bias, int relu,
                                                         float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BX_BM][BX_PITCH];  // [buffer][plane][row][k]: 2 x 30 720 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int wr = wave >> 2, wc = wave & 3;
  const long row0 = (long)blockIdx.y * BX_BM;
  const int col0 = blockIdx.x * BX_BN;
  // ---- this thread's part of an A tile: rows tid / 8 + 64 i (i < 2), the four floats k = 4 (tid % 8) .. + 3
  const int ar = tid >> 3, ak = (tid & 7) * 4;
  const float* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) arow[i] = x + (size_t)min(row0 + ar + 64 * i, (long)M - 1) * lda + ak;
  auto request = [&](int chunk, float4 (&av)[2]) {  // (past the end: the last chunk again -- split into the idle buffer, unused)
    const int k0 = min(chunk * BX_BK, K - BX_BK);
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const float4*>(arow[i] + k0);
  };
  auto split_store = [&](int buf, const float4 (&av)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
      __bf16 ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bx_split3(v[j], ph[j], pm[j], pl[j]);
      const int r = ar + 64 * i;
      *reinterpret_cast<uint2*>(&As[buf][0][r][ak]) = *reinterpret_cast<const uint2*>(ph);
      *reinterpret_cast<uint2*>(&As[buf][1][r][ak]) = *reinterpret_cast<const uint2*>(pm);
      *reinterpret_cast<uint2*>(&As[buf][2][r][ak]) = *reinterpret_cast<const uint2*>(pl);
    }
  };
  // ---- the wave's W operands of one k-step (16 indices), planes 0 .. 2: 16 bytes per lane and plane; requested one step ahead
  // of their products (the other three waves of the SIMD cover the L2 latency), two register sets, no copies
  const size_t plane8 = (size_t)(K / 8) * N;  // 16-byte items per plane
  const bx_bf16x8* wsv = reinterpret_cast<const bx_bf16x8*>(ws) + (size_t)h * N + col0 + wc * 32 + l32;
  const int nstep = K / 16;
  auto load_w = [&](int step, bx_bf16x8 (&wb)[3]) {
    const size_t g = (size_t)(2 * min(step, nstep - 1)) * N;
#pragma unroll
    for (int p = 0; p < 3; ++p) wb[p] = wsv[p * plane8 + g];
  };
  bx_f32x16 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  // the products of one k-step: the six terms, smallest first -- (al bh, ah bl, am bm), (am bh, ah bm), ah bh --, alternating
  // between the two accumulators
  auto products = [&](int buf, int s, const bx_bf16x8 (&wb)[3]) {
    bx_bf16x8 a[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        a[p][rb] = *reinterpret_cast<const bx_bf16x8*>(&As[buf][p][wr * 64 + rb * 32 + l32][s * 16 + h * 8]);
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TW[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][rb], wb[TW[t]], acc[rb], 0, 0, 0);
  };
  bx_bf16x8 w0[3], w1[3];  // W operands of the even / odd k-steps
  float4 ava[2], avb[2];   // A tiles in flight: requested two chunks ahead, split one chunk ahead -- next to the products,
                           // in one basic block, so that the split's vector work is scheduled between the matrix instructions
  request(0, ava);
  load_w(0, w0);
  request(1, avb);
  split_store(0, ava);
  __syncthreads();
  const int nchunk = K / BX_BK;
  for (int c = 0; c < nchunk; c += 2) {
    request(c + 2, ava);
    load_w(2 * c + 1, w1);
    products(0, 0, w0);
    split_store(1, avb);
    load_w(2 * c + 2, w0);
    products(0, 1, w1);
    __syncthreads();
    if (c + 1 >= nchunk) break;
    request(c + 3, avb);
    load_w(2 * c + 3, w1);
    products(1, 0, w0);
    split_store(0, ava);
    load_w(2 * c + 4, w0);
    products(1, 1, w1);
    __syncthreads();
  }
  // ---- bias, activation, store: D[row (r & 3) + 8 (r >> 2) + 4 h][column l32] -- 128 contiguous bytes per half-wave and row
  const int col = col0 + wc * 32 + l32;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = row0 + wr * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[rb][r] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (row < M) out[(size_t)row * N + col] = v;
    }
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_bf16x3_weights_bytes(int kdim, int n) {
  if (kdim <= 0 || n <= 0) return 0;
  return (size_t)3 * kdim * n * 2;
}

extern "C" int pasnl_bf16x3_split_weights(int kdim, int n, const float* w, void* wsplit, pasnl_stream_t stream) {
  PASNL_REQUIRE(kdim > 0 && n > 0, PASNL_EINVAL);
  PASNL_REQUIRE(w && wsplit, PASNL_ENULL);
  PASNL_REQUIRE(kdim % 8 == 0 && reinterpret_cast<uintptr_t>(wsplit) % 16 == 0, PASNL_EUNSUPPORTED);
  const long total = (long)(kdim / 8) * n;
  const long g = (total + 255) / 256;
  hipLaunchKernelGGL(bx_split_weights_kernel, dim3((unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g))), dim3(256), 0,
                     pasnl_hip_stream(stream), kdim, n, w, static_cast<__bf16*>(wsplit));
  return pasnl_launch_status();
}

extern "C" int pasnl_dense_bf16x3(int rows, int kdim, int n, int lda, const float* x, const void* wsplit, const float* bias, int relu,
                                  float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows >= 0 && kdim > 0 && n > 0 && lda >= kdim, PASNL_EINVAL);
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(x && wsplit && out, PASNL_ENULL);
  PASNL_REQUIRE(kdim % BX_BK == 0 && n % BX_BN == 0 && lda % 4 == 0, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(wsplit) % 16 == 0, PASNL_EUNSUPPORTED);
  const long mt = ((long)rows + BX_BM - 1) / BX_BM;
  PASNL_REQUIRE(mt <= 65535, PASNL_EUNSUPPORTED);
  hipLaunchKernelGGL(bx_gemm_kernel, dim3(n / BX_BN, (unsigned)mt), dim3(512), 0, pasnl_hip_stream(stream), rows, kdim, n, lda, x,
                     static_cast<const __bf16*>(wsplit), bias, relu, out);
  return pasnl_launch_status();
}
