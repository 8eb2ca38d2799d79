// The "group all" PointNet set-abstraction module of the classifier in ONE kernel (+ a pooling pass over tile maxima):
// reference utils/pointnet_util.py:87-137 as called at models/pointasnl_cls.py:39-40 (layer3_1: 512 points x [128, 256, 512],
// layer3_2: 128 points x [256, 512, 1024]) -- sample_and_group_all (every point of the cloud is one group), three 1x1
// convolutions (BN folded, ReLU) and tf.reduce_max over the group.  Round 4 ran it as three vendor GEMMs + a pooling kernel
// per module: eight launches at 58-123 TF whose 256 tiles (one per CU) stretch by whatever shares a CU with them (the next
// batch's sampler: 78 -> 141 us for the widest product).
//
// One workgroup of 8 waves owns a tile of 32 points of one cloud and carries it through all three convolutions:
//   X (32 x k0) -> LDS buffer A;  H1 = relu(X W0 + b0) -> LDS buffer B;  H2 = relu(H1 W1 + b1) -> buffer A;
//   H3 = relu(H2 W2 + b2) never leaves the registers: its column maxima over the tile's 32 rows go to partial[cloud][tile][:].
// v_mfma_f32_32x32x2_f32 with A = the activations (row ql of the tile, LDS, odd row pitch: conflict-free), B = the weights
// (rows 2 s + h of W, straight from global memory / L2: a half-wave reads 128 contiguous bytes, the row's address is a scalar),
// D[m = row kappa(r, h)][n = channel]: the bias is one value per lane, the store of a block is column-contiguous, and the
// pooled maximum is a maximum over a lane's 16 accumulators and one exchange between the wave's halves.  A wave owns the
// 32-channel blocks wave, wave + 8, ...: one LDS operand feeds up to four products.  Operands of the NEXT eight steps are
// requested while eight steps multiply (two register sets).  The grid has 4 (n = 128) or 16 (n = 512) tiles per cloud: more
// workgroups than CUs, handed out by the dispatcher as CUs become free -- a CU that shares its time with the sampler simply
// takes fewer tiles.  fp32 MFMA: an exact fmaf chain (another summation order than the vendor GEMM's: parity 1e-5 of scale).
#include "common.hpp"

namespace pasnl {

typedef float mp_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mp_kappa(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// acc[i] += X[32 rows][2 S] . W[2 S][block i]  for the wave's NB blocks.  xrow = in + ql * pitch + h;  W row 2 s + h of block i is
// at W + s * 2 * WOUT + loff[i].  TAIL: S is not a multiple of the batch -- steps beyond S re-read step S - 1 with zero weights.
template <int WOUT, int NB, int RB, bool TAIL>
__device__ __forceinline__ void mp_mm(const float* xrow, int rbstride, int S, const float* __restrict__ W, const int (&loff)[NB],
                                      mp_f32x16 (&acc)[RB][NB]) {
  constexpr int BT = 8;
  float xb[2][RB][BT], wa[2][NB][BT];
  auto load = [&](int set, int sb) {
#pragma unroll
    for (int u = 0; u < BT; ++u) {
      const int s = sb + u, sc = TAIL ? min(s, S - 1) : s;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xb[set][rb][u] = xrow[rb * rbstride + 2 * sc];
      const float* wrow = W + (size_t)sc * (2 * WOUT);  // (uniform: a scalar base, the lane's offset in a register)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const float wv = wrow[loff[i]];
        wa[set][i][u] = (TAIL && s >= S) ? 0.f : wv;
      }
    }
  };
  load(0, 0);
  for (int sb = 0; sb < S; sb += 2 * BT) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int s0 = sb + half * BT;
      if (s0 < S) {
        load(half ^ 1, TAIL ? s0 + BT : min(s0 + BT, S - BT));  // the next batch (a dummy re-read behind the last one)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BT; ++u)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < NB; ++i)
              acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[half][rb][u], wa[half][i][u], acc[rb][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// one convolution of the tile: out[row][channel] = relu(in . W + bias), blocks of 32 channels dealt to the waves round robin
template <int WOUT, int RB, bool TAIL>
__device__ __forceinline__ void mp_layer(const float* in, int pin, int K, const float* __restrict__ W, const float* __restrict__ bias,
                                         float* out, int pout, int wave, int ql, int h) {
  constexpr int NW = 8, BLOCKS = WOUT / 32, NB = BLOCKS >= NW ? BLOCKS / NW : 1;
  if constexpr (BLOCKS * RB == NW && RB > 1) {
    // fewer blocks than waves (layer 0 of the narrower module): the tile's row blocks go to different waves instead of
    // idling half of them -- wave w owns block w mod BLOCKS of row block w / BLOCKS
    const int rb = wave / BLOCKS;
    mp_layer<WOUT, 1, TAIL>(in + rb * 32 * pin, pin, K, W, bias, out + rb * 32 * pout, pout, wave % BLOCKS, ql, h);
    return;
  }
  if (wave >= BLOCKS) return;  // (fewer blocks than waves)
  int loff[NB];
  mp_f32x16 acc[RB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    loff[i] = h * WOUT + (wave + NW * i) * 32 + ql;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
  }
  mp_mm<WOUT, NB, RB, TAIL>(in + ql * pin + h, 32 * pin, K >> 1, W, loff, acc);
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int ch = (wave + NW * i) * 32 + ql;
    const float bb = bias[ch];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(rb * 32 + mp_kappa(r, h)) * pout + ch] = fmaxf(acc[rb][i][r] + bb, 0.f);
  }
}

template <int C1, int C2, int C3, int RB>
__global__ __launch_bounds__(512) void mlp3_pool_kernel(int n, int k0, const float* __restrict__ x, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ partial) {
  constexpr int NW = 8, PB = C1 + 1;
  static_assert(C1 % 32 == 0 && C2 % 256 == 0 && C3 % 256 == 0, "blocks of 32 channels; the two wide layers fill all eight waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pa = max(k0, C2) | 1;  // odd row pitch (k0 and C2 are even)
  constexpr int TR = 32 * RB;                 // rows per tile
  float* A = reinterpret_cast<float*>(smem);  // [TR][pa]: X, then H2
  float* Bf = A + TR * pa;                    // [TR][PB]: H1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ql = lane & 31;
  const int tile = blockIdx.x, cloud = blockIdx.y;
  // ---- the tile's rows (a row beyond the cloud repeats its last point: no maximum changes)
  {
    const int q4 = k0 >> 2;
    const float4* xc = reinterpret_cast<const float4*>(x + (size_t)cloud * n * k0);
    for (int r = wave; r < TR; r += NW) {
      const int row = min(tile * TR + r, n - 1);
      for (int q = lane; q < q4; q += 64) {
        const float4 v = xc[(size_t)row * q4 + q];
        float* d = A + r * pa + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
  }
  __syncthreads();
  mp_layer<C1, RB, true>(A, pa, k0, w0, b0, Bf, PB, wave, ql, h);
  __syncthreads();
  mp_layer<C2, RB, false>(Bf, PB, C1, w1, b1, A, pa, wave, ql, h);
  __syncthreads();
  // ---- the last convolution, pooled: column maxima over the tile's rows (max and relu(. + bias) commute)
  {
    constexpr int NB = C3 / 32 / NW;
    int loff[NB];
    mp_f32x16 acc[RB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      loff[i] = h * C3 + (wave + NW * i) * 32 + ql;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
    }
    mp_mm<C3, NB, RB, false>(A + ql * pa + h, 32 * pa, C2 >> 1, w2, loff, acc);
    float* po = partial + ((size_t)cloud * gridDim.x + tile) * C3;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      float mx = acc[0][i][0];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[rb][i][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));  // the other half of the wave holds the other 16 rows
      const int ch = (wave + NW * i) * 32 + ql;
      if (h == 0) po[ch] = fmaxf(mx + b2[ch], 0.f);
    }
  }
}

template <int C1, int C2, int C3, int RB>
static int mlp3_launch(int b, int n, int k0, const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* partial, hipStream_t st) {
  const int pa = (k0 > C2 ? k0 : C2) | 1;
  const size_t lds = ((size_t)32 * RB * pa + (size_t)32 * RB * (C1 + 1)) * sizeof(float);
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  auto kern = mlp3_pool_kernel<C1, C2, C3, RB>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((n + 32 * RB - 1) / (32 * RB), b), dim3(512), lds, st, n, k0, x, w0, b0, w1, b1, w2, b2, partial);
  return pasnl_launch_status();
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_mlp3_max_pool_workspace_bytes(int b, int n, int c3) {
  if (b <= 0 || n <= 0 || c3 <= 0) return 0;
  return (size_t)b * ((n + 31) / 32) * c3 * sizeof(float);
}

extern "C" int pasnl_mlp3_max_pool(int b, int n, int k0, int c1, int c2, int c3, const float* x, const float* w0, const float* b0,
                                   const float* w1, const float* b1, const float* w2, const float* b2, float* out, long out_stride,
                                   void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && k0 > 0 && c1 > 0 && c2 > 0 && c3 > 0 && out_stride >= c3, PASNL_EINVAL);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(x && w0 && b0 && w1 && b1 && w2 && b2 && out && workspace, PASNL_ENULL);
  PASNL_REQUIRE(workspace_bytes >= pasnl_mlp3_max_pool_workspace_bytes(b, n, c3), PASNL_EWORKSPACE);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  // rows read in 16-byte pieces; an even contraction length per MFMA step pair
  PASNL_REQUIRE(k0 % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0, PASNL_EUNSUPPORTED);
  hipStream_t st = pasnl_hip_stream(stream);
  float* partial = static_cast<float*>(workspace);
  int rc;
  int tr;  // rows per tile: two row blocks where the activations of 64 rows fit the LDS (every weight then feeds two products)
  if (c1 == 128 && c2 == 256 && c3 == 512) { tr = 64; rc = mlp3_launch<128, 256, 512, 2>(b, n, k0, x, w0, b0, w1, b1, w2, b2, partial, st); }
  else if (c1 == 256 && c2 == 512 && c3 == 1024) { tr = 32; rc = mlp3_launch<256, 512, 1024, 1>(b, n, k0, x, w0, b0, w1, b1, w2, b2, partial, st); }
  else return PASNL_EUNSUPPORTED;
  if (rc != PASNL_OK) return rc;
  // maxima over the tiles of a cloud -> out[cloud * out_stride + channel]
  return pasnl_max_pool_rows_strided(b, (n + tr - 1) / tr, c3, partial, out, out_stride, stream);
}
