// The "group all" PointNet set-abstraction module of the classifier in ONE kernel (+ a pooling pass over tile maxima):
// reference utils/pointnet_util.py:87-137 as called at models/pointasnl_cls.py:39-40 (layer3_1: 512 points x [128, 256, 512],
// layer3_2: 128 points x [256, 512, 1024]) -- sample_and_group_all (every point of the cloud is one group), three 1x1
// convolutions (BN folded, ReLU) and tf.reduce_max over the group.  Round 4 ran it as three vendor GEMMs + a pooling kernel
// per module: eight launches at 58-123 TF whose 256 tiles (one per CU) stretch by whatever shares a CU with them (the next
// batch's sampler: 78 -> 141 us for the widest product).
//
// One workgroup of 8 waves owns a tile of 32 RB points of one cloud (RB = 2 row blocks where 64 rows of activations fit the
// LDS -- every weight then feeds two products --, else 1) and carries it through all three convolutions:
//   X (rows x k0) -> LDS buffer A;  H1 = relu(X W0 + b0) -> LDS buffer B;  H2 = relu(H1 W1 + b1) -> buffer A;
//   H3 = relu(H2 W2 + b2) never leaves the registers: its column maxima over the tile's rows go to partial[cloud][tile][:].
// v_mfma_f32_32x32x2_f32 with A = the activations (row ql of the tile, LDS, odd row pitch: conflict-free), B = the weights
// (straight from global memory / L2, PACKED in operand order: 32 bytes per lane and batch of eight steps, see mp_mm),
// D[m = row kappa(r, h)][n = channel]: the bias is one value per lane, the store of a block is column-contiguous, and the
// pooled maximum is a maximum over a lane's 16 accumulators and one exchange between the wave's halves.  A wave owns the
// 32-channel blocks wave, wave + 8, ...: one LDS operand feeds up to four products.  Operands of the NEXT eight steps are
// requested while eight steps multiply (two register sets).  The grid has 4 (n = 128, 32-row tiles) or 8 (n = 512, 64-row tiles) tiles
// per cloud: more workgroups than CUs, handed out by the dispatcher as CUs become free -- a CU that shares its time with the sampler simply
// takes fewer tiles.  fp32 MFMA: an exact fmaf chain (another summation order than the vendor GEMM's: parity 1e-5 of scale).
#include "common.hpp"

#ifndef PASNL_MLP3_ABL
#define PASNL_MLP3_ABL 0  // (diagnostic builds: 1 = no weight loads, 2 = no LDS operand reads, 4 = no tile load; results are wrong)
#endif
namespace pasnl {

typedef float mp_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mp_kappa(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// Weights in OPERAND order (pasnl_mlp3_pack_weights): for a batch of 8 matrix steps = 16 contraction indices 16 bt .. + 15, lane
// (column n, half h) needs W[16 bt + 2 u + h][n], u = 0 .. 7 -- packed as P[bt][h][n][u]: 32 contiguous bytes per lane and
// batch = two 16-byte loads (round 5 first read them as eight 4-byte loads of the row-major matrix: a quarter of the bytes
// per instruction; with the loads ablated the kernel ran 25 / 50 us faster), a half-wave's 32 columns 1 KiB contiguous.  The
// contraction length is padded to a multiple of 16 with zero weights (the activations there only have to be finite).
//
// acc[i] += X[32 rows][16 NBT] . W[16 NBT][block i]  for the wave's NB blocks.  xrow = in + ql * pitch + h;  the lane's packed
// weights of block i start at Wp + loff[i] (floats; h and the column folded in), a batch further every 16 WOUT floats.
template <int WOUT, int NB, int RB>
__device__ __forceinline__ void mp_mm(const float* xrow, int rbstride, int NBT, const float* __restrict__ Wp, const int (&loff)[NB],
                                      mp_f32x16 (&acc)[RB][NB]) {
  constexpr int BT = 8;
  float xb[2][RB][BT];
  float4 wa[2][NB][2];
  auto load = [&](int set, int bt) {
    const float* wrow = Wp + (size_t)bt * (16 * WOUT);  // (uniform: a scalar base, the lane's offset in a register)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (PASNL_MLP3_ABL & 1) {
        wa[set][i][0] = make_float4((float)bt, 1.f, 2.f, (float)loff[i]); wa[set][i][1] = wa[set][i][0];
      } else {
        wa[set][i][0] = *reinterpret_cast<const float4*>(wrow + loff[i]);
        wa[set][i][1] = *reinterpret_cast<const float4*>(wrow + loff[i] + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < BT; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        xb[set][rb][u] = (PASNL_MLP3_ABL & 2) ? (float)(bt + rb) : xrow[rb * rbstride + 2 * (bt * BT + u)];
  };
  load(0, 0);
  for (int bt = 0; bt < NBT; bt += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int b0 = bt + half;
      if (b0 < NBT) {
        load(half ^ 1, min(b0 + 1, NBT - 1));  // the next batch (a dummy re-read behind the last one)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BT; ++u) {
          float wv[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const float4 q = wa[half][i][u >> 2];
            wv[i] = (u & 3) == 0 ? q.x : ((u & 3) == 1 ? q.y : ((u & 3) == 2 ? q.z : q.w));
          }
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < NB; ++i)
              acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[half][rb][u], wv[i], acc[rb][i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// one convolution of the tile: out[row][channel] = relu(in . W + bias), blocks of 32 channels dealt to the waves round robin
template <int WOUT, int RB>
__device__ __forceinline__ void mp_layer(const float* in, int pin, int K, const float* __restrict__ W, const float* __restrict__ bias,
                                         float* out, int pout, int wave, int ql, int h) {
  constexpr int NW = 8, BLOCKS = WOUT / 32, NB = BLOCKS >= NW ? BLOCKS / NW : 1;
  if constexpr (BLOCKS * RB == NW && RB > 1) {
    // fewer blocks than waves (layer 0 of the narrower module): the tile's row blocks go to different waves instead of
    // idling half of them -- wave w owns block w mod BLOCKS of row block w / BLOCKS
    const int rb = wave / BLOCKS;
    mp_layer<WOUT, 1>(in + rb * 32 * pin, pin, K, W, bias, out + rb * 32 * pout, pout, wave % BLOCKS, ql, h);
    return;
  }
  if (wave >= BLOCKS) return;  // (fewer blocks than waves)
  int loff[NB];
  mp_f32x16 acc[RB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    loff[i] = (h * WOUT + (wave + NW * i) * 32 + ql) * 8;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
  }
  mp_mm<WOUT, NB, RB>(in + ql * pin + h, 32 * pin, (K + 15) >> 4, W, loff, acc);
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
  const int pa = max((k0 + 15) & ~15, C2) | 1;  // odd row pitch (k0 and C2 are even)
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
        const float4 v = (PASNL_MLP3_ABL & 4) ? make_float4(1.f, 2.f, 3.f, 4.f) : xc[(size_t)row * q4 + q];
        float* d = A + r * pa + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      // the contraction of layer 0 runs to the next multiple of 16 (zero weights there): finite operands
      if (lane < ((k0 + 15) & ~15) - k0) A[r * pa + k0 + lane] = 0.f;
    }
  }
  __syncthreads();
  mp_layer<C1, RB>(A, pa, k0, w0, b0, Bf, PB, wave, ql, h);
  __syncthreads();
  mp_layer<C2, RB>(Bf, PB, C1, w1, b1, A, pa, wave, ql, h);
  __syncthreads();
  // ---- the last convolution, pooled: column maxima over the tile's rows (max and relu(. + bias) commute)
  {
    constexpr int NB = C3 / 32 / NW;
    int loff[NB];
    mp_f32x16 acc[RB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      loff[i] = (h * C3 + (wave + NW * i) * 32 + ql) * 8;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
    }
    mp_mm<C3, NB, RB>(A + ql * pa + h, 32 * pa, C2 >> 4, w2, loff, acc);
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

// W (K, N) row-major -> P[bt][h][n][u] = W[16 bt + 2 u + h][n] (zero beyond K): one thread per 16-byte piece
__global__ __launch_bounds__(256) void mlp3_pack_kernel(int K, int N, const float* __restrict__ W, float* __restrict__ P) {
  const long total = (long)((K + 15) >> 4) * 2 * N * 2;  // 16-byte pieces
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int half = (int)(e & 1);
    const long lane = e >> 1;  // (bt * 2 + h) * N + n
    const int n = (int)(lane % N);
    const long bh = lane / N;
    const int h = (int)(bh & 1), bt = (int)(bh >> 1);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * bt + 2 * (4 * half + j) + h;
      v[j] = k < K ? W[(size_t)k * N + n] : 0.f;
    }
    reinterpret_cast<float4*>(P)[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int C1, int C2, int C3, int RB>
static int mlp3_launch(int b, int n, int k0, const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* partial, hipStream_t st) {
  const int k0p = (k0 + 15) & ~15;
  const int pa = (k0p > C2 ? k0p : C2) | 1;
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

extern "C" size_t pasnl_mlp3_packed_weights_bytes(int k, int n) {
  if (k <= 0 || n <= 0) return 0;
  return (size_t)((k + 15) & ~15) * n * sizeof(float);
}

extern "C" int pasnl_mlp3_pack_weights(int k, int n, const float* w, float* packed, pasnl_stream_t stream) {
  PASNL_REQUIRE(k > 0 && n > 0, PASNL_EINVAL);
  PASNL_REQUIRE(w && packed, PASNL_ENULL);
  PASNL_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0, PASNL_EUNSUPPORTED);
  const long pieces = (long)((k + 15) >> 4) * 4 * n;
  const long g = (pieces + 255) / 256;
  hipLaunchKernelGGL(mlp3_pack_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, pasnl_hip_stream(stream), k, n, w, packed);
  return pasnl_launch_status();
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
  PASNL_REQUIRE((reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) % 16 == 0,
                PASNL_EUNSUPPORTED);  // (packed weights: read in 16-byte pieces)
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
