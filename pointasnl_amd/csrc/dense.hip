// dense.hip -- the dense layers with FEW rows (the classifier head: fc1/fc2/fc3 on one row per cloud,
// models/pointasnl_cls.py:45-51 -> tf_util.fully_connected, tf_util.py:327-365; BN folded by the caller).
//
// out (R,N) = act(X (R,K) . W (K,N) + bias), R <= 128.  A vendor GEMM tiles such a product by its LARGE dimensions and ends up
// with one or two workgroups (measured on the cls head at B = 64: 12.8 + 36.8 + 11.4 us for 0.12 GFLOP); here the work is
// cut along N (32-column blocks) AND along K, so that ~128 workgroups each stream a (kchunk x 32) slab of W once:
//   * a wave owns kchunk/4 of the slab's K range and all R rows: 8 contraction indices per step -- lane (r, h) loads the
//     float4 X[r][k0+4h .. k0+4h+3] and the four W words W[k0+4h+t][col] (128 contiguous bytes per (t, h) across a half-wave);
//     MFMA step t contracts k0+t and k0+4+t (v_mfma_f32_32x32x2_f32: any pairing is legal as long as A and B agree);
//   * the four waves' accumulators meet in LDS and are added in wave order; the K-slices of different workgroups meet in a
//     workspace and are added in slice order by whichever workgroup of the column block finishes last (a counter per column
//     block, left at zero for the next launch): the summation order is fixed, the result bit-reproducible from run to run.
// Rows / columns beyond R / N are computed on clamped addresses and never stored (a D element depends on its own row and
// column only); K must be a multiple of 8.
#include "common.hpp"

namespace pasnl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int dense_kappa(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

template <int RB>  // 32-row blocks
__global__ __launch_bounds__(256) void dense_rows_kernel(int rows, int kdim, int n, int kchunk, int ksplit,
                                                        const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, int relu, float* __restrict__ out,
                                                        float* part, unsigned* counters) {
  extern __shared__ float red[];  // [4][RB * 1024]
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int cb = blockIdx.x, ks = blockIdx.y, ncb = gridDim.x;
  const int col = min(cb * 32 + l32, n - 1);
  // every element a thread finishes lies in column `col` (e = q * 256 + tid keeps its low five bits): ONE bias word, requested
  // here -- as `bias[c]` inside finish() it was a load -> wait -> store chain per element at the very end of the kernel
  const float bias_col = bias[col];
  const int per_wave = kchunk >> 2;  // multiple of 8
  const int kbeg = ks * kchunk + wave * per_wave;
  const int kend = min(kbeg + per_wave, kdim);

  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

  const float* xrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) xrow[rb] = X + (size_t)min(rb * 32 + l32, rows - 1) * kdim + 4 * h;
  const float* wcol = W + (size_t)(4 * h) * n + col;

  // Groups of D chunks of 8 contraction indices; the loads of group g+1 are issued before the MFMAs of group g, every load
  // of a group back to back (one round trip to L2 / HBM per group, not per chunk).  The plan makes a wave's K range a whole
  // number of groups whenever kdim allows; what is left over runs chunk by chunk.
  constexpr int D = RB <= 2 ? 4 : 2;
  struct Group { float4 x[D][RB]; float w[D][4]; };
  auto load = [&](int k0, Group& g) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) g.x[d][rb] = *reinterpret_cast<const float4*>(xrow[rb] + k0 + 8 * d);
#pragma unroll
      for (int t = 0; t < 4; ++t) g.w[d][t] = wcol[(size_t)(k0 + 8 * d + t) * n];
    }
  };
  auto fma = [&](const Group& g) {
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x[d][rb].x, g.w[d][0], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x[d][rb].y, g.w[d][1], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x[d][rb].z, g.w[d][2], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x[d][rb].w, g.w[d][3], acc[rb], 0, 0, 0);
      }
  };
  int k0 = kbeg;
  if (k0 + 8 * D <= kend) {
    Group ga, gb;
    load(k0, ga);
    while (true) {
      const bool more = k0 + 16 * D <= kend;
      if (more) load(k0 + 8 * D, gb);
      fma(ga);
      k0 += 8 * D;
      if (!more) break;
      const bool more2 = k0 + 16 * D <= kend;
      if (more2) load(k0 + 8 * D, ga);
      fma(gb);
      k0 += 8 * D;
      if (!more2) break;
    }
  }
  for (; k0 < kend; k0 += 8) {  // left-over chunks
    float4 x[RB];
    float w[4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) x[rb] = *reinterpret_cast<const float4*>(xrow[rb] + k0);
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = wcol[(size_t)(k0 + t) * n];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[rb].x, w[0], acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[rb].y, w[1], acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[rb].z, w[2], acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[rb].w, w[3], acc[rb], 0, 0, 0);
    }
  }

  // ---- the four waves' partial sums, added in wave order
  constexpr int E = RB * 1024;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave * E + (rb * 16 + i) * 64 + lane] = acc[rb][i];
  __syncthreads();
  float s[E / 256];
#pragma unroll
  for (int q = 0; q < E / 256; ++q) {
    const int e = q * 256 + tid;
    s[q] = ((red[e] + red[E + e]) + red[2 * E + e]) + red[3 * E + e];
  }

  // a full tile (the usual case) stores without a test per element: tested stores are a branch each, and behind a store that
  // may or may not have been issued the compiler can only wait for EVERYTHING (vmcnt(0): the previous store's round trip)
  const bool full_tile = rows == RB * 32 && cb * 32 + 32 <= n;  // uniform
  auto finish_all = [&](float (&v)[E / 256]) {
    size_t at[E / 256];
    bool ok[E / 256];
#pragma unroll
    for (int q = 0; q < E / 256; ++q) {
      const int e = q * 256 + tid;  // = (rb * 16 + i) * 64 + lane'
      const int ln = e & 63, i = (e >> 6) & 15, rb = e >> 10;
      const int r = rb * 32 + dense_kappa(i, ln >> 5), c = cb * 32 + (ln & 31);
      v[q] += bias_col;
      v[q] = relu ? fmaxf(v[q], 0.f) : v[q];
      at[q] = (size_t)r * n + c;
      ok[q] = r < rows && c < n;
    }
    if (full_tile) {
#pragma unroll
      for (int q = 0; q < E / 256; ++q) out[at[q]] = v[q];
    } else {
#pragma unroll
      for (int q = 0; q < E / 256; ++q)
        if (ok[q]) out[at[q]] = v[q];
    }
  };

  if (ksplit == 1) {
    finish_all(s);
    return;
  }
  // ---- K slices of the other workgroups: through the workspace, summed in slice order by the last one to arrive
  float* mine = part + ((size_t)ks * ncb + cb) * E;
#pragma unroll
  for (int q = 0; q < E / 256; ++q) mine[q * 256 + tid] = s[q];
  __threadfence();  // release: the slice is visible device-wide before the counter moves
  __syncthreads();
  if (tid == 0) {
    const unsigned seen = atomicAdd(&counters[cb], 1u);
    last_flag = seen == (unsigned)(ksplit - 1);
    if (last_flag) counters[cb] = 0;  // every other workgroup of this column block is done with it
  }
  __syncthreads();
  if (!last_flag) return;
  __threadfence();  // acquire: the other slices' stores are visible to the plain loads below
  float v[E / 256];
#pragma unroll
  for (int q = 0; q < E / 256; ++q) v[q] = 0.f;
  const float* col_part = part + (size_t)cb * E + tid;
  const size_t slice = (size_t)ncb * E;
  constexpr int SB = RB <= 2 ? 8 : 4;  // slices in flight per thread (E/256 words each); still added in slice order
  for (int k = 0; k < ksplit; k += SB) {
    float t[SB][E / 256];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int ku = min(k + u, ksplit - 1);  // (clamped: an unconditional load; the surplus is not added)
#pragma unroll
      for (int q = 0; q < E / 256; ++q) t[u][q] = col_part[ku * slice + q * 256];
    }
#pragma unroll
    for (int u = 0; u < SB; ++u)
      if (k + u < ksplit) {
#pragma unroll
        for (int q = 0; q < E / 256; ++q) v[q] += t[u][q];
      }
  }
  finish_all(v);
}

struct DensePlan {
  int rb, ncb, ksplit, kchunk;
  size_t counter_bytes, bytes;
};

static DensePlan dense_plan(int rows, int kdim, int n) {
  DensePlan p;
  p.rb = (rows + 31) / 32;
  p.ncb = (n + 31) / 32;
  int want = 128 / p.ncb;  // ~128 workgroups, slices of at least 64 contraction indices
  if (want > kdim / 64) want = kdim / 64;
  if (want < 1) want = 1;
  p.kchunk = ((kdim + want - 1) / want + 127) & ~127;  // a wave's quarter = whole groups of 4 chunks of 8
  if (const char* e = tune_env("PASNL_DENSE_KCHUNK")) p.kchunk = atoi(e);  // (tuning build only)
  p.ksplit = (kdim + p.kchunk - 1) / p.kchunk;
  p.counter_bytes = ((size_t)p.ncb * 4 + 255) & ~(size_t)255;
  p.bytes = p.counter_bytes + (p.ksplit > 1 ? (size_t)p.ksplit * p.ncb * p.rb * 1024 * 4 : 0);
  return p;
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_dense_rows_workspace_bytes(int rows, int kdim, int n) {
  if (rows <= 0 || kdim <= 0 || n <= 0 || rows > 128) return 0;
  return dense_plan(rows, kdim, n).bytes;
}

extern "C" int pasnl_dense_rows(int rows, int kdim, int n, const float* x, const float* w, const float* bias, int relu,
                                float* out, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows >= 0 && kdim > 0 && n > 0, PASNL_EINVAL);
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(x && w && bias && out && workspace, PASNL_ENULL);
  PASNL_REQUIRE(rows <= 128 && kdim % 8 == 0, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0, PASNL_EUNSUPPORTED);
  const DensePlan p = dense_plan(rows, kdim, n);
  PASNL_REQUIRE(workspace_bytes >= p.bytes, PASNL_EWORKSPACE);
  unsigned* counters = reinterpret_cast<unsigned*>(workspace);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + p.counter_bytes);
  hipStream_t st = pasnl_hip_stream(stream);
  const dim3 grid(p.ncb, p.ksplit);
  const size_t lds = (size_t)4 * p.rb * 1024 * 4;
#define PASNL_DENSE(RB)                                                                                                   \
  {                                                                                                                        \
    auto kern = dense_rows_kernel<RB>;                                                                                     \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                (int)lds) != hipSuccess)                                                  \
      return PASNL_ELAUNCH;                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, rows, kdim, n, p.kchunk, p.ksplit, x, w, bias, relu, out, part,     \
                       counters);                                                                                          \
  }
  switch (p.rb) {
    case 1: PASNL_DENSE(1) break;
    case 2: PASNL_DENSE(2) break;
    case 3: PASNL_DENSE(3) break;
    default: PASNL_DENSE(4) break;
  }
#undef PASNL_DENSE
  return pasnl_launch_status();
}

// ---------------------------------------------------------------------------------------------
// THIN products with a long contraction: out (M,N) = act(A (M,K) . W (K,N) + bias) where M*N is a few hundred 128 x 128
// tiles at most and K is thousands -- the after_conv / decode_after_conv layers of the deep levels of the segmentation
// models ((M,K,N) = (320,16384,512), (2560,4096,128), (640,8192,256), ...: pointasnl_util.py:277-280, :329-331).  The vendor
// library does not split K for these (40-53 TF, and 3.9 TF on (4096,384,256)); this kernel does:
//   * a workgroup (4 waves) owns one 128 x 128 output tile and one K slice; a wave owns 64 x 64 of it as 2 x 2
//     v_mfma_f32_32x32x2_f32 accumulators;
//   * chunks of 32 contraction indices (16: one chunk of loads in flight did not cover the A stream's HBM latency -- 40-54 TF)
//     go through LDS, double-buffered: A rows as they are (k contiguous: 64-byte runs per
//     row from global), W TRANSPOSED on the way in (a thread loads 8 words of one column, 256-byte runs per wave, and
//     writes 16-byte pieces of Ws[col][k]); both tiles have rows of KC + 4 floats, so the 16-byte operand reads of 16
//     lanes hit 16 different bank quads;
//   * MFMA step t of a group of 8 indices contracts k0+t and k0+4+t (any pairing is legal as long as A and B agree), so a
//     lane's operand for four steps is ONE ds_read_b128;
//   * ksplit > 1: the slices' partial tiles go to a workspace and a second kernel adds them IN SLICE ORDER, with bias and
//     activation (bit-reproducible; one workgroup adding all slices of a tile itself would be a serial tail of ~1 MB).
// K % 16 == 0, lda % 4 == 0, A 16-byte aligned, else PASNL_EUNSUPPORTED (the caller then takes the vendor GEMM).
// ---------------------------------------------------------------------------------------------
namespace pasnl {

constexpr int SK_BM = 128, SK_BN = 128;
// SK_KC contraction indices per chunk (32; 16 when K is only a multiple of 16); tile rows of KC + 4 floats: 16-byte reads of 16
// lanes hit 16 bank quads; a thread moves SK_H = KC / 2 indices of one A row and of one W column per chunk

template <int SK_KC>
__global__ __launch_bounds__(256) void dense_splitk_kernel(int M, int K, int N, int lda, int kslice, int ksplit,
                                                          const float* __restrict__ A, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int relu, float* __restrict__ out,
                                                          float* __restrict__ part) {
  constexpr int SK_LD = SK_KC + 4, SK_H = SK_KC / 2;
  __shared__ __attribute__((aligned(16))) float As[2][SK_BM * SK_LD];
  __shared__ __attribute__((aligned(16))) float Ws[2][SK_BN * SK_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * SK_BN, m0 = blockIdx.y * SK_BM, ks = blockIdx.z;
  const int kbeg = ks * kslice, kend = min(kbeg + kslice, K);
  const int nchunks = (kend - kbeg) / SK_KC;

  // global -> registers: A: thread t takes row t/2, KC/2 contraction indices (KC/8 float4); W: column t%128, KC/2 indices
  const int arow = tid >> 1, akq = (tid & 1) * SK_H;
  const float* ap = A + (size_t)min(m0 + arow, M - 1) * lda + kbeg + akq;
  const int wcol = tid & 127, wkh = (tid >> 7) * SK_H;
  const float* wp = W + (size_t)(kbeg + wkh) * N + min(n0 + wcol, N - 1);
  float4 ra[SK_H / 4];
  float rw[SK_H];
  auto gload = [&](int c) {
#pragma unroll
    for (int j = 0; j < SK_H / 4; ++j) ra[j] = *reinterpret_cast<const float4*>(ap + c * SK_KC + 4 * j);
#pragma unroll
    for (int j = 0; j < SK_H; ++j) rw[j] = wp[(size_t)(c * SK_KC + j) * N];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < SK_H / 4; ++j) {
      *reinterpret_cast<float4*>(&As[buf][arow * SK_LD + akq + 4 * j]) = ra[j];
      *reinterpret_cast<float4*>(&Ws[buf][wcol * SK_LD + wkh + 4 * j]) = make_float4(rw[4 * j], rw[4 * j + 1], rw[4 * j + 2], rw[4 * j + 3]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nchunks > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(c + 1);  // in flight under this chunk's products
    const float* as = &As[buf][(wm * 64 + l32) * SK_LD + 4 * h];
    const float* ws = &Ws[buf][(wn * 64 + l32) * SK_LD + 4 * h];
#pragma unroll
    for (int g = 0; g < SK_KC / 8; ++g) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float4*>(as + i * 32 * SK_LD + g * 8);
        b[i] = *reinterpret_cast<const float4*>(ws + i * 32 * SK_LD + g * 8);
      }
      // the four accumulators in turn inside every contraction step: no product waits for the one issued just before it
#define PASNL_SK_STEP(f)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].f, b[j].f, acc[i][j], 0, 0, 0);
      PASNL_SK_STEP(x) PASNL_SK_STEP(y) PASNL_SK_STEP(z) PASNL_SK_STEP(w)
#undef PASNL_SK_STEP
    }
    if (c + 1 < nchunks) lstore(buf ^ 1);  // the other buffer: its readers finished before the barrier that opened this chunk
    __syncthreads();
  }

  // D element e of lane (l32, h) of block (i, j): row = wm*64 + i*32 + kappa(e, h), column = wn*64 + j*32 + l32
  if (ksplit == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      const float bv = bias[min(col, N - 1)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 64 + i * 32 + dense_kappa(e, h);
          float v = acc[i][j][e] + bv;
          v = relu ? fmaxf(v, 0.f) : v;
          if (row < M && col < N) out[(size_t)row * N + col] = v;
        }
    }
    return;
  }
  // partial tile, in (row, column) order of the OUTPUT matrix inside a slice: part[ks][row][col] over the padded M x N
  const int Mp = gridDim.y * SK_BM, Np = gridDim.x * SK_BN;
  float* mine = part + (size_t)ks * Mp * Np;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + l32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + dense_kappa(e, h);
        mine[(size_t)row * Np + col] = acc[i][j][e];
      }
  }
}

// out[r][c] = act(bias[c] + sum over the slices, in slice order, of part[s][r][c]); a thread owns 4 consecutive columns
__global__ __launch_bounds__(256) void dense_splitk_reduce_kernel(int M, int N, int Mp, int Np, int ksplit,
                                                                 const float* __restrict__ part, const float* __restrict__ bias,
                                                                 int relu, float* __restrict__ out) {
  const int q = N >> 2;  // N % 4 == 0
  const long total = (long)M * q;
  const size_t slice = (size_t)Mp * Np;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / q), c = (int)(e - (long)r * q) * 4;
    const float* p = part + (size_t)r * Np + c;
    float4 v = *reinterpret_cast<const float4*>(p);
    for (int s = 1; s < ksplit; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(p + s * slice);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + (size_t)r * N + c) = v;
  }
}

struct SplitKPlan {
  int mt, nt, ksplit, kslice;
  size_t bytes;
};

static SplitKPlan splitk_plan(int M, int K, int N) {
  SplitKPlan p;
  const int SK_KC = K % 32 == 0 ? 32 : 16;
  p.mt = (M + SK_BM - 1) / SK_BM;
  p.nt = (N + SK_BN - 1) / SK_BN;
  const int tiles = p.mt * p.nt;
  // at most one workgroup per CU: a slice is long enough to amortise prologue, partial tile and reduce.  Rounded DOWN: 13
  // slices x 20 tiles = 260 workgroups put two on four of the 256 CUs and the launch took twice one workgroup's time
  int want = 256 / tiles;
  const int most = K / 64 > 0 ? K / 64 : 1;  // slices of at least 64 contraction indices
  if (want > most) want = most;
  if (want < 1) want = 1;
  p.kslice = (((K + want - 1) / want) + SK_KC - 1) / SK_KC * SK_KC;
  p.ksplit = (K + p.kslice - 1) / p.kslice;
  p.bytes = p.ksplit > 1 ? (size_t)p.ksplit * p.mt * SK_BM * p.nt * SK_BN * 4 : 0;
  return p;
}

}  // namespace pasnl

extern "C" size_t pasnl_dense_splitk_workspace_bytes(int rows, int kdim, int n) {
  if (rows <= 0 || kdim <= 0 || n <= 0) return 0;
  return splitk_plan(rows, kdim, n).bytes;
}

extern "C" int pasnl_dense_splitk(int rows, int kdim, int n, int lda, const float* x, const float* w, const float* bias, int relu,
                                  float* out, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows >= 0 && kdim > 0 && n > 0 && lda >= kdim, PASNL_EINVAL);
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(x && w && bias && out, PASNL_ENULL);
  PASNL_REQUIRE(kdim % 16 == 0 && lda % 4 == 0 && n % 4 == 0, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) % 16 == 0,
                PASNL_EUNSUPPORTED);
  const SplitKPlan p = splitk_plan(rows, kdim, n);
  PASNL_REQUIRE(p.mt <= 65535 && p.ksplit <= 65535, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(p.bytes == 0 || (workspace && workspace_bytes >= p.bytes), PASNL_EWORKSPACE);
  hipStream_t st = pasnl_hip_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  if (kdim % 32 == 0)
    hipLaunchKernelGGL(dense_splitk_kernel<32>, dim3(p.nt, p.mt, p.ksplit), dim3(256), 0, st, rows, kdim, n, lda, p.kslice,
                       p.ksplit, x, w, bias, relu, out, part);
  else
    hipLaunchKernelGGL(dense_splitk_kernel<16>, dim3(p.nt, p.mt, p.ksplit), dim3(256), 0, st, rows, kdim, n, lda, p.kslice,
                       p.ksplit, x, w, bias, relu, out, part);
  if (p.ksplit > 1) {
    const long total = (long)rows * (n / 4);
    long g = (total + 255) / 256;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(dense_splitk_reduce_kernel, dim3((int)g), dim3(256), 0, st, rows, n, p.mt * SK_BM, p.nt * SK_BN, p.ksplit,
                       part, bias, relu, out);
  }
  return pasnl_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Projections of NARROW rows (kdim <= 16: coordinates, coordinates + normals): out (rows,n) = x (rows,kdim) . w + bias.
// The first layer's non-local cell projects 3 / 6 input channels to its keys|values and queries (pointasnl_util.py:186-193):
// as GEMMs these are two launches bound by their own start-up (11.7 + 6.8 us at cls B = 64 for 21 MB of output).  Here
// both run in one launch as what they are -- a streaming write: a thread owns 4 output columns (its weights in registers)
// and walks the rows; a half-wave covers whole 64-byte runs of a row, so stores are full lines.  fmaf chains in ascending k
// (a GEMM's association is not reproducible anyway; the tests hold this to 1e-6 of fp64).
// ---------------------------------------------------------------------------------------------
namespace pasnl {

struct NarrowJob {
  const float* x; const float* w; const float* bias; float* out;
  long rows; int kdim, n;
};

// KD = the contraction length when it is one the models use (3: coordinates, 6: coordinates + centre / normals, 9), else 0:
// 16 steps on clamped addresses with zero weights beyond kdim -- every load unconditional, so all of a row's words (and the
// R rows a thread walks per pass) are in flight together.
template <int KD>
__device__ __forceinline__ void narrow_rows(const NarrowJob& j, int blk, int nblk) {
  constexpr int KS = KD ? KD : 16;
  constexpr int R = 4;                       // rows per thread per pass
  const int groups = j.n >> 2;               // 4-column groups per row; 256 % groups == 0 (n in {32, 64, 128, 256})
  const int g = threadIdx.x % groups, rlane = threadIdx.x / groups, rpb = 256 / groups;
  float4 wr[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const float4 wv = *reinterpret_cast<const float4*>(j.w + (size_t)min(k, j.kdim - 1) * j.n + 4 * g);
    wr[k] = k < j.kdim ? wv : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4 bv = *reinterpret_cast<const float4*>(j.bias + 4 * g);
  const long step = (long)nblk * rpb;
  for (long r0 = (long)blk * rpb + rlane; r0 < j.rows; r0 += R * step) {
    float xv[R][KS];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const long r = min(r0 + i * step, j.rows - 1);
      const float* xr = j.x + r * j.kdim;
#pragma unroll
      for (int k = 0; k < KS; ++k) xv[i][k] = xr[KD ? k : min(k, j.kdim - 1)];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float4 a = bv;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        a.x = fmaf(xv[i][k], wr[k].x, a.x); a.y = fmaf(xv[i][k], wr[k].y, a.y);
        a.z = fmaf(xv[i][k], wr[k].z, a.z); a.w = fmaf(xv[i][k], wr[k].w, a.w);
      }
      const long r = r0 + i * step;
      if (r < j.rows) *reinterpret_cast<float4*>(j.out + r * j.n + 4 * g) = a;
    }
  }
}

__device__ __forceinline__ void narrow_dispatch(const NarrowJob& j, int blk, int nblk) {
  if (j.kdim == 3) narrow_rows<3>(j, blk, nblk);
  else if (j.kdim == 6) narrow_rows<6>(j, blk, nblk);
  else if (j.kdim == 9) narrow_rows<9>(j, blk, nblk);
  else narrow_rows<0>(j, blk, nblk);
}

__global__ __launch_bounds__(256) void narrow_project_kernel(NarrowJob j0, NarrowJob j1, int blocks0) {
  if ((int)blockIdx.x < blocks0) narrow_dispatch(j0, blockIdx.x, blocks0);
  else narrow_dispatch(j1, blockIdx.x - blocks0, gridDim.x - blocks0);
}

static bool narrow_ok(long rows, int kdim, int n, const void* w, const void* bias, const void* out) {
  return rows >= 0 && kdim >= 1 && kdim <= 16 && (n == 32 || n == 64 || n == 128 || n == 256) &&
         ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
}

}  // namespace pasnl

extern "C" int pasnl_narrow_project2(long rows0, int kdim0, int n0, const float* x0, const float* w0, const float* bias0,
                                     float* out0, long rows1, int kdim1, int n1, const float* x1, const float* w1,
                                     const float* bias1, float* out1, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows0 >= 0 && rows1 >= 0 && kdim0 > 0 && n0 > 0, PASNL_EINVAL);
  PASNL_REQUIRE(rows1 == 0 || (kdim1 > 0 && n1 > 0), PASNL_EINVAL);
  if (rows0 == 0 && rows1 == 0) return PASNL_OK;
  PASNL_REQUIRE(rows0 == 0 || (x0 && w0 && bias0 && out0), PASNL_ENULL);
  PASNL_REQUIRE(rows1 == 0 || (x1 && w1 && bias1 && out1), PASNL_ENULL);
  PASNL_REQUIRE(rows0 == 0 || narrow_ok(rows0, kdim0, n0, w0, bias0, out0), PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(rows1 == 0 || narrow_ok(rows1, kdim1, n1, w1, bias1, out1), PASNL_EUNSUPPORTED);
  auto blocks = [](long rows, int n) -> int {
    if (rows == 0) return 0;
    const long rpb = 256 / (n >> 2);
    long b = (rows + rpb * 8 - 1) / (rpb * 8);  // ~8 rows per thread
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
  };
  const int b0 = blocks(rows0, n0), b1 = blocks(rows1, n1);
  NarrowJob j0{x0, w0, bias0, out0, rows0, kdim0, n0}, j1{x1, w1, bias1, out1, rows1, kdim1, n1};
  if (rows0 == 0) { j0 = j1; }  // (blocks0 == 0: every workgroup takes the second job)
  hipLaunchKernelGGL(narrow_project_kernel, dim3(b0 + b1), dim3(256), 0, pasnl_hip_stream(stream), j0, j1, b0);
  return pasnl_launch_status();
}
