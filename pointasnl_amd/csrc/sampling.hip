// Sampling ops for gfx950: farthest point sampling, 3-wide gather (+grad), probability sampling.
// Behaviour contract: reference tf_ops/sampling/tf_sampling_g.cu (kernels :7-192), restated in oracle/.
// The design is NOT the reference's: FPS keeps every point and its running min-distance in VGPRs,
// reduces with DPP inside a wave and one LDS hop across waves, one workgroup per cloud (details below).
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"

namespace pasnl {

// ---------------------------------------------------------------------------------------------
// Farthest point sampling.
//   One workgroup (WAVES x 64 lanes) owns one cloud; lane t keeps points k = i*T + t (i < PPL, T = WAVES*64)
//   and their running min-distances in VGPRs for the whole kernel.  The loop is a chain of npoint DEPENDENT
//   rounds (a dependent VALU op costs ~6.5 cycles with one wave per SIMD), so the design minimises the depth
//   of one round:
//     * distances for two points per instruction (v_pk_* fp32), all PPL points independent;
//     * in-lane argmax as a tournament tree (depth log2 PPL) with strict '>' and left preference; the leaves
//       are ordered by (i*T mod 512, i), which makes "leftmost maximum" exactly the reference tie rule
//       "largest d2, then lowest k mod 512, then lowest k" (tf_sampling_g.cu:142-164) -- T divides
//       512 or 512 divides T, so the order is a compile-time permutation;
//     * wave argmax = six single-instruction v_max_i32_dpp steps on the distance bits (real distances are
//       >= +0 and order like signed ints; padding carries distinct negative sentinels) + one ballot; several
//       lanes holding the maximum is the rare uniform slow path that compares tie keys;
//     * cross-wave (WAVES > 1): one {d2,k} slot per wave, double-buffered, ONE barrier per round; afterwards
//       lane w of every wave reads slot w and a 4-step row DPP max finishes -- cost independent of WAVES;
//     * no global memory traffic inside the loop: picks are buffered in LDS (a store in the loop would make
//       every barrier wait for its acknowledgement) and the cloud sits in LDS as 16-byte records for the one
//       b128 broadcast read "coordinates of the pick".
//   Single-wave clouds (n <= 1024) need no barrier at all.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t fps_tiekey(int k) { return (((uint32_t)k & 511u) << 22) | (uint32_t)k; }

// max over the wave of an int; result in lane 63 (rows of 16: shr 1,2,4,8 then row broadcasts)
__device__ __forceinline__ int wave_max_i32_to_lane63(int x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(x));
  return x;
}
// max over each row of 16 lanes; result in lane 15 of the row
__device__ __forceinline__ int row_max_i32_to_lane15(int x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(x));
  return x;
}

template <int T, int PPL>
struct FpsOrder {
  // tournament leaf order of a lane's points: ascending ((i*T) mod 512, i)
  int idx[PPL];
  constexpr FpsOrder() : idx{} {
    for (int i = 0; i < PPL; ++i) idx[i] = i;
    for (int a = 0; a < PPL; ++a)
      for (int b = a + 1; b < PPL; ++b) {
        int ka = ((idx[a] * T) % 512) * 1024 + idx[a], kb = ((idx[b] * T) % 512) * 1024 + idx[b];
        if (kb < ka) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
      }
  }
};

// uniform helper: among the lanes in `tie`, the one whose candidate index has the smallest tie key
__device__ __forceinline__ int fps_break_tie(unsigned long long tie, int cand_k) {
  uint32_t bestkey = 0xffffffffu;
  int win = 0;
  while (tie) {
    int l = (int)__builtin_ctzll(tie);
    tie &= tie - 1;
    uint32_t key = fps_tiekey(__builtin_amdgcn_readlane(cand_k, l));
    if (key < bestkey) { bestkey = key; win = l; }
  }
  return win;
}

// Tournament over leaves [LO, LO+N) of the tie order; every index is a compile-time constant (plain recursion
// instead of loops over arrays: selects between array elements inside unrolled loops get rewritten by the
// compiler into "element [select(index)]" gathers, i.e. 16-deep select chains).
template <int T, int PPL, int LO, int N>
__device__ __forceinline__ void fps_tournament(const int (&td)[PPL], int tid, int& d, int& k) {
  if constexpr (N == 1) {
    constexpr FpsOrder<T, PPL> ORDER{};
    d = td[ORDER.idx[LO]];
    k = ORDER.idx[LO] * T + tid;
  } else {
    int dl, kl, dr, kr;
    fps_tournament<T, PPL, LO, N / 2>(td, tid, dl, kl);
    fps_tournament<T, PPL, LO + N / 2, N - N / 2>(td, tid, dr, kr);
    const bool right = dr > dl;
    d = right ? dr : dl;
    k = right ? kr : kl;
  }
}

// The sampled coordinates (gather_point of the picks) written by the sampler itself: (m,3) rows of the cloud, copied from
// the (L2-resident) input so that the bits are the gather's.  picks = the LDS list of the workgroup, complete and visible.
template <int T>
__device__ __forceinline__ void fps_emit_xyz(const float* __restrict__ cloud, const int* picks, int m, float* __restrict__ o, int tid) {
  if (!o) return;
  for (int f = tid; f < m * 3; f += T) {
    const int j = f / 3;
    o[f] = cloud[(size_t)picks[j] * 3 + (f - 3 * j)];
  }
}

template <int WAVES, int PPL, int STRIDE>  // STRIDE = floats per LDS point record (4: one b128 read; 3: 10240-point clouds)
__global__ __launch_bounds__(WAVES * 64) void fps_kernel(int n, int m, const float* __restrict__ xyz,
                                                        int* __restrict__ idx, float* __restrict__ out_xyz) {
  static_assert(PPL % 2 == 0, "points are processed in packed pairs");
  constexpr int T = WAVES * 64;
  constexpr int NP = PPL / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* slots = reinterpret_cast<float2*>(smem);                 // [2][16] {d2 bits, k}
  float* spt = reinterpret_cast<float*>(smem + 2 * 16 * 8);        // [n][STRIDE] {x,y,z[,-]}: broadcast read per pick
  int* picks = reinterpret_cast<int*>(spt + (size_t)n * STRIDE);   // [m]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cloud = xyz + (size_t)blockIdx.x * n * 3;
  // A sampler is m dependent rounds of a few hundred instructions on ONE wave per SIMD: beside a matrix kernel (a serving loop
  // runs the next batch's sampler beside the current batch's dense layers) every instruction it loses in the issue
  // arbitration lengthens the chain (212 -> 535 us measured), while what it takes from the other kernel is a few per cent of
  // a SIMD's issue slots.  Highest wave priority.
  __builtin_amdgcn_s_setprio(3);

  for (int f = tid; f < n * 3; f += T) {  // coalesced flat copy of the (n,3) array into 16-byte LDS records
    float v = cloud[f];
    int p = f / 3, c = f - p * 3;
    spt[p * STRIDE + c] = v;
  }
  if (tid == 0) picks[0] = 0;
  __syncthreads();

  f32x2 px[NP], py[NP], pz[NP];
  int td[PPL];    // running min-distance as float BITS: non-negative floats order like signed ints, so min/max are
                  // single integer ops (a float min would first canonicalise both operands)
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int k = (2 * q + e) * T + tid;
      bool ok = k < n;
      px[q][e] = ok ? spt[k * STRIDE] : 0.f;
      py[q][e] = ok ? spt[k * STRIDE + 1] : 0.f;
      pz[q][e] = ok ? spt[k * STRIDE + 2] : 0.f;
      // padding: negative and distinct per lane -> never wins, never ties
      td[2 * q + e] = ok ? __float_as_int(1e38f) : __float_as_int(-(float)(tid + 1));
    }
  float x1 = spt[0], y1 = spt[1], z1 = spt[2];

  for (int j = 1; j < m; ++j) {
    // ---- running distances (no FMA contraction; ((dx*dx)+(dy*dy))+(dz*dz))
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f32x2 dx = px[q] - x1, dy = py[q] - y1, dz = pz[q] - z1;
      f32x2 d = (dx * dx + dy * dy) + dz * dz;
      td[2 * q] = min(td[2 * q], __float_as_int(d[0]));
      td[2 * q + 1] = min(td[2 * q + 1], __float_as_int(d[1]));
    }
    // ---- in-lane tournament over the leaves in tie order; left wins ties
    int bd, bk;
    fps_tournament<T, PPL, 0, PPL>(td, tid, bd, bk);
    // ---- wave argmax
    const int wmaxi = __builtin_amdgcn_readlane(wave_max_i32_to_lane63(bd), 63);
    unsigned long long tie = __ballot(bd == wmaxi);
    int win = (int)__builtin_ctzll(tie);
    if (__builtin_popcountll(tie) > 1) win = fps_break_tie(tie, bk);
    int old = __builtin_amdgcn_readlane(bk, win);
    if constexpr (WAVES > 1) {
      float2* slot = slots + (j & 1) * 16;
      if (lane == 0) slot[wave] = make_float2(__int_as_float(wmaxi), __int_as_float(old));
      __syncthreads();
      float2 sv = lane < WAVES ? slot[lane] : make_float2(__int_as_float((int)0x80000000), 0.f);
      const int di = __float_as_int(sv.x), ki = __float_as_int(sv.y);
      const int gmax = __builtin_amdgcn_readlane(row_max_i32_to_lane15(di), 15);
      unsigned long long wt = __ballot(di == gmax) & 0xffffull;
      int ww = (int)__builtin_ctzll(wt);
      if (__builtin_popcountll(wt) > 1) ww = fps_break_tie(wt, ki);
      old = __builtin_amdgcn_readlane(ki, ww);
    }
    if (tid == 0) picks[j] = old;
    if constexpr (STRIDE == 4) {
      float4 pick = *reinterpret_cast<const float4*>(spt + old * 4);
      x1 = pick.x; y1 = pick.y; z1 = pick.z;
    } else {
      x1 = spt[old * 3]; y1 = spt[old * 3 + 1]; z1 = spt[old * 3 + 2];
    }
  }
  __syncthreads();
  int* out = idx + (size_t)blockIdx.x * m;
  for (int j = tid; j < m; j += T) out[j] = picks[j];
  fps_emit_xyz<T>(xyz + (size_t)blockIdx.x * n * 3, picks, m, out_xyz ? out_xyz + (size_t)blockIdx.x * m * 3 : nullptr, tid);
}


// ---------------------------------------------------------------------------------------------
// Small clouds (n <= 4096, WAVES <= 4): the same algorithm with a shorter dependent chain per round.
// What a round of fps_kernel spends after the in-lane tournament is synchronisation, not arithmetic:
// wave max -> ballot -> readlane -> LDS slot -> barrier -> LDS read -> row DPP max -> ballot -> readlane ->
// LDS read of the pick's coordinates.  Here the tournament carries the candidate's COORDINATES along with (d, k)
// (a few more v_cndmask per level, off the critical path), the lane that holds the wave maximum writes
// {d, tie key, x, y, z} to its wave's slot itself (exec-masked store: no ballot/readlane on the common path), and
// after the ONE barrier every lane reads the WAVES slots (broadcast reads, one wait) and reduces them in registers
// with 64-bit compares on (d bits << 32 | ~tie key): largest distance, then smallest tie key = the reference rule.
// The winner's coordinates arrive with it: no second LDS round trip.
// ---------------------------------------------------------------------------------------------
template <int T, int PPL, int LO, int N>
__device__ __forceinline__ void fps_tournament_xyz(const int (&td)[PPL], const f32x2 (&px)[PPL / 2], const f32x2 (&py)[PPL / 2],
                                                   const f32x2 (&pz)[PPL / 2], int tid, int& d, int& k, float& x, float& y,
                                                   float& z) {
  if constexpr (N == 1) {
    constexpr FpsOrder<T, PPL> ORDER{};
    constexpr int I = ORDER.idx[LO];
    d = td[I];
    k = I * T + tid;
    x = px[I / 2][I % 2];
    y = py[I / 2][I % 2];
    z = pz[I / 2][I % 2];
  } else {
    int dl, kl, dr, kr;
    float xl, yl, zl, xr, yr, zr;
    fps_tournament_xyz<T, PPL, LO, N / 2>(td, px, py, pz, tid, dl, kl, xl, yl, zl);
    fps_tournament_xyz<T, PPL, LO + N / 2, N - N / 2>(td, px, py, pz, tid, dr, kr, xr, yr, zr);
    const bool right = dr > dl;
    d = right ? dr : dl;
    k = right ? kr : kl;
    x = right ? xr : xl;
    y = right ? yr : yl;
    z = right ? zr : zl;
  }
}

struct __attribute__((aligned(16))) FpsSlot {
  uint32_t nkey;  // ~tie key  (low half of the 64-bit order key)
  int d;          // distance bits (high half)
  float x, y;
  float z;
  int pad[3];
};

// best of slots [LO, LO+N): 64-bit order key (d bits << 32 | ~tie key) and the candidate's coordinates
template <int LO, int N>
__device__ __forceinline__ void fps_reduce_slots(const FpsSlot* slot, unsigned long long& key, float& x, float& y, float& z) {
  if constexpr (N == 1) {
    const float4 a = *reinterpret_cast<const float4*>(&slot[LO]);
    key = ((unsigned long long)__float_as_uint(a.y) << 32) | __float_as_uint(a.x);
    x = a.z;
    y = a.w;
    z = slot[LO].z;
  } else {
    unsigned long long kl, kr;
    float xl, yl, zl, xr, yr, zr;
    fps_reduce_slots<LO, N / 2>(slot, kl, xl, yl, zl);
    fps_reduce_slots<LO + N / 2, N - N / 2>(slot, kr, xr, yr, zr);
    const bool r = kr > kl;
    key = r ? kr : kl;
    x = r ? xr : xl;
    y = r ? yr : yl;
    z = r ? zr : zl;
  }
}

#ifdef PASNL_TUNING
#define FPS_MARK(i) do { if (dbg) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                                    __builtin_amdgcn_sched_barrier(0); acc[i] += t_ - tprev; tprev = t_; } } while (0)
#define FPS_DBG_PARAM , long long* dbg
#define FPS_DBG_ARG , dbg
#else
#define FPS_MARK(i)
#define FPS_DBG_PARAM
#define FPS_DBG_ARG
#endif

template <int WAVES, int PPL, int ABL = 0>  // ABL (tuning build, timing only): 1 = no cross-wave exchange, 2 = no wave reduction either
__global__ __launch_bounds__(WAVES * 64) void fps_small_kernel(int n, int m, const float* __restrict__ xyz,
                                                              int* __restrict__ idx FPS_DBG_PARAM) {
  static_assert(PPL % 2 == 0 && WAVES <= 4, "");
#ifdef PASNL_TUNING
  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  constexpr int T = WAVES * 64;
  constexpr int NP = PPL / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FpsSlot* slots = reinterpret_cast<FpsSlot*>(smem);                      // [2][WAVES]
  int* picks = reinterpret_cast<int*>(smem + 2 * WAVES * sizeof(FpsSlot));  // [m]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cloud = xyz + (size_t)blockIdx.x * n * 3;

  f32x2 px[NP], py[NP], pz[NP];
  int td[PPL];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = (2 * q + e) * T + tid;
      const bool ok = k < n;
      px[q][e] = ok ? cloud[k * 3] : 0.f;
      py[q][e] = ok ? cloud[k * 3 + 1] : 0.f;
      pz[q][e] = ok ? cloud[k * 3 + 2] : 0.f;
      td[2 * q + e] = ok ? __float_as_int(1e38f) : __float_as_int(-(float)(tid + 1));
    }
  float x1 = cloud[0], y1 = cloud[1], z1 = cloud[2];
  if (tid == 0) picks[0] = 0;

  for (int j = 1; j < m; ++j) {
    FPS_MARK(0);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f32x2 dx = px[q] - x1, dy = py[q] - y1, dz = pz[q] - z1;
      f32x2 d = (dx * dx + dy * dy) + dz * dz;
      td[2 * q] = min(td[2 * q], __float_as_int(d[0]));
      td[2 * q + 1] = min(td[2 * q + 1], __float_as_int(d[1]));
    }
    int bd, bk;
    float bx, by, bz;
    fps_tournament_xyz<T, PPL, 0, PPL>(td, px, py, pz, tid, bd, bk, bx, by, bz);
    FPS_MARK(1);
    if constexpr (ABL == 2) {
      x1 = bx; y1 = by; z1 = bz;
      if (tid == 0) picks[j] = bk;
      continue;
    }
    const int wmaxi = __builtin_amdgcn_readlane(wave_max_i32_to_lane63(bd), 63);
    FpsSlot* slot = slots + (j & 1) * WAVES;
    const bool mine = bd == wmaxi;
    unsigned long long tie = __ballot(mine);
    bool writer = mine;
    if (__builtin_popcountll(tie) > 1) writer = lane == fps_break_tie(tie, bk);  // rare, wave-uniform branch
    if (writer) {
      *reinterpret_cast<float4*>(&slot[wave]) =
          make_float4(__uint_as_float(~fps_tiekey(bk)), __int_as_float(bd), bx, by);
      slot[wave].z = bz;
    }
    FPS_MARK(2);
    if constexpr (ABL == 1) {
      x1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx)));
      y1 = by; z1 = bz;
      if (tid == 0) picks[j] = bk;
      continue;
    }
    if constexpr (WAVES > 1) __syncthreads();
    else __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own store
    FPS_MARK(3);
    // every lane reduces the WAVES candidates in registers (compile-time recursion: see fps_tournament)
    unsigned long long wkey;
    fps_reduce_slots<0, WAVES>(slot, wkey, x1, y1, z1);
    if (tid == 0) picks[j] = (int)(~(uint32_t)wkey & 0x3fffffu);
    FPS_MARK(4);
  }
#ifdef PASNL_TUNING
  if (dbg && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 5; ++i) dbg[i] = (long long)acc[i];
#endif
  __syncthreads();
  int* out = idx + (size_t)blockIdx.x * m;
  for (int j = tid; j < m; j += T) out[j] = picks[j];
}

template <int WAVES, int PPL, int ABL = 0>
static int fps_small_launch(int b, int n, int m, const float* xyz, int* idx, hipStream_t st) {
  size_t lds = (size_t)2 * WAVES * sizeof(FpsSlot) + (size_t)m * 4;
  if (lds > 64 * 1024) return PASNL_EUNSUPPORTED;
#ifdef PASNL_TUNING
  long long* dbg = nullptr;  // PASNL_FPS_PROBE=<device pointer to 8 int64, hex>: phase cycles of workgroup 0, wave 0
  if (const char* pe = tune_env("PASNL_FPS_PROBE")) dbg = reinterpret_cast<long long*>(strtoull(pe, nullptr, 16));
#endif
  hipLaunchKernelGGL((fps_small_kernel<WAVES, PPL, ABL>), dim3(b), dim3(WAVES * 64), lds, st, n, m, xyz, idx FPS_DBG_ARG);
  return pasnl_launch_status();
}

template <int WAVES, int PPL>
static int fps_launch(int b, int n, int m, const float* xyz, int* idx, hipStream_t st, float* oxyz = nullptr) {
  size_t lds = (size_t)2 * 16 * 8 + (size_t)n * 16 + (size_t)m * 4;
  auto kern = fps_kernel<WAVES, PPL, 4>;
  if (lds > 160 * 1024) {  // fall back to 12-byte records
    lds = (size_t)2 * 16 * 8 + (size_t)n * 12 + (size_t)m * 4;
    kern = fps_kernel<WAVES, PPL, 3>;
  }
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return PASNL_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(b), dim3(WAVES * 64), lds, st, n, m, xyz, idx, oxyz);
  return pasnl_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Large clouds (2048 < n <= 10240): the same sampling with the round's arithmetic PRUNED.
// fps_kernel updates every running distance in every round: n x npoint updates (8192 x 1024 x 16 clouds), one compute unit
// per cloud, ~0.9 us per round of pure vector work.  But a new pick p changes min(td[x], |x-p|^2) only for points closer
// to p than sqrt(td[x]) -- after a few hundred picks a small neighbourhood.  So:
//   * once, at the start: the points are sorted along a Morton curve (16^3 cells over the bounding box, counting sort in
//     LDS) and dealt out so that a wave owns 64*NB consecutive points = a compact blob, a lane NB consecutive ones; each
//     wave keeps its blob's centre c and radius R (wave-uniform);
//   * every round a wave first tests  |p - c| >= R + sqrt(M)  (M = its current largest running distance; margins of 1e-5
//     dwarf the rounding of the test and of the fp32 distances): then NO point of the wave can change, its cached candidate
//     (largest distance, tie key) is still exact, and the wave skips the update, the tournament and the wave reduction;
//   * every wave -- active or not -- folds its candidate into one LDS word with ds_max_u64 on
//     (distance bits << 32 | ~tie key): largest distance, then lowest k mod 512, then lowest k: the reference rule
//     (tf_sampling_g.cu:142-164) as an unsigned 64-bit order.  One barrier; every lane reads the winner and its coordinates.
// Exactly the reference's picks for every input (tests: lattice clouds, duplicates, metre-scale KITTI coordinates).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fps_spread3(uint32_t v) {  // 4 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

#ifdef PASNL_TUNING
__device__ unsigned long long fps_dbg[8];  // [active wave-rounds, wave-rounds | fps_multi, workgroup 0: touched wave-rounds, rounds, cycles: wave 1 pre-barrier, wave 0 merge, wave 1 round total, picks]
#endif

#ifndef PASNL_FPS_ABL
#define PASNL_FPS_ABL 0  // tuning builds only: 1 = no wave is ever active (exchange cost alone), 2 = every wave always active
#endif
template <int WAVES, int NB>
__global__ __launch_bounds__(WAVES * 64) void fps_pruned_kernel(int n, int m, const float* __restrict__ xyz, int* __restrict__ idx,
                                                               float* __restrict__ out_xyz) {
  constexpr int T = WAVES * 64;
  constexpr int NCELL = 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* spt = reinterpret_cast<float*>(smem);                                   // [n][3], ORIGINAL order: pick coordinates
  unsigned short* order = reinterpret_cast<unsigned short*>(spt + (size_t)n * 3);  // [n]: sorted position -> original index
  // 16-byte aligned whatever n is, exactly as fps_pruned_launch sizes the allocation
  int* hist = reinterpret_cast<int*>(smem + (((size_t)n * 12 + (size_t)((n + 1) & ~1) * 2 + 15) & ~(size_t)15));  // [4096] during the sort ...
  unsigned long long* best = reinterpret_cast<unsigned long long*>(hist);         // ... then [2][16] {distance, key} slots
  int* picks = hist + 64;                                                         // ... and [m] picks
  __shared__ float red[6][WAVES];
  __shared__ int wsum[WAVES];
#if defined(PASNL_TUNING) && PASNL_FPS_ABL == 0
  __shared__ int dbg_act[3][4];
  if (threadIdx.x < 12) (&dbg_act[0][0])[threadIdx.x] = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cloud = xyz + (size_t)blockIdx.x * n * 3;

  // ---- stage the cloud, bounding box
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int f = tid; f < n * 3; f += T) spt[f] = cloud[f];  // coalesced flat copy
  __syncthreads();
  for (int k = tid; k < n; k += T) {
    const float x = spt[k * 3], y = spt[k * 3 + 1], z = spt[k * 3 + 2];
    mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
    mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
    mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], sft));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], sft));
    }
    if (lane == 0) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
  }
  for (int c = tid; c < NCELL; c += T) hist[c] = 0;
  __syncthreads();
  float lo[3], sc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = red[a][0], u = red[3 + a][0];
    for (int w = 1; w < WAVES; ++w) { l = fminf(l, red[a][w]); u = fmaxf(u, red[3 + a][w]); }
    lo[a] = l;
    sc[a] = u > l ? 15.999f / (u - l) : 0.f;
  }
  // ---- Morton counting sort: histogram, scan, scatter of the original indices
  auto code_of = [&](int k) {
    const uint32_t cx = (uint32_t)min(15, max(0, (int)((spt[k * 3] - lo[0]) * sc[0])));
    const uint32_t cy = (uint32_t)min(15, max(0, (int)((spt[k * 3 + 1] - lo[1]) * sc[1])));
    const uint32_t cz = (uint32_t)min(15, max(0, (int)((spt[k * 3 + 2] - lo[2]) * sc[2])));
    return (int)(fps_spread3(cx) | (fps_spread3(cy) << 1) | (fps_spread3(cz) << 2));
  };
  for (int k = tid; k < n; k += T) atomicAdd(&hist[code_of(k)], 1);
  __syncthreads();
  {
    constexpr int PER = NCELL / T;  // cells per thread (T divides 4096)
    int v[PER], tsum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { v[i] = hist[tid * PER + i]; tsum += v[i]; }
    int incl = tsum;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int o = __shfl_up(incl, sft);
      if (lane >= sft) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - tsum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) { hist[tid * PER + i] = run; run += v[i]; }
  }
  __syncthreads();
  for (int k = tid; k < n; k += T) order[atomicAdd(&hist[code_of(k)], 1)] = (unsigned short)k;
  __syncthreads();

  // ---- this lane's NB consecutive points of the sorted order
  f32x2 px[NB / 2], py[NB / 2], pz[NB / 2];
  uint32_t td[NB], nkey[NB];  // running distance bits; ~tie key (0 = padding: loses every comparison)
  float bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int pos = (wave * 64 + lane) * NB + i;
    const bool ok = pos < n;
    const int k = ok ? (int)order[pos] : 0;
    const float x = ok ? spt[k * 3] : 0.f, y = ok ? spt[k * 3 + 1] : 0.f, z = ok ? spt[k * 3 + 2] : 0.f;
    px[i / 2][i % 2] = x; py[i / 2][i % 2] = y; pz[i / 2][i % 2] = z;
    td[i] = ok ? __float_as_uint(1e38f) : 0u;
    nkey[i] = ok ? ~fps_tiekey(k) : 0u;
    if (ok) {
      bmn[0] = fminf(bmn[0], x); bmx[0] = fmaxf(bmx[0], x);
      bmn[1] = fminf(bmn[1], y); bmx[1] = fmaxf(bmx[1], y);
      bmn[2] = fminf(bmn[2], z); bmx[2] = fmaxf(bmx[2], z);
    }
  }
  __syncthreads();  // everybody has read hist-as-fill-pointers / order: the region becomes best[] + picks[]
  // this LANE's bounding box (its NB consecutive points of the Morton order: a tight cluster), inflated by 1e-5 of its size
  // (rounding of the test below).  The wave skips a round iff NO lane's box is closer to the pick than that lane's largest
  // running distance -- tighter than one box per wave, at the same cost (the test is one value per lane either way).
  float blo[3], bhi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pad = 1e-5f * fmaxf(fabsf(bmn[a]), fabsf(bmx[a])) + 1e-30f;
    blo[a] = bmn[a] - pad;
    bhi[a] = bmx[a] + pad;
  }
  const bool empty_lane = !(bmn[0] <= bmx[0]);
  float2* slots = reinterpret_cast<float2*>(best);  // [2][16] {distance bits, ~tie key}
  if (tid == 0) picks[0] = 0;
  float x1 = spt[0], y1 = spt[1], z1 = spt[2];
  float thr = empty_lane ? -1.f : INFINITY;  // the lane is active while dist^2(p, its box) < thr  (= its largest running distance, inflated)
  uint32_t cand_d = PASNL_FPS_ABL == 1 ? (uint32_t)(wave + 1) : 0u, cand_k = 0u;  // the wave's candidate: largest running distance (bits), ~tie key
  __syncthreads();

  for (int j = 1; j < m; ++j) {
    const float ex = fmaxf(fmaxf(blo[0] - x1, x1 - bhi[0]), 0.f), ey = fmaxf(fmaxf(blo[1] - y1, y1 - bhi[1]), 0.f),
                ez = fmaxf(fmaxf(blo[2] - z1, z1 - bhi[2]), 0.f);
    const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
#if defined(PASNL_TUNING) && PASNL_FPS_ABL == 0
    if ((blockIdx.x & 3) == 0) {
      const bool act = __ballot(lb < thr) != 0ull;
      if (lane == 0) { atomicAdd(&fps_dbg[1], 1ull); if (act) { atomicAdd(&fps_dbg[0], 1ull); atomicAdd(&dbg_act[j % 3][wave & 3], 1); } }
    }
#endif
    if ((PASNL_FPS_ABL == 2 || __ballot(lb < thr) != 0ull) && PASNL_FPS_ABL != 1) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < NB / 2; ++q) {
        const f32x2 dx = px[q] - x1, dy = py[q] - y1, dz = pz[q] - z1;
        const f32x2 d = (dx * dx + dy * dy) + dz * dz;
        td[2 * q] = min(td[2 * q], __float_as_uint(d[0]));
        td[2 * q + 1] = min(td[2 * q + 1], __float_as_uint(d[1]));
      }
      // in-lane maximum of (td, ~tie key): the distance by a max tree, then the largest key among the slots that hold it
      // (short dependent chains: a linear scan of 64-bit compares was the longest piece of an active round)
      uint32_t t2[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) t2[i] = td[i];
#pragma unroll
      for (int stp = 1; stp < NB; stp *= 2)
#pragma unroll
        for (int i = 0; i + stp < NB; i += 2 * stp) t2[i] = max(t2[i], t2[i + stp]);
      const uint32_t bd = t2[0];
      uint32_t k2[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) k2[i] = td[i] == bd ? nkey[i] : 0u;
#pragma unroll
      for (int stp = 1; stp < NB; stp *= 2)
#pragma unroll
        for (int i = 0; i + stp < NB; i += 2 * stp) k2[i] = max(k2[i], k2[i + stp]);
      const uint32_t bkey = k2[0];
      const int wmaxi = __builtin_amdgcn_readlane(wave_max_i32_to_lane63((int)bd), 63);  // distances >= 0: signed order is fine
      unsigned long long tie = __ballot(bd == (uint32_t)wmaxi);
      uint32_t wkey = (uint32_t)__builtin_amdgcn_readlane((int)bkey, (int)__builtin_ctzll(tie));
      if (__builtin_popcountll(tie) > 1) {
        tie &= tie - 1;
        while (tie) {
          const int l = (int)__builtin_ctzll(tie);
          tie &= tie - 1;
          const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)bkey, l);
          wkey = kk > wkey ? kk : wkey;
        }
      }
      cand_d = (uint32_t)wmaxi;
      cand_k = wkey;
      // no point of this lane changes while dist^2(p, its box) >= (its largest running distance) (1 + 1e-5)
      thr = empty_lane ? -1.f : __uint_as_float(bd) * 1.00001f;
    }
    float2* slot = slots + (j & 1) * 16;
    if (lane == 0) slot[wave] = make_float2(__uint_as_float(cand_d), __uint_as_float(cand_k));
    __syncthreads();
    const float2 sv = lane < WAVES ? slot[lane] : make_float2(0.f, 0.f);
#if defined(PASNL_TUNING) && PASNL_FPS_ABL == 0
    if (tid == 0 && (blockIdx.x & 3) == 0) {  // rounds by the number of active waves on the busiest SIMD (waves w, w+4, .. share one)
      const int mxa = max(max(dbg_act[j % 3][0], dbg_act[j % 3][1]), max(dbg_act[j % 3][2], dbg_act[j % 3][3]));
      atomicAdd(&fps_dbg[2 + min(mxa, 4)], 1ull);
      for (int q = 0; q < 4; ++q) dbg_act[(j + 2) % 3][q] = 0;
    }
#endif
    const int di = (int)__float_as_uint(sv.x);
    const uint32_t ki = __float_as_uint(sv.y);
    const int gmax = __builtin_amdgcn_readlane(row_max_i32_to_lane15(di), 15);
    unsigned long long wt = __ballot(di == gmax && lane < WAVES) & 0xffffull;
    uint32_t gkey = (uint32_t)__builtin_amdgcn_readlane((int)ki, (int)__builtin_ctzll(wt));
    if (__builtin_popcountll(wt) > 1) {
      wt &= wt - 1;
      while (wt) {
        const int l = (int)__builtin_ctzll(wt);
        wt &= wt - 1;
        const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)ki, l);
        gkey = kk > gkey ? kk : gkey;
      }
    }
    const int old = (int)(~gkey & 0x3fffffu);
    x1 = spt[old * 3]; y1 = spt[old * 3 + 1]; z1 = spt[old * 3 + 2];
    if (tid == 0) picks[j] = old;
  }
  __syncthreads();
  int* out = idx + (size_t)blockIdx.x * m;
  for (int j = tid; j < m; j += T) out[j] = picks[j];
  fps_emit_xyz<T>(xyz + (size_t)blockIdx.x * n * 3, picks, m, out_xyz ? out_xyz + (size_t)blockIdx.x * m * 3 : nullptr, tid);
}

template <int WAVES, int NB>
static int fps_pruned_launch(int b, int n, int m, const float* xyz, int* idx, hipStream_t st, float* oxyz = nullptr) {
  size_t lds = (size_t)n * 12 + (size_t)((n + 1) & ~1) * 2;
  lds = (lds + 15) & ~(size_t)15;
  const size_t tail = (size_t)4096 * 4 > (size_t)(64 + m) * 4 ? (size_t)4096 * 4 : (size_t)(64 + m) * 4;
  lds += tail;
  if (lds > 160 * 1024 - 1024 || n > 65535) return PASNL_EUNSUPPORTED;
  auto kern = fps_pruned_kernel<WAVES, NB>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(b), dim3(WAVES * 64), lds, st, n, m, xyz, idx, oxyz);
  return pasnl_launch_status();
}

#ifdef PASNL_TUNING
#include "../../tools/experimental/fps_multi.inc"  // several picks per round: measured slower, tuning build only (EXPERIMENTS.md)
#endif

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

#ifdef PASNL_TUNING
extern "C" int pasnl_fps_dbg_read(unsigned long long* host4) {
  if (hipMemcpyFromSymbol(host4, HIP_SYMBOL(pasnl::fps_dbg), sizeof(pasnl::fps_dbg)) != hipSuccess) return -1;
  unsigned long long zero[8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pasnl::fps_dbg), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

static int fps_dispatch(int b, int n, int m, const float* xyz, int* idx, float* oxyz, pasnl_stream_t stream) {
  PASNL_REQUIRE(m > 0, PASNL_EINVAL);  // "FarthestPointSample expects positive npoint"
  PASNL_REQUIRE(b >= 0 && n > 0, PASNL_EINVAL);
  PASNL_REQUIRE(n < (1 << 22), PASNL_EUNSUPPORTED);
  if (b == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && idx, PASNL_ENULL);
  hipStream_t st = pasnl_hip_stream(stream);
  // lanes x points-per-lane must cover n.  PASNL_FPS_CFG="waves,ppl" overrides the table (tuning only).
  const char* cfg = tune_env("PASNL_FPS_CFG");
  if (cfg) {
    int w = 0, p = 0;
    if (sscanf(cfg, "%d,%d", &w, &p) == 2 && (long)w * 64 * p >= n) {
#define PASNL_FPS_TRY(W, P) if (w == W && p == P) return fps_launch<W, P>(b, n, m, xyz, idx, st, oxyz);
      PASNL_FPS_TRY(1, 2) PASNL_FPS_TRY(1, 4) PASNL_FPS_TRY(1, 8) PASNL_FPS_TRY(1, 16) PASNL_FPS_TRY(2, 4) PASNL_FPS_TRY(2, 8)
      PASNL_FPS_TRY(2, 16) PASNL_FPS_TRY(4, 2) PASNL_FPS_TRY(4, 4) PASNL_FPS_TRY(4, 8) PASNL_FPS_TRY(4, 16) PASNL_FPS_TRY(8, 2)
      PASNL_FPS_TRY(8, 4) PASNL_FPS_TRY(8, 8) PASNL_FPS_TRY(8, 16) PASNL_FPS_TRY(16, 2) PASNL_FPS_TRY(16, 4) PASNL_FPS_TRY(16, 8)
      PASNL_FPS_TRY(16, 10)
#undef PASNL_FPS_TRY
    }
  }
#ifdef PASNL_TUNING
  if (cfg && cfg[0] == 'a') return fps_small_launch<4, 4, 1>(b, n, m, xyz, idx, st);
  if (cfg && cfg[0] == 'b') return fps_small_launch<4, 4, 2>(b, n, m, xyz, idx, st);
  if (cfg && cfg[0] == 's') {  // "s<waves>,<ppl>": the small-cloud kernel with an explicit shape
    int w = 0, p = 0;
    if (sscanf(cfg + 1, "%d,%d", &w, &p) == 2 && (long)w * 64 * p >= n && m <= 8192) {
#define PASNL_FPS_TRY(W, P) if (w == W && p == P) return fps_small_launch<W, P>(b, n, m, xyz, idx, st);
      PASNL_FPS_TRY(1, 2) PASNL_FPS_TRY(1, 4) PASNL_FPS_TRY(1, 8) PASNL_FPS_TRY(1, 16) PASNL_FPS_TRY(2, 2) PASNL_FPS_TRY(2, 4)
      PASNL_FPS_TRY(2, 8) PASNL_FPS_TRY(2, 16) PASNL_FPS_TRY(4, 2) PASNL_FPS_TRY(4, 4) PASNL_FPS_TRY(4, 8) PASNL_FPS_TRY(4, 16)
#undef PASNL_FPS_TRY
    }
  }
#endif
  if (n <= 128) return fps_launch<1, 2>(b, n, m, xyz, idx, st, oxyz);
  if (n <= 256) return fps_launch<1, 4>(b, n, m, xyz, idx, st, oxyz);
  if (n <= 512) return fps_launch<1, 8>(b, n, m, xyz, idx, st, oxyz);
  // 513..1024 points.  Up to ~2 clouds per CU the four-wave workgroup wins (the round is a latency chain and four waves share
  // its arithmetic: 199 vs 246 us at B = 64, 203 vs 250 at B = 256, 1024 -> 512); beyond that the CUs are full and ONE wave
  // per cloud -- 16 points per lane, no barrier, no LDS hop between waves -- is cheaper per cloud: 269 vs 318 us at B = 1024,
  // 387 vs 524 at B = 2048 (tools/fps_batch_sweep.py, profiles/r04_g_fps_batch_sweep.txt)
  if (n <= 1024) return b > 640 ? fps_launch<1, 16>(b, n, m, xyz, idx, st, oxyz) : fps_launch<4, 4>(b, n, m, xyz, idx, st, oxyz);
  if (n <= 2048) return fps_launch<4, 8>(b, n, m, xyz, idx, st, oxyz);   // measured (tools/fps_cfg_sweep.py): 146 vs 207 us for (2,16) at 8x1280->320, 231 vs 329 us at 16x2048->512
  if (!tune_env("PASNL_FPS_NOPRUNE")) {
    // pruned rounds (fps_pruned_kernel): the unpruned kernels below stay as the A/B reference
    int rc = PASNL_EUNSUPPORTED;
#ifdef PASNL_TUNING
    const char* kenv = tune_env("PASNL_FPS_K");  // 2..4 = picks per round (fps_multi_kernel, a measurement); default: one
    const int kk = kenv ? atoi(kenv) : 0;
#define PASNL_FPS_BIG(NB)                                                            \
    (kk == 2 ? fps_multi_launch<16, NB, 2>(b, n, m, xyz, idx, st, oxyz)               \
     : kk == 3 ? fps_multi_launch<16, NB, 3>(b, n, m, xyz, idx, st, oxyz)             \
     : kk == 4 ? fps_multi_launch<16, NB, 4>(b, n, m, xyz, idx, st, oxyz)             \
               : fps_pruned_launch<16, NB>(b, n, m, xyz, idx, st, oxyz))
#else
#define PASNL_FPS_BIG(NB) fps_pruned_launch<16, NB>(b, n, m, xyz, idx, st, oxyz)
#endif
    if (n <= 4096) rc = PASNL_FPS_BIG(4);
    else if (n <= 8192) rc = PASNL_FPS_BIG(8);
    else if (n <= 10240) rc = PASNL_FPS_BIG(10);
#undef PASNL_FPS_BIG
    if (rc != PASNL_EUNSUPPORTED) return rc;
  }
  if (n <= 4096) return fps_launch<4, 16>(b, n, m, xyz, idx, st, oxyz);
  if (n <= 8192) return fps_launch<16, 8>(b, n, m, xyz, idx, st, oxyz);
  if (n <= 10240) return fps_launch<16, 10>(b, n, m, xyz, idx, st, oxyz);
  return PASNL_EUNSUPPORTED;  // > 10240 points/cloud (or cloud + picks > 160 KiB LDS): LDS-resident design limit = the
                              // largest reference config (SemanticKITTI, 10240 points)
}

extern "C" int pasnl_farthest_point_sample(int b, int n, int m, const float* xyz, int* idx, pasnl_stream_t stream) {
  return fps_dispatch(b, n, m, xyz, idx, nullptr, stream);
}

#ifdef PASNL_TUNING
extern "C" void pasnl_tuning_stamp(int slot, hipStream_t st);
#endif
extern "C" int pasnl_farthest_point_sample_gather(int b, int n, int m, const float* xyz, int* idx, float* new_xyz,
                                                  pasnl_stream_t stream) {
  PASNL_REQUIRE(b == 0 || new_xyz, PASNL_ENULL);
#ifdef PASNL_TUNING
  const bool stamp = pasnl::tune_env("PASNL_STAMP_N") && atoi(pasnl::tune_env("PASNL_STAMP_N")) == n;  // (tools/step_stamps.py)
  if (stamp) pasnl_tuning_stamp(4, pasnl_hip_stream(stream));
  const int rc = fps_dispatch(b, n, m, xyz, idx, new_xyz, stream);
  if (stamp) pasnl_tuning_stamp(5, pasnl_hip_stream(stream));
  return rc;
#else
  return fps_dispatch(b, n, m, xyz, idx, new_xyz, stream);
#endif
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
