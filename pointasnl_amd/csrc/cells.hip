// PointASNL cells for gfx950: the attention cores of PointNonLocalCell and SampleWeights, and the
// AdaptiveSampling re-weighting tail.  Behaviour contract: reference utils/pointasnl_util.py:112-219
// (chains of separate TF ops that materialise the (B,P,N) attention map); restated in oracle/cells.py.
// The 1x1 convolutions around these cores are plain GEMMs and stay with the vendor BLAS on the host side.
//
// fp32 in, fp32 out, tolerance 1e-5 against the fp32 oracle.  The matrix products run on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains at the vector-FMA rate, which
// leaves the VALU free for the softmax); a vector-FMA variant of the non-local kernel is kept so the
// choice is measured, not assumed (bench.py --ops, profiles/).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.hpp"

namespace pasnl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// A value combined across the two halves of the wave (lane l with lane l ^ 32): v_permlane32_swap_b32 with both operands the
// same register leaves {value of lane l & 31, value of lane (l & 31) + 32} in every lane -- one vector instruction where
// __shfl_xor(v, 32) is a ds_bpermute_b32 (an LDS round trip).  lo + hi in both halves: the same bits as own + other.
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// row index held in register r of lane-half h of a 32x32 MFMA C/D tile
__device__ __forceinline__ int kappa(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// =============================================================================================
// Non-local attention, MFMA variant with LDS staging ("swapped" flash form, keys split across waves).  The kernel
// for cb = 128; cb <= 64 runs nl_attention_direct_kernel (below), which takes its operands straight from L2.
//   A workgroup owns 32 queries; its SPLIT waves each walk every SPLIT-th block of 32 keys, so even a layer
//   with few query tiles (cls layer2: 256) has enough waves in flight.  (More waves do NOT overlap one wave's
//   softmax with another's MFMAs -- both run on the SIMD's vector lanes -- nor, as it turned out, its memory
//   round trips: DESIGN.md 4.)  Staging is wave-private (each wave copies its own 32 K/V rows into its own LDS
//   region): no workgroup barrier inside the loop.  The partial (max, sum, O) of the SPLIT waves are merged
//   through LDS at the end (one barrier).
//   Per key block a wave forms S^T = K_blk . Q^T with cb/2 32x32x2 MFMAs, so lane l holds, for ITS query
//   (l & 31), the 16 keys kappa(r,h) = (r&3) + 8*(r>>2) + 4*h, h = l>>5: the softmax statistics of a query are
//   16 in-lane values plus one exchange with lane l^32 -- no transposes.  P^T is then already in B-operand
//   position for O^T += V^T . P^T when MFMA step t is defined to contract key kappa(t,h): the A operand
//   V[kappa(t,h)][c0 + (l&31)] is a conflict-free LDS row read.
//   K rows are stored with stride CB+1 so that the A-operand read K[l&31][2t+h] is conflict-free.
// =============================================================================================
// Diagnostic builds only (make probe: -DPASNL_SA_CELL_PROBE=<level> [-DPASNL_SA_ABLATE=<mask>] -> libpasnl_hip_probe*.so,
// tools/sa_cell_probe.py): s_memtime marks at the phase boundaries of a tile, summed over all waves into sa_probe[].
// Level 2 adds marks around explicit vmcnt(0) waits (tile start, every chunk), which separates "waiting for gathered
// rows" from matrix work at the price of perturbing the LDS prefetch.  The ablation mask removes one ingredient at a
// time (results are then wrong; only the time matters): 1 skip maxima, 2 global operand loads, 4 output stores.
#ifdef PASNL_SA_CELL_PROBE
__device__ unsigned long long sa_probe[16];
__device__ unsigned long long nl_probe[16];
#define SA_MARK0(t) do { __builtin_amdgcn_sched_barrier(0); t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#if PASNL_SA_CELL_PROBE >= 1
#define SA_MARK(t) SA_MARK0(t)
#else
#define SA_MARK(t) t = 0   /* level 0: only the wave's total (two marks per wave: the code is the production code) */
#endif
#define SA_PROBE(...) __VA_ARGS__
#if PASNL_SA_CELL_PROBE >= 2
#define SA_MARK2(t) SA_MARK(t)
#define SA_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SA_MARK2(t) t = 0
#define SA_WAIT_VM()
#endif
#else
#define SA_MARK(t)
#define SA_MARK2(t)
#define SA_WAIT_VM()
#define SA_PROBE(...)
#endif
// streaming (non-temporal) stores of the cell's (C2 x 32)-per-group output: written once, read by the next kernel from HBM
// anyway, and without the hint 67-268 MB per launch wash through every XCD's L2 (cls step 1.497 -> 1.483 ms; the cell itself
// takes the same time)
#ifndef PASNL_SA_NT
#define PASNL_SA_NT 1
#endif
#ifndef PASNL_SA_ABLATE
#define PASNL_SA_ABLATE 0
#endif

constexpr int NL_KB = 32;  // keys per block

template <int CB, int SPLIT>
__global__ __launch_bounds__(SPLIT * 64) void nl_attention_mfma_kernel(int p, int n, float qscale,
                                                                     const float* __restrict__ q,
                                                                     const float* __restrict__ kv,
                                                                     float* __restrict__ out) {
  constexpr int KS = CB + 1;                              // padded K row stride
  constexpr int WAVE_FLOATS = NL_KB * KS + NL_KB * CB + 3;  // +3: room to align V to 16 bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Ks = reinterpret_cast<float*>(smem) + (size_t)wave * ((WAVE_FLOATS + 3) & ~3);  // [NL_KB][KS]
  float* Vs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(Ks + NL_KB * KS) + 15) & ~uintptr_t(15));  // [NL_KB][CB]

  const int h = lane >> 5, ql = lane & 31;
  // XCD-aware mapping: workgroups are dispatched round-robin over the 8 XCDs in linear order; all query tiles of a
  // cloud go to ONE XCD (cloud % 8), so its K/V block (256 KiB at cls layer1) is fetched into one L2 once instead of
  // into all eight (measured before: 143 MB of L2-miss fetches per launch for 17 MB of K/V)
  int bi = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, slot = id >> 3;
    bi = (id & 7) + 8 * (slot / (int)gridDim.x);
    qt = slot % (int)gridDim.x;
  }
  const int q0 = qt * 32;
  const int qi = min(q0 + ql, p - 1);
  const float* kvb = kv + (size_t)bi * n * 2 * CB;

  // Q as MFMA B operand: B[k = h][j = ql] = Q[q][2t+h] * (log2e / sqrt(cb))
  float qreg[CB / 2];
  {
    const float* qp = q + ((size_t)bi * p + qi) * CB;
#pragma unroll
    for (int t = 0; t < CB / 2; ++t) qreg[t] = qp[2 * t + h] * qscale;
  }
  f32x16 O[CB / 32];
#pragma unroll
  for (int c = 0; c < CB / 32; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  SA_PROBE(unsigned long long n0, n1, n2, n3, n4, nk0, a_st = 0, a_s = 0, a_sm = 0, a_pv = 0, a_blk = 0; SA_MARK0(nk0);)

  // Register prefetch (cb <= 64): the [K | V] rows of block i+1 are requested before the products of block i and written
  // to the wave's LDS region after them, so a wave overlaps ITS OWN memory round trip with its own matrix work.  (More
  // waves per SIMD do not: the waves of a launch start together and run phases of identical length, so they wait for
  // memory together and compete for the pipe together -- the loop took the same 249 k cycles per SIMD with 1, 2 or 4
  // waves on it.)  Lane l always copies float4 column l % F4 of rows l / F4 + i * RPI: offsets are loop constants.
  constexpr bool PF = CB <= 64;
  constexpr int F4 = 2 * CB / 4, RPI = PF ? 64 / F4 : 1, NIT = PF ? NL_KB / RPI : 1;
  const int pr0 = lane / F4, pc4 = (lane - pr0 * F4) * 4;
  float4 pf[NIT];
  auto request_block = [&](int base) {
    const int cnt = min(NL_KB, n - base);
#pragma unroll
    for (int i = 0; i < NIT; ++i)
      pf[i] = *reinterpret_cast<const float4*>(kvb + (size_t)(base + min(pr0 + i * RPI, cnt - 1)) * 2 * CB + pc4);
  };
  if constexpr (PF) {
    if (wave * NL_KB < n) request_block(wave * NL_KB);
  }

  for (int base = wave * NL_KB; base < n; base += SPLIT * NL_KB) {
    SA_MARK(n0);
    const int cnt = min(NL_KB, n - base);
    if constexpr (PF) {
      // rows past cnt are zero-filled: their probabilities are 0, and 0 * stale-LDS-NaN must not reach the accumulator
      if (cnt < NL_KB) {  // only the last block of a cloud whose size is not a multiple of 32
#pragma unroll
        for (int i = 0; i < NIT; ++i)
          if (pr0 + i * RPI >= cnt) pf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (pc4 < CB) {
        float* d = Ks + pr0 * KS + pc4;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          d[i * RPI * KS] = pf[i].x; d[i * RPI * KS + 1] = pf[i].y; d[i * RPI * KS + 2] = pf[i].z; d[i * RPI * KS + 3] = pf[i].w;
        }
      } else {
        float* d = Vs + pr0 * CB + (pc4 - CB);
#pragma unroll
        for (int i = 0; i < NIT; ++i) *reinterpret_cast<float4*>(d + i * RPI * CB) = pf[i];
      }
      if (base + SPLIT * NL_KB < n) request_block(base + SPLIT * NL_KB);
    } else {
    // wave-private staging of 32 [K | V] rows (coalesced float4 reads); rows past cnt are zero-filled: their
    // probabilities are 0, and 0 * stale-LDS-NaN must not reach the accumulator
#pragma unroll 2
    for (int f = lane; f < NL_KB * (2 * CB / 4); f += 64) {
      int row = f / (2 * CB / 4), c4 = (f - row * (2 * CB / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < cnt) v = *reinterpret_cast<const float4*>(kvb + (size_t)(base + row) * 2 * CB + c4);
      if (c4 < CB) {
        float* d = Ks + row * KS + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      } else {
        *reinterpret_cast<float4*>(Vs + row * CB + (c4 - CB)) = v;
      }
    }
    }
    // (LDS operations of one wave execute in order: no barrier needed before reading the region back)
    SA_MARK(n1);
    // ---- S^T = K_blk . Q^T
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const float* krow = Ks + ql * KS + h;
#pragma unroll
    for (int t = 0; t < CB / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * t], qreg[t], S, 0, 0, 0);
    // the V operands of the first 32-channel block are requested now: they arrive under the softmax
    float vop[16];
    {
      const float* vcol0 = Vs + ql;
#pragma unroll
      for (int t = 0; t < 16; ++t) vop[t] = vcol0[kappa(t, h) * CB];
    }
    SA_MARK(n2);
    // ---- online softmax over this lane's 16 keys (+ the other half-wave's 16)
    float tmax = -INFINITY;
    if (cnt < NL_KB) {  // keys to mask exist only in the last block of a cloud whose size is not a multiple of 32
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = kappa(r, h) < cnt ? S[r] : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(mrun, tmax);  // finite: every block has >= 1 valid key
    const float alpha = fast_exp2(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = fast_exp2(S[r] - mnew);
      psum += S[r];
    }
    psum += __shfl_xor(psum, 32);
    lrun = lrun * alpha + psum;
    mrun = mnew;
    SA_MARK(n3);
    // ---- O^T = alpha * O^T + V^T . P^T
#pragma unroll
    for (int c = 0; c < CB / 32; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
      const float* vcol = Vs + c * 32 + ql;
#pragma unroll
      for (int t = 0; t < 16; ++t)
        O[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(c == 0 ? vop[t] : vcol[kappa(t, h) * CB], S[t], O[c], 0, 0, 0);
    }
    SA_MARK(n4);
    SA_PROBE(a_st += n1 - n0; a_s += n2 - n1; a_sm += n3 - n2; a_pv += n4 - n3; a_blk += 1;)
  }
#ifdef PASNL_SA_CELL_PROBE
  SA_MARK0(n4);
  if (lane == 0) {
    atomicAdd(&nl_probe[0], a_st); atomicAdd(&nl_probe[1], a_s); atomicAdd(&nl_probe[2], a_sm); atomicAdd(&nl_probe[3], a_pv);
    atomicAdd(&nl_probe[4], a_blk); atomicAdd(&nl_probe[5], n4 - nk0); atomicAdd(&nl_probe[6], 1ull);
  }
  SA_MARK0(nk0);
#endif

  if constexpr (SPLIT > 1) {
    // ---- merge the SPLIT partial results: wave w > 0 parks (m, l, O) in its own LDS region, wave 0 folds them in
    float* park = Ks;  // [CB/2 + 2][64]
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < CB / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) park[(c * 16 + r) * 64 + lane] = O[c][r];
      park[(CB / 2) * 64 + lane] = mrun;
      park[(CB / 2 + 1) * 64 + lane] = lrun;
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < SPLIT; ++w) {
      const float* pw = reinterpret_cast<const float*>(smem) + (size_t)w * ((WAVE_FLOATS + 3) & ~3);
      const float mw = pw[(CB / 2) * 64 + lane], lw = pw[(CB / 2 + 1) * 64 + lane];
      const float mnew = fmaxf(mrun, mw);  // a wave that saw no key block has m = -inf, l = 0, O = 0
      const float a0 = mrun == -INFINITY ? 0.f : fast_exp2(mrun - mnew);
      const float a1 = mw == -INFINITY ? 0.f : fast_exp2(mw - mnew);
      lrun = lrun * a0 + lw * a1;
      mrun = mnew;
#pragma unroll
      for (int c = 0; c < CB / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = O[c][r] * a0 + pw[(c * 16 + r) * 64 + lane] * a1;
    }
  }
  // O^T[channel = c*32 + kappa(r,h)][query = ql] / l
  if (q0 + ql < p) {
    const float inv = 1.0f / lrun;
    float* op = out + ((size_t)bi * p + q0 + ql) * CB;
#pragma unroll
    for (int c = 0; c < CB / 32; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[c][4 * g] * inv, O[c][4 * g + 1] * inv, O[c][4 * g + 2] * inv, O[c][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + c * 32 + 8 * g + 4 * h) = v;
      }
  }
}

// =============================================================================================
// Non-local attention, MFMA variant WITHOUT LDS staging (cb <= 64).
//   A key block is used by exactly one wave, so the LDS round trip of the kernel above only re-distributes data
//   between lanes -- and both products can take their A operands from global memory in the layout the loads
//   deliver:
//     * S^T = K . Q^T: the pairing of channels inside an MFMA step is free as long as K and Q agree on it.  Step t
//       contracts channel t (lanes 0..31) and channel cb/2 + t (lanes 32..63): lane (ql, h) needs the cb/2
//       CONTIGUOUS floats K[key ql][h*cb/2 ..] -- cb/8 16-byte loads, no transposition, no bank-conflict padding;
//     * O^T += V^T . P^T: step t contracts key kappa(t, h), its A operand V[key][c*32 + ql] is a coalesced 128-byte
//       row read (lanes over channels) straight from L2.
//   The operand registers are refilled IN PLACE with the next key block as soon as the products that read them
//   have been issued (K under the softmax and P.V, V under the next block's K.Q and softmax): the wave overlaps
//   its own round trips, which more waves per SIMD do not do for it (equal-length phases).
//   LDS only for the final merge of the SPLIT partial results.
// =============================================================================================
// cb = 64: capped at 256 registers (two waves per SIMD; uncapped the compiler takes 210 + 48 accumulation registers = one wave):
// cls layer 2 19.2 -> 17.6 us.  cb = 32 is left alone: four waves per SIMD without accumulation registers, block addresses by
// scalar base + constant offsets (-60 vector instructions per block) and v_permlane32_swap instead of the two LDS shuffles each
// measured SLOWER on the long key loops (scannet 198 -> 207 / 212 / 203 us, all three 231; profiles/r04_q_nl_ab.txt) -- the loop
// is bound by its own chain of round trips, not by issue: fewer instructions between the loads lengthen the measured VMEM latency
template <int CB, int SPLIT>
__global__ __launch_bounds__(SPLIT * 64, CB == 64 ? 2 : 1) void nl_attention_direct_kernel(int p, int n, float qscale,
                                                                       const float* __restrict__ q,
                                                                       const float* __restrict__ kv,
                                                                       float* __restrict__ out) {
  constexpr int HC = CB / 2;                       // MFMA steps of S = channels per half-wave
  constexpr int PARK = (CB / 2 + 2) * 64;          // floats a wave parks for the merge
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;
  int bi = blockIdx.y, qt = blockIdx.x;  // XCD-aware mapping as in nl_attention_mfma_kernel
  if ((gridDim.y & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, slot = id >> 3;
    bi = (id & 7) + 8 * (slot / (int)gridDim.x);
    qt = slot % (int)gridDim.x;
  }
  const int q0 = qt * 32;
  const int qi = min(q0 + ql, p - 1);
  const float* kvb = kv + (size_t)bi * n * 2 * CB;

  float qreg[HC];  // B[k = h][j = ql] = Q[query][h*HC + t] * (log2e / sqrt(cb))
  {
    const float* qp = q + ((size_t)bi * p + qi) * CB + HC * h;
#pragma unroll
    for (int t = 0; t < HC; ++t) qreg[t] = qp[t] * qscale;
  }
  f32x16 O[CB / 32];
#pragma unroll
  for (int c = 0; c < CB / 32; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  // rows past the end of a ragged last block are clamped to its last row: finite values, their probabilities are 0
  float kreg[HC], vreg[CB / 32][16];
  auto load_k = [&](int base) {
    const int cnt = min(NL_KB, n - base);
    const float4* kp = reinterpret_cast<const float4*>(kvb + (size_t)(base + min(ql, cnt - 1)) * 2 * CB + HC * h);
#pragma unroll
    for (int g = 0; g < HC / 4; ++g) {
      const float4 v = kp[g];
      kreg[4 * g] = v.x; kreg[4 * g + 1] = v.y; kreg[4 * g + 2] = v.z; kreg[4 * g + 3] = v.w;
    }
  };
  auto load_v = [&](int base, int c) {
    const int cnt = min(NL_KB, n - base);
#pragma unroll
    for (int t = 0; t < 16; ++t) vreg[c][t] = kvb[(size_t)(base + min(kappa(t, h), cnt - 1)) * 2 * CB + CB + c * 32 + ql];
  };
  const int first = wave * NL_KB;
  if (first < n) {
    // K strictly BEFORE V, as in the loop: loads return in order, and the wait in front of the first K.Q step is the
    // weaker of "entered from here" and "came round the back edge".  With V first (the compiler's choice for SPLIT = 8) it
    // became vmcnt(0) -- every round drained the V refill it had just issued
    load_k(first);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CB / 32; ++c) load_v(first, c);
    __builtin_amdgcn_sched_barrier(0);
  }

  for (int base = first; base < n; base += SPLIT * NL_KB) {
    const int cnt = min(NL_KB, n - base);
    const int nxt = base + SPLIT * NL_KB < n ? base + SPLIT * NL_KB : base;  // the last refill of a wave is a dummy
    // ---- S^T = K_blk . Q^T
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int t = 0; t < HC; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[t], qreg[t], S, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_k(nxt);
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax over this lane's 16 keys (+ the other half-wave's 16)
    if (cnt < NL_KB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = kappa(r, h) < cnt ? S[r] : -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(mrun, tmax);  // finite: every block has >= 1 valid key
    const float alpha = fast_exp2(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = fast_exp2(S[r] - mnew);
      psum += S[r];
    }
    psum += __shfl_xor(psum, 32);
    lrun = lrun * alpha + psum;
    mrun = mnew;
    // ---- O^T = alpha * O^T + V^T . P^T
#pragma unroll
    for (int c = 0; c < CB / 32; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
#pragma unroll
      for (int t = 0; t < 16; ++t) O[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[c][t], S[t], O[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_v(nxt, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if constexpr (SPLIT > 1) {
    // ---- merge the SPLIT partial results: wave w > 0 parks (m, l, O) in its LDS region, wave 0 folds them in
    float* park = reinterpret_cast<float*>(smem) + (size_t)wave * PARK;
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < CB / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) park[(c * 16 + r) * 64 + lane] = O[c][r];
      park[(CB / 2) * 64 + lane] = mrun;
      park[(CB / 2 + 1) * 64 + lane] = lrun;
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < SPLIT; ++w) {
      const float* pw = reinterpret_cast<const float*>(smem) + (size_t)w * PARK;
      const float mw = pw[(CB / 2) * 64 + lane], lw = pw[(CB / 2 + 1) * 64 + lane];
      const float mnew = fmaxf(mrun, mw);  // a wave that saw no key block has m = -inf, l = 0, O = 0
      const float a0 = mrun == -INFINITY ? 0.f : fast_exp2(mrun - mnew);
      const float a1 = mw == -INFINITY ? 0.f : fast_exp2(mw - mnew);
      lrun = lrun * a0 + lw * a1;
      mrun = mnew;
#pragma unroll
      for (int c = 0; c < CB / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = O[c][r] * a0 + pw[(c * 16 + r) * 64 + lane] * a1;
    }
  }
  if (q0 + ql < p) {  // O^T[channel = c*32 + kappa(r,h)][query = ql] / l
    const float inv = 1.0f / lrun;
    float* op = out + ((size_t)bi * p + q0 + ql) * CB;
#pragma unroll
    for (int c = 0; c < CB / 32; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[c][4 * g] * inv, O[c][4 * g + 1] * inv, O[c][4 * g + 2] * inv, O[c][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + c * 32 + 8 * g + 4 * h) = v;
      }
  }
}

// =============================================================================================
// Non-local attention, cb = 32, TWO query tiles per wave (64 queries per workgroup).  A key block's K and V registers feed two
// independent score / output chains, so (a) every load serves twice the matrix work and (b) the wave has something for the
// matrix pipe while it computes a softmax: the program order is
//     S0 = K.Q0 | S1 = K.Q1 with softmax(S0) between its products | O0 += V.P0 with softmax(S1) between | O1 += V.P1
// (sched_group_barrier: one product, then a few vector instructions), where the one-tile kernel above leaves the pipe idle
// during its softmax unless another wave happens to be in a product phase.  Same arithmetic per tile as the kernel above (same
// key blocks per wave, same merge order).  cls layer 1 64.6 -> 55 us, ScanNet layer 1 202 -> 178, KITTI layer 1_1 222 -> 200
// (profiles/r04_q_nl_pair.txt); without its refills the loop runs 2-3 % faster: it is not waiting for memory.
// =============================================================================================
// PART ("flash-decoding"): where b * ceil(p / 64) workgroups leave CUs empty (KITTI layer 1_1: 160 for 256 CUs),
// the KEYS are split over gridDim.z workgroups as well: workgroup z owns the key blocks [z * bpz, (z + 1) * bpz), merges its waves
// as before and leaves (O, m, l) un-normalised in a workspace; nl_attention_merge_kernel combines the parts in ascending z -- a
// fixed order, so the result is a pure function of the inputs -- and normalises.
template <int SPLIT, bool PART = false>
__global__ __launch_bounds__(SPLIT * 64, 2) void nl_attention_pair_kernel(int p, int n, float qscale, const float* __restrict__ q,
                                                                     const float* __restrict__ kv, float* __restrict__ out,
                                                                     int bpz = 0, float* __restrict__ part = nullptr) {
  constexpr int CB = 32, HC = 16;
  constexpr int PARK = (CB / 2 + 2) * 64;  // floats a wave parks per tile for the merge
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ql = lane & 31;
  int bi = blockIdx.y, qt = blockIdx.x;  // XCD-aware mapping as in nl_attention_mfma_kernel
  if ((gridDim.y & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, slot = id >> 3;
    bi = (id & 7) + 8 * (slot / (int)gridDim.x);
    qt = slot % (int)gridDim.x;
  }
  const int q0 = qt * 64;
  const float* kvb = kv + (size_t)bi * n * 2 * CB;

  float qreg[2][HC];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float* qp = q + ((size_t)bi * p + min(q0 + 32 * u + ql, p - 1)) * CB + HC * h;
#pragma unroll
    for (int t = 0; t < HC; ++t) qreg[u][t] = qp[t] * qscale;
  }
  f32x16 O[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[u][r] = 0.f;
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};

  float kreg[HC], vreg[16];
  // n is a multiple of the 32-key block here (the launcher's condition): no ragged block, so a block's rows are ONE lane
  // pointer each for K and V plus compile-time offsets (V: rows kappa(t, h) - 14 around row 14, inside the load's 13-bit
  // immediate), advanced by a uniform stride per block -- 4 address registers instead of 40
  // the key blocks of this workgroup: all of them, or (PART) its slice
  const int kfirst = PART ? (int)blockIdx.z * bpz * NL_KB : 0;
  const int kend = PART ? min(n, kfirst + bpz * NL_KB) : n;
  const float* kp = kvb + (size_t)(kfirst + wave * NL_KB + ql) * 2 * CB + HC * h;
  const float* vp = kvb + (size_t)(kfirst + wave * NL_KB + 4 * h + 14) * 2 * CB + CB + ql;
  auto load_k = [&]() {
#pragma unroll
    for (int g = 0; g < HC / 4; ++g) {
      const float4 v = reinterpret_cast<const float4*>(kp)[g];
      kreg[4 * g] = v.x; kreg[4 * g + 1] = v.y; kreg[4 * g + 2] = v.z; kreg[4 * g + 3] = v.w;
    }
  };
  auto load_v = [&]() {
#pragma unroll
    for (int t = 0; t < 16; ++t) vreg[t] = vp[((t & 3) + 8 * (t >> 2) - 14) * 2 * CB];
  };
  // online softmax of one tile's 32 x 32 scores (this lane: 16 keys of one query; the other half-wave holds the other 16), cut
  // into 16 pieces of a few vector instructions: piece i is issued right behind product i of the OTHER tile's chain and runs in
  // its shadow (a 32x32x2 product occupies the matrix pipe for 64 cycles = 16 vector issue slots)
  float tmx, mnew_, psum_;
  auto softmax_piece = [&](f32x16& S, int u, float& alpha, int i) {
    if (i == 0) {
      tmx = fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3]));
    } else if (i < 4) {
      tmx = fmaxf(tmx, fmaxf(fmaxf(S[4 * i], S[4 * i + 1]), fmaxf(S[4 * i + 2], S[4 * i + 3])));
    } else if (i == 4) {
      tmx = half_max(tmx);  // v_permlane32_swap: a vector instruction -- an LDS shuffle here would stall the pieces (and the
                            // product) queued behind it for a round trip
      mnew_ = fmaxf(mrun[u], tmx);
      alpha = fast_exp2(mrun[u] - mnew_);
      psum_ = 0.f;
    } else if (i < 13) {  // pieces 5 .. 12: two probabilities each, summed in key order like the one-tile kernel
      const int r = 2 * (i - 5);
      S[r] = fast_exp2(S[r] - mnew_);
      psum_ += S[r];
      S[r + 1] = fast_exp2(S[r + 1] - mnew_);
      psum_ += S[r + 1];
    } else if (i == 13) {
      psum_ = half_sum(psum_);
      lrun[u] = lrun[u] * alpha + psum_;
      mrun[u] = mnew_;
    } else {  // pieces 14, 15: the running output rescaled
#pragma unroll
      for (int r = 8 * (i - 14); r < 8 * (i - 14) + 8; ++r) O[u][r] *= alpha;
    }
  };

  const int first = kfirst + wave * NL_KB;
  constexpr size_t STRIDE = (size_t)SPLIT * NL_KB * 2 * CB;  // floats between a wave's consecutive key blocks
  if (first < kend) {
    load_k();
    __builtin_amdgcn_sched_barrier(0);
    load_v();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int base = first; base < kend; base += SPLIT * NL_KB) {
    if (base + SPLIT * NL_KB < kend) { kp += STRIDE; vp += STRIDE; }  // (uniform) the last refill of a wave re-reads its block
    f32x16 S0, S1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S0[r] = 0.f; S1[r] = 0.f; }
    float alpha0, alpha1;
#pragma unroll
    for (int t = 0; t < HC; ++t) S0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[t], qreg[0][t], S0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S1's products with softmax(S0) in their shadow
#pragma unroll
    for (int t = 0; t < HC; ++t) {
      S1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[t], qreg[1][t], S1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      softmax_piece(S0, 0, alpha0, t);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef PASNL_NL_PAIR_NOLOAD  // (timing ablation: the loop without its refills)
    load_k();
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---- O0 += V . P0 with softmax(S1) in its shadow
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[t], S0[t], O[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      softmax_piece(S1, 1, alpha1, t);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[t], S1[t], O[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#ifndef PASNL_NL_PAIR_NOLOAD
    load_v();
#endif
    __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (SPLIT > 1) {
    float* park = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * PARK;
    if (wave > 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) park[u * PARK + r * 64 + lane] = O[u][r];
        park[u * PARK + 16 * 64 + lane] = mrun[u];
        park[u * PARK + 17 * 64 + lane] = lrun[u];
      }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 1; w < SPLIT; ++w)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* pw = reinterpret_cast<const float*>(smem) + (size_t)w * 2 * PARK + u * PARK;
        const float mw = pw[16 * 64 + lane], lw = pw[17 * 64 + lane];
        const float mnew = fmaxf(mrun[u], mw);  // a wave that saw no key block has m = -inf, l = 0, O = 0
        const float a0 = mrun[u] == -INFINITY ? 0.f : fast_exp2(mrun[u] - mnew);
        const float a1 = mw == -INFINITY ? 0.f : fast_exp2(mw - mnew);
        lrun[u] = lrun[u] * a0 + lw * a1;
        mrun[u] = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[u][r] = O[u][r] * a0 + pw[r * 64 + lane] * a1;
      }
  } else if (wave > 0) {
    return;
  }
  if constexpr (PART) {  // (wave 0) this slice's running state, in the layout the waves park theirs: [tile][16 O rows, m, l][lane]
    float* pz = part + (((size_t)bi * gridDim.x + qt) * gridDim.z + blockIdx.z) * 2 * PARK;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pz[u * PARK + r * 64 + lane] = O[u][r];
      pz[u * PARK + 16 * 64 + lane] = mrun[u];
      pz[u * PARK + 17 * 64 + lane] = lrun[u];
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (q0 + 32 * u + ql < p) {
      const float inv = 1.0f / lrun[u];
      float* op = out + ((size_t)bi * p + q0 + 32 * u + ql) * CB;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[u][4 * g] * inv, O[u][4 * g + 1] * inv, O[u][4 * g + 2] * inv, O[u][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + 8 * g + 4 * h) = v;
      }
    }
}

// the parts of nl_attention_pair_kernel<., true> combined in ascending z (the same rescaling as the merge of a workgroup's waves),
// normalised and written: one wave per pair of query tiles; grid = (ceil(p / 64), b)
__global__ __launch_bounds__(64) void nl_attention_merge_kernel(int p, int kparts, const float* __restrict__ part, float* __restrict__ out) {
  constexpr int CB = 32, PARK = (CB / 2 + 2) * 64;
  const int lane = threadIdx.x, h = lane >> 5, ql = lane & 31;
  const int bi = blockIdx.y, qt = blockIdx.x, q0 = qt * 64;
  const float* pb = part + (((size_t)bi * gridDim.x + qt) * kparts) * 2 * PARK;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float O[16], mrun = pb[u * PARK + 16 * 64 + lane], lrun = pb[u * PARK + 17 * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = pb[u * PARK + r * 64 + lane];
    for (int z = 1; z < kparts; ++z) {
      const float* pw = pb + (size_t)z * 2 * PARK + u * PARK;
      const float mw = pw[16 * 64 + lane], lw = pw[17 * 64 + lane];
      const float mnew = fmaxf(mrun, mw);  // a part that saw no key block has m = -inf, l = 0, O = 0
      const float a0 = mrun == -INFINITY ? 0.f : fast_exp2(mrun - mnew);
      const float a1 = mw == -INFINITY ? 0.f : fast_exp2(mw - mnew);
      lrun = lrun * a0 + lw * a1;
      mrun = mnew;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = O[r] * a0 + pw[r * 64 + lane] * a1;
    }
    if (q0 + 32 * u + ql < p) {
      const float inv = 1.0f / lrun;
      float* op = out + ((size_t)bi * p + q0 + 32 * u + ql) * CB;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[4 * g] * inv, O[4 * g + 1] * inv, O[4 * g + 2] * inv, O[4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + 8 * g + 4 * h) = v;
      }
    }
  }
}

// =============================================================================================
// Non-local attention, vector-FMA variant: one query per lane, K/V rows broadcast from LDS.
// Kept for the MFMA-vs-FMA comparison (cb <= 64: q and the accumulator live in VGPRs).
// =============================================================================================
template <int CB, int TB>
__global__ __launch_bounds__(TB) void nl_attention_valu_kernel(int p, int n, float qscale, const float* __restrict__ q,
                                                               const float* __restrict__ kv, float* __restrict__ out) {
  constexpr int TK = 64;
  __shared__ float4 Ks[TK * CB / 4];
  __shared__ float4 Vs[TK * CB / 4];
  const int tid = threadIdx.x;
  const int bi = blockIdx.y;
  const int qi = blockIdx.x * TB + tid;
  const bool ok = qi < p;
  const float* qp = q + ((size_t)bi * p + (ok ? qi : p - 1)) * CB;
  const float* kvb = kv + (size_t)bi * n * 2 * CB;
  float qr[CB], acc[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) { qr[c] = qp[c] * qscale; acc[c] = 0.f; }
  float mrun = -INFINITY, lrun = 0.f;
  for (int base = 0; base < n; base += TK) {
    const int cnt = min(TK, n - base);
    __syncthreads();
    for (int f = tid; f < cnt * (2 * CB / 4); f += TB) {
      int row = f / (2 * CB / 4), c4 = f - row * (2 * CB / 4);
      float4 v = *reinterpret_cast<const float4*>(kvb + (size_t)(base + row) * 2 * CB + c4 * 4);
      if (c4 < CB / 4) Ks[row * (CB / 4) + c4] = v; else Vs[row * (CB / 4) + c4 - CB / 4] = v;
    }
    __syncthreads();
    for (int j0 = 0; j0 < cnt; j0 += 4) {
      float s[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float a = 0.f;
        int j = min(j0 + jj, cnt - 1);
#pragma unroll
        for (int c4 = 0; c4 < CB / 4; ++c4) {
          float4 k4 = Ks[j * (CB / 4) + c4];
          a = __builtin_fmaf(qr[4 * c4], k4.x, a);
          a = __builtin_fmaf(qr[4 * c4 + 1], k4.y, a);
          a = __builtin_fmaf(qr[4 * c4 + 2], k4.z, a);
          a = __builtin_fmaf(qr[4 * c4 + 3], k4.w, a);
        }
        s[jj] = (j0 + jj < cnt) ? a : -INFINITY;
      }
      float mnew = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), mrun);
      float alpha = fast_exp2(mrun - mnew);
      mrun = mnew;
      float pw[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) pw[jj] = fast_exp2(s[jj] - mnew);
      lrun = lrun * alpha + ((pw[0] + pw[1]) + (pw[2] + pw[3]));
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[c] *= alpha;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        int j = min(j0 + jj, cnt - 1);
#pragma unroll
        for (int c4 = 0; c4 < CB / 4; ++c4) {
          float4 v4 = Vs[j * (CB / 4) + c4];
          acc[4 * c4] = __builtin_fmaf(pw[jj], v4.x, acc[4 * c4]);
          acc[4 * c4 + 1] = __builtin_fmaf(pw[jj], v4.y, acc[4 * c4 + 1]);
          acc[4 * c4 + 2] = __builtin_fmaf(pw[jj], v4.z, acc[4 * c4 + 2]);
          acc[4 * c4 + 3] = __builtin_fmaf(pw[jj], v4.w, acc[4 * c4 + 3]);
        }
      }
    }
  }
  if (ok) {
    float inv = 1.0f / lrun;
    float* op = out + ((size_t)bi * p + qi) * CB;
#pragma unroll
    for (int c4 = 0; c4 < CB / 4; ++c4)
      *reinterpret_cast<float4*>(op + 4 * c4) =
          make_float4(acc[4 * c4] * inv, acc[4 * c4 + 1] * inv, acc[4 * c4 + 2] * inv, acc[4 * c4 + 3] * inv);
  }
}

// =============================================================================================
// Adaptive-Sampling micro attention: one wave per group, as <= 16 neighbours padded to a 16x16 tile,
// v_mfma_f32_16x16x4_f32, operands straight from global memory (a group is a few KB, L2/L1 resident).
//   S^T[key][query]:  D[row = 4*(l>>4)+r][col = l&15],   A = K[key = l&15][4t + (l>>4)],
//                                                        B = Q[query = l&15][4t + (l>>4)]
//   softmax over keys = 4 in-lane values + exchanges with lanes l^16, l^32
//   O^T[ch][query] += V^T . P^T with step t contracting key 4*(l>>4)+t
// =============================================================================================
__global__ __launch_bounds__(256) void as_attention_kernel(long groups, int as, int cb, float qscale, int qs, int kvs,
                                                          const float* __restrict__ q, const float* __restrict__ kv,
                                                          float* __restrict__ out) {
  // qs / kvs = row strides of q and kv in floats (cb and 2cb for separate tensors; 3cb for one [K|V|Q] tensor)
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int col = lane & 15, grp = lane >> 4;
  const float* qg = q + (size_t)g * as * qs;
  const float* kg = kv + (size_t)g * as * kvs;
  const bool rowok = col < as;  // this lane's key row (as A operand) / query (as B operand) exists
  f32x4 S = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < cb; c0 += 4) {
    int c = c0 + grp;
    bool okc = rowok && c < cb;
    float a = okc ? kg[(size_t)col * kvs + c] : 0.f;
    float b = okc ? qg[(size_t)col * qs + c] * qscale : 0.f;
    S = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, S, 0, 0, 0);
  }
  float tmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    S[r] = (4 * grp + r) < as ? S[r] : -INFINITY;
    tmax = fmaxf(tmax, S[r]);
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    S[r] = fast_exp2(S[r] - tmax);
    psum += S[r];
  }
  psum += __shfl_xor(psum, 16);
  psum += __shfl_xor(psum, 32);
  const float inv = 1.0f / psum;
  const float* vg = kg + cb;  // V = second half of each kv row
  for (int c0 = 0; c0 < cb; c0 += 16) {
    f32x4 O = {0.f, 0.f, 0.f, 0.f};
    int ch = c0 + col;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int key = 4 * grp + t;
      float a = (key < as && ch < cb) ? vg[(size_t)key * kvs + ch] : 0.f;
      O = __builtin_amdgcn_mfma_f32_16x16x4f32(a, S[t], O, 0, 0, 0);
    }
    // O^T[ch = c0 + 4*grp + r][query = col]
    if (rowok) {
      float* op = out + ((size_t)g * as + col) * cb + c0 + 4 * grp;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c0 + 4 * grp + r < cb) op[r] = O[r] * inv;
    }
  }
}

// =============================================================================================
// AdaptiveSampling tail: softmax over the neighbour axis, then the weighted sums.
// One thread per (group, column) of the (1+ch)-wide logits; column 0 re-weights xyz.
// =============================================================================================
constexpr int AS_MAX = 16;
__global__ __launch_bounds__(256) void as_reweight_kernel(long groups, int as, int nsample, int ch, int xs, int fs,
                                                         const float* __restrict__ logits,
                                                         const float* __restrict__ gxyz, const float* __restrict__ gfeat,
                                                         float* __restrict__ new_xyz, float* __restrict__ new_feature) {
  // xs / fs = row strides of the grouped coordinates / features in floats (3 and ch for separate tensors)
  const int w = 1 + ch;
  const long total = groups * w;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long g = e / w;
    int c = (int)(e - g * w);
    float v[AS_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < AS_MAX; ++k) {
      v[k] = k < as ? logits[((size_t)g * as + k) * w + c] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < AS_MAX; ++k) {
      v[k] = k < as ? expf(v[k] - mx) : 0.f;
      sum += v[k];
    }
    if (c == 0) {
      float ax = 0.f, ay = 0.f, az = 0.f;
      const float* xp = gxyz + (size_t)g * nsample * xs;
#pragma unroll
      for (int k = 0; k < AS_MAX; ++k)
        if (k < as) {
          float wk = v[k] / sum;
          ax += xp[k * xs] * wk; ay += xp[k * xs + 1] * wk; az += xp[k * xs + 2] * wk;
        }
      new_xyz[g * 3] = ax; new_xyz[g * 3 + 1] = ay; new_xyz[g * 3 + 2] = az;
    } else {
      float a = 0.f;
      const float* fp = gfeat + (size_t)g * nsample * fs + (c - 1);
#pragma unroll
      for (int k = 0; k < AS_MAX; ++k)
        if (k < as) a += fp[(size_t)k * fs] * (v[k] / sum);
      new_feature[(size_t)g * ch + (c - 1)] = a;
    }
  }
}


// =============================================================================================
// Set-abstraction "local cell" (pointasnl_util.py:264-274), fused:
//     H1 = relu(X W0 + b0)   (K x C1)       X = new_point of one query: K neighbours x (6+C) channels
//     H2 = relu(H1 W1 + b1)  (K x C2)
//     G  = relu(X[:, 0:3] Ww + bw)  (K x 32)   weight net on the centred coordinates
//     M  = H2^T G            (C2 x 32)      -> out[group] (the input of the [1,C2] `after_conv` GEMM)
// One wave owns one query group and walks its neighbours 32 at a time; everything between X and M lives in
// registers.  The four products are chained on v_mfma_f32_32x32x2_f32 WITHOUT any transpose or LDS round trip
// by alternating transposed / untransposed forms, using that a D tile holds, in lane l, column (l & 31) and the
// 16 rows kappa(r, l>>5) = (r&3) + 8*(r>>2) + 4*(l>>5) -- exactly an A (or B) operand whose k index at MFMA
// step t is kappa(t, l>>5):
//     H1^T[c1][p] : A = W0[c][c1]        (LDS)   B = X[p][c]   (regs)      D rows c1, cols p
//     H2 [p][c2]  : A = H1^T tile regs           B = W1[c1][c2](LDS)       D rows p,  cols c2
//     G  [p][j]   : A = X[p][c] regs             B = Ww[c][j]  (LDS)       D rows p,  cols j
//     M  [c2][j]  : A = H2 tile regs             B = G tile regs           D rows c2, cols j
// LDS holds only the (BN-folded) weights, staged once per persistent workgroup.  fp32 in, fp32 accumulate
// (the MFMA is an exact fp32 fmaf chain); tolerance vs the fp32 oracle 1e-5 relative.
// =============================================================================================
// Sources of the gather-fused cell (pasnl_sa_cell): row s of group (b,j) is [xyz[i]-new_xyz[j] | xyz[i] | feature[i]],
// i = idx[b,j,s] (pointasnl_util.py:63-74,248-249)
struct SaGatherSrc {
  const float* xyz;      // (b,n,3)
  const float* feature;  // (b,n,c), c = w - 6
  const int* idx;        // (b,m,k)
  const float* new_xyz;  // (b,m,3); with centre0: any readable (b,m,3) array (requested, never used)
  float* skip_max;       // (b,m,w)
  int n, m;
  int centre0;           // 1: the centre of a group is its neighbour 0, xyz[b, idx[b,j,0]] (AdaptiveSampling with no neighbours)
  float* new_xyz_out;    // centre0 only, optional: (b,m,3) the centres, and
  float* new_feature_out;  //                          (b,m,3+c) [centre | feature row of neighbour 0]  (pasnl_take_neighbor0's outputs)
};

template <int C1, int C2>
__global__ __launch_bounds__(256) void sa_local_cell_kernel(long groups, int k, int w, const float* __restrict__ x,
                                                           const float* __restrict__ w0, const float* __restrict__ b0,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ ww, const float* __restrict__ bw,
                                                           float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wp = (w + 31) & ~31;                    // input channels padded to whole 32-chunks (zero rows)
  float* W0s = reinterpret_cast<float*>(smem);      // [wp][C1]
  float* W1s = W0s + (size_t)wp * C1;               // [C1][C2]
  float* Wws = W1s + C1 * C2;                       // [4][32]  (row 3 = 0: channel 3 is not a centred coordinate)
  float* B0s = Wws + 4 * 32;                        // [C1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;

  for (int i = tid; i < wp * C1; i += 256) W0s[i] = (i / C1) < w ? w0[i] : 0.f;
  for (int i = tid; i < C1 * C2; i += 256) W1s[i] = w1[i];
  for (int i = tid; i < 4 * 32; i += 256) Wws[i] = i < 3 * 32 ? ww[i] : 0.f;
  for (int i = tid; i < C1; i += 256) B0s[i] = b0[i];
  __syncthreads();

  float b1r[C2 / 32];
#pragma unroll
  for (int cb = 0; cb < C2 / 32; ++cb) b1r[cb] = b1[cb * 32 + ql];
  const float bwr = bw[ql];
  const int nchunk = wp / 32;

  for (long g = (long)blockIdx.x * 4 + wave; g < groups; g += (long)gridDim.x * 4) {
    f32x16 M[C2 / 32];
#pragma unroll
    for (int cb = 0; cb < C2 / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) M[cb][r] = 0.f;

    for (int tile = 0; tile < k; tile += 32) {
      const float* xrow = x + ((size_t)g * k + tile + ql) * w;  // this lane's neighbour row
      f32x16 H1T[C1 / 32];
#pragma unroll
      for (int ob = 0; ob < C1 / 32; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1T[ob][r] = 0.f;
      f32x16 G;
#pragma unroll
      for (int r = 0; r < 16; ++r) G[r] = 0.f;

      for (int ch = 0; ch < nchunk; ++ch) {
        float xr[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          int c = ch * 32 + 2 * t + h;
          xr[t] = c < w ? xrow[c] : 0.f;
        }
        const int live = min(16, (w - ch * 32 + 1) >> 1);  // MFMA steps with a non-zero k pair (uniform)
        if (ch == 0) {
          // weight net: channels 0..2 are the centred coordinates
          G = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0], Wws[h * 32 + ql], G, 0, 0, 0);
          G = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[1], Wws[(2 + h) * 32 + ql], G, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          if (t < live) {
            const float* wrow = W0s + (size_t)(ch * 32 + 2 * t + h) * C1 + ql;
#pragma unroll
            for (int ob = 0; ob < C1 / 32; ++ob)
              H1T[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[ob * 32], xr[t], H1T[ob], 0, 0, 0);
          }
        }
      }
      // bias + ReLU: H1^T rows are output channels c1 = ob*32 + kappa(r,h); G columns are j = ql
#pragma unroll
      for (int ob = 0; ob < C1 / 32; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1T[ob][r] = fmaxf(H1T[ob][r] + B0s[ob * 32 + kappa(r, h)], 0.f);
#pragma unroll
      for (int r = 0; r < 16; ++r) G[r] = fmaxf(G[r] + bwr, 0.f);

#pragma unroll
      for (int cb = 0; cb < C2 / 32; ++cb) {
        f32x16 H2;
#pragma unroll
        for (int r = 0; r < 16; ++r) H2[r] = 0.f;
        // The W1 operands are read in batches of 16 (one 32-channel block), one batch ahead of the MFMAs that use
        // them; sched_barriers keep the compiler from hoisting all C1*C2/32 LDS reads to the top (which spills).
        float wv[2][16];
        const float* w1p = W1s + (size_t)kappa(0, h) * C2 + cb * 32 + ql;
#pragma unroll
        for (int t = 0; t < 16; ++t) wv[0][t] = w1p[(size_t)(kappa(t, 0)) * C2];
#pragma unroll
        for (int blk = 0; blk < C1 / 32; ++blk) {
          if (blk + 1 < C1 / 32) {
#pragma unroll
            for (int t = 0; t < 16; ++t) wv[(blk + 1) & 1][t] = w1p[(size_t)((blk + 1) * 32 + kappa(t, 0)) * C2];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 16; ++t) H2 = __builtin_amdgcn_mfma_f32_32x32x2f32(H1T[blk][t], wv[blk & 1][t], H2, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) H2[r] = fmaxf(H2[r] + b1r[cb], 0.f);
#pragma unroll
        for (int t = 0; t < 16; ++t) M[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(H2[t], G[t], M[cb], 0, 0, 0);
      }
    }
    // M[c2 = cb*32 + kappa(r,h)][j = ql] -> out[g][c2*32 + j]
    float* o = out + (size_t)g * C2 * 32;
#pragma unroll
    for (int cb = 0; cb < C2 / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if PASNL_SA_NT
        __builtin_nontemporal_store(M[cb][r], &o[(size_t)(cb * 32 + kappa(r, h)) * 32 + ql]);
#else
        o[(size_t)(cb * 32 + kappa(r, h)) * 32 + ql] = M[cb][r];
#endif
      }
  }
}


// =============================================================================================
// pasnl_sa_cell: grouping + skip maxima + local cell, NW waves per workgroup sharing one LDS copy of the weights.
//   * The fp32 MFMA executes on the SIMD's vector lanes: VALU instructions do not overlap with it (tools/mfmaprobe.hip:
//     4 VALU ops behind every MFMA -> 88 instead of 64 cycles per MFMA, with one or two waves per SIMD), LDS and
//     global-memory instructions do.  The kernel is therefore written to MINIMISE VALU INSTRUCTIONS: reductions go
//     through LDS atomics, accumulators stay where they are (single-path loops, M pinned in AccVGPRs), operands are
//     refilled in place, group bookkeeping is scalar.
//   * NW = 8 puts TWO waves on every SIMD (<= 256 registers each): while one wave waits for its gathered rows the
//     other one keeps the pipe busy (latency only -- see above).  With the weights filling the LDS there can be only
//     one workgroup per CU, so the second wave has to come from inside the workgroup.
//   * every global load is UNCONDITIONAL with a clamped address (a conditional load compiles to its own
//     exec-masked basic block: serialised loads and vmcnt(0) at every join); values that must not be used are
//     masked where they are consumed.
//   * skip connection: every lane folds its operands into one of 4 replica rows of the wave with ds_max_f32 (no return
//     value: nothing waits for it); the rows are combined and written out once per group.
// =============================================================================================
// max(x, 0) as a signed-integer maximum of the bits (negative floats are negative integers; -0 -> +0): ONE plain instruction
// the compiler can see.  There is NO inline-assembly arithmetic in this file: the compiler's hazard recogniser does not look
// into inline assembly (EXPERIMENTS.md, round 4: an asm v_max_f32 sunk behind the v_mfma that read its target register gave
// results a few per cent off); tools/asm_scan.py fails on any inline-asm vector instruction in a kernel that issues v_mfma.
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// the larger of two floats as a compare + select (fmaxf() would canonicalise both operands first in IEEE mode); a NaN in `b`
// keeps `a`
__device__ __forceinline__ float max_sel(float a, float b) { return b > a ? b : a; }
// Internal column order of the cell input X (a permutation of the reference's [xyz-centre | xyz | feature], chosen
// so that a lane's 16 operands of a chunk are 64 contiguous, 16-byte aligned bytes of a feature row):
//     0..2 xyz - centre    3..5 xyz    6 the constant 1 (row 6 of W0 = b0: the conv0 bias rides on the MFMA)
//     7 zero               8.. feature[0..cf)      then zeros up to a multiple of 32
// VEC (cf % 4 == 0): MFMA step t of chunk ch contracts columns ch*32+t (lanes 0..31) and ch*32+16+t (lanes 32..63),
// so lane (ql, h) needs X[row ql][ch*32 + 16h .. +16): FOUR global_load_dwordx4 per chunk.  The first kernels loaded
// the same data as 16 scattered global_load_dword per chunk (80 per tile): the texture-addresser rate for 64
// scattered dwords per instruction, not the matrix pipe, set their speed (57 % of the fp32 MFMA peak).
// !VEC (cls layer1, cf = 3): step t contracts columns 2t (lanes 0..31) and 2t+1 (lanes 32..63), scalar loads.
// TAIL8: the row ends with exactly one partial chunk of <= 8 live steps (every shape of the three models: 8 + cf is
// 8 mod 32 for cf = 32, 64, 128 and 11 for cf = 3).  Known at compile time, the conv0 code is one straight path:
// [full-chunk loop][8-step tail], and the accumulators are not copied where paths would merge.

constexpr int SA_SKIP_REP = 4;

// SINGLE: the layer has ONE convolution (mlp = [c, c]): conv0 is formed with its operands swapped -- H1 = X . W0 with rows =
// neighbours, the layout the matmul takes as conv1's output -- and there is no conv1 (w1 / b1 are not read)
template <int C1, int C2, int NW, bool VEC, bool TAIL8, bool XYZ3 = false, bool SINGLE = false>  // XYZ3: the feature rows are 3 wide (xyz-only first layers)
__global__ __launch_bounds__(NW * 64) void sa_cell_kernel(long groups, int k, int w, SaGatherSrc src,
                                                         const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ ww, const float* __restrict__ bw,
                                                         float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SA_PROBE(unsigned long long pentry; SA_MARK0(pentry);)
  const int cf = w - 6;
  const int wi = 8 + cf;                            // internal width
  const int wp = (wi + 31) & ~31;
  float* W0s = reinterpret_cast<float*>(smem);      // [wp][C1], rows in internal column order
  float* W1s = W0s + (size_t)wp * C1;               // [C1][C2]
  float* Wws = W1s + C1 * C2;                       // [3 steps][2 halves][32]: weight net, zero rows for the unused half
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: the group bookkeeping below is scalar work
  const int h = lane >> 5, ql = lane & 31;
  // running column maxima of the current group: [NW][SA_SKIP_REP][sks]; row ql of a tile folds into replica ql % 4
  // (8 lanes per address instead of 32), sks = wp + 4 puts the 8 addresses of one ds_max_f32 on 8 different banks
  const int sks = wp + 4;
  float* skp = Wws + 6 * 32 + (size_t)wave * SA_SKIP_REP * sks;
  float* skl = skp + (ql & (SA_SKIP_REP - 1)) * sks + (VEC ? 16 * h : h);  // this lane's replica, its first column

  // ---- weights into LDS, once per (persistent) workgroup.  16-byte loads, SB of them in flight per thread before the
  // first LDS store: the copy is latency-bound (144 KiB per workgroup at C = 128: 57 k cycles = 7 % of the launch with
  // one dependent dword load + store per iteration, conditional on the row kind)
  constexpr int T = NW * 64;
  if (((reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(b0)) & 15) == 0) {
    constexpr int Q = C1 / 4, SB = 8;
    const int n0 = wp * Q;  // float4 items of W0s; row r of W0s = w0 row r (r < 6) | b0 (6) | 0 (7) | w0 row r - 2 | 0 (r >= wi)
    for (int base = tid; base < n0; base += SB * T) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int i = min(base + u * T, n0 - 1), r = i / Q, q = i - r * Q;
        const float* srow = r == 6 ? b0 : w0 + (size_t)(r < 6 ? r : min(max(r, 8), wi - 1) - 2) * C1;
        v[u] = reinterpret_cast<const float4*>(srow)[q];
        if (r == 7 || r >= wi) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (base + u * T < n0) reinterpret_cast<float4*>(W0s)[base + u * T] = v[u];
    }
    constexpr int N1 = C1 * C2 / 4;
    for (int base = tid; base < (SINGLE ? 0 : N1); base += SB * T) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) v[u] = reinterpret_cast<const float4*>(w1)[min(base + u * T, N1 - 1)];
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (base + u * T < N1) reinterpret_cast<float4*>(W1s)[base + u * T] = v[u];
    }
  } else {
    for (int i = tid; i < wp * C1; i += T) {
      const int r = i / C1, c1 = i - r * C1;
      float v = 0.f;
      if (r < 6) v = w0[(size_t)r * C1 + c1];
      else if (r == 6) v = b0[c1];
      else if (r >= 8 && r < wi) v = w0[(size_t)(r - 2) * C1 + c1];
      W0s[i] = v;
    }
    for (int i = tid; i < (SINGLE ? 0 : C1 * C2); i += T) W1s[i] = w1[i];
  }
  for (int i = tid; i < 6 * 32; i += NW * 64) {
    const int t = i / 64, hh = (i >> 5) & 1, j = i & 31;
    // VEC: step t pairs column t with column 16+t -> only the first half carries a coordinate.
    // !VEC: step t pairs columns 2t, 2t+1: (0,1), (2,3): column 3 (xyz.x) is not a weight-net input.
    float v = 0.f;
    if (VEC) v = hh == 0 ? ww[t * 32 + j] : 0.f;
    else v = (2 * t + hh) < 3 ? ww[(2 * t + hh) * 32 + j] : 0.f;
    Wws[i] = v;
  }
  __syncthreads();

  float b1r[C2 / 32];
#pragma unroll
  for (int cb = 0; cb < C2 / 32; ++cb) b1r[cb] = SINGLE ? 0.f : b1[cb * 32 + ql];
  const float bwr = bw[ql];
  const int nchunk = wp / 32;
  const int cf4 = cf >> 2;

  // XCD-aware work distribution: workgroup i runs on XCD i % 8 (round-robin dispatch) and every XCD has its own
  // 4 MiB L2, so XCD x is given whole clouds (x, x+8, ...): the tables it gathers from (8 clouds x <= 270 KiB at
  // cls B = 64) stay in ITS L2 instead of every XCD streaming every cloud's table through (measured: 368 MB of
  // L2-miss fetches per launch for 17 MB of tables with the linear mapping).
  // All of it is wave-uniform and kept in SGPRs (scalar instructions do not occupy the vector/matrix pipe); the
  // (cloud, point) pair advances incrementally, one division per wave instead of one per group.
  const int m = src.m;
  const int nclouds = (int)(groups / m);
  const bool xcd_map = (gridDim.x % 8 == 0) && nclouds >= 16;
  const int xcd = blockIdx.x & 7;
  const int my_groups = xcd_map ? ((nclouds - xcd + 7) >> 3) * m : (int)groups;
  const int first = xcd_map ? (int)(blockIdx.x >> 3) * NW + wave : (int)blockIdx.x * NW + wave;
  const int step = xcd_map ? (int)(gridDim.x >> 3) * NW : (int)gridDim.x * NW;
  const int step_q = step / m, step_r = step - step_q * m;
  int cl = first / m, pj = first - cl * m;  // position in this workgroup's list: cloud slot, point
  SA_PROBE(unsigned long long pk0, pg0, pt0, pt1, pt2, pt3, pc0, pc1;
           unsigned long long a_pro = 0, a_start = 0, a_conv0 = 0, a_cwait = 0, a_conv1 = 0, a_epi = 0, a_tiles = 0;)
  SA_PROBE(SA_MARK0(pk0);)
  // Operands of one chunk for this lane.  Every load is unconditional with a clamped address (a conditional load
  // compiles to its own exec-masked basic block); columns outside the feature row exist only in the LAST chunk of
  // a row that is not a multiple of 32 wide (and in chunk 0's first 8 columns, which are overwritten below), so
  // only that chunk is masked.
  float xr[16];
  float px = 0.f, py = 0.f, pz = 0.f;   // the tile's neighbour coordinates (one row per lane pair)
  float nf0 = 0.f, nf1 = 0.f;           // centre0 outputs: neighbour 0's feature row, in flight during a group's first tile
  float f3x = 0.f, f3y = 0.f, f3z = 0.f;  // XYZ3: the tile row's three features (one 12-byte load instead of clamped dword loads)
  unsigned frow_off = 0;                 // and its feature row: element offset into src.feature (one register, not a 64-bit
  const float* const fbase_ = src.feature;  // pointer per lane; a tensor of < 2^32 floats: sa_cell_entry checks)
#define frow (fbase_ + frow_off)
  // operands [u0, u1) of chunk ch (VEC: whole 16-byte groups); the loops unroll, u0 / u1 are constants at every call
  auto load_part = [&](int ch, int u0, int u1) {
    if constexpr ((PASNL_SA_ABLATE & 2) != 0) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (t >= u0 && t < u1) xr[t] = (float)(ch + t) * px;
    } else if constexpr (VEC) {
      const int g0 = ch * 8 + 4 * h - 2;  // first 16-byte group of this lane's 16 columns (-2 in chunk 0)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * q >= u0 && 4 * q < u1) {
          const float4 t4 = reinterpret_cast<const float4*>(frow)[min(max(g0 + q, 0), cf4 - 1)];
          xr[4 * q] = t4.x; xr[4 * q + 1] = t4.y; xr[4 * q + 2] = t4.z; xr[4 * q + 3] = t4.w;
        }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (t >= u0 && t < u1) xr[t] = frow[min(max(ch * 32 + 2 * t + h - 8, 0), cf - 1)];
    }
  };
  auto mask_chunk = [&](int ch, float (&v)[16]) {
    if constexpr (VEC) {
      const int g0 = ch * 8 + 4 * h - 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = g0 + q >= 0 && g0 + q < cf4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = ok ? v[4 * q + e] : 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int f = ch * 32 + 2 * t + h - 8;
        v[t] = (f >= 0 && f < cf) ? v[t] : 0.f;
      }
    }
  };
  // Request everything a tile starts with -- its rows' coordinates and the first chunk of operands -- for neighbour
  // index i of cloud bc.  Called while the PREVIOUS tile still has its conv1 / matmul phase ahead (the operand
  // registers are dead there), so that a tile starts without waiting for memory.
  auto request_rows = [&](long bc, int i) {
    const float* pp = src.xyz + ((size_t)bc * src.n + i) * 3;
    px = pp[0]; py = pp[1]; pz = pp[2];
    frow_off = (unsigned)(((size_t)bc * src.n + i) * (size_t)cf);
    if constexpr (XYZ3) {
      f3x = frow[0]; f3y = frow[1]; f3z = frow[2];
    } else {
      if (TAIL8 && wi < 32) load_part(0, 0, 8);  // a one-chunk row: only 8 steps exist
      else load_part(0, 0, 16);
    }
  };
  // Software pipeline over tiles: the neighbour indices of tile t+1 are requested when tile t starts, its rows when
  // tile t has finished conv0; the centre of the next group when a group starts.  Only a wave's first tile waits
  // for the two dependent round trips (index, then rows).
  int inext = 0;
  float cxn = 0.f, cyn = 0.f, czn = 0.f;
  if (first < my_groups) {
    const long b0 = xcd_map ? xcd + 8L * cl : (long)cl, g0 = b0 * m + pj;
    request_rows(b0, src.idx[g0 * k + ql]);
    cxn = src.new_xyz[g0 * 3]; cyn = src.new_xyz[g0 * 3 + 1]; czn = src.new_xyz[g0 * 3 + 2];
  }
  for (int li = first; li < my_groups; li += step) {
    SA_MARK(pg0);
    const long bi = xcd_map ? xcd + 8 * cl : cl;
    const long g = bi * m + pj;
    pj += step_r;
    cl += step_q;
    if (pj >= m) { pj -= m; ++cl; }
    // the group after this one (this one again when it is the last: a dummy request)
    const long bi_next = li + step < my_groups ? (xcd_map ? xcd + 8L * cl : (long)cl) : bi;
    const long g_next = li + step < my_groups ? bi_next * m + pj : g;
    float cx = cxn, cy = cyn, cz = czn;
    cxn = src.new_xyz[g_next * 3]; cyn = src.new_xyz[g_next * 3 + 1]; czn = src.new_xyz[g_next * 3 + 2];
    if constexpr (XYZ3) {  // sks = 36: 144 words, known at compile time
      skp[lane] = -INFINITY;
      skp[lane + 64] = -INFINITY;
      if (lane < SA_SKIP_REP * 36 - 128) skp[lane + 128] = -INFINITY;
    } else {
      for (int c = lane; c < SA_SKIP_REP * sks; c += 64) skp[c] = -INFINITY;
    }
    f32x16 M[C2 / 32];
#pragma unroll
    for (int cb = 0; cb < C2 / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) M[cb][r] = 0.f;

    // (XYZ3 is dispatched for k == 32 only: one tile per group, known at compile time -- the accumulators' zeros fold into
    // their first MFMA and nothing has to be pinned)
    const int ktile = XYZ3 ? 32 : k;
    for (int tile = 0; tile < ktile; tile += 32) {
      if constexpr (!XYZ3) {
#pragma unroll
        for (int cb = 0; cb < C2 / 32; ++cb) asm volatile("" : "+a"(M[cb]));  // M lives in AccVGPRs: no VALU ever reads it
      }
      SA_MARK(pt0);
      SA_PROBE(if (tile == 0) a_pro += pt0 - pg0;)
      // (this tile's rows were requested during the previous tile; now the indices of the following tile)
      inext = src.idx[(tile + 32 < ktile ? (size_t)g * k + tile + 32 : (size_t)g_next * k) + ql];
      SA_WAIT_VM();
      SA_MARK(pt1);
      SA_PROBE(a_start += pt1 - pt0;)

      f32x16 H1T[C1 / 32];
#pragma unroll
      for (int ob = 0; ob < C1 / 32; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1T[ob][r] = 0.f;
      // (pinned in AccVGPRs where registers are scarce; the narrow first-layer variant has room, and without the pin the
      // compiler folds the zeros into the first MFMA's accumulator operand instead of writing 16 registers per block)
      if constexpr (!XYZ3) {
#pragma unroll
        for (int ob = 0; ob < C1 / 32; ++ob) asm volatile("" : "+a"(H1T[ob]));
      }
      f32x16 G;
#pragma unroll
      for (int r = 0; r < 16; ++r) G[r] = 0.f;

      if (src.centre0 && tile == 0) {
        // the group's centre IS row 0 of its first tile (pointasnl_util.py:161-163): lane 0 holds it -- three readlanes
        // instead of a (b,m,3) table somebody had to gather first
        cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), 0));
        cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), 0));
        cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz), 0));
        if (src.new_feature_out) {  // neighbour 0's feature row: requested now, stored when the tile's matrix work has issued
          const unsigned long long fp = reinterpret_cast<unsigned long long>(frow);
          const float* f0 = reinterpret_cast<const float*>(
              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(fp >> 32), 0) << 32) |
              (unsigned)__builtin_amdgcn_readlane((int)fp, 0));
          nf0 = f0[min(lane, cf - 1)];
          nf1 = f0[min(lane + 64, cf - 1)];
        }
      }
      const int nfull = wi >> 5;  // chunks whose 32 columns all exist
      if constexpr (XYZ3) {
        // internal columns 8..10 = the three features, 11..15 padding; step t holds columns 2t (h = 0) and 2t + 1 (h = 1)
        xr[4] = h ? f3y : f3x;
        xr[5] = h ? 0.f : f3z;
        xr[6] = 0.f;
        xr[7] = 0.f;
      } else if (nfull == 0) {
        mask_chunk(0, xr);
      }
      // internal columns 0..7 = [xyz - centre | xyz | 1 | 0]; the weight net (3 -> 32) rides on the same operands
      if constexpr (VEC) {
        if (h == 0) {
          xr[0] = px - cx; xr[1] = py - cy; xr[2] = pz - cz; xr[3] = px;
          xr[4] = py; xr[5] = pz; xr[6] = 1.f; xr[7] = 0.f;
        }
      } else {
        xr[0] = h ? py - cy : px - cx;
        xr[1] = h ? px : pz - cz;
        xr[2] = h ? pz : py;
        xr[3] = h ? 0.f : 1.f;
      }
#pragma unroll
      for (int t = 0; t < (VEC ? 3 : 2); ++t)
        G = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[t], Wws[(t * 2 + h) * 32 + ql], G, 0, 0, 0);

      constexpr int RS = VEC ? 1 : 2;  // W0 row / column stride between consecutive MFMA steps
      // Skip connection: the column maxima over the group's rows.  The fp32 MFMA runs on the SIMD's vector lanes, so
      // every VALU instruction of this wave costs matrix time (tools/mfmaprobe.hip: 64 -> 88 cycles per MFMA with 4
      // VALU ops behind each); LDS instructions do not.  Each lane therefore folds its 16 operands straight into its
      // replica row with ds_max_f32 (no return value, nothing waits for it) instead of reducing over the 32 rows
      // with 80 DPP steps per chunk first.  Padding columns receive zeros: unread.
      // NS MFMA steps of chunk ch: the W0 operands of batch j+1 (BT steps x C1/32 blocks) are read from LDS while the
      // MFMAs of batch j run
      // STREAM: the operand registers of a finished group of steps are refilled in place with the same columns of
      // the next chunk (12 steps = 48 MFMAs of matrix time before they are used again): no second buffer, no copies.
      auto chunk_steps = [&](int ch, auto ns_c, auto stream_c) {
        constexpr int NS = decltype(ns_c)::value;
        constexpr bool STREAM = decltype(stream_c)::value;
        // an opaque LDS address: one address register + immediate offsets instead of one address add per ds_max
        // (the row lies past the 64 KiB an immediate offset could reach from the start of the LDS)
        typedef __attribute__((address_space(3))) float lds_float;
        lds_float* srow = (lds_float*)(skl + ch * 32);
        asm volatile("" : "+v"(srow));
#ifdef PASNL_SA_BT
        constexpr int BT = PASNL_SA_BT;
#else
        constexpr int BT = XYZ3 ? 2 : (C1 >= 128 ? 1 : 128 / C1);  // MFMA steps per batch (XYZ3: 6 steps = 3 batches of 2)
#endif
        constexpr int NB = NS / BT;
        const float* wbase = VEC ? W0s + (size_t)(ch * 32 + 16 * h) * C1 + ql : W0s + (size_t)(ch * 32 + h) * C1 + ql;
        float wa[2][BT][C1 / 32];
#pragma unroll
        for (int u = 0; u < BT; ++u)
#pragma unroll
          for (int ob = 0; ob < C1 / 32; ++ob) wa[0][u][ob] = wbase[(size_t)(RS * u) * C1 + ob * 32];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (j + 1 < NB) {
#pragma unroll
            for (int u = 0; u < BT; ++u)
#pragma unroll
              for (int ob = 0; ob < C1 / 32; ++ob)
                wa[(j + 1) & 1][u][ob] = wbase[(size_t)(RS * ((j + 1) * BT + u)) * C1 + ob * 32];
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(PASNL_SA_ABLATE & 1)) {
            // the skip maxima of THIS batch's operands, next to the MFMAs that consume the same registers: folded in one go at
            // the top of the chunk they made the wave wait for the operand refills the previous chunk had only just requested
            // (3.4 k of a 49.7-k-cycle tile: what the "no skip maxima" ablation gains); here they wait for what the MFMAs wait for
#pragma unroll
            for (int u = 0; u < BT; ++u) {
              const int t = j * BT + u;
              // XYZ3: steps 3 (the constant-1 / zero columns) and 6, 7 (padding) hold no column the skip connection reads
              if (XYZ3 && (t == 3 || t >= 6)) continue;
              __hip_atomic_fetch_max(srow + RS * t, xr[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
#pragma unroll
          for (int u = 0; u < BT; ++u)
#pragma unroll
            for (int ob = 0; ob < C1 / 32; ++ob)
              H1T[ob] = SINGLE ? __builtin_amdgcn_mfma_f32_32x32x2f32(xr[j * BT + u], wa[j & 1][u][ob], H1T[ob], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j & 1][u][ob], xr[j * BT + u], H1T[ob], 0, 0, 0);
          if constexpr (STREAM) {
            constexpr int GR = VEC ? 4 : 1;  // operands per load
            // the load groups whose last step the PREVIOUS batch finished: [floor((j-1)*BT / GR), floor(j*BT / GR)) * GR.
            // One batch late on purpose: across the chunk loop's back edge the compiler waits for EVERY outstanding load
            // (vmcnt(0)) before the first use of a refilled register; with the refill of a group issued right behind its last
            // step, that wait sat three instructions after a load had been issued -- a full memory round trip per chunk.
            // The LAST group's refill cannot be issued behind the last batch for the same reason (the next iteration's first
            // wait would follow it immediately): it is deferred to the first batch of the chunk that owns the data -- 12 batches
            // before its first use.  (Chunk 0's operands all come from request_rows.)
            if (j >= 1) load_part(min(ch + 1, nchunk - 1), (j - 1) * BT / GR * GR, j * BT / GR * GR);
            else if (ch > 0) load_part(ch, (NB - 1) * BT / GR * GR, NS);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };

      // ---- full chunks: one code path, so that the accumulators stay where they are across iterations
      for (int ch = 0; ch < nfull; ++ch) {
        chunk_steps(ch, std::integral_constant<int, 16>{}, std::true_type{});  // (the last refill of a tile is a dummy)
        SA_MARK2(pc0);
        SA_WAIT_VM();
        SA_MARK2(pc1);
        SA_PROBE(a_cwait += pc1 - pc0;)
      }
      // ---- the last chunk of a row whose width is not a multiple of 32: 8 or 16 steps (rows of W0 past the width and
      // the masked operands are zero, so steps past the last live column add nothing)
      if constexpr (XYZ3) {
        chunk_steps(0, std::integral_constant<int, 6>{}, std::false_type{});  // steps 6, 7 would multiply padding: skipped
      } else if constexpr (TAIL8) {
        if (nfull > 0) mask_chunk(nfull, xr);
        chunk_steps(nfull, std::integral_constant<int, 8>{}, std::false_type{});
      } else if (nfull < nchunk) {
        const int ch = nfull;
        if (nfull > 0) {
          // (the group the last full chunk left to its successor: see chunk_steps)
          constexpr int BTF = C1 >= 128 ? 1 : 128 / C1, GRF = VEC ? 4 : 1;
          load_part(ch, (16 / BTF - 1) * BTF / GRF * GRF, 16);
          mask_chunk(ch, xr);
        }
        const int rem = wi - ch * 32;
        const int live = VEC ? min(16, rem) : min(16, (rem + 1) >> 1);  // MFMA steps that touch a column < wi
        if (live <= 8) chunk_steps(ch, std::integral_constant<int, 8>{}, std::false_type{});
        else chunk_steps(ch, std::integral_constant<int, 16>{}, std::false_type{});
      }
      // rows of the following tile (same group, or the first tile of the next one): they arrive during conv1
      request_rows(tile + 32 < ktile ? bi : bi_next, inext);
      SA_MARK(pt2);
      SA_PROBE(a_conv0 += pt2 - pt1;)
      // ReLU (the bias came with the MFMA); G: bias + ReLU
#pragma unroll
      for (int ob = 0; ob < C1 / 32; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // the ReLU as a signed-integer maximum of the bits: a plain instruction between two matrix products that the
          // hazard recogniser sees (an inline-assembly v_max_f32 here was sunk right behind the product that reads the SAME
          // register as an operand, with no wait state in between: results off by a few per cent, round 4)
          H1T[ob][r] = relu_bits(H1T[ob][r]);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) G[r] = fmaxf(G[r] + bwr, 0.f);

      if constexpr (SINGLE) {
        // mlp = [c, c]: ONE convolution (the *_2 layers of pointasnl_sem_seg_res.py); its ReLU-ed output -- rows = neighbours,
        // thanks to the swapped conv0 -- IS the matmul's operand.  (As an identity conv1 -- relu(h * 1 + 0) = h bit for bit --
        // the same result cost 43-47 % more matrix work.)
        static_assert(C1 == C2, "the single-convolution form keeps the block structure");
#pragma unroll
        for (int cb = 0; cb < C2 / 32; ++cb)
#pragma unroll
          for (int t = 0; t < 16; ++t) M[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(H1T[cb][t], G[t], M[cb], 0, 0, 0);
      } else
#pragma unroll
      for (int cb = 0; cb < C2 / 32; ++cb) {
        f32x16 H2;
#pragma unroll
        for (int r = 0; r < 16; ++r) H2[r] = 0.f;
        const float* w1p = W1s + (size_t)kappa(0, h) * C2 + cb * 32 + ql;
        {
          float wv[2][16];
#pragma unroll
          for (int t = 0; t < 16; ++t) wv[0][t] = w1p[(size_t)(kappa(t, 0)) * C2];
#pragma unroll
          for (int blk = 0; blk < C1 / 32; ++blk) {
            if (blk + 1 < C1 / 32) {
#pragma unroll
              for (int t = 0; t < 16; ++t) wv[(blk + 1) & 1][t] = w1p[(size_t)((blk + 1) * 32 + kappa(t, 0)) * C2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) H2 = __builtin_amdgcn_mfma_f32_32x32x2f32(H1T[blk][t], wv[blk & 1][t], H2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) H2[r] = fmaxf(H2[r] + b1r[cb], 0.f);
#pragma unroll
        for (int t = 0; t < 16; ++t) M[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(H2[t], G[t], M[cb], 0, 0, 0);
      }
      if (src.centre0 && tile == 0 && src.new_feature_out) {
        float* nfo = src.new_feature_out + (size_t)g * (3 + cf);
        if (lane < 3) {
          const float v = lane == 0 ? cx : (lane == 1 ? cy : cz);
          nfo[lane] = v;
          src.new_xyz_out[(size_t)g * 3 + lane] = v;
        }
        if (lane < cf) nfo[3 + lane] = nf0;
        if (lane + 64 < cf) nfo[3 + lane + 64] = nf1;
      }
      SA_MARK(pt3);
      SA_PROBE(a_conv1 += pt3 - pt2; a_tiles += 1;)
    }
    // M[c2 = cb*32 + kappa(r,h)][j = ql] -> out[g][c2*32 + j]
    float* o = out + (size_t)g * C2 * 32;
    if ((PASNL_SA_ABLATE & 4) && cx == 12345.f) o = nullptr;  // ablation: keep M alive, store (almost) never
    if (!(PASNL_SA_ABLATE & 4) || o == nullptr)
#pragma unroll
    for (int cb = 0; cb < C2 / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if PASNL_SA_NT
        __builtin_nontemporal_store(M[cb][r], &o[(size_t)(cb * 32 + kappa(r, h)) * 32 + ql]);
#else
        o[(size_t)(cb * 32 + kappa(r, h)) * 32 + ql] = M[cb][r];
#endif
      }
    // the wave's LDS operations execute in order; the fences only keep the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // reference column c of the skip maxima = internal column c (c < 6) or c + 2 (features)
    if constexpr (XYZ3) {  // nine columns: one masked step instead of a general loop
      if (lane < 9) {
        const float* sc = skp + (lane < 6 ? lane : lane + 2);
        src.skip_max[(size_t)g * 9 + lane] = fmaxf(fmaxf(sc[0], sc[sks]), fmaxf(sc[2 * sks], sc[3 * sks]));
      }
    } else {
      for (int c = lane; c < w; c += 64) {
        const float* sc = skp + (c < 6 ? c : c + 2);
        src.skip_max[(size_t)g * w + c] = fmaxf(fmaxf(sc[0], sc[sks]), fmaxf(sc[2 * sks], sc[3 * sks]));
      }
    }
    __builtin_amdgcn_wave_barrier();
    SA_MARK(pt0);
    SA_PROBE(a_epi += pt0 - pt3;)
  }
#ifdef PASNL_SA_CELL_PROBE
  SA_MARK0(pt0);
  if (lane == 0) {
    atomicAdd(&sa_probe[0], a_pro); atomicAdd(&sa_probe[1], a_start); atomicAdd(&sa_probe[2], a_conv0);
    atomicAdd(&sa_probe[3], a_cwait); atomicAdd(&sa_probe[4], a_conv1); atomicAdd(&sa_probe[5], a_epi);
    atomicAdd(&sa_probe[6], a_tiles); atomicAdd(&sa_probe[7], pt0 - pk0); atomicAdd(&sa_probe[8], 1ull);
    atomicAdd(&sa_probe[9], pk0 - pentry);
  }
#endif
}

#undef frow
#ifdef PASNL_SA_CELL_PROBE
}  // namespace pasnl
// [prologue, start wait, conv0 (incl. chunk wait), chunk wait, conv1 + matmul, epilogue, tiles, wave total, waves,
// weight staging] cycles
// [staging, S = K.Q^T, softmax, O += V^T.P^T, blocks, wave loop total, waves] cycles
extern "C" int pasnl_nl_probe_read(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(pasnl::nl_probe), sizeof(pasnl::nl_probe)) != hipSuccess) return -1;
  unsigned long long zero[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pasnl::nl_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
extern "C" int pasnl_sa_cell_probe_read(unsigned long long* host16) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(pasnl::sa_probe), sizeof(pasnl::sa_probe)) != hipSuccess) return -1;
  unsigned long long zero[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pasnl::sa_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
namespace pasnl {
#endif


// =============================================================================================
// AdaptiveSampling input (pointasnl_util.py:121-124,165-166): for every group the first `as` neighbours as rows
//     [xyz[i_s] - xyz[i_0] | xyz[i_s] | feature[i_s]]   (as x (6+C)),   i_s = idx[b,j,s]
// = concat(normalized_xyz, shift_group_points) of the reference, gathered straight from the tables (one launch
// instead of two gathers, a slice, a subtraction and two concats).  Columns 3.. are also the (xyz | feature) rows
// the re-weighting tail needs, so nothing else of the grouped tensors is ever materialised.
// =============================================================================================
// thread per element: narrow rows (the xyz-only layers, 6 + c = 9 columns)
__global__ __launch_bounds__(256) void as_gather_elem_kernel(int n, int c, int m, int k, int as, long total,
                                                            const float* __restrict__ xyz, const float* __restrict__ feature,
                                                            const int* __restrict__ idx, float* __restrict__ out) {
  const int w = 6 + c;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / w;  // (b, j, s)
    const int col = (int)(e - row * w);
    const long gj = row / as;  // (b, j)
    const int sidx = (int)(row - gj * as);
    const long bi = gj / m;
    const int i = idx[gj * k + sidx];
    float v;
    if (col < 3) {
      const int i0 = idx[gj * k];
      v = xyz[((size_t)bi * n + i) * 3 + col] - xyz[((size_t)bi * n + i0) * 3 + col];
    } else if (col < 6) {
      v = xyz[((size_t)bi * n + i) * 3 + (col - 3)];
    } else {
      v = feature[((size_t)bi * n + i) * c + (col - 6)];
    }
    out[e] = v;
  }
}

// One wave per output row (b, j, s): the row's (cloud, neighbour index, first-neighbour index) are wave-uniform, the
// lanes run over its 6 + c columns (the thread-per-element version spent ~150 instructions per float on three 64-bit
// divisions: 44 us for 52 MB at cls layer2).
__global__ __launch_bounds__(256) void as_gather_kernel(int n, int c, int m, int k, int as, long rows,
                                                       const float* __restrict__ xyz, const float* __restrict__ feature,
                                                       const int* __restrict__ idx, float* __restrict__ out) {
  const int w = 6 + c;
  const int lane = threadIdx.x & 63;
  const long nwaves = (long)gridDim.x * 4;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const long gj = row / as;  // (b, j)
    const int sidx = (int)(row - gj * as);
    const long bi = gj / m;
    const int i = idx[gj * k + sidx], i0 = idx[gj * k];
    const float* ps = xyz + ((size_t)bi * n + i) * 3;
    const float* p0 = xyz + ((size_t)bi * n + i0) * 3;
    const float* fs = feature + ((size_t)bi * n + i) * c;
    float* o = out + (size_t)row * w;
    if (lane < 6) o[lane] = lane < 3 ? ps[lane] - p0[lane] : ps[lane - 3];
    for (int col = lane; col < c; col += 64) o[6 + col] = fs[col];
  }
}

// =============================================================================================
// Decoder local cell (PointASNLDecodingLayer, pointasnl_util.py:323-331): for every point p of the dense level,
//     F = [xyz[i_s] | feature[i_s]]            (k x (3+c))   i_s = idx[p, s], the point's k nearest neighbours
//     G = relu((xyz[i_s] - xyz[p]) Ww + bw)     (k x 32)      weight net on the centred coordinates
//     out[p] = F^T G                            ((3+c) x 32)  -> the input of `decode_after_conv`
// The reference materialises both gathers (1.1 GB at ScanNet fa_layer4), a transpose and a batched matmul of
// 131072 tiny matrices (4.4 ms in the vendor BLAS on MI355X).  Here one wave owns one point: G lives in 8 (k=16)
// registers per lane, F^T is read straight from the L2-resident feature table as the A operand of
// v_mfma_f32_32x32x2_f32 (lane = channel: 128-byte coalesced row segments), and each 32-channel output tile goes out as
// 128-byte rows.  Bound: HBM writes (16.8 KB per point; 2.2 GB per launch at fa_layer4).
// =============================================================================================
template <int K>
__global__ __launch_bounds__(256) void decode_cell_kernel(long points, int n, int c, const float* __restrict__ xyz,
                                                         const float* __restrict__ feature, const int* __restrict__ idx,
                                                         const float* __restrict__ ww, const float* __restrict__ bw,
                                                         float* __restrict__ out) {
  constexpr int T = K / 2;  // MFMA steps: step t contracts neighbours 2t (lanes 0..31) and 2t+1 (lanes 32..63)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, ql = lane & 31;
  const int w = 3 + c, ntile = (w + 31) >> 5;
  const float w0 = ww[ql], w1 = ww[32 + ql], w2 = ww[64 + ql], bj = bw[ql];

  // XCD-aware: XCD x owns clouds x, x+8, ... (their feature tables stay in its L2)
  const int nclouds = (int)(points / n);
  const bool xcd_map = (gridDim.x % 8 == 0) && nclouds >= 8;
  const int xcd = blockIdx.x & 7;
  const long mine = xcd_map ? (long)((nclouds - xcd + 7) >> 3) * n : points;
  const long first = xcd_map ? (long)(blockIdx.x >> 3) * 4 + wave : (long)blockIdx.x * 4 + wave;
  const long step = xcd_map ? (long)(gridDim.x >> 3) * 4 : (long)gridDim.x * 4;
  for (long li = first; li < mine; li += step) {
    long p = li, bi;
    if (xcd_map) {
      const int cl = (int)(li / n);
      bi = xcd + 8 * cl;
      p = bi * n + (li - (long)cl * n);
    } else {
      bi = p / n;
    }
    const float cx = xyz[p * 3], cy = xyz[p * 3 + 1], cz = xyz[p * 3 + 2];
    float G[T], qc[T];  // weight-net value of neighbour (2t+h) for column j = ql; coordinate ql (< 3) of that neighbour
    const float* frow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int is = idx[p * K + 2 * t + h];
      const float* q = xyz + ((size_t)bi * n + is) * 3;
      const float qx = q[0], qy = q[1], qz = q[2];
      // conv accumulates in input-channel order, then the folded bias, then ReLU
      G[t] = fmaxf(__builtin_fmaf(qz - cz, w2, __builtin_fmaf(qy - cy, w1, (qx - cx) * w0)) + bj, 0.f);
      qc[t] = ql == 0 ? qx : (ql == 1 ? qy : qz);
      frow[t] = feature + ((size_t)bi * n + is) * (size_t)c - 3;  // F column ch >= 3 is frow[ch]
    }
    float* o = out + (size_t)p * w * 32;
    float a[T], an[T];
#pragma unroll
    for (int t = 0; t < T; ++t) a[t] = frow[t][min(max(ql, 3), w - 1)];
    for (int ct = 0; ct < ntile; ++ct) {
      // next tile's operands in flight during this tile's MFMAs and stores (clamped addresses, masked at use)
      const int chn = min(ct + 1, ntile - 1) * 32 + ql;
#pragma unroll
      for (int t = 0; t < T; ++t) an[t] = frow[t][min(max(chn, 3), w - 1)];
      const int ch = ct * 32 + ql;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float av = ch < 3 ? qc[t] : (ch < w ? a[t] : 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, G[t], acc, 0, 0, 0);
      }
      // acc[r] = out[channel ct*32 + kappa(r,h)][j = ql]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = ct * 32 + kappa(r, h);
        if (oc < w) o[(size_t)oc * 32 + ql] = acc[r];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) a[t] = an[t];
    }
  }
}

}  // namespace pasnl

using namespace pasnl;

template <int CB, int SPLIT>
static int nl_mfma_launch(int b, int p, int n, float qscale, const float* q, const float* kv, float* out, bool staged,
                          hipStream_t st) {
  if constexpr (CB <= 64) {
    if (!staged) {  // (the LDS-staged kernel stays selectable -- variant 3 -- for A/B measurements)
      const size_t lds = (size_t)SPLIT * (CB / 2 + 2) * 64 * sizeof(float);
      hipLaunchKernelGGL((nl_attention_direct_kernel<CB, SPLIT>), dim3((p + 31) / 32, b), dim3(SPLIT * 64), lds, st, p, n, qscale,
                         q, kv, out);
      return pasnl_launch_status();
    }
  }
  constexpr int WAVE_FLOATS = NL_KB * (CB + 1) + NL_KB * CB + 3;
  size_t lds = (size_t)SPLIT * ((WAVE_FLOATS + 3) & ~3) * 4 + 16;
  auto kern = nl_attention_mfma_kernel<CB, SPLIT>;
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((p + 31) / 32, b), dim3(SPLIT * 64), lds, st, p, n, qscale, q, kv, out);
  return pasnl_launch_status();
}

template <int SPLIT>
static int nl_pair_launch(int b, int p, int n, float qscale, const float* q, const float* kv, float* out, hipStream_t st) {
  const size_t lds = (size_t)SPLIT * 2 * (32 / 2 + 2) * 64 * sizeof(float);
  auto kern = nl_attention_pair_kernel<SPLIT>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((p + 63) / 64, b), dim3(SPLIT * 64), lds, st, p, n, qscale, q, kv, out, 0, static_cast<float*>(nullptr));
  return pasnl_launch_status();
}

// Keys over workgroups (nl_attention_pair_kernel<., true>): for cb = 32 shapes whose b * ceil(p / 64) workgroups leave CUs empty
// and whose key loops are long (n >= 4096).  -> kparts (1: not used) and the waves per workgroup.  Measured on the shapes of the
// models (tools/nl_parts_sweep.py): four waves per workgroup; the number of parts that deals the workgroups evenly to the 256
// CUs -- the smallest k that minimises ceil(pairs k / 256) / k -- with >= 4 key blocks per wave: [8,1280,10240] 198 -> 128 us
// at k = 8 (1280 workgroups = 5 per CU), [4,1024,8192] 90 -> 49 us at k = 4; key sets of 40 blocks ([8,320,1280], 18 us) gain
// nothing and keep the plain form.
struct NlParts { int kparts, split; };
static NlParts nl_parts(int b, int p, int n, int cb) {
  NlParts r{1, 1};
  if (cb != 32 || n % NL_KB != 0) return r;
  const long pairs = (long)b * ((p + 63) / 64), blocks = n / NL_KB;
  if (pairs >= 224 || pairs <= 0 || blocks < 128) return r;  // (the plain form has a workgroup for ~every CU / short key loops)
  const int split = 4;
  double best = 1.0;  // the plain form: one round of whole pairs
  for (int k = 2; k <= 32 && (long)k * split * 4 <= blocks; ++k) {
    const double t = (double)((pairs * k + 255) / 256) / (double)k;
    if (t < best - 1e-9) { best = t; r.kparts = k; r.split = split; }
  }
  if (const char* e = tune_env("PASNL_NL_PARTS")) {  // tuning build only: "kparts,split"
    int k = 1, sp = 8;
    if (sscanf(e, "%d,%d", &k, &sp) == 2 && k >= 1 && (long)k * sp <= blocks) { r.kparts = k; r.split = sp; }
  }
  return r;
}
static size_t nl_parts_bytes(int b, int p, int kparts) {
  return (size_t)b * ((p + 63) / 64) * kparts * 2 * (32 / 2 + 2) * 64 * sizeof(float);
}

template <int SPLIT>
static int nl_pair_parts_launch(int b, int p, int n, float qscale, const float* q, const float* kv, float* out, int kparts, float* part,
                                hipStream_t st) {
  const size_t lds = (size_t)SPLIT * 2 * (32 / 2 + 2) * 64 * sizeof(float);
  auto kern = nl_attention_pair_kernel<SPLIT, true>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  const int blocks = n / NL_KB, bpz = (blocks + kparts - 1) / kparts;
  hipLaunchKernelGGL(kern, dim3((p + 63) / 64, b, kparts), dim3(SPLIT * 64), lds, st, p, n, qscale, q, kv, out, bpz, part);
  hipLaunchKernelGGL(nl_attention_merge_kernel, dim3((p + 63) / 64, b), dim3(64), 0, st, p, kparts, part, out);
  return pasnl_launch_status();
}

template <int CB>
static int nl_mfma_dispatch(int b, int p, int n, float qscale, const float* q, const float* kv, float* out, bool staged,
                            hipStream_t st) {
  if constexpr (CB == 32) {
    // two query tiles per wave when there are pairs for at least half of the CUs (cls layer 1: 512, ScanNet layer 1: 256,
    // KITTI layer 1_1: 160 -- 222 -> 200 us although 96 CUs stay empty) and no ragged key block
    const long pairs = (long)b * ((p + 63) / 64);
    const long blocks = (n + NL_KB - 1) / NL_KB;
    const char* one = tune_env("PASNL_NL_PAIR");  // tuning only: "0" = never
    const char* pm = tune_env("PASNL_NL_PAIR_MIN");  // tuning only
    const long pair_min = pm && *pm ? atol(pm) : 128;
    if (!staged && pairs >= pair_min && n % NL_KB == 0 && !(one && *one == '0')) {
      int want = 1;
      while (want < 8 && pairs * want < 2048 && want * 2 * 4 <= blocks) want *= 2;
      const char* force = tune_env("PASNL_NL_SPLIT");
      if (force && *force) want = atoi(force);
      if (want >= 8) return nl_pair_launch<8>(b, p, n, qscale, q, kv, out, st);
      if (want >= 4) return nl_pair_launch<4>(b, p, n, qscale, q, kv, out, st);
      if (want >= 2) return nl_pair_launch<2>(b, p, n, qscale, q, kv, out, st);
      return nl_pair_launch<1>(b, p, n, qscale, q, kv, out, st);
    }
  }
  // split the keys over enough waves to put ~4 on every SIMD (4096 waves; measured best on all reference shapes:
  // cls layer1 140 -> 82 us, cls layer2 128 -> 37 us, ScanNet layer1 1030 -> 279 us with the LDS-staged kernel), but
  // keep >= 4 key blocks per wave (cls layer2, 16 blocks: 19 us at 4 waves, 23 us at 8: the merge is not free) and
  // stay within the LDS (cb = 128: 4 wave regions fit)
  long tiles = (long)b * ((p + 31) / 32);
  long blocks = (n + NL_KB - 1) / NL_KB;
  int want = 1;
  while (want < 8 && tiles * want < 4096 && want * 2 * 4 <= blocks) want *= 2;
  const char* force = tune_env("PASNL_NL_SPLIT");  // tuning only
  if (force && *force) want = atoi(force);
  if (CB == 128 && want > 4) want = 4;
  if constexpr (CB < 128) {  // (cb = 128: 4 wave regions fill the LDS; the 8-wave form would also spill)
    if (want >= 8) return nl_mfma_launch<CB, 8>(b, p, n, qscale, q, kv, out, staged, st);
  }
  if (want >= 4) return nl_mfma_launch<CB, 4>(b, p, n, qscale, q, kv, out, staged, st);
  if (want >= 2) return nl_mfma_launch<CB, 2>(b, p, n, qscale, q, kv, out, staged, st);
  return nl_mfma_launch<CB, 1>(b, p, n, qscale, q, kv, out, staged, st);
}

template <int CB>
static int nl_valu_launch(int b, int p, int n, float qscale, const float* q, const float* kv, float* out, hipStream_t st) {
  hipLaunchKernelGGL((nl_attention_valu_kernel<CB, 64>), dim3((p + 63) / 64, b), dim3(64), 0, st, p, n, qscale, q, kv, out);
  return pasnl_launch_status();
}

static int nl_attention_entry(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                              void* workspace, size_t workspace_bytes, pasnl_stream_t stream);
extern "C" int pasnl_nl_attention(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                                  pasnl_stream_t stream) {
  return nl_attention_entry(b, p, n, cb, q, kv, out, variant, nullptr, 0, stream);
}
extern "C" size_t pasnl_nl_attention_workspace_bytes(int b, int p, int n, int cb) {
  if (b <= 0 || p <= 0 || n <= 0) return 0;
  const NlParts np = nl_parts(b, p, n, cb);
  return np.kparts > 1 ? nl_parts_bytes(b, p, np.kparts) : 0;
}
extern "C" int pasnl_nl_attention_ws(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                                     void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  return nl_attention_entry(b, p, n, cb, q, kv, out, variant, workspace, workspace_bytes, stream);
}
static int nl_attention_entry(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                              void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && p >= 0 && n > 0 && cb > 0, PASNL_EINVAL);
  PASNL_REQUIRE(variant >= 0 && variant <= 3, PASNL_EINVAL);
  PASNL_REQUIRE(cb == 32 || cb == 64 || cb == 128, PASNL_EUNSUPPORTED);
  if (b == 0 || p == 0) return PASNL_OK;
  PASNL_REQUIRE(q && kv && out, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(((reinterpret_cast<uintptr_t>(kv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, PASNL_EUNSUPPORTED);
  hipStream_t st = pasnl_hip_stream(stream);
  // scores are kept in the log2 domain: exp(x/sqrt(cb) - m) == exp2((x*log2e/sqrt(cb)) - m')
  const float qscale = LOG2E / sqrtf((float)cb);
  if (variant == 1) {
    if (cb == 32) return nl_valu_launch<32>(b, p, n, qscale, q, kv, out, st);
    if (cb == 64) return nl_valu_launch<64>(b, p, n, qscale, q, kv, out, st);
    return PASNL_EUNSUPPORTED;  // cb=128 does not fit the one-query-per-lane register budget
  }
  if (cb == 32 && variant == 0 && workspace != nullptr) {  // keys over workgroups where the plain form leaves CUs empty
    const NlParts np = nl_parts(b, p, n, cb);
    if (np.kparts > 1) {
      PASNL_REQUIRE(workspace_bytes >= nl_parts_bytes(b, p, np.kparts), PASNL_EWORKSPACE);
      PASNL_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, PASNL_EUNSUPPORTED);
      float* part = static_cast<float*>(workspace);
      if (np.split >= 8) return nl_pair_parts_launch<8>(b, p, n, qscale, q, kv, out, np.kparts, part, st);
      if (np.split >= 4) return nl_pair_parts_launch<4>(b, p, n, qscale, q, kv, out, np.kparts, part, st);
      if (np.split >= 2) return nl_pair_parts_launch<2>(b, p, n, qscale, q, kv, out, np.kparts, part, st);
      return nl_pair_parts_launch<1>(b, p, n, qscale, q, kv, out, np.kparts, part, st);
    }
  }
  if (cb == 32) return nl_mfma_dispatch<32>(b, p, n, qscale, q, kv, out, variant == 3, st);
  if (cb == 64) return nl_mfma_dispatch<64>(b, p, n, qscale, q, kv, out, variant == 3, st);
  return nl_mfma_dispatch<128>(b, p, n, qscale, q, kv, out, variant == 3, st);
}

namespace pasnl {
// ---------------------------------------------------------------------------------------------
// AdaptiveSampling micro attention with the K / V / Q projections computed on the fly (narrow inputs: w <= 15, the
// xyz-only first layers, w = 9).  The projection output would be (groups*as, 3cb) floats -- 151 MB at cls layer1,
// written by a GEMM that is pure HBM traffic at K = 9 and read once by the attention.  Here everything is a chain of
// v_mfma_f32_16x16x4_f32 whose D layouts are already the next product's operand layouts (lane = (col, grp)):
//     K^T = Wk'^T . X'^T  and  Q^T = Wq'^T . X'^T   -> lane (row = col, grp) holds channels 4*grp + r of a 16-block
//     V   = X' . Wv'                                -> lane (channel = col, grp) holds keys 4*grp + r
//     S   = sum over (block, r) of K^T[r] (A) x Q^T[r] (B): the pairing of channels inside a step is free as long
//           as K and Q agree on it               -> lane (query = col, grp) holds keys 4*grp + r
//     O^T = V[t] (A) x P[t] (B)
// X' = [x | 1 | 0..] (the constant column carries the biases), W' = the BN-folded weights with the bias as row w;
// the log2e/sqrt(cb) scale is folded into Wq'.  A lane's weights (<= 48 values) live in registers for the whole kernel.
//   one wave per group, persistent over groups
// ---------------------------------------------------------------------------------------------
constexpr int AS_PROJ_MAXW = 15;
template <int KS, int CBLK>  // k-steps of 4 inputs (w + 1 <= 4 KS), 16-channel blocks (cb = 16 CBLK)
__global__ __launch_bounds__(256) void as_attention_proj_kernel(long groups, int as, int w, float qscale,
                                                               const float* __restrict__ x, const float* __restrict__ wkvq,
                                                               const float* __restrict__ bkvq, float* __restrict__ out) {
  constexpr int CB = 16 * CBLK;
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, grp = lane >> 4;
  // W'[k][c] for this lane's k = 4s + grp, c = cb*16 + col; columns of wkvq: [K | V | Q]
  float wk[KS][CBLK], wv[KS][CBLK], wq[KS][CBLK];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + grp;
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      const int c = cb * 16 + col;
      const float* wrow = wkvq + (size_t)min(k, w - 1) * 3 * CB;
      const float k_ = k < w ? wrow[c] : (k == w ? bkvq[c] : 0.f);
      const float v_ = k < w ? wrow[CB + c] : (k == w ? bkvq[CB + c] : 0.f);
      const float q_ = k < w ? wrow[2 * CB + c] : (k == w ? bkvq[2 * CB + c] : 0.f);
      wk[s][cb] = k_; wv[s][cb] = v_; wq[s][cb] = q_ * qscale;
    }
  }
  const long nwaves = (long)gridDim.x * 4;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += nwaves) {
    // X'[row = col][k = 4s + grp]; rows past `as` are zero rows (their keys are masked, their queries not stored)
    float xv[KS];
    const float* xp = x + ((size_t)g * as + min(col, as - 1)) * w;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + grp;
      const float v = xp[min(k, w - 1)];
      xv[s] = col < as ? (k < w ? v : (k == w ? 1.f : 0.f)) : 0.f;
    }
    f32x4 Kt[CBLK], Qt[CBLK], V[CBLK];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      Kt[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; Qt[cb] = Kt[cb]; V[cb] = Kt[cb];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        Kt[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s][cb], xv[s], Kt[cb], 0, 0, 0);
        Qt[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s][cb], xv[s], Qt[cb], 0, 0, 0);
        V[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s][cb], V[cb], 0, 0, 0);
      }
    }
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) S = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[cb][r], Qt[cb][r], S, 0, 0, 0);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = (4 * grp + r) < as ? S[r] : -INFINITY;
      tmax = fmaxf(tmax, S[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = fast_exp2(S[r] - tmax);
      psum += S[r];
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    const float inv = 1.0f / psum;
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) O = __builtin_amdgcn_mfma_f32_16x16x4f32(V[cb][t], S[t], O, 0, 0, 0);
      if (col < as)  // O^T[ch = cb*16 + 4*grp + r][query = col]
        *reinterpret_cast<float4*>(out + ((size_t)g * as + col) * CB + cb * 16 + 4 * grp) =
            make_float4(O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv);
    }
  }
}
}  // namespace pasnl

namespace pasnl {
// all-reduce over the 16 lanes of a DPP row (quad xor 1, xor 2, then the two mirrors); every lane ends with the result
__device__ __forceinline__ float row16_max(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return v;
}

// ---------------------------------------------------------------------------------------------
// The whole AdaptiveSampling cell of a narrow layer (w = 6 + c <= 15 inputs, 1 + ch <= 16 weights per neighbour) in ONE
// kernel after the gather (pointasnl_util.py:112-173): projections + micro attention as in as_attention_proj_kernel, then
//     hid^T    = relu(Wa'^T . att^T + ba)   (cb -> 32; the attention output O^T is already the B operand, the bias
//                                            is the accumulator's initial value)
//     logit^T  = Wb'^T . hid^T + bb         (32 -> 1 + ch)   -> lane (neighbour s = col, grp) holds outputs o = 4*grp + r
//     softmax over the neighbours = over the 16 lanes of a DPP row; weighted sums of the gathered rows the same way.
// Nothing but the gathered rows is read and only new_xyz (g,3) / new_feature (g,ch) are written: the (groups*as, cb),
// (.., 32) and (.., 1+ch) intermediates of the op-by-op chain (50 + 50 + 11 MB at cls layer1) never exist.
// ---------------------------------------------------------------------------------------------
template <int KS, int CBLK>
__global__ __launch_bounds__(256) void as_cell_narrow_kernel(long groups, int as, int w, int ch, float qscale,
                                                            const float* __restrict__ x, const float* __restrict__ wkvq,
                                                            const float* __restrict__ bkvq, const float* __restrict__ wa,
                                                            const float* __restrict__ ba, const float* __restrict__ wb,
                                                            const float* __restrict__ bb, float* __restrict__ new_xyz,
                                                            float* __restrict__ new_feature) {
  constexpr int CB = 16 * CBLK;
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, grp = lane >> 4;
  const int nout = 1 + ch;
  float wk[KS][CBLK], wv[KS][CBLK], wq[KS][CBLK];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + grp;
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      const int c = cb * 16 + col;
      const float* wrow = wkvq + (size_t)min(k, w - 1) * 3 * CB;
      wk[s][cb] = k < w ? wrow[c] : (k == w ? bkvq[c] : 0.f);
      wv[s][cb] = k < w ? wrow[CB + c] : (k == w ? bkvq[CB + c] : 0.f);
      wq[s][cb] = (k < w ? wrow[2 * CB + c] : (k == w ? bkvq[2 * CB + c] : 0.f)) * qscale;
    }
  }
  // mlp2_0: A[m = h][k-slot grp] of step (cb, r) = Wa[cb*16 + 4*grp + r][hb*16 + col]; its bias as accumulator start
  float wa_r[CBLK][4][2];
  f32x4 ba_r[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ba_r[hb][r] = ba[hb * 16 + 4 * grp + r];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) wa_r[cb][r][hb] = wa[(size_t)(cb * 16 + 4 * grp + r) * 32 + hb * 16 + col];
  }
  // mlp2_1: A[m = o][k-slot grp] of step (hb, r) = Wb[hb*16 + 4*grp + r][o = col]
  float wb_r[2][4];
  f32x4 bb_r;
#pragma unroll
  for (int r = 0; r < 4; ++r) bb_r[r] = 4 * grp + r < nout ? bb[4 * grp + r] : 0.f;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int r = 0; r < 4; ++r) wb_r[hb][r] = col < nout ? wb[(size_t)(hb * 16 + 4 * grp + r) * nout + col] : 0.f;

  const long nwaves = (long)gridDim.x * 4;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += nwaves) {
    float xv[KS];
    const float* xp = x + ((size_t)g * as + min(col, as - 1)) * w;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + grp;
      const float v = xp[min(k, w - 1)];
      xv[s] = col < as ? (k < w ? v : (k == w ? 1.f : 0.f)) : 0.f;
    }
    // the columns this lane re-weights: output o = 4*grp + r multiplies x column 3 + (o - 1) (o >= 1), output 0 the xyz
    float xo[4], xc[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) xo[r] = xp[min(max(2 + 4 * grp + r, 3), w - 1)];
#pragma unroll
    for (int d = 0; d < 3; ++d) xc[d] = xp[3 + d];

    f32x4 Kt[CBLK], Qt[CBLK], V[CBLK];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      Kt[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; Qt[cb] = Kt[cb]; V[cb] = Kt[cb];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        Kt[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s][cb], xv[s], Kt[cb], 0, 0, 0);
        Qt[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s][cb], xv[s], Qt[cb], 0, 0, 0);
        V[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s][cb], V[cb], 0, 0, 0);
      }
    }
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) S = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[cb][r], Qt[cb][r], S, 0, 0, 0);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = (4 * grp + r) < as ? S[r] : -INFINITY;
      tmax = fmaxf(tmax, S[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = fast_exp2(S[r] - tmax);
      psum += S[r];
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    const float inv = 1.0f / psum;
    // att^T blocks (O^T / l) feed mlp2_0 directly
    f32x4 H[2] = {ba_r[0], ba_r[1]};
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
      f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) O = __builtin_amdgcn_mfma_f32_16x16x4f32(V[cb][t], S[t], O, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = O[r] * inv;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) H[hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_r[cb][r][hb], a, H[hb], 0, 0, 0);
      }
    }
    f32x4 L = bb_r;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int r = 0; r < 4; ++r) L = __builtin_amdgcn_mfma_f32_16x16x4f32(wb_r[hb][r], fmaxf(H[hb][r], 0.f), L, 0, 0, 0);
    // L[r] = logit of neighbour `col` for output o = 4*grp + r; softmax over the neighbours = over the row's lanes
    float wgt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = col < as ? L[r] : -INFINITY;
      const float mx = row16_max(v);
      const float e = col < as ? fast_exp2((v - mx) * LOG2E) : 0.f;
      wgt[r] = e / row16_sum(e);
    }
    // weighted sums over the neighbours; lane col == 0 of every row writes its outputs
    float sx[3], sf[4];
#pragma unroll
    for (int d = 0; d < 3; ++d) sx[d] = row16_sum(col < as ? wgt[0] * xc[d] : 0.f);  // (only grp 0 holds output 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) sf[r] = row16_sum(col < as ? wgt[r] * xo[r] : 0.f);
    if (col == 0) {
      if (grp == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) new_xyz[g * 3 + d] = sx[d];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 4 * grp + r;
        if (o >= 1 && o < nout) new_feature[(size_t)g * ch + (o - 1)] = sf[r];
      }
    }
  }
}
}  // namespace pasnl

namespace pasnl {
// ---------------------------------------------------------------------------------------------
// The AdaptiveSampling cell of a WIDE layer after its projection GEMM (w = 6 + c > 15: K = w is a GEMM worth running):
// kvq (g, as, 3cb) = [K | V | Q] rows -> attention -> mlp2 -> softmax over the neighbours -> re-weighted sums, as in
// as_cell_narrow_kernel, with the projected operands read from memory in the layouts that kernel computes them in
// (K^T / Q^T: 4 consecutive channels of the lane's row; V: coalesced row reads).  Wa, Wb (32 x (1+ch)) and the second bias
// sit in LDS.  Differences from the narrow cell, all about what a wave waits for:
//   * every load is unconditional and masked by a product (a select in front of a load makes the compiler predicate the
//     load: branch + load + full wait per matrix step -- DESIGN.md 6 "loads behind a select");
//   * the logits are formed TRANSPOSED (a lane owns an output and four neighbours): the softmax over the neighbours is
//     in-lane work plus two exchanges between the 16-lane rows per reduction, one division per output, and the 16 outputs of
//     a block leave as one 64-byte store;
//   * the features a block re-weights are requested one block ahead; 3 waves per SIMD (__launch_bounds__(256, 3)).
// ---------------------------------------------------------------------------------------------
template <int CBLK>  // 16 (CBLK - 1) < cb <= 16 CBLK (the reference's bottleneck widths are (3 + c) / 2: 33, 65, ...): only
                     // the LAST block of 16 channels needs clamped addresses and masks, the others load at constant offsets
__global__ __launch_bounds__(256, 3) void as_cell_wide_kernel(long groups, int as, int cb, int w, int ch, float qscale,
                                                          const float* __restrict__ kvq, int ld, const float* __restrict__ x,
                                                          const float* __restrict__ wa, const float* __restrict__ ba,
                                                          const float* __restrict__ wb, const float* __restrict__ bb,
                                                          float* __restrict__ new_xyz, float* __restrict__ new_feature) {
  const int cb3 = ld;  // row stride of kvq: 3 cb, or more when the projection GEMM was given a rounder width
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wbs = reinterpret_cast<float*>(smem);  // [32][nout]
  const int nout = 1 + ch;
  float* Bbs = Wbs + 32 * nout;                 // [nout] (+ 16 zeros: the last block of outputs reads past nout)
  float* Was = Bbs + nout + 16;                 // [16 CBLK][WAS_LD]: Wa, rows past cb zero (the padded channels contribute
  constexpr int WAS_LD = 36;                    //  nothing); stride 36: the four 16-lane rows of a read hit all 32 banks
  for (int i = threadIdx.x; i < 32 * nout; i += 256) Wbs[i] = wb[i];
  for (int i = threadIdx.x; i < nout + 16; i += 256) Bbs[i] = i < nout ? bb[i] : 0.f;
  for (int i = threadIdx.x; i < 16 * CBLK * 32; i += 256) Was[(i >> 5) * WAS_LD + (i & 31)] = (i >> 5) < cb ? wa[i] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, grp = lane >> 4;
  f32x4 ba_r[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int r = 0; r < 4; ++r) ba_r[hb][r] = ba[hb * 16 + 4 * grp + r];
  const int noblk = (nout + 15) >> 4;
  const long nwaves = (long)gridDim.x * 4;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += nwaves) {
    const float* kq = kvq + ((size_t)g * as + min(col, as - 1)) * cb3;  // this lane's [K | V | Q] row
    const float* vb = kvq + (size_t)g * as * cb3 + cb;                  // V rows of the group
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    const float mrow = col < as ? 1.f : 0.f;
#pragma unroll
    for (int cbk = 0; cbk < CBLK; ++cbk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cbk * 16 + 4 * grp + r;
        // unconditional loads (the last block's on clamped addresses), masked by a PRODUCT: behind `ok ? value : 0` the
        // compiler makes the load itself conditional -- a branch, the load and a full wait in front of every matrix step
        const bool last = cbk == CBLK - 1;
        const int cc = last ? min(c, cb - 1) : c;
        const float kk = kq[cc], qq = kq[2 * cb + cc];
        const float m = last ? (col < as && c < cb ? 1.f : 0.f) : mrow;
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(kk * m, qq * (qscale * m), S, 0, 0, 0);
      }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = (4 * grp + r) < as ? S[r] : -INFINITY;
      tmax = fmaxf(tmax, S[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[r] = fast_exp2(S[r] - tmax);
      psum += S[r];
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    const float inv = 1.0f / psum;
    // The re-weighting at the end works on TRANSPOSED logits: lane (col, grp) owns output ob*16 + col and the neighbours
    // 4 grp + r.  What it multiplies: column 2 + o of its four neighbours' rows (coalesced over col), and for output 0 their
    // coordinates.  The first block's values are requested here, a matrix chain ahead of their use (and after the K / Q
    // registers have been released) -- not one by one in front of each use.
    const float* xg = x + (size_t)g * as * w;
    int xrow[4];
    float x3[4][3], xo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xrow[r] = min(4 * grp + r, as - 1) * w;
#pragma unroll
      for (int d = 0; d < 3; ++d) x3[r][d] = xg[xrow[r] + min(3 + d, w - 1)];
      xo[r] = xg[xrow[r] + min(2 + col, w - 1)];
    }
    f32x4 H[2] = {ba_r[0], ba_r[1]};
    int vrow[4];
    float vmask[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      vrow[t] = min(4 * grp + t, as - 1) * cb3;
      vmask[t] = 4 * grp + t < as ? 1.f : 0.f;
    }
#pragma unroll
    for (int cbk = 0; cbk < CBLK; ++cbk) {
      f32x4 O = {0.f, 0.f, 0.f, 0.f};
      const int vc = cbk * 16 + col;  // V[key][vc]: lanes over channels
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key = 4 * grp + t;
        const bool last = cbk == CBLK - 1;
        const float v = vb[vrow[t] + (last ? min(vc, cb - 1) : vc)];
        O = __builtin_amdgcn_mfma_f32_16x16x4f32(v * (last ? (key < as && vc < cb ? 1.f : 0.f) : vmask[t]), S[t], O, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = O[r] * inv;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          H[hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Was[(cbk * 16 + 4 * grp + r) * WAS_LD + hb * 16 + col], a, H[hb], 0, 0, 0);
      }
    }
    float hr[2][4];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hr[hb][r] = fmaxf(H[hb][r], 0.f);
    // (a use the compiler cannot move: without it the coordinate loads are sunk into the one branch that reads them)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(x3[r][0]), "v"(x3[r][1]), "v"(x3[r][2]));
    for (int ob = 0; ob < noblk; ++ob) {
      float xn[4];  // the next block's column, in flight under this block's matrix steps and softmax
#pragma unroll
      for (int r = 0; r < 4; ++r) xn[r] = xg[xrow[r] + min(2 + (ob + 1) * 16 + col, w - 1)];
      // logits, transposed: L[r] = logit of output ob*16 + col for neighbour 4 grp + r (the operands of the reference-order
      // product, swapped).  The softmax over the neighbours is then three in-lane steps and two exchanges between the four
      // 16-lane rows per reduction, ONE division per output -- not a 16-lane DPP reduction per (output, reduction)
      f32x4 L;
      const float b0 = Bbs[ob * 16 + col];
#pragma unroll
      for (int r = 0; r < 4; ++r) L[r] = b0;
      const int oc = min(ob * 16 + col, nout - 1);
      const float mo = ob * 16 + col < nout ? 1.f : 0.f;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          L = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[hb][r], Wbs[(hb * 16 + 4 * grp + r) * nout + oc] * mo, L, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        L[r] = 4 * grp + r < as ? L[r] : -INFINITY;
        mx = fmaxf(mx, L[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));  // finite: neighbour 0 is always valid
      float den = 0.f, num = 0.f, n3[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = fast_exp2((L[r] - mx) * LOG2E);  // 0 for the padding neighbours; their (clamped) rows are finite
        den += e;
        num += e * xo[r];
        if (ob == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) n3[d] += e * x3[r][d];
        }
      }
      den += __shfl_xor(den, 16);
      num += __shfl_xor(num, 16);
      den += __shfl_xor(den, 32);
      num += __shfl_xor(num, 32);
      const int o = ob * 16 + col;
      if (ob == 0) {  // (uniform) output 0 re-weights the coordinates
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          n3[d] += __shfl_xor(n3[d], 16);
          n3[d] += __shfl_xor(n3[d], 32);
        }
        if (lane == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) new_xyz[g * 3 + d] = n3[d] / den;
        }
      }
      if (grp == 0 && o >= 1 && o < nout) new_feature[(size_t)g * ch + (o - 1)] = num / den;  // 16 consecutive floats
#pragma unroll
      for (int r = 0; r < 4; ++r) xo[r] = xn[r];
    }
  }
}
}  // namespace pasnl

extern "C" int pasnl_as_cell_wide_ld(int g, int as, int cb, int w, int ch, const float* kvq, int ld, const float* x,
                                     const float* wa, const float* ba, const float* wb, const float* bb, float* new_xyz,
                                     float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && cb > 0 && w > 0 && ch > 0 && ld >= 3 * cb, PASNL_EINVAL);
  PASNL_REQUIRE(as <= 16 && cb <= 144 && w == 3 + ch, PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(kvq && x && wa && ba && wb && bb && new_xyz && new_feature, PASNL_ENULL);
  const size_t lds = ((size_t)33 * (1 + ch) + 16 + (size_t)((cb + 15) / 16 > 9 ? 9 : (cb + 15) / 16) * 16 * 36) * sizeof(float);
  PASNL_REQUIRE(lds <= 64 * 1024, PASNL_EUNSUPPORTED);
  const float qscale = LOG2E / sqrtf((float)cb);
  const long wgs = ((long)g + 3) / 4;
  long cap = 768;  // persistent workgroups: a wave's weights (registers) and the workgroup's Wb (LDS) are loaded once
  if (const char* e = tune_env("PASNL_AS_GRID")) cap = atol(e) > 0 ? atol(e) : cap;  // tuning only
  const dim3 grid((unsigned)(wgs < cap ? wgs : cap)), block(256);
  hipStream_t st = pasnl_hip_stream(stream);
#define PASNL_AS_GO(CBLK)                                                                                                  \
  do {                                                                                                                     \
    auto kern = pasnl::as_cell_wide_kernel<CBLK>;                                                                          \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds) != hipSuccess)                                                    \
      return PASNL_ELAUNCH;                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, lds, st, (long)g, as, cb, w, ch, qscale, kvq, ld, x, wa, ba, wb, bb, new_xyz,    \
                       new_feature);                                                                                       \
  } while (0)
  switch ((cb + 15) / 16) {  // exact: the kernel treats every block of 16 channels but the last as full
    case 1: PASNL_AS_GO(1); break;
    case 2: PASNL_AS_GO(2); break;
    case 3: PASNL_AS_GO(3); break;
    case 4: PASNL_AS_GO(4); break;
    case 5: PASNL_AS_GO(5); break;
    case 6: PASNL_AS_GO(6); break;
    case 7: PASNL_AS_GO(7); break;
    case 8: PASNL_AS_GO(8); break;
    default: PASNL_AS_GO(9); break;
  }
#undef PASNL_AS_GO
  return pasnl_launch_status();
}

extern "C" int pasnl_as_cell_wide(int g, int as, int cb, int w, int ch, const float* kvq, const float* x, const float* wa,
                                  const float* ba, const float* wb, const float* bb, float* new_xyz, float* new_feature,
                                  pasnl_stream_t stream) {
  return pasnl_as_cell_wide_ld(g, as, cb, w, ch, kvq, 3 * cb, x, wa, ba, wb, bb, new_xyz, new_feature, stream);
}

extern "C" int pasnl_as_cell_narrow(int g, int as, int cb, int w, int ch, const float* x, const float* wkvq, const float* bkvq,
                                    const float* wa, const float* ba, const float* wb, const float* bb, float* new_xyz,
                                    float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && cb > 0 && w > 0 && ch > 0, PASNL_EINVAL);
  PASNL_REQUIRE(as <= 16 && w <= pasnl::AS_PROJ_MAXW && (cb == 32 || cb == 64) && 1 + ch <= 16 && w == 3 + ch,
                PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(x && wkvq && bkvq && wa && ba && wb && bb && new_xyz && new_feature, PASNL_ENULL);
  const float qscale = LOG2E / sqrtf((float)cb);
  const long wgs = ((long)g + 3) / 4;
  const dim3 grid((unsigned)(wgs < 2048 ? wgs : 2048)), block(256);
  hipStream_t st = pasnl_hip_stream(stream);
  const int ks = (w + 1 + 3) / 4;
#define PASNL_AS_GO(KS, CBLK)                                                                                            \
  hipLaunchKernelGGL((pasnl::as_cell_narrow_kernel<KS, CBLK>), grid, block, 0, st, (long)g, as, w, ch, qscale, x, wkvq, bkvq, wa, \
                     ba, wb, bb, new_xyz, new_feature)
  if (cb == 32) { if (ks <= 2) PASNL_AS_GO(2, 2); else if (ks == 3) PASNL_AS_GO(3, 2); else PASNL_AS_GO(4, 2); }
  else { if (ks <= 2) PASNL_AS_GO(2, 4); else if (ks == 3) PASNL_AS_GO(3, 4); else PASNL_AS_GO(4, 4); }
#undef PASNL_AS_GO
  return pasnl_launch_status();
}

extern "C" int pasnl_as_attention_proj(int g, int as, int cb, int w, const float* x, const float* wkvq, const float* bkvq,
                                       float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && cb > 0 && w > 0, PASNL_EINVAL);
  PASNL_REQUIRE(as <= 16 && w <= pasnl::AS_PROJ_MAXW && (cb == 32 || cb == 64), PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(x && wkvq && bkvq && out, PASNL_ENULL);
  PASNL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, PASNL_EINVAL);
  const float qscale = LOG2E / sqrtf((float)cb);
  const long wgs = ((long)g + 3) / 4;
  const dim3 grid((unsigned)(wgs < 2048 ? wgs : 2048)), block(256);  // persistent: a lane's weights are loaded once
  hipStream_t st = pasnl_hip_stream(stream);
  const int ks = (w + 1 + 3) / 4;
#define PASNL_AS_GO(KS, CBLK) \
  hipLaunchKernelGGL((pasnl::as_attention_proj_kernel<KS, CBLK>), grid, block, 0, st, (long)g, as, w, qscale, x, wkvq, bkvq, out)
  if (cb == 32) { if (ks <= 2) PASNL_AS_GO(2, 2); else if (ks == 3) PASNL_AS_GO(3, 2); else PASNL_AS_GO(4, 2); }
  else { if (ks <= 2) PASNL_AS_GO(2, 4); else if (ks == 3) PASNL_AS_GO(3, 4); else PASNL_AS_GO(4, 4); }
#undef PASNL_AS_GO
  return pasnl_launch_status();
}

extern "C" int pasnl_as_attention(int g, int as, int cb, const float* q, const float* kv, float* out,
                                  pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && cb > 0, PASNL_EINVAL);
  PASNL_REQUIRE(as <= 16 && cb <= 256, PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(q && kv && out, PASNL_ENULL);
  const float qscale = LOG2E / sqrtf((float)cb);
  hipLaunchKernelGGL(as_attention_kernel, dim3((unsigned)(((long)g + 3) / 4)), dim3(256), 0, pasnl_hip_stream(stream), (long)g,
                     as, cb, qscale, cb, 2 * cb, q, kv, out);
  return pasnl_launch_status();
}

extern "C" int pasnl_as_attention_qkv(int g, int as, int cb, const float* kvq, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && cb > 0, PASNL_EINVAL);
  PASNL_REQUIRE(as <= 16 && cb <= 256, PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(kvq && out, PASNL_ENULL);
  const float qscale = LOG2E / sqrtf((float)cb);
  hipLaunchKernelGGL(as_attention_kernel, dim3((unsigned)(((long)g + 3) / 4)), dim3(256), 0, pasnl_hip_stream(stream), (long)g,
                     as, cb, qscale, 3 * cb, 3 * cb, kvq + 2 * cb, kvq, out);
  return pasnl_launch_status();
}

extern "C" int pasnl_as_gather(int b, int n, int c, int m, int k, int as, const float* xyz, const float* feature, const int* idx,
                               float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && k > 0 && as > 0 && as <= k, PASNL_EINVAL);
  long rows = (long)b * m * as;
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && feature && idx && out, PASNL_ENULL);
  if (c < 26) {  // fewer than 32 columns: a wave per row would idle most of its lanes (74 vs 18 us at cls layer1)
    long grid = (rows * (6 + c) + 255) / 256;
    hipLaunchKernelGGL(as_gather_elem_kernel, dim3((unsigned)(grid > 16384 ? 16384 : grid)), dim3(256), 0, pasnl_hip_stream(stream),
                       n, c, m, k, as, rows * (6 + c), xyz, feature, idx, out);
    return pasnl_launch_status();
  }
  long grid = (rows + 3) / 4;
  hipLaunchKernelGGL(as_gather_kernel, dim3((unsigned)(grid > 16384 ? 16384 : grid)), dim3(256), 0, pasnl_hip_stream(stream), n, c,
                     m, k, as, rows, xyz, feature, idx, out);
  return pasnl_launch_status();
}

extern "C" int pasnl_as_reweight(int g, int as, int nsample, int ch, const float* logits, const float* grouped_xyz,
                                 const float* grouped_feature, float* new_xyz, float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && nsample >= as && ch > 0, PASNL_EINVAL);
  PASNL_REQUIRE(as <= AS_MAX, PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(logits && grouped_xyz && grouped_feature && new_xyz && new_feature, PASNL_ENULL);
  long total = (long)g * (1 + ch);
  long grid = (total + 255) / 256;
  hipLaunchKernelGGL(as_reweight_kernel, dim3((unsigned)(grid > 16384 ? 16384 : grid)), dim3(256), 0, pasnl_hip_stream(stream),
                     (long)g, as, nsample, ch, 3, ch, logits, grouped_xyz, grouped_feature, new_xyz, new_feature);
  return pasnl_launch_status();
}

extern "C" int pasnl_as_reweight_x(int g, int as, int ch, const float* logits, const float* x, float* new_xyz,
                                   float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE(g >= 0 && as > 0 && ch > 3, PASNL_EINVAL);
  PASNL_REQUIRE(as <= AS_MAX, PASNL_EUNSUPPORTED);
  if (g == 0) return PASNL_OK;
  PASNL_REQUIRE(logits && x && new_xyz && new_feature, PASNL_ENULL);
  // x rows = [xyz - xyz0 | xyz | feature] (3 + ch floats): coordinates and the (xyz | feature) rows both start at column 3
  const int w = 3 + ch;
  long total = (long)g * (1 + ch);
  long grid = (total + 255) / 256;
  hipLaunchKernelGGL(as_reweight_kernel, dim3((unsigned)(grid > 16384 ? 16384 : grid)), dim3(256), 0, pasnl_hip_stream(stream),
                     (long)g, as, as, ch, w, w, logits, x + 3, x + 3, new_xyz, new_feature);
  return pasnl_launch_status();
}

template <int C1, int C2>
static int local_cell_launch(long groups, int k, int w, const float* x, const float* w0, const float* b0,
                             const float* w1, const float* b1, const float* ww, const float* bw, float* out, hipStream_t st) {
  const int wp = (w + 31) & ~31;
  size_t lds = ((size_t)wp * C1 + (size_t)C1 * C2 + 4 * 32 + C1) * sizeof(float);
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  auto kern = sa_local_cell_kernel<C1, C2>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  // persistent workgroups: the weights are staged into LDS once per workgroup
  long wgs = (groups + 3) / 4;
  int per_cu = lds > 80 * 1024 ? 1 : (lds > 40 * 1024 ? 2 : 3);
  long cap = 256L * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(256), lds, st, groups, k, w, x, w0, b0, w1, b1,
                     ww, bw, out);
  return pasnl_launch_status();
}

static int local_cell_dispatch(long groups, int k, int w, int c1, int c2, const float* x, const float* w0,
                               const float* b0, const float* w1, const float* b1, const float* ww, const float* bw,
                               float* out, hipStream_t st) {
  if (c1 == 32 && c2 == 32) return local_cell_launch<32, 32>(groups, k, w, x, w0, b0, w1, b1, ww, bw, out, st);
  if (c1 == 64 && c2 == 64) return local_cell_launch<64, 64>(groups, k, w, x, w0, b0, w1, b1, ww, bw, out, st);
  if (c1 == 128 && c2 == 128) return local_cell_launch<128, 128>(groups, k, w, x, w0, b0, w1, b1, ww, bw, out, st);
  return PASNL_EUNSUPPORTED;
}

template <int C1, int C2, int NW, bool VEC, bool TAIL8, bool XYZ3 = false, bool SINGLE = false>
static int sa_cell_launch(long groups, int k, int w, SaGatherSrc src, const float* w0, const float* b0, const float* w1,
                          const float* b1, const float* ww, const float* bw, float* out, hipStream_t st) {
  const int wp = (8 + (w - 6) + 31) & ~31;  // internal width: [xyz-c | xyz | 1 | 0 | feature], padded to 32-chunks
  size_t lds = ((size_t)wp * C1 + (size_t)C1 * C2 + 6 * 32 + (size_t)NW * SA_SKIP_REP * (wp + 4)) * sizeof(float);
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  auto kern = sa_cell_kernel<C1, C2, NW, VEC, TAIL8, XYZ3, SINGLE>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  // persistent workgroups (the weights are staged into LDS once per workgroup): exactly as many as are resident at once
  long wgs = (groups + NW - 1) / NW;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), NW * 64, lds) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  // Over-subscribing the resident count (so that the hardware dispatcher balances the load when another stream's
  // kernels keep some CUs busy) was measured and lost (cls layer2: 429 / 455 / 505 / 528 us at 1x / 2x / 4x / 8x, and
  // the two-lane cls step 1.52 / 1.54 / 1.60 / 1.65 ms).
  long cap = 256L * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(NW * 64), lds, st, groups, k, w, src, w0, b0, w1, b1, ww,
                     bw, out);
  return pasnl_launch_status();
}

// =============================================================================================
// pasnl_sa_cell for the 16-channel first layer of pointasnl_sem_seg_res (mlp [16, 16, 32] on xyz-only rows, k = 32:
// pointasnl_sem_seg_res.py:32): the same four products on v_mfma_f32_16x16x4_f32, whose 16-row tiles a 16-channel layer
// fills -- on the 32x32x2 kernel above half of every conv0 / matmul tile and three quarters of conv1 multiply zero
// padding, and the padded 32-channel output doubles the bytes written.  34 MFMAs of 32 cycles per group instead of 40 of
// 64, 2 KiB stored per group instead of 4.
//   A tile of the 16x16x4 MFMA: lane l = (n = l & 15, q = l >> 4) holds A[n][k = q], B[k = q][n], D[4 q + r][n] in
//   register r -- so a D tile IS the next product's A (or B) operand when that product's k index at step s, lane group q
//   is defined as 4 q + s (the chaining trick of the big kernel, one size down):
//     H1^T[c][p]   : A = W0[col][c] (regs)  B = X[p][col] (regs)   3 steps: cols [dx dy dz x | y z fx fy | fz 1 0 0]
//     G   [p][j]   : A = [dx dy dz 1][p]    B = [Ww ; bw][j]       1 step per 16 columns (tile jt holds columns 2 n + jt)
//     H2  [p][c2]  : A = H1^T regs          B = W1[c][c2] (regs)   4 steps
//     M   [c2][j]  : A = H2 regs            B = G regs             4 steps per row tile
//   The weights are 3 + 2 + 4 registers per lane for the whole kernel (no LDS at all); a group is two row tiles of 16
//   neighbours; the skip maxima are three 16-lane DPP reductions (lane group q owns columns q, 4 + q, 8 + q -- its own
//   conv0 operands); a row of M leaves as 8-byte stores that cover 128 contiguous bytes per lane group.
// =============================================================================================
template <int NW>
__global__ __launch_bounds__(NW * 64) void sa_cell16_kernel(long groups, SaGatherSrc src, const float* __restrict__ w0,
                                                           const float* __restrict__ b0, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ ww,
                                                           const float* __restrict__ bw, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const bool q0 = q == 0, q1 = q == 1, q2 = q == 2;
  auto pick = [&](float a, float b, float c, float d) { return q0 ? a : (q1 ? b : (q2 ? c : d)); };
  // ---- the weights, once per wave.  conv0: internal column 4 s + q of [xyz - c | xyz | feature | 1 | 0 | 0] -> channel n
  float a0[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int col = 4 * s + q;
    a0[s] = col < 9 ? w0[col * 16 + n] : (col == 9 ? b0[n] : 0.f);
  }
  float wn[2];  // weight net: row q of [Ww ; bw], column 2 n + jt
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) wn[jt] = q < 3 ? ww[q * 32 + 2 * n + jt] : bw[2 * n + jt];
  float w1r[4];  // conv1: input channel 4 q + s -> output channel n
#pragma unroll
  for (int s = 0; s < 4; ++s) w1r[s] = w1[(4 * q + s) * 16 + n];
  const float b1n = b1[n];

  const int m = src.m;
  const uint32_t first = blockIdx.x * NW + (threadIdx.x >> 6), stride = gridDim.x * NW;  // groups < 2^31 (sa_cell_entry)
  for (uint32_t g = first; g < (uint32_t)groups; g += stride) {
    const uint32_t bi = g / (uint32_t)m;
    const int* gi = src.idx + (size_t)g * 32;
    const int i0 = gi[n], i1 = gi[16 + n];  // the rows of the two tiles this lane stands for
    const float* cen = src.new_xyz + (size_t)g * 3;
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    const float* base = src.xyz + (size_t)bi * src.n * 3;
    const float* fbase = src.feature + (size_t)bi * src.n * 3;
    const float* pr[2] = {base + (size_t)i0 * 3, base + (size_t)i1 * 3};
    const float* fr[2] = {fbase + (size_t)i0 * 3, fbase + (size_t)i1 * 3};
    f32x4 M[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float sk[3] = {-INFINITY, -INFINITY, -INFINITY};  // running maxima of columns q, 4 + q, 8 + q over the group's rows
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float px = pr[t][0], py = pr[t][1], pz = pr[t][2];
      const float fx = fr[t][0], fy = fr[t][1], fz = fr[t][2];
      const float dx = px - cx, dy = py - cy, dz = pz - cz;
      // this lane's operand of the three conv0 steps = columns q, 4 + q, 8 + q of its row
      const float x0 = pick(dx, dy, dz, px), x1 = pick(py, pz, fx, fy), x2 = pick(fz, 1.f, 0.f, 0.f);
      sk[0] = max_sel(sk[0], x0); sk[1] = max_sel(sk[1], x1); sk[2] = max_sel(sk[2], x2);
      f32x4 H1 = {0.f, 0.f, 0.f, 0.f};
      H1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], x0, H1, 0, 0, 0);
      H1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], x1, H1, 0, 0, 0);
      H1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], x2, H1, 0, 0, 0);
      // weight net on the centred coordinates (+ bias through the constant 1): rows x columns 2 n + jt
      const float gx = q < 3 ? x0 : 1.f;
      f32x4 G[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        G[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        G[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gx, wn[jt], G[jt], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) G[jt][r] = relu_bits(G[jt][r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) H1[r] = relu_bits(H1[r]);  // (the conv0 bias came with the MFMA)
      f32x4 H2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) H2 = __builtin_amdgcn_mfma_f32_16x16x4f32(H1[s], w1r[s], H2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) H2[r] = fmaxf(H2[r] + b1n, 0.f);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int s = 0; s < 4; ++s) M[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(H2[s], G[jt][s], M[jt], 0, 0, 0);
    }
    // M[jt][r] = channel 4 q + r, column 2 n + jt -> out[g][c * 32 + j]: lane group q stores 128 contiguous bytes per channel
    float* o = out + (size_t)g * 16 * 32 + (size_t)(4 * q) * 32 + 2 * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(o + r * 32) = make_float2(M[0][r], M[1][r]);
    // skip maxima: reference column c = internal column c (c < 9); lane group q holds columns q, 4 + q, 8 + q (all-reduced)
#pragma unroll
    for (int s = 0; s < 3; ++s) sk[s] = row16_max(sk[s]);
    if (n == 15) {
      float* so = src.skip_max + (size_t)g * 9;
      so[q] = sk[0];
      so[4 + q] = sk[1];
      if (q == 0) so[8] = sk[2];
    }
  }
}

static int sa_cell16_launch(long groups, SaGatherSrc src, const float* w0, const float* b0, const float* w1, const float* b1,
                            const float* ww, const float* bw, float* out, hipStream_t st) {
  constexpr int NW = 4;
  // no LDS, ~64 registers: 8 waves per SIMD hide the two dependent round trips (indices, rows) of a group; the grid is
  // sized to fill the chip a few times over and strides over the groups
  const long wgs = (groups + NW - 1) / NW, cap = 256L * 8 * 2;
  hipLaunchKernelGGL(sa_cell16_kernel<NW>, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(NW * 64), 0, st, groups, src, w0, b0, w1,
                     b1, ww, bw, out);
  return pasnl_launch_status();
}

// =============================================================================================
// pasnl_sa_cell for the WIDE layers (c1 = c2 = 256 or 512: pointasnl_sem_seg.py:34 layer4, pointasnl_sem_seg_res.py:46-51
// layer3_2 / 4_1 / 4_2): their weights (262 x 256 + 256 x 256 floats and up) do not fit the LDS, and they have few groups
// (320 .. 640), so the persistent one-wave-per-group kernel above has nothing to amortise a weight copy over.  Here ONE
// WORKGROUP owns one group and its c / 32 waves each own a 32-channel block of both convolutions:
//   1. the group's 32 rows [xyz - centre | xyz | 1 | 0 | feature] are gathered into LDS once (Xs, odd row pitch);
//      the skip maxima are column maxima of that tile;
//   2. conv0: wave v accumulates H1^T block v (32 channels x 32 rows): A = W0 rows straight from global memory (L2: every
//      workgroup streams the same 270 .. 1 060 KiB), B = X from LDS; ReLU; the block goes to LDS as H1[row][channel];
//   3. conv1 (when the layer has one): H2 block v (32 rows x 32 channels) from H1 (LDS) and W1 (global); bias, ReLU;
//      a layer with a single convolution (mlp = [c, c]) reads its H2 block back from H1 instead -- no identity product;
//   4. G = relu(X[:, 0:3] Ww + bw) per wave (3 MFMA steps), M block v = H2^T G, stored as in the kernel above.
//   v_mfma_f32_32x32x2_f32 throughout, D tiles chained as operands exactly as above (kappa).  k = 32 (one tile per group).
// =============================================================================================
// PACKED: the feature rows of w0 (rows 6 ..) and w1 also arrive in the matrix instruction's operand order
// (pasnl_mlp3_pack_weights' layout: P[batch of 8 steps][h][channel][8]): two 16-byte loads per lane and batch instead of eight
// 4-byte ones (what that is worth: EXPERIMENTS "Round 5", mlp3_pool / sa_tail).
template <int C, bool CONV1, bool PACKED>
__global__ __launch_bounds__(C / 32 * 64) void sa_cell_wide_kernel(int w, SaGatherSrc src, const float* __restrict__ w0,
                                                                  const float* __restrict__ b0, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ ww,
                                                                  const float* __restrict__ bw, float* __restrict__ out,
                                                                  const float* __restrict__ w0p, const float* __restrict__ w1p) {
  constexpr int NW = C / 32, T = NW * 64, HP = C + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cf = w - 6, wi = 8 + cf;        // internal width (a multiple of 8: cf % 4 == 0 and the launcher asks for cf % 8 == 0)
  const int xp = wi + 1;                    // odd row pitch: conflict-free column-pair reads by 32 rows
  float* Xs = reinterpret_cast<float*>(smem);          // [32][xp]
  float* H1s = Xs + 32 * xp;                           // [32][HP]
  int* rows = reinterpret_cast<int*>(CONV1 ? H1s + 32 * HP : H1s);  // [32] neighbour indices of the group (no H1 tile without conv1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ql = lane & 31;
  const int g = blockIdx.x;
  const int bi = g / src.m;
  if (tid < 32) rows[tid] = src.idx[(size_t)g * 32 + tid];
  __syncthreads();
  // ---- 1. gather
  {
    const float* fb = src.feature + (size_t)bi * src.n * cf;
    const int q4 = cf >> 2;
    for (int i = tid; i < 32 * q4; i += T) {
      const int r = i / q4, q = i - r * q4;
      const float4 v = reinterpret_cast<const float4*>(fb + (size_t)rows[r] * cf)[q];
      float* d = Xs + r * xp + 8 + 4 * q;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (tid < 32) {
      const float* pp = src.xyz + ((size_t)bi * src.n + rows[tid]) * 3;
      // centre0: the centre of the group is its neighbour 0 (pointasnl_util.py:161-163)
      const float* cc = src.centre0 ? src.xyz + ((size_t)bi * src.n + rows[0]) * 3 : src.new_xyz + (size_t)g * 3;
      const float px = pp[0], py = pp[1], pz = pp[2];
      float* d = Xs + tid * xp;
      d[0] = px - cc[0]; d[1] = py - cc[1]; d[2] = pz - cc[2];
      d[3] = px; d[4] = py; d[5] = pz; d[6] = 1.f; d[7] = 0.f;
    }
  }
  __syncthreads();
  if (src.centre0 && src.new_feature_out) {
    // what pasnl_take_neighbor0 would have written: new_xyz = the centre, new_feature = [centre | feature row of neighbour 0]
    float* nfo = src.new_feature_out + (size_t)g * (3 + cf);
    for (int c = tid; c < 3 + cf; c += T) {
      const float v = Xs[c < 3 ? 3 + c : 8 + (c - 3)];  // row 0 of the tile: columns 3..5 = xyz, 8.. = the feature row
      nfo[c] = v;
      if (c < 3) src.new_xyz_out[(size_t)g * 3 + c] = v;
    }
  }
  // skip maxima (pointasnl_util.py:258): reference column c = internal column c (c < 6) or c + 2
  for (int c = tid; c < w; c += T) {
    const float* col = Xs + (c < 6 ? c : c + 2);
    float mx = col[0];
#pragma unroll 8
    for (int r = 1; r < 32; ++r) mx = fmaxf(mx, col[r * xp]);
    src.skip_max[(size_t)g * w + c] = mx;
  }
  // ---- 2. conv0: H1^T block `wave`.  Step s contracts internal columns 2 s (lanes 0..31) and 2 s + 1 (lanes 32..63);
  // internal rows of W0: 0..5 = w0 rows 0..5, 6 = b0 (the bias rides on the constant-1 column), 7 = zero, 8.. = w0 rows 6..
  // With conv1 the block is wanted as H1^T (channels x rows: the weights are the A operand); WITHOUT it the block IS H2 and
  // is wanted as rows x channels -- the same products with the operands swapped, no exchange through LDS, no barrier.
  auto mm = [](float wv, float xv, f32x16 d) {
    return CONV1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv, d, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(xv, wv, d, 0, 0, 0);
  };
  const int ch = wave * 32 + ql;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const float* xrow = Xs + ql * xp + h;
    // the first four steps (columns 0..7)
    const float a3 = h ? 0.f : b0[ch];
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = mm(w0[(size_t)(2 * s + h) * C + ch], xrow[2 * s], acc);
    acc = mm(a3, xrow[6], acc);
    // the features: w0 row 2 s + h - 2, in batches of BT steps; the operands of the batch after the NEXT one are requested
    // while this one multiplies (three register sets: a weight row is a round trip to L2 of about two batches of MFMAs when
    // four waves share the matrix pipe)
    constexpr int BT = 8;   // (16 measured slower: 139 vs 126 us at 320 groups of 512 channels)
    const float* wp = w0 + (size_t)(6 + h) * C + ch;  // w0 row of internal column 8 + h
    const int nsteps = cf >> 1;                        // a multiple of BT (cf % 16 == 0), >= 2 BT
    float wa[3][BT], xb[3][BT];
    // the weights of steps sb .. sb + BT - 1 (sb a multiple of BT) into register set `set`
    auto load_w = [&](int set, int sb, const float* rowmajor, const float* packed) {
      if constexpr (PACKED) {
        const float4* q = reinterpret_cast<const float4*>(packed + ((size_t)((sb >> 3) * 2 + h) * C + ch) * 8);
        const float4 a = q[0], b = q[1];
        wa[set][0] = a.x; wa[set][1] = a.y; wa[set][2] = a.z; wa[set][3] = a.w;
        wa[set][4] = b.x; wa[set][5] = b.y; wa[set][6] = b.z; wa[set][7] = b.w;
      } else {
#pragma unroll
        for (int u = 0; u < BT; ++u) wa[set][u] = rowmajor[(size_t)(2 * (sb + u)) * C];
      }
    };
    static_assert(BT == 8, "a packed batch is eight steps");
    load_w(0, 0, wp, w0p);
    load_w(1, BT, wp, w0p);
#pragma unroll
    for (int u = 0; u < BT; ++u) { xb[0][u] = xrow[8 + 2 * u]; xb[1][u] = xrow[8 + 2 * (BT + u)]; }
    for (int s0 = 0; s0 < nsteps; s0 += 3 * BT) {
#pragma unroll
      for (int third = 0; third < 3; ++third) {
        const int sb = s0 + third * BT;
        if (sb < nsteps) {
          const int sn = min(sb + 2 * BT, nsteps - BT);  // two batches ahead (a dummy re-read at the end)
#pragma unroll
          for (int u = 0; u < BT; ++u) xb[(third + 2) % 3][u] = xrow[8 + 2 * (sn + u)];
          load_w((third + 2) % 3, sn, wp, w0p);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < BT; ++u) acc = mm(wa[third][u], xb[third][u], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // ---- 3. conv1 (or the block of H1 itself): H2[m = row kappa(t,h)][n = channel ch]
  f32x16 H2;
  if constexpr (CONV1) {
    // ReLU; D[m = channel wave*32 + kappa(r,h)][n = row ql] -> H1s[row][channel]
#pragma unroll
    for (int r = 0; r < 16; ++r) H1s[ql * HP + wave * 32 + kappa(r, h)] = fmaxf(acc[r], 0.f);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) H2[r] = 0.f;
    constexpr int BT = 8, NS = C / 2;
    const float* hrow = H1s + ql * HP + h;
    const float* wp = w1 + (size_t)h * C + ch;
    float wa[2][BT], xb[2][BT];  // (two register sets: a third, as in conv0, gained nothing here -- H1 comes from LDS)
    auto load_w = [&](int set, int sb) {
      if constexpr (PACKED) {
        const float4* q = reinterpret_cast<const float4*>(w1p + ((size_t)((sb >> 3) * 2 + h) * C + ch) * 8);
        const float4 a = q[0], b = q[1];
        wa[set][0] = a.x; wa[set][1] = a.y; wa[set][2] = a.z; wa[set][3] = a.w;
        wa[set][4] = b.x; wa[set][5] = b.y; wa[set][6] = b.z; wa[set][7] = b.w;
      } else {
#pragma unroll
        for (int u = 0; u < BT; ++u) wa[set][u] = wp[(size_t)(2 * (sb + u)) * C];
      }
    };
    load_w(0, 0);
#pragma unroll
    for (int u = 0; u < BT; ++u) xb[0][u] = hrow[2 * u];
    for (int s0 = 0; s0 < NS; s0 += 2 * BT) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int sb = s0 + half * BT;
        const int sn = min(sb + BT, NS - BT);
#pragma unroll
        for (int u = 0; u < BT; ++u) xb[half ^ 1][u] = hrow[2 * (sn + u)];
        load_w(half ^ 1, sn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BT; ++u) H2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[half][u], wa[half][u], H2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float bb = b1[ch];
#pragma unroll
    for (int r = 0; r < 16; ++r) H2[r] = fmaxf(H2[r] + bb, 0.f);
  } else {
#pragma unroll
    for (int t = 0; t < 16; ++t) H2[t] = fmaxf(acc[t], 0.f);  // D[m = row kappa(t,h)][n = channel ch]
  }
  // ---- 4. weight net on the centred coordinates: G[m = row kappa][n = j]; columns 0..3 = dx dy dz x (x is no input: zero row)
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = 0.f;
  {
    const float* xrow = Xs + ql * xp + h;
    const float b0w = ww[h * 32 + ql], b1w = h ? 0.f : ww[2 * 32 + ql];
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(xrow[0], b0w, G, 0, 0, 0);
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(xrow[2], b1w, G, 0, 0, 0);
    const float bj = bw[ql];
#pragma unroll
    for (int r = 0; r < 16; ++r) G[r] = fmaxf(G[r] + bj, 0.f);
  }
  f32x16 M;
#pragma unroll
  for (int r = 0; r < 16; ++r) M[r] = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) M = __builtin_amdgcn_mfma_f32_32x32x2f32(H2[t], G[t], M, 0, 0, 0);
  float* o = out + (size_t)g * C * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[(size_t)(wave * 32 + kappa(r, h)) * 32 + ql] = M[r];
}

template <int C, bool CONV1>
static int sa_cell_wide_launch(long groups, int w, SaGatherSrc src, const float* w0, const float* b0, const float* w1,
                               const float* b1, const float* ww, const float* bw, float* out, hipStream_t st,
                               const float* w0p = nullptr, const float* w1p = nullptr) {
  const int wi = 8 + (w - 6);
  const size_t lds = ((size_t)32 * (wi + 1) + (CONV1 ? (size_t)32 * (C + 1) : 0) + 32) * sizeof(float);
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  // packed operands: both matrices (the one convolution a layer has, when it has one), 16-byte aligned
  const bool packed = w0p && (!CONV1 || w1p) && (reinterpret_cast<uintptr_t>(w0p) | reinterpret_cast<uintptr_t>(w1p)) % 16 == 0;
  auto kern = packed ? sa_cell_wide_kernel<C, CONV1, true> : sa_cell_wide_kernel<C, CONV1, false>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(C / 32 * 64), lds, st, w, src, w0, b0, w1, b1, ww, bw, out, w0p, w1p);
  return pasnl_launch_status();
}

// Waves per workgroup: two waves per SIMD (8 per workgroup, one LDS copy of the weights) where 256 registers per wave
// suffice (c1 <= 64; 201 vs 240 us at cls layer1, 44 vs 59 us at ScanNet layer2 when measured); the 128-channel cell
// needs ~350 registers (at 8 waves it spilled: 3 % faster, 2.4x the HBM bytes) and runs one wave per SIMD.
template <int C1, int C2>
static int sa_cell_cfg(bool vec, bool tail8, long groups, int k, int w, SaGatherSrc src, const float* w0, const float* b0,
                       const float* w1, const float* b1, const float* ww, const float* bw, float* out, hipStream_t st) {
  constexpr int NW = C1 >= 128 ? 4 : 8;
  // NG: the 64-channel forms no model uses (no 8-step tail, or scalar loads of a row that is not xyz-only) need more than the
  // 256 registers two waves per SIMD leave each -- they ran with 150-300 bytes of spills; one wave per SIMD instead
  constexpr int NG = C1 >= 64 ? 4 : 8;
  if (!w1) {  // one convolution: the form every model's *_2 layers have (16-byte feature rows, an 8-step last chunk)
    if (vec && tail8) return sa_cell_launch<C1, C2, NW, true, true, false, true>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
    return PASNL_EUNSUPPORTED;
  }
  if (vec) return tail8 ? sa_cell_launch<C1, C2, NW, true, true>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st)
                        : sa_cell_launch<C1, C2, NG, true, false>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
  if constexpr (C1 <= 64) {
    if (w == 9 && tail8 && k == 32)  // the xyz-only first layer of every model: rows [xyz - c | xyz | xyz-as-feature]
      return sa_cell_launch<C1, C2, NW, false, true, true>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
  }
  return tail8 ? sa_cell_launch<C1, C2, NG, false, true>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st)
               : sa_cell_launch<C1, C2, NG, false, false>(groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
}

extern "C" int pasnl_sa_local_cell(int groups, int k, int w, int c1, int c2, const float* x, const float* w0, const float* b0,
                                   const float* w1, const float* b1, const float* ww, const float* bw, float* out,
                                   pasnl_stream_t stream) {
  PASNL_REQUIRE(groups >= 0 && k > 0 && w >= 3 && c1 > 0 && c2 > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k % 32 == 0, PASNL_EUNSUPPORTED);
  if (groups == 0) return PASNL_OK;
  PASNL_REQUIRE(x && w0 && b0 && w1 && b1 && ww && bw && out, PASNL_ENULL);
  return local_cell_dispatch(groups, k, w, c1, c2, x, w0, b0, w1, b1, ww, bw, out, pasnl_hip_stream(stream));
}

static int sa_cell_entry(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                         const int* idx, const float* new_xyz, const float* w0, const float* b0, const float* w1,
                         const float* b1, const float* ww, const float* bw, float* out, float* skip_max, float* new_xyz_out,
                         float* new_feature_out, pasnl_stream_t stream, const float* w0p = nullptr, const float* w1p = nullptr) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && m >= 0 && k > 0 && c1 > 0 && c2 > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k % 32 == 0, PASNL_EUNSUPPORTED);
  const long groups = (long)b * m;
  if (groups == 0) return PASNL_OK;
  PASNL_REQUIRE(groups < (1L << 31), PASNL_EUNSUPPORTED);
  PASNL_REQUIRE((long)b * n * c < (1L << 32), PASNL_EUNSUPPORTED);  // feature rows are addressed by 32-bit element offsets
  PASNL_REQUIRE(xyz && feature && idx && w0 && b0 && ww && bw && out && skip_max, PASNL_ENULL);
  // w1 == NULL: the layer has ONE convolution (mlp = [c, c]: the *_2 layers of pointasnl_sem_seg_res.py) -- the wide kernel only
  PASNL_REQUIRE((w1 && b1) || c1 >= 32, PASNL_ENULL);  // (the 16-channel kernel has both convolutions)
  // new_xyz == NULL: the centre of a group is its neighbour 0.  The kernel's centre prefetch stays unconditional and is
  // pointed at xyz, whose b*n*3 floats cover the b*m*3 it touches when m <= n
  PASNL_REQUIRE(new_xyz || m <= n, PASNL_EUNSUPPORTED);
  SaGatherSrc src{xyz, feature, idx, new_xyz ? new_xyz : xyz, skip_max, n, m, new_xyz ? 0 : 1, new_xyz_out, new_feature_out};
  hipStream_t st = pasnl_hip_stream(stream);
  const int w = 6 + c;
  // 16-byte operand loads need 16-byte aligned feature rows
  const bool vec = (c % 4 == 0) && (reinterpret_cast<uintptr_t>(feature) % 16 == 0);
  // the row's last chunk: live MFMA steps (0 = the width is a multiple of 32); see TAIL8
  const int wi = 8 + c, rem = wi & 31;
  const bool tail8 = rem != 0 && (vec ? rem : (rem + 1) >> 1) <= 8;
  if (c1 == 128 && c2 == 128 && !w1 && k == 32 && c % 16 == 0 && c >= 32 && new_xyz && reinterpret_cast<uintptr_t>(feature) % 16 == 0) {
    // a 128-channel layer with ONE convolution (mlp = [128, 128]: pointasnl_sem_seg_res.py layer2_2) on the wide kernel's
    // single-convolution form: 60 us at 2560 groups (86 with an identity conv1 on the persistent kernel below, which takes
    // the layer -- without the identity -- where the wide kernel's conditions do not hold)
    return sa_cell_wide_launch<128, false>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p);
  }
  // A 128-channel layer with FEW groups (pointasnl_sem_seg_res.py layer3_1: 640 groups, pointasnl_sem_seg.py layer3: 1024) on
  // the wide kernel as well: the persistent kernel stages 144 KB of weights into the LDS of each of 256 workgroups before a
  // wave sees its first -- and here only -- group (267 us for 640 groups, 121 us for 1 024; 354 us for the classifier's 8 192)
  constexpr long SA_WIDE128_MAX_GROUPS = 2048;
  if (c1 == 128 && c2 == 128 && w1 && groups <= SA_WIDE128_MAX_GROUPS && k == 32 && c % 16 == 0 && c >= 32 &&
      reinterpret_cast<uintptr_t>(feature) % 16 == 0)
    return sa_cell_wide_launch<128, true>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p);
  if ((c1 == 256 && c2 == 256) || (c1 == 512 && c2 == 512)) {  // the wide layers: one workgroup per group, weights from L2
    // c >= 32: the wide kernel preloads TWO batches of conv0's weight rows and of the LDS row unconditionally (nsteps = c / 2 >= 2 BT)
    PASNL_REQUIRE(k == 32 && c % 16 == 0 && c >= 32 && reinterpret_cast<uintptr_t>(feature) % 16 == 0, PASNL_EUNSUPPORTED);
    if (c1 == 256)
      return w1 ? sa_cell_wide_launch<256, true>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p)
                : sa_cell_wide_launch<256, false>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p);
    return w1 ? sa_cell_wide_launch<512, true>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p)
              : sa_cell_wide_launch<512, false>(groups, w, src, w0, b0, w1, b1, ww, bw, out, st, w0p, w1p);
  }
  if (c1 == 16 && c2 == 16) {  // the 16-channel first layer: xyz-only rows, 32 neighbours, centres from a table
    PASNL_REQUIRE(c == 3 && k == 32 && new_xyz, PASNL_EUNSUPPORTED);
    return sa_cell16_launch(groups, src, w0, b0, w1, b1, ww, bw, out, st);
  }
  PASNL_REQUIRE(!new_feature_out || c <= 128, PASNL_EUNSUPPORTED);  // persistent kernel: two words per lane carry neighbour 0's row
  if (c1 == 32 && c2 == 32) return sa_cell_cfg<32, 32>(vec, tail8, groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
  if (c1 == 64 && c2 == 64) return sa_cell_cfg<64, 64>(vec, tail8, groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
  if (c1 == 128 && c2 == 128) return sa_cell_cfg<128, 128>(vec, tail8, groups, k, w, src, w0, b0, w1, b1, ww, bw, out, st);
  return PASNL_EUNSUPPORTED;
}

extern "C" int pasnl_sa_cell(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                             const int* idx, const float* new_xyz, const float* w0, const float* b0, const float* w1,
                             const float* b1, const float* ww, const float* bw, float* out, float* skip_max,
                             pasnl_stream_t stream) {
  return sa_cell_entry(b, n, c, m, k, c1, c2, xyz, feature, idx, new_xyz, w0, b0, w1, b1, ww, bw, out, skip_max, nullptr,
                       nullptr, stream);
}

extern "C" int pasnl_sa_cell_packed(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                                    const int* idx, const float* new_xyz, const float* w0, const float* b0, const float* w1,
                                    const float* b1, const float* ww, const float* bw, const float* w0_features_packed,
                                    const float* w1_packed, float* out, float* skip_max, float* new_xyz_out,
                                    float* new_feature_out, pasnl_stream_t stream) {
  PASNL_REQUIRE(new_xyz || (long)b * m == 0 || (new_xyz_out && new_feature_out), PASNL_ENULL);
  return sa_cell_entry(b, n, c, m, k, c1, c2, xyz, feature, idx, new_xyz, w0, b0, w1, b1, ww, bw, out, skip_max,
                       new_xyz ? nullptr : new_xyz_out, new_xyz ? nullptr : new_feature_out, stream, w0_features_packed, w1_packed);
}

extern "C" int pasnl_sa_cell_centre0(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                                     const int* idx, const float* w0, const float* b0, const float* w1, const float* b1,
                                     const float* ww, const float* bw, float* out, float* skip_max, float* new_xyz,
                                     float* new_feature, pasnl_stream_t stream) {
  PASNL_REQUIRE((long)b * m == 0 || (new_xyz && new_feature), PASNL_ENULL);
  return sa_cell_entry(b, n, c, m, k, c1, c2, xyz, feature, idx, nullptr, w0, b0, w1, b1, ww, bw, out, skip_max, new_xyz,
                       new_feature, stream);
}

namespace pasnl {
// =============================================================================================
// Tail of a set-abstraction layer (pointasnl_util.py:258-261, 213-216, 282-290), fused:
//     out = relu( ( A + relu(S Ws + bs) + relu(N Wb + bb) ) Wagg + bagg )
//   A (rows, C)  the after_conv output (already ReLU'd by its GEMM epilogue),   S (rows, w)  the skip maxima of the groups,
//   N (rows, cb) the non-local attention output (absent when the layer has no non-local cell),
//   Ws / Wb / Wagg  the BN-folded weights of the `skip`, `conv_back_project` and `aggregation` layers.
// In the reference these are three 1x1 convolutions, two adds; on the vendor BLAS three GEMMs of 7-20 us each (launch- and
// latency-bound: 0.1-0.5 GFLOP) plus two element-wise passes -- 67 + 50 us per classification forward.  Here a workgroup
// owns a tile of 32 rows and chains v_mfma_f32_32x32x2_f32 the way sa_cell does: the products are formed TRANSPOSED
// (T^T = Ws^T S^T), so that a D tile holds, per lane, channels kappa(i, h) of the lane's row -- exactly the B operand of the
// next product O^T = Wagg^T V^T.  V = A + relu(T) + relu(U) passes through LDS once (channel-major, stride 33: conflict-free
// both ways) because the four waves of the workgroup split the channel blocks of both stages; A enters and O leaves through
// the same LDS tile so that global memory only sees coalesced 128-byte rows.  Weights stream from L2 (coalesced rows).
// =============================================================================================
// A-operand stream of one 32-channel output block: weight rows k (lanes 0-31) / k+1 (lanes 32-63), 16 k-steps per chunk,
// the next chunk requested while the current one feeds the MFMAs.
constexpr int TAIL_KS = 16;  // k-steps (pairs of input channels) per operand chunk
// PACKED: the matrix in operand order (pasnl_sa_tail_pack_weights): P[chunk][h][column][t] = W[32 chunk + 2 t + h][column], zero
// beyond kdim -- a lane's 16 words of a chunk are 64 contiguous bytes = four 16-byte loads instead of sixteen 4-byte ones (the
// weight loads were 5 of a launch's 31 us: EXPERIMENTS "Round 5")
template <bool PACKED>
struct TailW {
  const float* __restrict__ base;  // W + cbase + l32  (PACKED: the matrix itself)
  int C, kdim, h;
  int col;                         // PACKED: cbase + l32
  __device__ __forceinline__ void load(int k0, float (&a)[TAIL_KS]) const {
    if constexpr (PACKED) {
      const int chunk = min(k0 >> 5, ((kdim + 31) >> 5) - 1);  // (the request behind the last chunk re-reads it)
      const float4* p = reinterpret_cast<const float4*>(base + ((size_t)(chunk * 2 + h) * C + col) * TAIL_KS);
#pragma unroll
      for (int q = 0; q < TAIL_KS / 4; ++q) {
        const float4 v = p[q];
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < TAIL_KS; ++t) {
      // rows past kdim are CLAMPED, not skipped: the X tile is zero there, so any finite weight contributes nothing, and
      // unconditional loads need no branch each and let the waits count (a skipped load costs an exec test per load and
      // the waits in front of the products all became vmcnt(0))
      const int kk = min(k0 + 2 * t + h, kdim - 1);
      a[t] = base[(size_t)kk * C];
    }
  }
};

// acc += W[:, block]^T . X^T for X rows held in LDS as xs[k * 33 + row] (k-major, zero-padded to a multiple of 2 TAIL_KS
// channels).  Chunks of TAIL_KS unconditional MFMAs (zero X beyond kdim), the next chunk's weights in flight.
template <bool PACKED>
__device__ __forceinline__ f32x16 tail_product(const TailW<PACKED>& W, const float* __restrict__ xs, int l32, f32x16 acc) {
  // `a` feeds the products while `b` (the next chunk) is in flight, covered by TAIL_KS products (~1000 cycles: an L2 round
  // trip).  One loop body without an early exit -- an exit between a load and its use lets the compiler sink the load behind
  // the exit, right in front of its use -- and the rotation as register copies after the products: by then `b` has arrived.
  float a[TAIL_KS], b[TAIL_KS];
  W.load(0, a);
  for (int k0 = 0; k0 < W.kdim; k0 += 2 * TAIL_KS) {
    W.load(k0 + 2 * TAIL_KS, b);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < TAIL_KS / 4; ++g)
      if (k0 + 8 * g < W.kdim) {  // uniform: a narrow product (9 skip columns) does not pay for a whole chunk
#pragma unroll
        for (int t = 4 * g; t < 4 * g + 4; ++t)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], xs[(k0 + 2 * t + W.h) * 33 + l32], acc, 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TAIL_KS; ++t) a[t] = b[t];
  }
  return acc;
}

#ifdef PASNL_TUNING  // timing ablations of sa_tail (tools/tail_probe.py): bit 0 no tile loads, 1 no stage 1, 2 no stage 2, 3 no stores
#define TAIL_ABL_PARAM , int abl
#define TAIL_ABL(bit) (abl & (1 << (bit)))
#else
#define TAIL_ABL_PARAM
#define TAIL_ABL(bit) 0
#endif
template <int NW, bool PACKED>  // waves per workgroup = 32-channel output blocks in flight (one per wave): C <= 32 NW
__global__ __launch_bounds__(NW * 64) void sa_tail_kernel(long rows, int w, int cb, int C, const float* __restrict__ A,
                                                         const float* __restrict__ S, const float* __restrict__ N,
                                                         const float* __restrict__ Ws, const float* __restrict__ bs,
                                                         const float* __restrict__ Wb, const float* __restrict__ bb,
                                                         const float* __restrict__ Wagg, const float* __restrict__ bagg,
                                                         float* __restrict__ out, const float* __restrict__ xyz3,
                                                         float* __restrict__ out_cat, const float* __restrict__ res TAIL_ABL_PARAM) {
  constexpr int RPW = 32 / NW;  // tile rows a wave stages / writes back
  extern __shared__ float lds[];
  float* vt = lds;                         // [C][33]      V^T, then O^T
  const int wp = (w + 2 * TAIL_KS - 1) & ~(2 * TAIL_KS - 1), cbp = (cb + 2 * TAIL_KS - 1) & ~(2 * TAIL_KS - 1);
  float* st = vt + (size_t)C * 33;         // [wp][33]     S^T tile
  float* nt_ = st + (size_t)wp * 33;       // [cbp][33]    N^T tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
  const long row0 = (long)blockIdx.x * 32;
  const int nblk = C >> 5;
  // ---- tiles into LDS, transposed (coalesced rows in, stride-33 columns out: conflict-free).  The three tiles are one list
  // of 64-column segments, STAGE_DEPTH of them requested before the first is written: a workgroup's segments are one memory
  // round trip, not one each (the grid is a single wave of workgroups -- its time IS a workgroup's chain of round trips).
  // Everything unconditional, so that the loads cannot be sunk in front of their use: rows, columns and the segment index are
  // clamped (a clamped lane / segment re-writes the value its twin writes), padding columns are zeroed by a select.
  float cat_xyz[RPW];  // lanes 0..3 of a wave: [0 | xyz] of its rows for out_cat, requested now, stored at the very end
  if (out_cat) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const long row = min(row0 + wave * RPW + i, rows - 1);
      cat_xyz[i] = xyz3[row * 3 + max(min(lane, 3) - 1, 0)];
    }
  }
  // the residual the caller adds to the layer's output (pointasnl_sem_seg_res.py:37,42,47,52), requested with the tiles:
  // the rows this wave writes at the very end, 64 columns per segment (C <= 32 NW: at most NW / 2 segments)
  float resv[NW / 2][RPW];
  if (res) {
#pragma unroll
    for (int sgm = 0; sgm < NW / 2; ++sgm)
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const long row = min(row0 + wave * RPW + i, rows - 1);
        resv[sgm][i] = res[row * C + min(sgm * 64 + lane, C - 1)];
      }
  }
  if (!TAIL_ABL(0)) {
    constexpr int STAGE_DEPTH = 4;
    const int nA = (C + 63) >> 6, nS = (wp + 63) >> 6, nN = N ? (cbp + 63) >> 6 : 0, total = nA + nS + nN;
    for (int it0 = 0; it0 < total; it0 += STAGE_DEPTH) {
      float v[STAGE_DEPTH][RPW];
      int cl[STAGE_DEPTH], width[STAGE_DEPTH];
      float* dst[STAGE_DEPTH];
#pragma unroll
      for (int d = 0; d < STAGE_DEPTH; ++d) {
        const int it = min(it0 + d, total - 1);
        const bool isA = it < nA, isS = !isA && it < nA + nS;
        const float* __restrict__ src = isA ? A : (isS ? S : N);
        const int padded = isA ? C : (isS ? wp : cbp), c0 = (isA ? it : (isS ? it - nA : it - nA - nS)) * 64;
        width[d] = isA ? C : (isS ? w : cb);
        dst[d] = isA ? vt : (isS ? st : nt_);
        cl[d] = min(c0 + lane, padded - 1);
        const int cc = min(cl[d], width[d] - 1);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const long row = min(row0 + wave * RPW + i, rows - 1);
          v[d][i] = src[row * width[d] + cc];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < STAGE_DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < RPW; ++i) dst[d][cl[d] * 33 + wave * RPW + i] = cl[d] < width[d] ? v[d][i] : 0.f;
    }
  }
  __syncthreads();
  // ---- stage 1: V^T += relu(Ws^T S^T + bs) + relu(Wb^T N^T + bb); wave = channel block
  const int cbase = wave * 32;
  const bool mine = wave < nblk;
  if (mine && !TAIL_ABL(1)) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bs[cbase + kappa(i, h)];
    acc = tail_product(PACKED ? TailW<PACKED>{Ws, C, w, h, cbase + l32} : TailW<PACKED>{Ws + cbase + l32, C, w, h, 0}, st, l32, acc);
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i], 0.f);
    if (N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = bb[cbase + kappa(i, h)];
      acc = tail_product(PACKED ? TailW<PACKED>{Wb, C, cb, h, cbase + l32} : TailW<PACKED>{Wb + cbase + l32, C, cb, h, 0}, nt_, l32, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] += fmaxf(acc[i], 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float* cell = vt + (cbase + kappa(i, h)) * 33 + l32;
      *cell = *cell + v[i];
    }
  }
  __syncthreads();
  // ---- stage 2: O^T = relu(Wagg^T V^T + bagg); k-step t contracts channels 2t (lanes 0-31) and 2t+1 (lanes 32-63)
  f32x16 o;
  if (mine) {
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = bagg[cbase + kappa(i, h)];
    if (!TAIL_ABL(2))
      o = tail_product(PACKED ? TailW<PACKED>{Wagg, C, C, h, cbase + l32} : TailW<PACKED>{Wagg + cbase + l32, C, C, h, 0}, vt, l32, o);
  }
  __syncthreads();  // every wave has read V^T: the tile becomes O^T
  if (mine) {
#pragma unroll
    for (int i = 0; i < 16; ++i) vt[(cbase + kappa(i, h)) * 33 + l32] = fmaxf(o[i], 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int sgm = 0; sgm < NW / 2; ++sgm) {
    const int c0 = sgm * 64;
    if (c0 >= C) break;
    const int c = c0 + lane;
    float v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) v[i] = c < C ? vt[c * 33 + wave * RPW + i] : 0.f;
    if (res) {
#pragma unroll
      for (int i = 0; i < RPW; ++i) v[i] = v[i] + resv[sgm][i];
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const long row = row0 + wave * RPW + i;
      if (row < rows && c < C && !(TAIL_ABL(3) && v[i] != 12345.f)) out[row * C + c] = v[i];
    }
    if (out_cat) {  // the same rows again as [0 | xyz | O] (C + 4 wide, 16-byte aligned): the next module's concat for free
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const long row = row0 + wave * RPW + i;
        if (row < rows && c < C) out_cat[row * (C + 4) + 4 + c] = v[i];
        if (c0 == 0 && lane < 4 && row < rows) out_cat[row * (C + 4) + lane] = lane ? cat_xyz[i] : 0.f;
      }
    }
  }
}
}  // namespace pasnl

#ifdef PASNL_TUNING
#define TAIL_ABL_ARG , (tune_env("PASNL_TAIL_ABL") ? atoi(tune_env("PASNL_TAIL_ABL")) : 0)
#else
#define TAIL_ABL_ARG
#endif
static int sa_tail_entry(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                         const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                         const float* bagg, float* out, const float* xyz3, float* out_cat, pasnl_stream_t stream,
                         const float* residual = nullptr, bool packed = false) {
  PASNL_REQUIRE(rows >= 0 && w > 0 && cb >= 0 && c > 0, PASNL_EINVAL);
  PASNL_REQUIRE(c % 32 == 0 && c <= 512, PASNL_EUNSUPPORTED);
  if (rows == 0) return PASNL_OK;
  PASNL_REQUIRE(after && skip_max && ws && bs && wagg && bagg && out, PASNL_ENULL);
  PASNL_REQUIRE(cb == 0 || (att && wb && bb), PASNL_ENULL);
  const size_t lds = ((size_t)c + ((w + 31) & ~31) + ((cb + 31) & ~31)) * 33 * sizeof(float);  // tiles padded to 2 TAIL_KS rows
  PASNL_REQUIRE(lds <= 160 * 1024, PASNL_EUNSUPPORTED);
  const int nw = c <= 128 ? 4 : (c <= 256 ? 8 : 16);  // one wave per 32-channel block (>= 4 waves stage the tiles)
  if (packed)
    PASNL_REQUIRE((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(wagg) | (cb ? reinterpret_cast<uintptr_t>(wb) : 0)) % 16 == 0,
                  PASNL_EUNSUPPORTED);
  auto kern = packed ? (nw == 4 ? pasnl::sa_tail_kernel<4, true> : (nw == 8 ? pasnl::sa_tail_kernel<8, true> : pasnl::sa_tail_kernel<16, true>))
                     : (nw == 4 ? pasnl::sa_tail_kernel<4, false> : (nw == 8 ? pasnl::sa_tail_kernel<8, false> : pasnl::sa_tail_kernel<16, false>));
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 31) / 32)), dim3(nw * 64), lds, pasnl_hip_stream(stream), (long)rows, w, cb, c,
                     after, skip_max, cb ? att : nullptr, ws, bs, wb, bb, wagg, bagg, out, xyz3, out_cat, residual TAIL_ABL_ARG);
  return pasnl_launch_status();
}

extern "C" int pasnl_sa_tail(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                             const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                             const float* bagg, float* out, pasnl_stream_t stream) {
  return sa_tail_entry(rows, w, cb, c, after, skip_max, att, ws, bs, wb, bb, wagg, bagg, out, nullptr, nullptr, stream);
}

extern "C" int pasnl_sa_tail_res(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                                 const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                                 const float* bagg, const float* residual, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows == 0 || residual, PASNL_ENULL);
  return sa_tail_entry(rows, w, cb, c, after, skip_max, att, ws, bs, wb, bb, wagg, bagg, out, nullptr, nullptr, stream, residual);
}

namespace pasnl {
// W (K, C) row-major -> P[chunk][h][column][t] = W[32 chunk + 2 t + h][column] (zero beyond K): one thread per 16-byte piece
__global__ __launch_bounds__(256) void sa_tail_pack_kernel(int K, int C, const float* __restrict__ W, float* __restrict__ P) {
  const long total = (long)((K + 31) >> 5) * 2 * C * (TAIL_KS / 4);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int q = (int)(e & (TAIL_KS / 4 - 1));
    const long lane = e / (TAIL_KS / 4);  // (chunk * 2 + h) * C + column
    const int col = (int)(lane % C);
    const long ch = lane / C;
    const int h = (int)(ch & 1), chunk = (int)(ch >> 1);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 32 * chunk + 2 * (4 * q + j) + h;
      v[j] = k < K ? W[(size_t)k * C + col] : 0.f;
    }
    reinterpret_cast<float4*>(P)[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace pasnl

extern "C" size_t pasnl_sa_tail_packed_weights_bytes(int k, int c) {
  if (k <= 0 || c <= 0) return 0;
  return (size_t)((k + 31) & ~31) * c * sizeof(float);
}

extern "C" int pasnl_sa_tail_pack_weights(int k, int c, const float* w, float* packed, pasnl_stream_t stream) {
  PASNL_REQUIRE(k > 0 && c > 0, PASNL_EINVAL);
  PASNL_REQUIRE(w && packed, PASNL_ENULL);
  PASNL_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0, PASNL_EUNSUPPORTED);
  const long pieces = (long)((k + 31) >> 5) * 2 * c * 4;
  const long g = (pieces + 255) / 256;
  hipLaunchKernelGGL(pasnl::sa_tail_pack_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, pasnl_hip_stream(stream), k, c, w, packed);
  return pasnl_launch_status();
}

extern "C" int pasnl_sa_tail_packed(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                                    const float* ws_packed, const float* bs, const float* wb_packed, const float* bb,
                                    const float* wagg_packed, const float* bagg, const float* residual, const float* new_xyz,
                                    float* out_cat, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows == 0 || !out_cat || new_xyz, PASNL_ENULL);
  return sa_tail_entry(rows, w, cb, c, after, skip_max, att, ws_packed, bs, wb_packed, bb, wagg_packed, bagg, out, out_cat ? new_xyz : nullptr,
                       out_cat, stream, residual, true);
}

extern "C" int pasnl_sa_tail_cat(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                                 const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                                 const float* bagg, float* out, const float* new_xyz, float* out_cat, pasnl_stream_t stream) {
  PASNL_REQUIRE(rows == 0 || (new_xyz && out_cat), PASNL_ENULL);
  return sa_tail_entry(rows, w, cb, c, after, skip_max, att, ws, bs, wb, bb, wagg, bagg, out, new_xyz, out_cat, stream);
}

namespace pasnl {
// ---------------------------------------------------------------------------------------------
// The decoder cell again, with the output in a TILED order of the (3+c)*32 values of a point (pasnl_decode_cell_tiled).
// decode_cell_kernel is bound by how many vector-memory INSTRUCTIONS a compute unit can issue (152 per point: 72 loads,
// 80 stores of 256 bytes each -- 3.0-3.5 TB/s of the ~7.9 TB/s a plain fill reaches), not by bytes.  The only consumer of
// its output is the `decode_after_conv` GEMM, which contracts ALL (3+c)*32 values of a point: any fixed permutation of them
// is as good as the reference's (channel, j) order once the GEMM's weight rows are permuted the same way.  So:
//   * the product is formed transposed (the same two operand registers, swapped): a lane holds, for ONE channel, j = 8g + 4h
//     + (0..3), g = 0..3 -- four float4;
//   * a tile of 32 channels x 32 j is stored as  [g][h][channel-in-tile][4]  (1024 floats): store g of a wave is 64 lanes x 16
//     bytes = 1 KiB CONTIGUOUS -- 4 stores per tile instead of 16;
//   * with c % 128 == 0 a lane reads FOUR consecutive features of a neighbour (one 16-byte load) and spends them on four
//     tiles: tile e of a 128-feature group holds features 4 m + e, m = 0..31 -- 8 loads per group instead of 32.
// Layout of a point's (3+c)*32 floats (the Python mirror builds the matching row permutation of the weights):
//   [0, 96)                                   the three coordinate channels, (channel, j) order as before
//   96 + T*1024 + (2g+h)*128 + 4m + i         tile T = Gq*V + e (V = 4 or 1):  channel 3 + 32 V Gq + V m + e,  j = 8g + 4h + i
// ---------------------------------------------------------------------------------------------
template <int K, int V, bool NT>  // NT: streaming (non-temporal) output stores
__global__ __launch_bounds__(256) void decode_cell_tiled_kernel(long points, int n, int c, const float* __restrict__ xyz,
                                                               const float* __restrict__ feature, const int* __restrict__ idx,
                                                               const float* __restrict__ ww, const float* __restrict__ bw,
                                                               float* __restrict__ out) {
  constexpr int T = K / 2;  // MFMA steps: step t contracts neighbours 2t (lanes 0..31) and 2t+1 (lanes 32..63)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, ql = lane & 31;
  const int w = 3 + c, ngroup = c / (32 * V);
  const float w0 = ww[ql], w1 = ww[32 + ql], w2 = ww[64 + ql], bj = bw[ql];
  const int nclouds = (int)(points / n);
  const bool xcd_map = (gridDim.x % 8 == 0) && nclouds >= 8;
  const int xcd = blockIdx.x & 7;
  const long mine = xcd_map ? (long)((nclouds - xcd + 7) >> 3) * n : points;
  const long first = xcd_map ? (long)(blockIdx.x >> 3) * 4 + wave : (long)blockIdx.x * 4 + wave;
  const long step = xcd_map ? (long)(gridDim.x >> 3) * 4 : (long)gridDim.x * 4;
  typedef float fvec __attribute__((ext_vector_type(V)));
  for (long li = first; li < mine; li += step) {
    long p = li, bi;
    if (xcd_map) {
      const int cl = (int)(li / n);
      bi = xcd + 8 * cl;
      p = bi * n + (li - (long)cl * n);
    } else {
      bi = p / n;
    }
    const float cx = xyz[p * 3], cy = xyz[p * 3 + 1], cz = xyz[p * 3 + 2];
    float G[T], qc[T];
    const float* frow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int is = idx[p * K + 2 * t + h];
      const float* q = xyz + ((size_t)bi * n + is) * 3;
      const float qx = q[0], qy = q[1], qz = q[2];
      G[t] = fmaxf(__builtin_fmaf(qz - cz, w2, __builtin_fmaf(qy - cy, w1, (qx - cx) * w0)) + bj, 0.f);
      qc[t] = ql == 0 ? qx : (ql == 1 ? qy : (ql == 2 ? qz : 0.f));
      frow[t] = feature + ((size_t)bi * n + is) * (size_t)c + V * ql;  // this lane's V features of a group
    }
    float* o = out + (size_t)p * w * 32;
    fvec a[T], an[T];
#pragma unroll
    for (int t = 0; t < T; ++t) a[t] = *reinterpret_cast<const fvec*>(frow[t]);
    {  // coordinate channels (rows 0..2 of an MFMA tile), the reference's order
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[t], G[t], acc, 0, 0, 0);
      if (h == 0) {  // kappa(r, 0) = r for r < 3
        o[ql] = acc[0];
        o[32 + ql] = acc[1];
        o[64 + ql] = acc[2];
      }
    }
    for (int gq = 0; gq < ngroup; ++gq) {
      // next group's features in flight during this group's tiles (none behind the last group: a dummy request there
      // would be 8 wasted loads per point -- and the wait for this group's operands would wait for them too)
      if (gq + 1 < ngroup) {
        const int gn = (gq + 1) * 32 * V;
#pragma unroll
        for (int t = 0; t < T; ++t) an[t] = *reinterpret_cast<const fvec*>(frow[t] + gn);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          float fv;
          fv = a[t][e];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(G[t], fv, acc, 0, 0, 0);  // rows = j, columns = this tile's 32 channels
        }
        float* ot = o + 96 + (size_t)(gq * V + e) * 1024 + h * 128 + ql * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
        {
          // NT = streaming stores: the output is written once and read by the next kernel from HBM anyway; without the hint
          // it washes the neighbours' feature table (4 MiB per cloud at ScanNet fa_layer4 = the whole L2 of an XCD) out of
          // L2 (650 -> 490 us there; tables that fit beside the output stream anyway lose a little, so the launcher decides)
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 v4 = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          if constexpr (NT) __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(ot + g * 256));
          else *reinterpret_cast<f4*>(ot + g * 256) = v4;
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) a[t] = an[t];
    }
  }
}
}  // namespace pasnl

extern "C" int pasnl_decode_cell_tiled(int b, int n, int c, int k, const float* xyz, const float* feature, const int* idx,
                                       const float* ww, const float* bw, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k == 16 && c % 32 == 0, PASNL_EUNSUPPORTED);
  const long points = (long)b * n;
  if (points == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && feature && idx && ww && bw && out, PASNL_ENULL);
  PASNL_REQUIRE(points * k < (1L << 40), PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(reinterpret_cast<uintptr_t>(out) % 16 == 0, PASNL_EUNSUPPORTED);
  const bool v4 = c % 128 == 0 && reinterpret_cast<uintptr_t>(feature) % 16 == 0;
  hipStream_t st = pasnl_hip_stream(stream);
  const bool nt = (size_t)n * c * sizeof(float) >= (1u << 20);  // a cloud's feature table is worth protecting in L2
  const void* kern = nullptr;
#define PASNL_DT(V, NT) reinterpret_cast<const void*>(pasnl::decode_cell_tiled_kernel<16, V, NT>)
  kern = v4 ? (nt ? PASNL_DT(4, true) : PASNL_DT(4, false)) : (nt ? PASNL_DT(1, true) : PASNL_DT(1, false));
#undef PASNL_DT
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  long wgs = (points + 3) / 4;
  const long cap = 256L * per_cu;
  unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
  void* args[] = {(void*)&points, (void*)&n, (void*)&c, (void*)&xyz, (void*)&feature, (void*)&idx, (void*)&ww, (void*)&bw, (void*)&out};
  if (hipLaunchKernel(kern, dim3(grid), dim3(256), args, 0, st) != hipSuccess) return PASNL_ELAUNCH;
  return pasnl_launch_status();
}

/* 1 when pasnl_decode_cell_tiled spends one 16-byte load on four tiles for this width and feature pointer (V = 4), else 0
 * (V = 1): the tiled order depends on it. */
extern "C" int pasnl_decode_cell_tiled_v4(int c, const float* feature) {
  return (c % 128 == 0 && reinterpret_cast<uintptr_t>(feature) % 16 == 0) ? 1 : 0;
}

extern "C" int pasnl_decode_cell(int b, int n, int c, int k, const float* xyz, const float* feature, const int* idx,
                                 const float* ww, const float* bw, float* out, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && c > 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k == 16 || k == 32, PASNL_EUNSUPPORTED);
  const long points = (long)b * n;
  if (points == 0) return PASNL_OK;
  PASNL_REQUIRE(xyz && feature && idx && ww && bw && out, PASNL_ENULL);
  PASNL_REQUIRE(points * k < (1L << 40), PASNL_EUNSUPPORTED);
  hipStream_t st = pasnl_hip_stream(stream);
  // persistent workgroups, as many as are resident at once (a multiple of 8 keeps the XCD map on)
  const void* kern = k == 16 ? reinterpret_cast<const void*>(decode_cell_kernel<16>) : reinterpret_cast<const void*>(decode_cell_kernel<32>);
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  long wgs = (points + 3) / 4;
  const long cap = 256L * per_cu;
  unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
  if (k == 16) hipLaunchKernelGGL(decode_cell_kernel<16>, dim3(grid), dim3(256), 0, st, points, n, c, xyz, feature, idx, ww, bw, out);
  else hipLaunchKernelGGL(decode_cell_kernel<32>, dim3(grid), dim3(256), 0, st, points, n, c, xyz, feature, idx, ww, bw, out);
  return pasnl_launch_status();
}
