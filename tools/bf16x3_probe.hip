// Exploratory: fp32-equivalent products on the bf16 matrix pipe of gfx950 by splitting every fp32 operand
// into three bf16 terms (hi + mid + lo = 24 mantissa bits) and keeping the six products whose weight is >= 2^-16 of the
// leading one:  a b ~= ah bh + (ah bm + am bh) + (ah bl + al bh + am bm),  each bf16 x bf16 product exact in fp32, fp32
// accumulate.  The bf16 MFMA has 16x the rate of the fp32 MFMA, so six of them should cost 6/16 of one fp32 product.
// This probe measures, on an MI355X,
//   1. sustained matrix rates, operands in registers, four independent accumulators per wave, 4 waves per SIMD:
//      v_mfma_f32_32x32x2_f32 against the six-product group of v_mfma_f32_32x32x16_bf16 (counted as ONE fp32 product of the
//      same 32x32x16 volume), with and without the on-the-fly split of one operand (the activations; weights split once);
//   2. accuracy on real-valued data: C = A B for A (256 x K), B (K x 128), K = 4192 (the decoder's after-conv contraction)
//      and K = 256, entries ~ N(0,1) and a ReLU-like half-zero variant, against fp64: worst |err| / (|A| |B|)_ij and worst
//      relative error, for (a) an fp32 fmaf chain in K order (what v_mfma_f32_*_f32 computes), (b) the six-product scheme,
//      (c) the three-product scheme (ah bh + ah bm + am bh) for comparison.
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o tools/bf16x3_probe && tools/bf16x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;  // round to nearest even
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// MODE 0: fp32 MFMA 32x32x2, 8 per "unit" (= one 32x32x16 volume).  MODE 1: six bf16 MFMAs per unit, operands pre-split.
// MODE 2: as 1, plus the split of the A operand (8 values per lane and unit) on the vector ALU.
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float xf[8];
  for (int i = 0; i < 8; ++i) xf[i] = 1e-3f * (float)(lane * 8 + i) + 0.37f;
  bf16x8 ah, am, al, bh, bm, bl;
  for (int i = 0; i < 8; ++i) {
    __bf16 h, m, l;
    split3(xf[i], h, m, l);
    ah[i] = h; am[i] = m; al[i] = l;
    split3(xf[7 - i] * 0.5f, h, m, l);
    bh[i] = h; bm[i] = m; bl[i] = l;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[s], xf[(s + a) & 7], acc[a], 0, 0, 0);
      } else {
        if (MODE == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __bf16 h, m, l;
            split3(xf[i] + acc[a][i] * 1e-30f, h, m, l);  // depends on live data: the split cannot be hoisted
            ah[i] = h; am[i] = m; al[i] = l;
          }
        }
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[a], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run_rate(float* out, const char* name) {
  const int iters = 2000, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  rate_kernel<MODE><<<blocks, 256>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  rate_kernel<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double units = (double)blocks * 4 /*waves*/ * iters * 4 /*accumulators*/;
  const double tf = units * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("%-58s %8.3f ms  %8.1f fp32-equivalent TFLOP/s\n", name, ms, tf);
  return tf;
}

// ---- accuracy: plain one-thread-per-output kernels (clarity over speed), K in ascending order like an MFMA chain
__global__ void acc_kernel(int M, int K, int N, const float* A, const float* B, float* c32, float* c6, float* c3) {
  const int i = blockIdx.x, j = threadIdx.x;
  if (i >= M || j >= N) return;
  float s32 = 0.f, s6 = 0.f, s3 = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = A[(size_t)i * K + k], b = B[(size_t)k * N + j];
    s32 = fmaf(a, b, s32);
    __bf16 ah, am, al, bh, bm, bl;
    split3(a, ah, am, al);
    split3(b, bh, bm, bl);
    const float fah = (float)ah, fam = (float)am, fal = (float)al, fbh = (float)bh, fbm = (float)bm, fbl = (float)bl;
    // every product of two bf16 values is exact in fp32; the accumulation rounds once per product like the MFMA's adder
    s6 += fah * fbh; s6 += fah * fbm; s6 += fam * fbh; s6 += fah * fbl; s6 += fal * fbh; s6 += fam * fbm;
    s3 += fah * fbh; s3 += fah * fbm; s3 += fam * fbh;
  }
  c32[(size_t)i * N + j] = s32; c6[(size_t)i * N + j] = s6; c3[(size_t)i * N + j] = s3;
}

static float gauss(unsigned& st) {
  float s = 0.f;
  for (int i = 0; i < 12; ++i) { st = st * 1664525u + 1013904223u; s += (float)(st >> 8) * (1.0f / 16777216.0f); }
  return s - 6.0f;
}

static void accuracy(int M, int K, int N, bool relu_like) {
  std::vector<float> A((size_t)M * K), B((size_t)K * N);
  unsigned st = 12345u + K;
  for (auto& v : A) { v = gauss(st); if (relu_like && v < 0.f) v = 0.f; }
  for (auto& v : B) v = gauss(st) * 0.05f;
  float *dA, *dB, *d32, *d6, *d3;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4);
  (void)hipMalloc(&d32, (size_t)M * N * 4); (void)hipMalloc(&d6, (size_t)M * N * 4); (void)hipMalloc(&d3, (size_t)M * N * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  acc_kernel<<<M, N>>>(M, K, N, dA, dB, d32, d6, d3);
  std::vector<float> c32((size_t)M * N), c6((size_t)M * N), c3((size_t)M * N);
  (void)hipMemcpy(c32.data(), d32, c32.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c6.data(), d6, c6.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c3.data(), d3, c3.size() * 4, hipMemcpyDeviceToHost);
  double w32 = 0, w6 = 0, w3 = 0, r32 = 0, r6 = 0, r3 = 0, scale = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)k * N + j]; ref += p; mag += fabs(p); }
      const size_t o = (size_t)i * N + j;
      w32 = fmax(w32, fabs(c32[o] - ref) / mag); w6 = fmax(w6, fabs(c6[o] - ref) / mag); w3 = fmax(w3, fabs(c3[o] - ref) / mag);
      scale = fmax(scale, fabs(ref));
      r32 = fmax(r32, fabs(c32[o] - ref)); r6 = fmax(r6, fabs(c6[o] - ref)); r3 = fmax(r3, fabs(c3[o] - ref));
    }
  printf("K = %5d %-9s worst |err| / sum|a b|: fp32 chain %.2e  six bf16 products %.2e  three %.2e ;  worst |err| / max|C|: %.2e  %.2e  %.2e\n",
         K, relu_like ? "(relu-ed)" : "", w32, w6, w3, r32 / scale, r6 / scale, r3 / scale);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(d32); (void)hipFree(d6); (void)hipFree(d3);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  const double f32 = run_rate<0>(out, "v_mfma_f32_32x32x2_f32 (8 per 32x32x16 volume)");
  const double b6 = run_rate<1>(out, "6 x v_mfma_f32_32x32x16_bf16, operands pre-split");
  const double b6s = run_rate<2>(out, "6 x bf16 MFMA + the A operand split on the vector ALU");
  printf("ratio to the fp32 MFMA: pre-split %.2fx, with the on-the-fly split %.2fx\n", b6 / f32, b6s / f32);
  accuracy(256, 4192, 128, false);
  accuracy(256, 4192, 128, true);
  accuracy(256, 256, 128, false);
  accuracy(64, 16480, 128, true);
  return 0;
}
