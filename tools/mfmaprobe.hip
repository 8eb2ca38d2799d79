// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 under the operand patterns of the local-cell kernel.
//   mode 0: 4 independent accumulators, operands in registers          (issue-bound peak)
//   mode 1: 1 accumulator, 64-deep dependent chain, operands in registers
//   mode 2: 4 independent accumulators, A operand read from LDS one batch ahead (conv0 pattern)
//   mode 3: 1 accumulator chain, B operands read from LDS one batch of 16 ahead (conv1 pattern)
//   mode 4: as 0, with 4 independent plain VALU ops (v_max_f32) pinned behind every MFMA
//   mode 5: as 1 (chain), with 4 VALU ops behind every MFMA
//   mode 6: as 0, with a cluster of 16 VALU ops behind every 4th MFMA
//   mode 7: as 0, with 2 DPP ops (v_max_f32_dpp row_shr) behind every MFMA
//   mode 8: as 0, with 12 VALU ops behind every MFMA (48 issue cycles of a 64-cycle MFMA)
// Prints cycles per MFMA (clock64), TFLOP/s over the whole chip (HIP events) and the implied clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, bool RND = false>
__global__ void probe(float* out, long long* t, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
    unsigned hsh = (unsigned)i * 2654435761u + 12345u;   // RND: operands with random mantissas (toggle rate of real data)
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    lds[i] = RND ? (float)(int)(hsh >> 8) * (1.0f / 8388608.0f) - 1.0f : 1e-6f * i;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x[16], y[8];
  for (int i = 0; i < 16; ++i) x[i] = RND ? lds[(lane * 16 + i) & 8191] : 1e-3f * (lane + i);
  for (int i = 0; i < 8; ++i) y[i] = 1e-2f * (lane - i);
  const float* wp = lds + lane;
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int tt = 0; tt < 16; ++tt)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[tt], x[(tt + a) & 15], acc[a], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int tt = 0; tt < 64; ++tt) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[tt & 15], x[(tt + 3) & 15], acc[0], 0, 0, 0);
    } else if (MODE >= 4) {
      constexpr int NV = MODE == 8 ? 12 : MODE == 7 ? 2 : 4;
#pragma unroll
      for (int tt = 0; tt < 16; ++tt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int ai = MODE == 5 ? 0 : a;
          acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[tt], x[(tt + a) & 15], acc[ai], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (MODE != 6 || a == 3) {
#pragma unroll
            for (int v = 0; v < (MODE == 6 ? 16 : NV); ++v) {
              if (MODE == 7) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y[v & 7]));
              else asm volatile("v_max_f32 %0, %0, %1" : "+v"(y[v & 7]) : "v"(y[(v + 3) & 7]));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 2) {
      float wa[2][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wa[0][a] = wp[a * 64];
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
        if (tt + 1 < 16) {
#pragma unroll
          for (int a = 0; a < 4; ++a) wa[(tt + 1) & 1][a] = wp[(tt + 1) * 256 + a * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[tt & 1][a], x[tt], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      float wv[2][16];
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) wv[0][tt] = wp[tt * 64];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        if (blk + 1 < 4) {
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) wv[(blk + 1) & 1][tt] = wp[(blk + 1) * 1024 + tt * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[tt], wv[blk & 1][tt], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += y[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, bool RND = false> void run(const char* name, float* out, long long* t, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<MODE, RND><<<256, threads>>>(out, t, 10);
  (void)hipEventRecord(e0);
  probe<MODE, RND><<<256, threads>>>(out, t, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  double mf = (double)iters * 64.0;                       // MFMAs per wave
  double waves = 256.0 * threads / 64.0;
  double tf = mf * waves * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-52s %4d thr/WG  %6.1f clk/MFMA/wave  %7.3f ms  %6.1f TFLOP/s  clock64 rate %.0f MHz\n", name, threads,
         (double)h / mf, ms, tf, (double)h / (ms * 1e3));
}
int main() {
  float* out; long long* t;
  (void)hipMalloc(&out, 4096 * 1024 * 4); (void)hipMalloc(&t, 4096 * 16);
  for (int threads : {256, 512}) {
    for (int iters : {2000, 40000}) {
      run<0>("4 accumulators, register operands", out, t, threads, iters);
      run<1>("1 accumulator chain, register operands", out, t, threads, iters);
      run<2>("4 accumulators, A from LDS one batch ahead", out, t, threads, iters);
      run<3>("1 accumulator chain, B from LDS 16 ahead", out, t, threads, iters);
      if (iters == 40000) {
        run<2, true>("4 accumulators, A from LDS, RANDOM operand data", out, t, threads, iters);
        run<3, true>("1 chain, B from LDS 16 ahead, RANDOM operand data", out, t, threads, iters);
      }
      if (iters == 2000) {
        run<4>("4 accumulators + 4 VALU behind every MFMA", out, t, threads, iters);
        run<5>("1 chain + 4 VALU behind every MFMA", out, t, threads, iters);
        run<6>("4 accumulators + 16 VALU behind every 4th MFMA", out, t, threads, iters);
        run<7>("4 accumulators + 2 DPP behind every MFMA", out, t, threads, iters);
        run<8>("4 accumulators + 12 VALU behind every MFMA", out, t, threads, iters);
      }
    }
  }
  return 0;
}
