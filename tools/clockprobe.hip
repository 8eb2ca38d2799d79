// Diagnostic: per-wave VALU issue behaviour on gfx950 (one wave per SIMD), measured with clock64().
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void chain(float* out, long long* t, int iters) {
  float a[8]; f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-9f + i; p[i] = f2{a[i], a[i] + 1}; }
  float b = 1.0000001f;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 16 dependent fma
#pragma unroll
      for (int u = 0; u < 16; ++u) a[0] = __builtin_fmaf(a[0], b, 1e-7f);
    } else if (MODE == 1) {  // 16 fma, 8 independent chains
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u & 7] = __builtin_fmaf(a[u & 7], b, 1e-7f);
    } else if (MODE == 2) {  // 16 packed mul, 8 independent chains
#pragma unroll
      for (int u = 0; u < 16; ++u) p[u & 7] = p[u & 7] * f2{b, b};
    } else if (MODE == 3) {  // 16 dependent packed mul
#pragma unroll
      for (int u = 0; u < 16; ++u) p[0] = p[0] * f2{b, b};
    } else if (MODE == 4) {  // 16 int max, 8 independent chains
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u & 7] = __int_as_float(max(__float_as_int(a[u & 7]), __float_as_int(a[(u + 1) & 7]) ^ u));
    }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out, long long* t, int threads) {
  int iters = 20000;
  chain<MODE><<<64, threads>>>(out, t, iters);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads/block %4d: %.2f cycles per instruction\n", name, threads, (double)h / (iters * 16.0));
}
int main() {
  float* out; long long* t;
  (void)hipMalloc(&out, 4096 * 1024 * 4); (void)hipMalloc(&t, 4096 * 16);
  for (int threads : {64, 256, 512, 1024}) {
    run<0>("dependent v_fma_f32", out, t, threads);
    run<1>("independent v_fma_f32 (8 chains)", out, t, threads);
    run<2>("independent v_pk_mul_f32 (8 chains)", out, t, threads);
    run<3>("dependent v_pk_mul_f32", out, t, threads);
    run<4>("independent v_max_i32 (8 chains)", out, t, threads);
  }
  return 0;
}
