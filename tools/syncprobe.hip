// Diagnostic: cost of the synchronisation primitives an FPS round is made of, on gfx950 (clock64 = s_memtime ticks).
//   hipcc --offload-arch=gfx950 -O3 tools/syncprobe.hip -o tools/syncprobe && tools/syncprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ int wave_max_dpp(int x) {
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(x));
  return x;
}
template <int MODE>
__global__ void probe(int* out, long long* t, int iters) {
  __shared__ int slot[2][16];
  __shared__ unsigned long long best[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
  int v = tid * 7 + 3;
  if (tid < 4) best[tid] = 0;
  __syncthreads();
  long long c0 = clock64();
  for (int j = 0; j < iters; ++j) {
    if (MODE == 0) {  // barrier only
      __syncthreads();
    } else if (MODE == 1) {  // lane-0 LDS write, barrier, broadcast LDS read
      if (lane == 0) slot[j & 1][wave] = v;
      __syncthreads();
      v += slot[j & 1][(wave + 1) % W];
    } else if (MODE == 2) {  // wave max by DPP + readlane
      v = __builtin_amdgcn_readlane(wave_max_dpp(v), 63) + lane;
    } else if (MODE == 3) {  // ds_max_u64 by lane 0, barrier, read
      if (lane == 0) atomicMax(&best[j & 3], (unsigned long long)(unsigned)v << 32 | j);
      if (tid == 0) best[(j + 1) & 3] = 0;
      __syncthreads();
      v += (int)(best[j & 3] >> 32) & 1;
    } else if (MODE == 4) {  // two dependent LDS reads (slot, then indexed by it)
      v += slot[0][slot[1][v & 15] & 15];
    } else if (MODE == 5) {  // the full exchange of fps_kernel: DPP max, readlane, ballot, readlane, write, barrier, read, row DPP, readlane x2, LDS read
      int m = __builtin_amdgcn_readlane(wave_max_dpp(v), 63);
      unsigned long long tie = __ballot(v == m);
      int old = __builtin_amdgcn_readlane(v, (int)__builtin_ctzll(tie));
      if (lane == 0) slot[j & 1][wave] = old;
      __syncthreads();
      int sv = lane < W ? slot[j & 1][lane] : (int)0x80000000;
      int g = __builtin_amdgcn_readlane(wave_max_dpp(sv), 63);
      v = slot[0][g & 15] + lane + j;
    }
  }
  long long c1 = clock64();
  if (tid == 0) t[blockIdx.x] = c1 - c0;
  out[blockIdx.x * blockDim.x + tid] = v;
}
template <int MODE> void run(const char* name, int* out, long long* t, int threads, double tick_ns) {
  int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<MODE><<<64, threads>>>(out, t, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE><<<64, threads>>>(out, t, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-58s waves %2d: %7.1f ns per round (%.1f clock64 ticks)\n", name, threads / 64, ms * 1e6 / iters, (double)h / iters);
}
// Cross-WORKGROUP exchange through L2 (two workgroups per cloud exchanging one 64-bit word per round): workgroups
// 2 i and 2 i + 1 each post a 64-bit key with a device-scope atomic max and wait until both have posted (a generation counter
// bumped with a device-scope atomic add, polled with s_sleep 1 between loads), then read the winner.  One slot pair per parity.
__global__ void pair_probe(unsigned long long* slots, unsigned int* gens, long long* t, int* out, int iters) {
  const int tid = threadIdx.x, pair = blockIdx.x >> 1;
  unsigned long long* slot = slots + pair * 4;
  unsigned int* gen = gens + pair * 4;
  int v = tid * 7 + 3 + blockIdx.x;
  __syncthreads();
  long long c0 = clock64();
  for (int j = 0; j < iters; ++j) {
    if (tid == 0) {
      __hip_atomic_fetch_max(&slot[j & 1], ((unsigned long long)(unsigned)v << 32) | (unsigned)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&gen[j & 1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = 2u * (unsigned)(j / 2 + 1);
      while (__hip_atomic_load(&gen[j & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      v += (int)(__hip_atomic_load(&slot[j & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) & 3;
    }
    __syncthreads();  // the workgroup waits for its lane 0 (the round's local barrier)
  }
  long long c1 = clock64();
  if (tid == 0) t[blockIdx.x] = c1 - c0;
  out[blockIdx.x * blockDim.x + tid] = v;
}
static void run_pair(int* out, long long* t, int threads) {
  unsigned long long* slots; unsigned int* gens;
  (void)hipMalloc(&slots, 64 * 4 * 8); (void)hipMalloc(&gens, 64 * 4 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(slots, 0, 64 * 4 * 8); (void)hipMemset(gens, 0, 64 * 4 * 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    pair_probe<<<16, threads>>>(slots, gens, t, out, iters);  // 8 pairs: all sixteen workgroups are resident at once
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s waves %2d: %7.1f ns per round\n", "cross-workgroup: atomic max + generation counter through L2", threads / 64, ms * 1e6 / iters);
}
int main() {
  int* out; long long* t;
  (void)hipMalloc(&out, 64 * 1024 * 4); (void)hipMalloc(&t, 64 * 8);
  for (int threads : {64, 256, 1024}) {
    run<0>("s_barrier", out, t, threads, 0);
    run<1>("lane-0 ds_write + barrier + broadcast ds_read", out, t, threads, 0);
    run<2>("wave max: 6 DPP steps + readlane", out, t, threads, 0);
    run<3>("ds_max_u64 + barrier + ds_read_b64", out, t, threads, 0);
    run<4>("two dependent ds_reads", out, t, threads, 0);
    run<5>("full fps_kernel exchange", out, t, threads, 0);
    run_pair(out, t, threads);
  }
  return 0;
}
