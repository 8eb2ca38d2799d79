// kNN / radius crop of a scan around ONE centre (second half): the search inside `crop_pc`,
// reference SemanticKITTI/semantic_kitti_dataset_grid.py:265-272 -- `search_tree.query(center, k=num_point+buffer)` or
// `search_tree.query_radius(center, r=in_radius)` on an sklearn KDTree built over the voxel-subsampled scan (~1e5 points,
// k ~ 13 000).  sklearn stores the float32 points as float64 and ranks by the reduced distance ((0+dx*dx)+dy*dy)+dz*dz
// evaluated in double: the same arithmetic is the KEY here (the bits of a non-negative double order like the number), so
// the selected SET is sklearn's whenever the k-th distance is not tied; a tie at the boundary goes to the LOWEST indices
// (sklearn: the tree's visit order).  The radius form keeps d2 <= r*r (inclusive, as sklearn).  Restated in oracle/.
//
// No tree: one centre per crop makes the search an exact k-SELECTION of n keys -- an MSD radix select on the 63 key bits,
// 13 bits per pass, every pass over all n keys of the crop (0.8 MB per 1e5 points, L2-resident) by as many workgroups as
// the scan has blocks of 2048 points:
//   init   zero the five per-pass histograms of every crop
//   pass p keys whose leading bits equal the prefix found so far count into an LDS histogram of the next 13 bits (pass 0
//          also computes and stores the keys); non-empty bins are added to the crop's global histogram p.  Which bin holds
//          the k-th key is found by EVERY workgroup of the next kernel from histogram p (8192 bins, 32 per thread and a
//          workgroup scan): no ticket, no fence, no spinning -- kernel boundaries are the only synchronisation.  A pass whose
//          boundary bin is taken whole ends the selection early (later passes return at once): on lidar scans the third.
//   count  per block: keys below the boundary, keys on it
//   write  selected = below the boundary, or on it and among its first r in index order; a block's output offset is the
//          sum of the counts of the blocks before it: indices come out ASCENDING, deterministically, optionally with d2.
// Eight launches, no host synchronisation, capturable; all state in the caller's workspace.
#include <math.h>
#include "common.hpp"

namespace pasnl {

constexpr int CR_THREADS = 256;
constexpr int CR_ITEMS = 8;                       // consecutive points per thread
constexpr int CR_BLOCK = CR_THREADS * CR_ITEMS;   // points per workgroup
constexpr int CR_BITS = 13;
constexpr int CR_BINS = 1 << CR_BITS;
constexpr int CR_PASSES = 5;                      // 13 + 13 + 13 + 13 + 11 = 63 bits (bit 63, the sign, is always clear)
__host__ __device__ constexpr int cr_shift(int p) { return p == 0 ? 50 : (p == 1 ? 37 : (p == 2 ? 24 : (p == 3 ? 11 : 0))); }
__host__ __device__ constexpr int cr_width(int p) { return p == 4 ? 11 : 13; }

// Selection state of one crop after pass p (state[p + 1]; state[0] is the start): keys with key >> shift < prefix are selected,
// and the first `r` (ascending index) of those with key >> shift == prefix.  done: the later passes have nothing to do.
struct CrState {
  unsigned long long prefix;
  int shift, r, done, pad;
};

struct CrWs {
  unsigned long long* keys;  // [b][n]
  unsigned* hist;            // [b][CR_PASSES][CR_BINS]
  CrState* state;            // [b][CR_PASSES + 1]
  int* blk;                  // [b][nblk][2]: below, on the boundary
};
static inline size_t cr_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t cr_layout(int b, long n, char* base, CrWs* w) {
  const long nblk = (n + CR_BLOCK - 1) / CR_BLOCK;
  size_t off = 0;
  w->keys = reinterpret_cast<unsigned long long*>(base + off); off += cr_align((size_t)b * n * 8);
  w->hist = reinterpret_cast<unsigned*>(base + off); off += cr_align((size_t)b * CR_PASSES * CR_BINS * 4);
  w->state = reinterpret_cast<CrState*>(base + off); off += cr_align((size_t)b * (CR_PASSES + 1) * sizeof(CrState));
  w->blk = reinterpret_cast<int*>(base + off); off += cr_align((size_t)b * nblk * 2 * 4);
  return off;
}

__device__ __forceinline__ unsigned long long cr_key(const float* __restrict__ p, double cx, double cy, double cz) {
  // sklearn's EuclideanDistance.rdist on float64 copies of the float32 coordinates: d = 0; d += t*t per axis, in order
  const double dx = (double)p[0] - cx, dy = (double)p[1] - cy, dz = (double)p[2] - cz;
  const double d2 = (dx * dx + dy * dy) + dz * dz;  // (-ffp-contract=off: no fused operations)
  return (unsigned long long)__double_as_longlong(d2);
}

// start state from k (or the radius): grid = b
__global__ __launch_bounds__(CR_THREADS) void crop_init_kernel(long n, int kcap, const int* __restrict__ kdev, double r2,
                                                               unsigned* __restrict__ hist, CrState* __restrict__ state) {
  const int c = blockIdx.x;
  unsigned* h = hist + (size_t)c * CR_PASSES * CR_BINS;
  for (int i = threadIdx.x; i < CR_PASSES * CR_BINS; i += CR_THREADS) h[i] = 0u;
  if (threadIdx.x == 0) {
    CrState s;
    s.pad = 0;
    if (r2 >= 0.0) {  // radius form: key <= bits(r2)  <=>  key < bits(r2) + 1; nothing to select by rank
      s.prefix = (unsigned long long)__double_as_longlong(r2) + 1ull; s.shift = 0; s.r = 0; s.done = 1;
    } else {
      long k = kdev ? (long)kdev[c] : (long)kcap;
      k = k < (long)kcap ? k : (long)kcap;
      if (k <= 0) { s.prefix = 0ull; s.shift = 0; s.r = 0; s.done = 1; }
      else if (k >= n) { s.prefix = 1ull; s.shift = 63; s.r = 0; s.done = 1; }  // every key (bit 63 is clear, NaNs included)
      else { s.prefix = 0ull; s.shift = 63; s.r = (int)k; s.done = 0; }        // all keys match the empty prefix; k to find
    }
    state[(size_t)c * (CR_PASSES + 1)] = s;
  }
}

// The boundary bin of histogram `h` for the r-th key (r >= 1, r <= total): every thread returns the same (bin, keys before it,
// keys in it).  sh: CR_THREADS / 64 + 3 ints of LDS.
__device__ __forceinline__ void cr_find_bin(const unsigned* __restrict__ h, int r, int* sh, int& bin, int& before, int& inbin) {
  constexpr int PER = CR_BINS / CR_THREADS;  // 32 consecutive bins per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned c[PER];
  int sum = 0;
  const uint4* h4 = reinterpret_cast<const uint4*>(h + tid * PER);
#pragma unroll
  for (int i = 0; i < PER / 4; ++i) {
    const uint4 v = h4[i];
    c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
    sum += (int)(v.x + v.y + v.z + v.w);
  }
  const int incl = wave_inclusive_sum_i32(sum);
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < wave; ++w) base += sh[w];
  if (base < r && r <= base + sum) {  // exactly one thread
    int cum = base;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (cum < r && r <= cum + (int)c[i]) { sh[CR_THREADS / 64] = tid * PER + i; sh[CR_THREADS / 64 + 1] = cum; sh[CR_THREADS / 64 + 2] = (int)c[i]; }
      cum += (int)c[i];
    }
  }
  __syncthreads();
  bin = sh[CR_THREADS / 64]; before = sh[CR_THREADS / 64 + 1]; inbin = sh[CR_THREADS / 64 + 2];
  __syncthreads();
}

// state[p] from state[p - 1] and histogram p - 1 (p >= 1); identical in every thread of every workgroup
__device__ __forceinline__ CrState cr_advance(int p, const CrState prev, const unsigned* __restrict__ hist_prev, int* sh) {
  if (prev.done) return prev;
  int bin, before, inbin;
  cr_find_bin(hist_prev, prev.r, sh, bin, before, inbin);
  CrState s;
  s.pad = 0;
  // prev.prefix covers the bits above cr_shift(p - 1) + cr_width(p - 1) (pass 0: the empty prefix 0 above bit 62)
  s.prefix = (prev.prefix << cr_width(p - 1)) | (unsigned long long)bin;
  s.shift = cr_shift(p - 1);
  s.r = prev.r - before;
  s.done = (s.r == inbin || p == CR_PASSES) ? 1 : 0;  // the boundary bin is taken whole / the key is complete
  return s;
}

// pass P: grid = (nblk, b)
template <int P>
__global__ __launch_bounds__(CR_THREADS) void crop_pass_kernel(long n, long scan_stride, const float* __restrict__ points,
                                                               const float* __restrict__ centres, unsigned long long* __restrict__ keys,
                                                               unsigned* __restrict__ hist, CrState* __restrict__ state) {
  __shared__ unsigned lh[CR_BINS];
  __shared__ int sh[CR_THREADS / 64 + 3];
  const int c = blockIdx.y, tid = threadIdx.x;
  CrState* st = state + (size_t)c * (CR_PASSES + 1);
  unsigned* h = hist + ((size_t)c * CR_PASSES + P) * CR_BINS;
  CrState s = st[P > 0 ? P - 1 : 0];
  if constexpr (P > 0) {
    if (s.done) {  // (uniform: the state is the same in every thread)
      if (blockIdx.x == 0 && tid == 0) st[P] = s;
      return;
    }
    s = cr_advance(P, s, h - CR_BINS, sh);
    if (blockIdx.x == 0 && tid == 0) st[P] = s;
    if (s.done) return;
  } else if (s.done) {
    // k <= 0, k >= n or the radius form: only the keys are needed
  }
  for (int i = tid; i < CR_BINS; i += CR_THREADS) lh[i] = 0u;
  __syncthreads();
  unsigned long long* kc = keys + (size_t)c * n;
  const long i0 = (long)blockIdx.x * CR_BLOCK + (long)tid * CR_ITEMS;
  if constexpr (P == 0) {
    const float* pc = points + (size_t)c * scan_stride * 3;
    const double cx = (double)centres[c * 3], cy = (double)centres[c * 3 + 1], cz = (double)centres[c * 3 + 2];
#pragma unroll
    for (int e = 0; e < CR_ITEMS; ++e) {
      const long i = i0 + e;
      if (i < n) {
        const unsigned long long key = cr_key(pc + i * 3, cx, cy, cz);
        kc[i] = key;
        if (!s.done) atomicAdd(&lh[(unsigned)(key >> cr_shift(0)) & (CR_BINS - 1)], 1u);
      }
    }
    if (s.done) return;
  } else {
    constexpr int SH = cr_shift(P), MASK = (1 << cr_width(P)) - 1;
#pragma unroll
    for (int e = 0; e < CR_ITEMS; ++e) {
      const long i = i0 + e;
      if (i < n) {
        const unsigned long long key = kc[i];
        if ((key >> s.shift) == s.prefix) atomicAdd(&lh[(unsigned)(key >> SH) & MASK], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < CR_BINS; i += CR_THREADS) {
    const unsigned v = lh[i];
    if (v != 0u) atomicAdd(&h[i], v);
  }
}

// final state of a crop (every thread the same): the last pass's histogram may still have to be read
__device__ __forceinline__ CrState cr_final(const CrState* __restrict__ st, const unsigned* __restrict__ hist_c, int* sh) {
  const CrState s = st[CR_PASSES - 1];
  return cr_advance(CR_PASSES, s, hist_c + (size_t)(CR_PASSES - 1) * CR_BINS, sh);
}

__global__ __launch_bounds__(CR_THREADS) void crop_count_kernel(long n, const unsigned long long* __restrict__ keys,
                                                                const unsigned* __restrict__ hist, CrState* __restrict__ state,
                                                                int* __restrict__ blk) {
  __shared__ int sh[CR_THREADS / 64 + 3];
  __shared__ int tot[2];
  const int c = blockIdx.y, tid = threadIdx.x;
  CrState* st = state + (size_t)c * (CR_PASSES + 1);
  const CrState s = cr_final(st, hist + (size_t)c * CR_PASSES * CR_BINS, sh);
  if (blockIdx.x == 0 && tid == 0) st[CR_PASSES] = s;
  if (tid < 2) tot[tid] = 0;
  __syncthreads();
  const unsigned long long* kc = keys + (size_t)c * n;
  const long i0 = (long)blockIdx.x * CR_BLOCK + (long)tid * CR_ITEMS;
  int below = 0, on = 0;
#pragma unroll
  for (int e = 0; e < CR_ITEMS; ++e) {
    const long i = i0 + e;
    if (i < n) {
      const unsigned long long t = kc[i] >> s.shift;
      below += t < s.prefix;
      on += t == s.prefix;
    }
  }
  below = wave_inclusive_sum_i32(below);
  on = wave_inclusive_sum_i32(on);
  if ((tid & 63) == 63) { atomicAdd(&tot[0], below); atomicAdd(&tot[1], on); }  // (integer sums: order-free)
  __syncthreads();
  if (tid < 2) blk[((size_t)c * gridDim.x + blockIdx.x) * 2 + tid] = tot[tid];
}

__global__ __launch_bounds__(CR_THREADS) void crop_write_kernel(long n, int kcap, const unsigned long long* __restrict__ keys,
                                                                const CrState* __restrict__ state, const int* __restrict__ blk,
                                                                int* __restrict__ out_idx, double* __restrict__ out_d2,
                                                                int* __restrict__ out_count) {
  __shared__ int shb[2][CR_THREADS / 64];
  __shared__ int base[2];
  const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const CrState s = state[(size_t)c * (CR_PASSES + 1) + CR_PASSES];
  const int* bc = blk + (size_t)c * gridDim.x * 2;
  // counts of the blocks before this one (block 0 of a crop also totals all of them for out_count)
  if (wave == 0) {
    const int upto = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;
    int a = 0, e = 0;
    for (int j = lane; j < upto; j += 64) { a += bc[2 * j]; e += bc[2 * j + 1]; }
    a = wave_inclusive_sum_i32(a); e = wave_inclusive_sum_i32(e);
    if (lane == 63) {
      if (blockIdx.x == 0) {
        out_count[c] = a + (e < s.r ? e : s.r);  // (k form: <= kcap by construction; radius form: the true count)
        base[0] = 0; base[1] = 0;
      } else {
        base[0] = a; base[1] = e;
      }
    }
  }
  const unsigned long long* kc = keys + (size_t)c * n;
  const long i0 = (long)blockIdx.x * CR_BLOCK + (long)tid * CR_ITEMS;
  unsigned long long key[CR_ITEMS];
  int below = 0, on = 0;
#pragma unroll
  for (int e = 0; e < CR_ITEMS; ++e) {
    const long i = i0 + e;
    key[e] = i < n ? kc[i] : ~0ull;
    const unsigned long long t = key[e] >> s.shift;
    below += (i < n && t < s.prefix);
    on += (i < n && t == s.prefix);
  }
  const int ib = wave_inclusive_sum_i32(below), io = wave_inclusive_sum_i32(on);
  if (lane == 63) { shb[0][wave] = ib; shb[1][wave] = io; }
  __syncthreads();
  int pb = base[0] + ib - below, po = base[1] + io - on;  // keys below / on the boundary in front of this thread's items
  for (int w = 0; w < wave; ++w) { pb += shb[0][w]; po += shb[1][w]; }
  int* oi = out_idx + (size_t)c * kcap;
  double* od = out_d2 ? out_d2 + (size_t)c * kcap : nullptr;
#pragma unroll
  for (int e = 0; e < CR_ITEMS; ++e) {
    const long i = i0 + e;
    const unsigned long long t = key[e] >> s.shift;
    const bool lt = i < n && t < s.prefix, eq = i < n && t == s.prefix;
    if (lt || (eq && po < s.r)) {
      const long pos = (long)pb + (long)(po < s.r ? po : s.r);
      if (pos < (long)kcap) {
        oi[pos] = (int)i;
        if (od) od[pos] = __longlong_as_double((long long)key[e]);
      }
    }
    pb += lt; po += eq;
  }
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_knn_crop_workspace_bytes(int b, long n) {
  if (b <= 0 || n <= 0) return 0;
  CrWs w;
  return cr_layout(b, n, nullptr, &w);
}

extern "C" int pasnl_knn_crop(int b, long n, long scan_stride, const float* points, const float* centres, const int* k, int kcap,
                              double radius, int* out_idx, double* out_d2, int* out_count, void* workspace, size_t workspace_bytes,
                              pasnl_stream_t stream) {
  PASNL_REQUIRE(b > 0 && n > 0 && n < (1l << 31) && kcap > 0 && (scan_stride == 0 || scan_stride >= n), PASNL_EINVAL);
  PASNL_REQUIRE(!(radius != radius), PASNL_EINVAL);  // NaN radius
  PASNL_REQUIRE(points && centres && out_idx && out_count && workspace, PASNL_ENULL);
  const long nblk = (n + CR_BLOCK - 1) / CR_BLOCK;
  CrWs w;
  PASNL_REQUIRE(workspace_bytes >= cr_layout(b, n, static_cast<char*>(workspace), &w), PASNL_EWORKSPACE);
  hipStream_t s = pasnl_hip_stream(stream);
  const double r2 = radius > 0.0 ? radius * radius : -1.0;  // sklearn: reduced radius r*r in double, inclusive
  hipLaunchKernelGGL(crop_init_kernel, dim3(b), dim3(CR_THREADS), 0, s, n, kcap, k, r2, w.hist, w.state);
  const dim3 grid((unsigned)nblk, (unsigned)b);
  hipLaunchKernelGGL(crop_pass_kernel<0>, grid, dim3(CR_THREADS), 0, s, n, scan_stride, points, centres, w.keys, w.hist, w.state);
  hipLaunchKernelGGL(crop_pass_kernel<1>, grid, dim3(CR_THREADS), 0, s, n, scan_stride, points, centres, w.keys, w.hist, w.state);
  hipLaunchKernelGGL(crop_pass_kernel<2>, grid, dim3(CR_THREADS), 0, s, n, scan_stride, points, centres, w.keys, w.hist, w.state);
  hipLaunchKernelGGL(crop_pass_kernel<3>, grid, dim3(CR_THREADS), 0, s, n, scan_stride, points, centres, w.keys, w.hist, w.state);
  hipLaunchKernelGGL(crop_pass_kernel<4>, grid, dim3(CR_THREADS), 0, s, n, scan_stride, points, centres, w.keys, w.hist, w.state);
  hipLaunchKernelGGL(crop_count_kernel, grid, dim3(CR_THREADS), 0, s, n, w.keys, w.hist, w.state, w.blk);
  hipLaunchKernelGGL(crop_write_kernel, grid, dim3(CR_THREADS), 0, s, n, kcap, w.keys, w.state, w.blk, out_idx, out_d2, out_count);
  return pasnl_launch_status();
}
