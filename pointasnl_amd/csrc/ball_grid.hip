// Ball query, grid-pruned, for clouds that fit LDS (n <= 2048): the kernel behind pasnl_query_ball_point for every
// in-model use and the north-star shape.  Contract: reference tf_ops/grouping/tf_grouping_g.cu:3-36 -- the FIRST
// nsample points in ascending index with max(sqrtf(d2),1e-20f) < radius, padded with the first hit; restated in
// oracle/pasnl_oracle.c.  The brute-force kernel for larger clouds lives in grouping.hip.
//
// One workgroup of 8 waves bins its cloud into a uniform grid held in LDS (cell edge h >= 1.001*radius in y and z, h/2 in
// x; counting sort into 16-byte records {x,y,z,index}).  The kernel is a chain of LDS round trips and vector issue, so
// what it needs is waves per SIMD: a workgroup is sized to THREE per CU for clouds of up to 1024 points (<= 53 760 bytes
// of LDS: records 16 KiB + cell starts 4 KiB + 8 wave regions of exactly 4 KiB; <= 80 VGPRs = 6 waves per SIMD).
// ONE LANE OWNS ONE QUERY: its candidates are the 3x3 runs of x-adjacent cells around it (each as wide in x as the
// ball's chord in that row of cells, contiguous in the cell-sorted array), evaluated with the canonical arithmetic, so
// the hit SET is bit-identical to the brute-force scan; what remains is to put it in index order.
//
//  tier 1 (sparse clouds, the usual case): the nine runs go into a small lane-private table (16-bit entries: start |
//    length << 11) and are walked as ONE flat sequence, two candidates per step (a wave's step count is the largest
//    candidate TOTAL over its lanes, not the sum of the largest runs); a hit appends its 16-bit index to the lane's list
//    in LDS (slot-major, lane-minor) with an unconditional store + carry add, the count clamped instead of tested;
//    afterwards the list is loaded into registers and sorted by a fixed compare-exchange network (8 / 16 / 20 inputs,
//    chosen per wave by the largest count) -- no data-dependent branch, no scan; the padded rows leave through the
//    wave's 4 KiB region, 32 rows at a time, as 16-byte stores that cover whole 128-byte rows.
//  tier 1.5 (a LANE with more than 19 hits or a run longer than 31 records, or a dense cloud; at most 128 candidates):
//    the whole wave serves that query -- one candidate (two) per lane, the hits' indices sorted across the wave.
//  tier 2 (more than 128 candidates): a hit sets bit k of the lane's n-bit row (ds_or_b32, word-major / lane-minor:
//    conflict-free; 4 KiB hold the rows of 64 / 32 / 16 lanes at a time) and scanning the row yields ascending order.
//    Exact for any input; only the lanes that need it take it.
//
// Why no hit can be missed: cell coordinate u = fl(fl(x - min) * fl(1/h)); two points closer than radius along an axis
// have |u_q - u_p| <= radius/h + 2*(G+1)*2^-23 <= 0.99901 < 1 in y and z (cells differ by at most one) and
// <= 1.99801 < 2 in x (half cells: at most two); points are clamped into the grid and queries into one (two) cells
// beyond it, which only widens the search.  A cloud whose extent is not finite (or an infinite radius) degenerates to a
// single cell whose one run every query walks in full (= brute force, through tier 2), never to a wrong answer.
#include <math.h>
#include <algorithm>
#include "common.hpp"
#include "sortnet.inc"

namespace pasnl {

constexpr int BG_WAVES = 8;
constexpr int BG_THREADS = BG_WAVES * 64;
constexpr int BG_G = 10;                       // cells per axis in y and z
constexpr int BG_XS = 2;                       // x cells per cell edge
constexpr int BG_GX = BG_G * BG_XS;
constexpr int BG_NC = BG_GX * BG_G * BG_G;     // 2000 cells at most
constexpr int BG_NMAX = 2048;                  // points per cloud (11-bit record positions, 16-bit list entries)
constexpr int BG_CAP = 20;                     // tier 1: a lane's count is clamped here; counts below it are exact
constexpr int BG_SLOTS = BG_CAP + 2;           // u16 [slot][lane]: appends go to slots <= BG_CAP + 1
constexpr int BG_TAB_OFF = BG_SLOTS * 128;     // u16 [10][lane]: the lane's non-empty runs (start | length << 11), then zeros
constexpr int BG_TAB_SLOTS = 10;
constexpr int BG_REGION = BG_TAB_OFF + BG_TAB_SLOTS * 128;  // 4096 bytes per wave: lists | tables
constexpr int BG_STAGE_ROWS = 32;              // rows of 32 entries a region stages at a time
constexpr int BG_SPLIT_BELOW = 64;             // at most this many whole chunks of queries -> quarter chunks
constexpr int BG_U = 4;                        // tier 2: candidates per lane and step
constexpr float BG_DENSE_HITS = 12.f;          // expected hits per query above which a workgroup starts in tier 2
static_assert(BG_REGION == 4096 && BG_REGION >= BG_STAGE_ROWS * 128, "a wave region stages 32 rows of 128 bytes and holds 4 KiB of bit rows");

#ifdef PASNL_TUNING
// phase probe (tuning build only; tools/ballprobe.py): cycles of wave 0 of workgroup (0, gridDim.y / 2), summed over its rounds
// [0] build  [1] -  [2] round set-up + run table  [3] walk  [4] sorting network + rows out  [5] tier 2 / round end
// [6] tier-2 passes  [7] steps
__device__ unsigned long long bg_probe[8];
#define BG_MARK(i)                                                                         \
  do {                                                                                     \
    if (probe) { const long long t_ = clock64(); atomicAdd(&bg_probe[i], (unsigned long long)(t_ - tmark)); tmark = clock64(); } \
  } while (0)
#define BG_COUNT(i, v) do { if (probe) atomicAdd(&bg_probe[i], (unsigned long long)(v)); } while (0)
// workgroup timeline (tools/ballprobe.py --timeline): {start, end, HW_ID, XCC_ID} of the first 8192 workgroups of a launch
__device__ unsigned long long bg_trace[8192 * 4];
#else
#define BG_MARK(i) do { } while (0)
#define BG_COUNT(i, v) do { } while (0)
#endif

__device__ __forceinline__ float bg_dist2(float qx, float qy, float qz, const float4 v) {
  // ((dx*dx)+(dy*dy))+(dz*dz) with x and y on one packed instruction each (identical IEEE operations per component)
  const pasnl_f32x2 d = pasnl_f32x2{v.x, v.y} - pasnl_f32x2{qx, qy};
  const pasnl_f32x2 s = d * d;
  const float dz = v.z - qz;
  return (s[0] + s[1]) + dz * dz;
}

// float -> int, truncating and SATURATING (v_cvt_i32_f32; +-inf from a degenerate grid's unbounded windows are meant)
__device__ __forceinline__ int bg_cvt_i32(float x) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// A value every lane of the wave holds (read from LDS, or computed from such values) moved to a scalar register: the
// grid's geometry would otherwise occupy ~20 vector registers for the whole kernel
__device__ __forceinline__ float bg_uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int bg_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Order-preserving integer key of a float (signed-integer order = float order; -0 < +0; NaNs beyond the infinities) and back
// (the map is its own inverse): the bounding box is reduced on keys with ONE v_min_i32_dpp per step -- fminf() on floats
// compiles to four instructions per step (identity move, dpp move, canonicalisation, minimum): 144 for six values
__device__ __forceinline__ int bg_key(float f) { const int x = __float_as_int(f); return x ^ ((x >> 31) & 0x7fffffff); }
__device__ __forceinline__ float bg_unkey(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
// minimum over the wave, valid in lane 63 (rows of 16: shr 1, 2, 4, 8, then the two row broadcasts).  (Inline assembly is fine
// in THIS file: no matrix instructions here; the s_nop keep the dpp read two wait states behind the write.)
__device__ __forceinline__ int bg_wave_min_to_lane63(int x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(x));
  return x;
}
// minimum over every group of eight lanes, valid in the group's last lane
__device__ __forceinline__ int bg_min8_to_last(int x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(x));
  return x;
}

// (Round 5: there was an inline-assembly v_addc_co_u32 here that took a lane mask as carry-in.  The hazard recogniser does not
// look into inline assembly: where the scheduler put it closer than the required wait states behind the vector compare that
// had written the mask, the add read the OLD mask and a candidate outside its run was counted.  Correct by scheduling luck in
// the shipped form; it surfaced as duplicated hits the moment the kernel body stood inside a loop (a persistent variant:
// another schedule).  `c += cond ? 1 : 0` compiles to the same v_addc, visibly -- 30.5 instead of 28 instructions per step.)
// two cell-adjacent records as the walk wants them: (x, y) of each as a pair, the two z as a pair, the two indices.  Two
// ds_read_b128 (the compiler's choice; 4 LDS cycles each) and two moves for the z pair: three ds_read2 that deliver the pairs
// directly cost 16 LDS cycles and were measured slower (155 -> 185 us at B = 4096: the LDS pipe is half busy as it is)
struct BgPair {
  pasnl_f32x2 xy0, xy1, zz;
  uint32_t k0, k1;
};

template <int NW32>  // bit-row words per lane in tier 2: n <= 32*NW32
__global__ __launch_bounds__(BG_THREADS, NW32 <= 32 ? 6 : 4) void ball_grid_kernel(
    int n, int m, float rpad, float thr2, float r3, int nsample, uint32_t ns_magic, int qchunk,
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, int* __restrict__ idx, int* __restrict__ pts_cnt) {
  constexpr int PPT = NW32 * 32 / BG_THREADS > 0 ? NW32 * 32 / BG_THREADS : 1;  // points per thread
  constexpr int LP = 1024 / NW32 > 64 ? 64 : 1024 / NW32;                        // tier 2: lanes whose bit rows fit 4 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* spt = reinterpret_cast<float4*>(smem);                                  // [n] cell-sorted {x,y,z,index bits}
  char* regions = reinterpret_cast<char*>(spt + n);                               // [BG_WAVES][BG_REGION bytes]
  unsigned short* cstart = reinterpret_cast<unsigned short*>(regions + BG_WAVES * BG_REGION);  // [BG_NC + 3]
  int* ccount = reinterpret_cast<int*>(regions);  // cell counters during the build (the regions are not live yet)
  float* red = reinterpret_cast<float*>(ccount + BG_NC);  // [BG_WAVES][6] bbox partials, [BG_WAVES] scan partials (build only)

  // The build is a chain of five barriers and a memory round trip with ~270 instructions per wave in it, beside two other
  // workgroups of the CU that are in their walks (~1 300 instructions per wave, no barrier): its waves go first at the issue
  // ports, the walk takes what is left (a workgroup's slot is held for build + walk: -2 .. 3 % at b = 1024 .. 8192)
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.y;
  const float* cloud = xyz1 + (size_t)bi * n * 3;
#ifdef PASNL_TUNING
  const bool probe = tid == 0 && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2;
  long long tmark = clock64();
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0 && wgid < 8192) {
    bg_trace[wgid * 4] = __builtin_amdgcn_s_memrealtime();
    bg_trace[wgid * 4 + 2] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    bg_trace[wgid * 4 + 3] = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
  }
#endif

  // the thread's query, requested before anything else: it arrives during the build (under load a global load takes
  // microseconds, and the round would start by waiting for it)
  const int qbase = blockIdx.x * qchunk;  // qchunk = BG_THREADS: one query per thread
  const int qend = min(m, qbase + qchunk);
  const int j = qbase + tid;
  const bool live = j < qend;
  float qx, qy, qz;
  {
    const float* qp = xyz2 + ((size_t)bi * m + (live ? j : qbase)) * 3;
    qx = qp[0]; qy = qp[1]; qz = qp[2];
  }
  // ---- A. the cloud: ONE 12-byte load per point (a wave's 64 loads cover 768 contiguous bytes), every load of the thread
  // unconditional with a clamped index and requested before the first one is used (a conditional load is waited for where it
  // is issued: round 4's two 16-byte pieces per thread were two memory round trips in a row, then a pass through LDS and a
  // rotation of the axes to undo the flat layout)
  float px[PPT], py[PPT], pz[PPT];
  float lo[3], hi[3];
  {
    const int lastp = n - 1;  // n > 0 (the entry point requires it)
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float* pp = cloud + (size_t)min(i * BG_THREADS + tid, lastp) * 3;
      px[i] = pp[0]; py[i] = pp[1]; pz[i] = pp[2];
    }
    // bounding box on integer keys: per thread the minimum of key(x) and of key(-x) over its points -- a clamped index loaded
    // a point of the cloud again, which changes no minimum: no masks -- then six single-instruction dpp reductions.  (A NaN
    // coordinate ends up in the box and the grid degenerates to one cell, like an infinite one: exact, slower.)
    int kmin[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) kmin[a] = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      kmin[0] = min(kmin[0], bg_key(px[i])); kmin[3] = min(kmin[3], bg_key(-px[i]));
      kmin[1] = min(kmin[1], bg_key(py[i])); kmin[4] = min(kmin[4], bg_key(-py[i]));
      kmin[2] = min(kmin[2], bg_key(pz[i])); kmin[5] = min(kmin[5], bg_key(-pz[i]));
    }
    int* redk = reinterpret_cast<int*>(red);  // [6][BG_WAVES]: the minima of the keys of x, y, z, -x, -y, -z per wave
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      kmin[a] = bg_wave_min_to_lane63(kmin[a]);
      if (lane == 63) redk[a * BG_WAVES + wave] = kmin[a];
    }
    for (int c = tid; c < BG_NC; c += BG_THREADS) ccount[c] = 0;
    __syncthreads();
    // the 48 partials in lanes 0..47 (axis-major, one wave's value per lane): three dpp steps reduce every group of eight lanes
    // into its last lane, six readlanes put the box into SGPRs, and the keys become floats again on the scalar unit
    static_assert(BG_WAVES == 8, "the groups of the reduction are eight lanes wide");
    const int v = bg_min8_to_last(lane < 6 * BG_WAVES ? redk[lane] : 0x7fffffff);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = bg_unkey(__builtin_amdgcn_readlane(v, a * 8 + 7));
      hi[a] = -bg_unkey(__builtin_amdgcn_readlane(v, (3 + a) * 8 + 7));
    }
  }
  // ---- B. grid geometry (identical in every thread)
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  const float maxext = fmaxf(ex, fmaxf(ey, ez));
  float inv_h = 0.f, inv_hx = 0.f, hcell = INFINITY;
  int gx = 1, gy = 1, gz = 1;
  if (maxext < INFINITY && rpad < INFINITY) {  // false for NaN / inf extents and for empty clouds (-inf)
    // (a product and a reciprocal of ~1 ulp instead of two IEEE divisions -- 20 instructions in every thread: any h and any
    // inv_h do as long as points and queries go through the SAME monotone map; a cell count that comes out one too large is
    // clamped below, and the margins of a thousandth of a cell dwarf an ulp)
    const float h = fmaxf(rpad, maxext * (1.0f / (float)BG_G));
    if (h > 0.f && h < INFINITY) {
      inv_h = __builtin_amdgcn_rcpf(h);
      inv_hx = inv_h * (float)BG_XS;
      if (inv_hx < INFINITY) {
        hcell = h;
        gx = min(BG_GX, (int)(ex * inv_hx) + 1);
        gy = min(BG_G, (int)(ey * inv_h) + 1);
        gz = min(BG_G, (int)(ez * inv_h) + 1);
      } else {
        inv_h = inv_hx = 0.f;
      }
    }
  }
  inv_h = bg_uni(inv_h); inv_hx = bg_uni(inv_hx); hcell = bg_uni(hcell);
  gx = bg_uni(gx); gy = bg_uni(gy); gz = bg_uni(gz);
  const int ncell = gx * gy * gz;
  // expected hits per query if the points were uniform in the box: decides the tier the workgroup starts in
  // (a heuristic only: both tiers are exact)
  const float vol = fmaxf(ex, hcell) * fmaxf(ey, hcell) * fmaxf(ez, hcell);
  const bool dense = bg_uni((int)!((float)n * 4.18879f * r3 <= BG_DENSE_HITS * vol)) != 0;  // also true for NaN / inf
  // ---- C. counting sort of the points by cell (the order inside a cell is irrelevant)
  int pcell[PPT], prank[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = i * BG_THREADS + tid;
    pcell[i] = 0; prank[i] = 0;
    if (k < n) {
      const int cx = min((int)fmaxf((px[i] - lo[0]) * inv_hx, 0.f), gx - 1);
      const int cy = min((int)fmaxf((py[i] - lo[1]) * inv_h, 0.f), gy - 1);
      const int cz = min((int)fmaxf((pz[i] - lo[2]) * inv_h, 0.f), gz - 1);
      pcell[i] = __mul24(__mul24(cz, gy) + cy, gx) + cx;
      prank[i] = atomicAdd(&ccount[pcell[i]], 1);
    }
  }
  __syncthreads();
  {
    // exclusive scan of the cell counts: thread t owns cells [t*CPT, t*CPT+CPT)
    constexpr int CPT = (BG_NC + BG_THREADS - 1) / BG_THREADS;
    int cnts[CPT], sum = 0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = tid * CPT + j;
      cnts[j] = c < ncell ? ccount[c] : 0;
      sum += cnts[j];
    }
    const int incl = wave_inclusive_sum_i32(sum);
    int* wsum = reinterpret_cast<int*>(red + BG_WAVES * 6);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = tid * CPT + j;
      if (c < ncell) cstart[c] = (unsigned short)base;
      base += cnts[j];
    }
    if (tid == 0) { cstart[ncell] = (unsigned short)n; cstart[ncell + 1] = (unsigned short)n; cstart[ncell + 2] = (unsigned short)n; }
  }
  __syncthreads();  // the cell starts are complete
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = i * BG_THREADS + tid;
    if (k < n) spt[(int)cstart[pcell[i]] + prank[i]] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
  }
  BG_MARK(0);


  // The nine runs of x-adjacent cells around a query.
  // Per row of cells the x window is as wide as the ball is THERE: a point of row (dy, dz) is at least day / daz cell edges
  // away from the query in y / z (its distance to the row's slab), so it can only be a hit within
  // sqrt(rho^2 - day^2 - daz^2) edges in x (rho = 1.001 radius / h <= 1); rows beyond rho are dropped.  Margins of a
  // thousandth of a cell dwarf the rounding of the cell coordinates (points and queries go through the same monotone map).
  // A degenerate grid (inv_h == 0: one cell) has no geometry to prune with: rho = inf keeps its single row and run.
  const float rho2 = bg_uni(inv_h > 0.f ? (rpad * inv_h) * (rpad * inv_h) : INFINITY);
  const float gxf1 = bg_uni((float)(gx + 1)), gyf = bg_uni((float)gy), gzf = bg_uni((float)gz);
  const int cbase2 = (int)((reinterpret_cast<char*>(cstart) - smem) >> 1);  // cstart's offset in 16-bit units
  const int rowz = gy * gx;
  // -> rs[r] = first record of run r, rl[r] = its length (0: empty, outside the grid, or beyond the ball; rs is then arbitrary
  //    but <= min(n, BG_NMAX - 1) -- it fits the 11 position bits of a table entry, which is then a run of ZERO records.  A cell
  //    start can equal n (an empty window in the tail of the last row of cells): at n == BG_NMAX that is 2048 = bit 11 = "one
  //    record at position 0" in an entry, so the largest instantiation clamps it)
  auto runs_of = [&](float qx, float qy, float qz, bool live, uint32_t (&rs)[9], uint32_t (&rl)[9]) {
    const float ux = (qx - lo[0]) * inv_hx, uy = (qy - lo[1]) * inv_h, uz = (qz - lo[2]) * inv_h;
    const float uxc = fminf(fmaxf(ux, -2.f), gxf1);  // a query outside the grid looks from its border: a superset
    const int cx = (int)floorf(uxc);
    const int cy = (int)floorf(fminf(fmaxf(uy, -1.f), gyf));
    const int cz = (int)floorf(fminf(fmaxf(uz, -1.f), gzf));
    // squared distance (cell edges) from the query to the slabs of rows cy-1, cy, cy+1 (clamped cell coordinates keep this
    // right for queries outside the grid: the rows that exist are then all on one side); what is left of rho^2 per row,
    // on pairs (one packed instruction per two rows)
    float rem[3][3];  // [dz + 1][dy + 1]
    {
      const pasnl_f32x2 a{fmaxf(uy - (float)cy - 1e-3f, 0.f), fmaxf((float)(cy + 1) - uy - 1e-3f, 0.f)};
      const pasnl_f32x2 b{fmaxf(uz - (float)cz - 1e-3f, 0.f), fmaxf((float)(cz + 1) - uz - 1e-3f, 0.f)};
      const pasnl_f32x2 a2 = a * a, b2 = b * b, r2{rho2, rho2};
      const pasnl_f32x2 ry = r2 - a2, rz = r2 - b2;                                                      // dz = 0 / dy = 0
      const pasnl_f32x2 c0 = r2 - (a2 + pasnl_f32x2{b2[0], b2[0]}), c2 = r2 - (a2 + pasnl_f32x2{b2[1], b2[1]});  // dz = -1 / +1
      rem[1][1] = rho2; rem[1][0] = ry[0]; rem[1][2] = ry[1]; rem[0][1] = rz[0]; rem[2][1] = rz[1];
      rem[0][0] = c0[0]; rem[0][2] = c0[1]; rem[2][0] = c2[0]; rem[2][2] = c2[1];
    }
    const int xl = max(cx - BG_XS, 0), xh = min(cx + BG_XS, gx - 1);  // the proven outer bounds (header); xl <= xh
    // rows that exist: y = cy + dy in [0, gy), z = cz + dz in [0, gz)   (cy in [-1, gy], cz in [-1, gz])
    const bool yok[3] = {cy >= 1, cy >= 0 && cy < gy, cy + 1 < gy};
    const bool zok[3] = {live && cz >= 1, live && cz >= 0 && cz < gz, live && cz + 1 < gz};
    const int cbc = __mul24(__mul24(cz, gy) + cy, gx) + cbase2;  // 16-bit index of the query's own row of cells in LDS
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int r = (dz + 1) * 3 + dy + 1;
        const float rm = rem[dz + 1][dy + 1];
        const float wc = __builtin_fmaf(__builtin_amdgcn_sqrtf(fmaxf(rm, 0.f)), (float)BG_XS, 2e-3f * (float)BG_XS);
        // truncation instead of floor: the lower bound is clamped at xl >= 0 anyway, the upper bound only gets wider
        const int x0 = max(bg_cvt_i32(uxc - wc), xl), x1 = min(bg_cvt_i32(uxc + wc), xh);
        const bool ok = yok[dy + 1] && zok[dz + 1] && rm > 0.f;
        const int cb = ok ? cbc + dy * gx + dz * rowz : cbase2;
        // x0 in [0, gx + 1], x1 + 1 in [-2 + 1, gx]: an inverted window reads e <= s (the cell starts are monotone, three
        // entries of padding follow the last cell) and is dropped below
        const int sv = reinterpret_cast<const unsigned short*>(smem)[cb + x0];
        const int ev = reinterpret_cast<const unsigned short*>(smem)[cb + max(x1 + 1, 0)];
        const int len = ev - sv;
        rs[r] = NW32 * 32 >= BG_NMAX ? (uint32_t)min(sv, BG_NMAX - 1) : (uint32_t)sv;  // (len <= 0 whenever sv == n)
        rl[r] = (ok && len > 0) ? (uint32_t)len : 0u;
      }
  };

  __syncthreads();  // last workgroup barrier: `ccount` (aliasing the regions) is dead, spt / cstart are complete

  // ---- D. queries: one per lane and round; a wave works in its own region only (no workgroup barrier from here on)
  char* wr = regions + wave * BG_REGION;                                           // this wave's region
  // tier 1: hit lists and run tables, slots of 128 bytes (one 16-bit entry per lane).  Lane l's entry sits at byte
  // 4 (l mod 32) + 2 (l / 32): the 32 lanes the LDS serves per cycle then fall on 32 DIFFERENT banks whatever slots they
  // address (lane-minor order put lanes 2 b and 2 b + 1 on bank b: a 2-way conflict whenever their counts differed)
  const int lofs = ((lane & 31) << 1) | (lane >> 5);                               // in 16-bit units
  unsigned short* hl = reinterpret_cast<unsigned short*>(wr) + lofs;               // hit lists: hl[slot * 64]
  unsigned short* tab = reinterpret_cast<unsigned short*>(wr + BG_TAB_OFF) + lofs; // run tables: tab[slot * 64]
  uint32_t* brow = reinterpret_cast<uint32_t*>(wr);                                // tier 2: bit rows [word][lane % LP]
  uint32_t* stage = reinterpret_cast<uint32_t*>(wr);                               // tier 1: 32 padded rows of 32 entries
  const int last = n > 0 ? n - 1 : 0;
  const bool vec4 = (nsample & 3) == 0;
  const int nchunk = (nsample + 3) >> 2;
  {  // one round: a workgroup owns qchunk = BG_THREADS queries (no loop: nothing for the compiler to hoist into registers)
    __builtin_amdgcn_s_setprio(0);
    if (!__any(live)) return;
    bool need2 = live;     // the lane's row still has to come from tier 2

    if (!dense) {
      // ---- tier 1.  (1) closing sentinels of the lists (0xFFFF), zeros (= "no more runs") in the tables: 22 + 10 slots of
      // 128 bytes = 4 KiB = four 16-byte stores per lane
      {
        const uint4 ff = make_uint4(~0u, ~0u, ~0u, ~0u), zz = make_uint4(0u, 0u, 0u, 0u);
        static_assert(BG_TAB_OFF == 2 * 1024 + 768 && BG_REGION == 4096, "the fill below is written for 22 list and 10 table slots");
        *reinterpret_cast<uint4*>(wr + lane * 16) = ff;
        *reinterpret_cast<uint4*>(wr + 1024 + lane * 16) = ff;
        const uint32_t edge = lane < 48 ? ~0u : 0u;  // the lists end 768 bytes into the third KiB
        *reinterpret_cast<uint4*>(wr + 2048 + lane * 16) = make_uint4(edge, edge, edge, edge);
        *reinterpret_cast<uint4*>(wr + 3072 + lane * 16) = zz;
      }
      // (2) the lane's non-empty runs, compacted: 16-bit entries start | length << 11
      uint32_t seen = 0u;
      {
        uint32_t rs[9], rl[9];
        runs_of(qx, qy, qz, live, rs, rl);
        int cntr = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const uint32_t ent = rs[r] | (rl[r] << 11);
          seen |= ent;
          // an empty run writes an entry of ZERO records where the next run goes: overwritten, or -- behind the lane's last
          // run -- crossed in the walk in one step whose pair is masked (the slots behind it are the fill's zeros)
          tab[cntr * 64] = (unsigned short)ent;
          cntr += rl[r] != 0u;
        }
      }
      const bool longrun = (seen >> 16) != 0u;  // a run of more than 31 records does not fit its entry: this lane -> tier 2
      if (longrun) {                            // ... and walks nothing here (rare: the block is skipped when no lane is)
#pragma unroll
        for (int t = 0; t < 9; ++t) tab[t * 64] = 0;
      }
      BG_MARK(2);
      // (3) the flat walk: two candidates per step, software-pipelined: the records of the NEXT pair are requested before the
      // current pair is evaluated (a wave alone on its SIMD otherwise waits out one LDS round trip per step), and the
      // table entry one step earlier still.  The walk's state is ONE register in the table's own format,
      // cur = position | records left << 11: a step is one add, one compare and one select (round 4: position and count apart,
      // unpacked from the entry at every step: 5 instructions more per step).
      uint32_t cur = 0, ti = 0, nx = tab[0];
      int c = 0;
      const pasnl_f32x2 qxy{qx, qy}, qzz{qz, qz};
      constexpr uint32_t ONE = 1u << 11;  // one record left
      auto advance = [&](bool adv, uint32_t stepped) {  // cur <- the pair to examine next; a lane that reached its zeros stays there
        cur = adv ? nx : stepped;
        ti += adv ? 1u : 0u;  // (v_addc with the compare's lane mask as carry-in, formed by the compiler)
        nx = tab[min(ti, (uint32_t)(BG_TAB_SLOTS - 1)) * 64];  // slot 9 is always zero
      };
      auto fetch = [&](BgPair& r) {  // position + 1 <= n: at worst the 16 bytes behind the records, read and not used
        uint32_t pos = cur & (ONE - 1u);
        asm("" : "+v"(pos));  // (keeps the mask in front of the shift-and-add that forms the address: one instruction less)
        const char* a = reinterpret_cast<const char*>(spt + pos);
        r.xy0 = *reinterpret_cast<const pasnl_f32x2*>(a);
        r.xy1 = *reinterpret_cast<const pasnl_f32x2*>(a + 16);
        r.zz = pasnl_f32x2{*reinterpret_cast<const float*>(a + 8), *reinterpret_cast<const float*>(a + 24)};
        r.k0 = *reinterpret_cast<const uint32_t*>(a + 12);
        r.k1 = *reinterpret_cast<const uint32_t*>(a + 28);
      };
      // one step: evaluate the pair `cur`; request the next pair's records into `nxt`.  A run ends when at most two of its
      // records were left BEFORE the step (tested on the old state: the position never carries into the count)
#define PASNL_BG_STEP(cur_, nxt_)                                                                                       \
      {                                                                                                                 \
        const bool l0 = cur >= ONE, l1 = cur >= 2u * ONE; /* records of the pair that belong to the run */               \
        advance(cur < 3u * ONE, cur + 2u - 2u * ONE);                                                                   \
        fetch(nxt_);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0); /* the requests go out before the current pair's arithmetic */               \
        const pasnl_f32x2 e0 = cur_.xy0 - qxy, e1 = cur_.xy1 - qxy, ez2 = cur_.zz - qzz;                                \
        const pasnl_f32x2 s0 = e0 * e0, s1 = e1 * e1, sz = ez2 * ez2;                                                   \
        float t0 = s0[0] + s0[1], t1 = s1[0] + s1[1];                                                                   \
        asm("" : "+v"(t0), "+v"(t1)); /* two plain adds into a register pair, not a packed add behind three moves */    \
        const pasnl_f32x2 dd = pasnl_f32x2{t0, t1} + sz; /* ((dx*dx)+(dy*dy))+(dz*dz), twice */                         \
        hl[c * 64] = (unsigned short)cur_.k0; /* unconditional: a miss is overwritten */                                \
        c += (l0 && dd[0] < thr2) ? 1 : 0;                                                                              \
        hl[c * 64] = (unsigned short)cur_.k1;                                                                           \
        c += (l1 && dd[1] < thr2) ? 1 : 0;                                                                              \
        c = min(c, BG_CAP); /* a count that reaches BG_CAP stays there: the lane's row then comes from tier 2 */        \
        m0 = __builtin_amdgcn_ballot_w64(cur >= ONE); /* lanes with a pair to examine in the NEXT step: the exit test */ \
        BG_COUNT(7, 1);                                                                                                 \
      }
      advance(true, 0u);
      BgPair ra, rb;
      fetch(ra);
      // two steps per trip and ONE exit test (on the mask the next step needs anyway, a scalar carried around the loop): a
      // wave's last trip may run one idle step, cheaper than a second test per trip and the copies that merge two exits
      unsigned long long m0 = __builtin_amdgcn_ballot_w64(cur >= ONE);
      while (m0 != 0ull) {
        PASNL_BG_STEP(ra, rb)
        PASNL_BG_STEP(rb, ra)
      }
#undef PASNL_BG_STEP
      BG_MARK(3);
      const bool done = live && !longrun && c < BG_CAP;  // the lane's list is complete and exact
      need2 = live && !done;
      const unsigned long long donemask = __builtin_amdgcn_ballot_w64(done);
      if (donemask != 0ull) {
        hl[c * 64] = 0xFFFFu;  // whatever a miss left behind the last hit
        // (4) lists -> registers -> sorting network.  The network size follows the wave's largest count.
        uint32_t v[20];
        const int cs = done ? c : 0;
        const bool big = __any(cs > 16), mid = __any(cs > 8);
#define PASNL_CE(a, b) { const uint32_t lo_ = min(v[a], v[b]), hi_ = max(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
        // entries are read sign-extended: the closing 0xFFFF becomes 0xFFFFFFFF, last for the unsigned network and -1 for
        // the signed maximum that turns it into the row's padding below
        const short* hs = reinterpret_cast<const short*>(hl);  // (the lane's offset inside a slot is part of the pointer)
        if (big) {
#pragma unroll
          for (int s = 0; s < 20; ++s) v[s] = (uint32_t)(int)hs[s * 64];
          PASNL_SORTNET_20
        } else if (mid) {
#pragma unroll
          for (int s = 0; s < 16; ++s) v[s] = (uint32_t)(int)hs[s * 64];
#pragma unroll
          for (int s = 16; s < 20; ++s) v[s] = 0xFFFFFFFFu;
          PASNL_SORTNET_16
        } else {
#pragma unroll
          for (int s = 0; s < 8; ++s) v[s] = (uint32_t)(int)hs[s * 64];
#pragma unroll
          for (int s = 8; s < 20; ++s) v[s] = 0xFFFFFFFFu;
          PASNL_SORTNET_8
        }
#undef PASNL_CE
        const int first = cs > 0 ? (int)v[0] : 0;  // zero-hit rows -> 0
        const int top = big ? 5 : (mid ? 4 : 2);    // 16-byte chunks that can hold anything but `first`
        // (5) the rows leave through the region, 32 rows (lanes) at a time: row q of a half is 8 chunks of 16 bytes, chunk g
        // stored at g ^ (q & 7) (conflict-free for the writers and for the readers)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          const uint32_t rowmask = (uint32_t)(donemask >> (half * 32));  // rows of this half that tier 1 owns
          if (rowmask == 0u) continue;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // lists are in registers / the previous half has been copied out
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if ((lane >> 5) == half) {
            uint4* r4 = reinterpret_cast<uint4*>(stage + (lane & 31) * 32);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
              uint4 o = make_uint4((uint32_t)first, (uint32_t)first, (uint32_t)first, (uint32_t)first);
              if (ch < 5 && ch < top) {  // a hit is >= the first hit, the sentinel is -1: one signed maximum pads the row
                o.x = (uint32_t)max((int)v[4 * ch], first);
                o.y = (uint32_t)max((int)v[4 * ch + 1], first);
                o.z = (uint32_t)max((int)v[4 * ch + 2], first);
                o.w = (uint32_t)max((int)v[4 * ch + 3], first);
              }
              r4[ch ^ (lane & 7)] = o;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (nsample == 32) {
            // the usual row of 32 entries: lane l copies chunk l & 7 of rows (l >> 3) + 8 it, it = 0 .. 3 -- every address is a
            // per-lane constant plus an immediate
            const int q0 = lane >> 3, g = lane & 7;
            const uint4* src = reinterpret_cast<const uint4*>(stage + q0 * 32 + ((g ^ q0) << 2));
            uint4* dst = reinterpret_cast<uint4*>(idx + ((size_t)bi * m + (j - lane + half * 32 + q0)) * 32 + (g << 2));
            const uint32_t rm = rowmask >> q0;
#pragma unroll
            for (int it = 0; it < 4; ++it)
              if ((rm >> (8 * it)) & 1u) dst[it * 8 * 8] = src[it * 8 * 8];  // 8 rows of 8 chunks further
          } else if (vec4) {
            const int total = BG_STAGE_ROWS * nchunk;
            for (int e0 = 0; e0 < total; e0 += 64) {
              const int ee = min(e0 + lane, total - 1);
              const int q = ns_magic ? (int)__umulhi((uint32_t)ee, ns_magic) : ee;  // ee / nchunk
              const int g = ee - q * nchunk;
              const int jq = j - lane + half * 32 + q;         // the query of row q of this half (consecutive in a wave)
              const uint4 w4 = *reinterpret_cast<const uint4*>(stage + q * 32 + ((min(g, 7) ^ (q & 7)) << 2));
              const uint32_t f0 = stage[q * 32 + ((q & 7) << 2)];  // entry 0 of the row = its first hit
              const uint4 o = g < 8 ? w4 : make_uint4(f0, f0, f0, f0);
              if (e0 + lane < total && ((rowmask >> q) & 1u) != 0u)
                *reinterpret_cast<uint4*>(idx + ((size_t)bi * m + jq) * nsample + (g << 2)) = o;
            }
          } else {
            const int total = BG_STAGE_ROWS * nsample;
            for (int e0 = 0; e0 < total; e0 += 64) {
              const int ee = min(e0 + lane, total - 1);
              const int q = ns_magic ? (int)__umulhi((uint32_t)ee, ns_magic) : ee;  // ee / nsample
              const int sidx = ee - q * nsample;
              const int jq = j - lane + half * 32 + q;
              const uint32_t o = stage[q * 32 + (min(sidx, 31) ^ ((q & 7) << 2))];
              const uint32_t f0 = stage[q * 32 + ((q & 7) << 2)];
              if (e0 + lane < total && ((rowmask >> q) & 1u) != 0u)
                idx[((size_t)bi * m + jq) * nsample + sidx] = (int)(sidx < 32 ? o : f0);
            }
          }
        }
        if (done) pts_cnt[(size_t)bi * m + j] = min(c, nsample);
        BG_MARK(4);
      }
    }

    if (__any(need2)) {
      // ---- tier 2: bit rows, LP lanes at a time, for the lanes that need it.  Word w of an active lane's row is
      // brow[w*LP + lane % LP]; the rows go straight to global memory (the rare path: no staging)
      int* orow = idx + ((size_t)bi * m + (live ? j : qbase)) * nsample;
      uint32_t rpk[9];  // start | end << 16, zero when empty
      {
        uint32_t rs[9], rl[9];
        runs_of(qx, qy, qz, need2, rs, rl);
#pragma unroll
        for (int r = 0; r < 9; ++r) rpk[r] = rl[r] != 0u ? (rs[r] | ((rs[r] + rl[r]) << 16)) : 0u;
      }
      // ---- tier 1.5: a lane whose list overflowed but whose runs hold at most 128 candidates is served by the WHOLE wave:
      // lane l takes candidates l and l + 64 of the query's runs, the hits' indices are sorted across the wave (bitonic,
      // DPP) and lane s stores entry s of the row -- ~200 instructions per such query where a bit-row pass costs thousands
      {
        int tot = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) tot += (int)(rpk[r] >> 16) - (int)(rpk[r] & 0xFFFFu);
        unsigned long long coop = __builtin_amdgcn_ballot_w64(need2 && tot <= 128);
        need2 = need2 && tot > 128;
        while (coop != 0ull) {
          const int src = (int)__builtin_ctzll(coop);
          coop &= coop - 1ull;
          const float ax = readlane_f(qx, src), ay = readlane_f(qy, src), az = readlane_f(qz, src);
          int rs[9], rl[9], total = 0;
#pragma unroll
          for (int r = 0; r < 9; ++r) {
            const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rpk[r], src);
            rs[r] = (int)(rk & 0xFFFFu);
            rl[r] = (int)(rk >> 16) - rs[r];
            total += rl[r];
          }
          auto key_of = [&](int t) {  // the index of candidate t of the query if it is a hit, else the sentinel that sorts last
            int pos = 0, cum = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
              pos = (t >= cum && t < cum + rl[r]) ? rs[r] + (t - cum) : pos;
              cum += rl[r];
            }
            const float4 rec = spt[t < total ? pos : 0];
            return (t < total && bg_dist2(ax, ay, az, rec) < thr2) ? (uint32_t)__float_as_int(rec.w) : 0xFFFFFFFFu;
          };
          uint32_t key[2];
          key[0] = key_of(lane);
          int cnt = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[0] != 0xFFFFFFFFu));
          if (total > 64) {
            key[1] = key_of(lane + 64);
            cnt += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[1] != 0xFFFFFFFFu));
            wave_bitonic_sort<2, uint32_t>(key, lane);
          } else {
            key[1] = 0xFFFFFFFFu;
            uint32_t k1[1] = {key[0]};
            wave_bitonic_sort<1, uint32_t>(k1, lane);
            key[0] = k1[0];
          }
          const uint32_t first = cnt > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)key[0], 0) : 0u;
          int* row = idx + ((size_t)bi * m + (j - lane + src)) * nsample;
          if (lane < nsample) row[lane] = (int)(lane < cnt ? key[0] : first);
          if (lane + 64 < nsample) row[lane + 64] = (int)(lane + 64 < cnt ? key[1] : first);
          for (int sp = lane + 128; sp < nsample; sp += 64) row[sp] = (int)first;  // cnt <= 128
          if (lane == 0) pts_cnt[(size_t)bi * m + (j - lane + src)] = min(cnt, nsample);
        }
      }
#pragma unroll 1
      for (int pass = 0; pass < 64 / LP; ++pass) {
        const bool act = need2 && (lane / LP) == pass;
        if (!__any(act)) continue;
        BG_COUNT(6, 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
          const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wr + i * 1024 + lane * 16) = zz;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t* myrow = brow + (lane % LP);
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
          uint32_t rk = rpk[0];  // rpk[r] without a dynamically indexed register array
#pragma unroll
          for (int t = 1; t < 9; ++t) rk = r == t ? rpk[t] : rk;
          const int rs = (int)(rk & 0xFFFFu), re = (int)(rk >> 16);
          for (int pp = rs; __any(act && pp < re); pp += BG_U) {
            float4 cv[BG_U];
#pragma unroll
            for (int u = 0; u < BG_U; ++u) cv[u] = spt[min(pp + u, last)];
#pragma unroll
            for (int u = 0; u < BG_U; ++u) {
              if (act && pp + u < re && bg_dist2(qx, qy, qz, cv[u]) < thr2) {
                const int k = __float_as_int(cv[u].w);
                atomicOr(&myrow[(k >> 5) * LP], 1u << (k & 31));  // ds_or_b32, no return: nothing waits for it
              }
            }
          }
        }
        if (act) {
          const int mw = (n + 31) >> 5;
          int first = 0, cc = 0;
          for (int w = 0; w < mw && cc < nsample; ++w) {
            uint32_t bits = myrow[w * LP];
            while (bits != 0u && cc < nsample) {
              const int k = w * 32 + (int)__builtin_ctz(bits);
              bits &= bits - 1u;
              if (cc == 0) first = k;
              orow[cc] = k;
              ++cc;
            }
          }
          for (int sp = cc; sp < nsample; ++sp) orow[sp] = first;
          pts_cnt[(size_t)bi * m + j] = cc;
        }
      }
    }
    BG_MARK(5);
  }
#ifdef PASNL_TUNING
  if (tid == 0 && wgid < 8192) bg_trace[wgid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Host side.  Returns PASNL_OK / PASNL_ELAUNCH, or PASNL_EUNSUPPORTED when the shape is not covered (the caller then
// launches the brute-force kernel).
int ball_grid_launch(int b, int n, int m, float radius, float thr2, int nsample, const float* xyz1, const float* xyz2, int* idx,
                     int* pts_cnt, hipStream_t stream) {
  if (n > BG_NMAX || nsample > 1024) return PASNL_EUNSUPPORTED;
  const int nw32 = n <= 256 ? 8 : (n <= 512 ? 16 : (n <= 1024 ? 32 : 64));
  const size_t lds = (size_t)n * 16 + BG_WAVES * BG_REGION + (size_t)((BG_NC + 3 + 1) & ~1) * 2;
  static_assert(BG_WAVES * BG_REGION >= BG_NC * 4 + (BG_WAVES * 7 + 8) * 4, "the cell counters and the build's partials alias the wave regions");
  static_assert(1024 * 16 + BG_WAVES * BG_REGION + ((BG_NC + 3 + 1) & ~1) * 2 <= 53760, "three workgroups per CU at n <= 1024");
  if (lds > 160 * 1024) return PASNL_EUNSUPPORTED;
  // queries per workgroup: one round of 64 per wave; FEW clouds: a cloud's queries over four workgroups (each builds the
  // cloud's grid again, ~3 us of 8 waves, and only two of its waves walk), so that b = 64 reaches every CU: 16.9 -> 15.8 us;
  // from b = 256 on (1024 quarter chunks) it loses: 20.2 -> 28.3 us
  const long whole = (long)b * ((m + BG_THREADS - 1) / BG_THREADS);
  const int qchunk = whole <= BG_SPLIT_BELOW ? BG_THREADS / 4 : BG_THREADS;
  dim3 grid((m + qchunk - 1) / qchunk, b);
  const float rpad = radius * 1.001f;
  const float r3 = radius * radius * radius;
  // e / d for e < 2^16 as umulhi(e, magic); d = 16-byte chunks (or entries) per row
  const unsigned div = (nsample & 3) == 0 ? (unsigned)nsample / 4 : (unsigned)nsample;
  const uint32_t ns_magic = div == 1 ? 0u : (uint32_t)((0x100000000ull / div) + 1ull);  // 0: divisor 1
#define PASNL_BG(NW)                                                                                                     \
  {                                                                                                                      \
    auto gk = ball_grid_kernel<NW>;                                                                                      \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(gk),                                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)      \
      return PASNL_ELAUNCH;                                                                                              \
    hipLaunchKernelGGL(gk, grid, dim3(BG_THREADS), lds, stream, n, m, rpad, thr2, r3, nsample, ns_magic, qchunk,          \
                       xyz1, xyz2, idx, pts_cnt);                                                                        \
  }
  if (nw32 == 8) PASNL_BG(8)
  else if (nw32 == 16) PASNL_BG(16)
  else if (nw32 == 32) PASNL_BG(32)
  else PASNL_BG(64)
#undef PASNL_BG
  return pasnl_launch_status();
}

}  // namespace pasnl

#ifdef PASNL_TUNING
extern "C" int pasnl_ball_trace_read(unsigned long long* host, int count) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pasnl::bg_trace), sizeof(unsigned long long) * 4 * count) == hipSuccess ? 0 : -1;
}
extern "C" int pasnl_ball_occupancy(int n) {  // workgroups per CU the runtime computes for the n <= 1024 instantiation
  int nb = -1;
  const size_t lds = (size_t)n * 16 + pasnl::BG_WAVES * pasnl::BG_REGION + (size_t)((pasnl::BG_NC + 3 + 1) & ~1) * 2;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pasnl::ball_grid_kernel<32>, pasnl::BG_THREADS, lds) != hipSuccess) return -1;
  return nb;
}
extern "C" int pasnl_ball_probe_read(unsigned long long* host8) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(pasnl::bg_probe), sizeof(pasnl::bg_probe)) != hipSuccess) return -1;
  unsigned long long zero[8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pasnl::bg_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
