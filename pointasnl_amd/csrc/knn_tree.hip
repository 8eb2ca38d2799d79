// kNN in the REFERENCE'S OWN ORDER among equal distances (optional: nearest_neighbors.knn_batch(..., tie_order="nanoflann")).
//
// The reference searches a nanoflann KD-tree (utils/nearest_neighbors/knn_.cxx:72-135: KDTreeTableAdaptor<float,float>, leaf
// size 10, KNNResultSet, SearchParams(10) -> eps 0).  Its result set keeps candidates sorted by distance and puts a candidate
// BEHIND the entries of equal distance (nanoflann.hpp:115-134, NANOFLANN_FIRST_MATCH undefined), and a leaf is scanned against
// the worst distance read once at its entry (:1357-1368): among exactly equal distances the order -- and which of several tied
// candidates is the K-th -- is the ORDER OF VISITS, i.e. a function of the tree.  The product's kNN kernels return the canonical
// (distance, index) order, identical whenever distances are distinct.  This file reproduces the reference's order
// bit for bit, for data with ties, by re-doing what nanoflann does -- the same tree (divideTree / middleSplit_ / planeSplit,
// nanoflann.hpp:916-1043, re-stated here with explicit stacks; every float expression in the reference's association, no
// contraction) and the same search (searchLevel :1351-1410, near child first) -- on the GPU:
//   * knn_tree_build_lds_kernel (n <= 8192, the points in LDS) / knn_tree_build_par_kernel (n <= 10240, gathers): a workgroup
//     per cloud builds the tree level by level, a wave per node, with planeSplit's Hoare loop in closed form; knn_tree_build_kernel: the literal one-lane restatement (larger clouds, and the
//     checker of the parallel build in the tuning build);
//   * knn_tree_search_kernel: one lane per query walks it as a flat state machine, the result set as the k smallest
//     (distance, arrival) keys in LDS, sorted once at the end.
//   16 clouds of 8192 points, 1024 queries, k = 32: 60.2 ms (round 3: one lane per cloud) -> 1.1 ms; the canonical-order
//   kernels take 62 us -- an exactness mode that is usable, not a fast path.
// Trees or searches deeper than KT_DEPTH levels (pathological, exponentially clustered data) raise a flag in the workspace,
// which the Python wrapper turns into PasnlUnsupported.
#include <algorithm>
#include "common.hpp"

namespace pasnl {

constexpr int KT_LEAF = 10;     // knn_.cxx:83 KDTree mat_index(npts, dim, points, 10)
constexpr int KT_DEPTH = 96;    // frames of the explicit stacks

struct KtNode {   // leaf: child1 < 0, a = left, b = right (as int bits); inner: a = divfeat, divlow, divhigh
  int child1, child2, a;
  float divlow, divhigh;
};

struct KtFrame {  // one activation of divideTree
  unsigned left, right, idx;
  int node, cutfeat, phase;
  float cutval;
  float bbox[6];   // in: the box handed down; out: the tight box of the subtree      [low0, high0, low1, high1, low2, high2]
  float lbox[6];   // the left child's box (in / out)
  float rbox[6];
};

__device__ __forceinline__ size_t kt_align(size_t x) { return (x + 15) & ~(size_t)15; }
static inline size_t kt_align_h(size_t x) { return (x + 15) & ~(size_t)15; }
struct KtWork {  // an inner node waiting for its split (parallel build): the activation's arguments
  int node;
  unsigned left, right;
  float box[6];
};
// A queue entry written by another wave of the workgroup one level ago, read by a whole wave: its nine words by nine LANES in ONE
// load (volatile: never the scalar cache, never a stale line), handed round by readlane.  Lane-uniform volatile loads of the
// struct compiled to nine system-coherent loads each waited for in turn -- most of the "3.5 us of fixed cost per node".
static_assert(sizeof(KtWork) == 36, "nine words");
__device__ __forceinline__ KtWork kt_load_work(const KtWork* qe, const int lane) {
  const uint32_t v = reinterpret_cast<const volatile uint32_t*>(qe)[lane < 9 ? lane : 0];
  KtWork wk;
  wk.node = __builtin_amdgcn_readlane((int)v, 0);
  wk.left = (unsigned)__builtin_amdgcn_readlane((int)v, 1);
  wk.right = (unsigned)__builtin_amdgcn_readlane((int)v, 2);
#pragma unroll
  for (int i = 0; i < 6; ++i) wk.box[i] = __int_as_float(__builtin_amdgcn_readlane((int)v, 3 + i));
  return wk;
}
constexpr int KT_HDR = 32;   // header bytes: [flag, root, nodes used, depth, deep work items, the queue they are in, -, -]
constexpr int KT_TOPIDS = 1024;  // node ids of the first phase of a two-phase build; a deep subtree at leaf position `left` owns the ids
                                 // KT_TOPIDS + 2 left .. (a subtree of c points has < 2 c nodes)
#define KT_NNODES(n) (2 * (n) + KT_TOPIDS)
constexpr int KTD_MAXWORK = 128;  // subtrees handed to the second phase at most (16 after KTD_TOP levels; clouds above 8192 points: every
                                  // subtree that fits a workgroup's LDS)
constexpr int KTD_TOP = 4;
constexpr int KTB_WAVES = 16;          // waves of the parallel build's workgroup (one workgroup per cloud)
constexpr int KTB_LDS_NMAX = 8192;     // points (of a cloud, of a subtree) whose 18-byte records + positions fit the LDS (147 KB)
constexpr int KTB_LDS_NMAX_ = KTB_LDS_NMAX;
constexpr int KTD_LDSQ_MAX = 6000;  // points of a subtree whose two level queues still fit in LDS behind its records
constexpr int KTB_NMAX = 10240;        // points per cloud it holds in LDS (vind + the cut coordinate + a scratch slice)
static inline int kt_queue_cap(int n) { return n / (KT_LEAF + 1) + 2; }  // inner nodes of one level: more than KT_LEAF points each
// workspace of one cloud: [flag, root, nodes used, depth] | vind[n] | nodes[2n] | build frames[KT_DEPTH] | 2 level queues |
// recs[n]: {x, y, z, index} of vind[i] -- the points in LEAF ORDER, so that a leaf's scan is one contiguous read
static inline size_t kt_recs_offset(int n) {
  return kt_align_h(KT_HDR) + kt_align_h((size_t)n * 4) + kt_align_h((size_t)KT_NNODES(n) * sizeof(KtNode)) + kt_align_h((size_t)KT_DEPTH * sizeof(KtFrame)) +
         2 * kt_align_h((size_t)kt_queue_cap(n) * sizeof(KtWork)) + 2 * kt_align_h((size_t)(kt_queue_cap(n) + 3 * KTD_MAXWORK) * sizeof(KtWork));
}
static inline size_t kt_cloud_bytes(int n) { return kt_recs_offset(n) + kt_align_h((size_t)n * 16); }

__device__ void knn_tree_build_body(int cloud, int n, const float* __restrict__ pts_all, char* __restrict__ ws_all,
                                    size_t stride, size_t recs_off) {
  const float* pts = pts_all + (size_t)cloud * n * 3;
  char* ws = ws_all + (size_t)cloud * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  unsigned* vind = reinterpret_cast<unsigned*>(ws + kt_align(KT_HDR));
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  KtFrame* st = reinterpret_cast<KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  hdr[0] = 0;
  for (int i = 0; i < n; ++i) vind[i] = (unsigned)i;                       // init_vind (:1318)
  auto get = [&](unsigned idx, int d) { return pts[(size_t)idx * 3 + d]; };  // dataset_get -> kdtree_get_pt
  // computeBoundingBox (:1321-1346)
  float root[6];
  for (int d = 0; d < 3; ++d) root[2 * d] = root[2 * d + 1] = get(0, d);
  for (int k = 1; k < n; ++k)
    for (int d = 0; d < 3; ++d) {
      const float v = get((unsigned)k, d);
      if (v < root[2 * d]) root[2 * d] = v;
      if (v > root[2 * d + 1]) root[2 * d + 1] = v;
    }
  auto min_max = [&](const unsigned* ind, unsigned count, int el, float& mn, float& mx) {  // computeMinMax (:898-907)
    mn = get(ind[0], el);
    mx = mn;
    for (unsigned i = 1; i < count; ++i) {
      const float v = get(ind[i], el);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  };
  int nnodes = 0, sp = 0, maxdepth = 0;
  // root activation
  st[0].left = 0; st[0].right = (unsigned)n; st[0].phase = 0;
  for (int i = 0; i < 6; ++i) st[0].bbox[i] = root[i];
  while (sp >= 0) {
    KtFrame& f = st[sp];
    if (f.phase == 0) {
      f.node = nnodes++;
      KtNode& nd = nodes[f.node];
      if (f.right - f.left <= (unsigned)KT_LEAF) {  // leaf (:921-936): its box shrinks to its points
        nd.child1 = nd.child2 = -1;
        nd.a = (int)f.left;
        nd.divlow = __int_as_float((int)f.right);
        nd.divhigh = 0.f;
        for (int d = 0; d < 3; ++d) f.bbox[2 * d] = f.bbox[2 * d + 1] = get(vind[f.left], d);
        for (unsigned k = f.left + 1; k < f.right; ++k)
          for (int d = 0; d < 3; ++d) {
            const float v = get(vind[k], d);
            if (f.bbox[2 * d] > v) f.bbox[2 * d] = v;
            if (f.bbox[2 * d + 1] < v) f.bbox[2 * d + 1] = v;
          }
        --sp;
        continue;
      }
      // middleSplit_ (:966-1005)
      unsigned* ind = vind + f.left;
      const unsigned count = f.right - f.left;
      const float EPS = 0.00001f;
      float max_span = f.bbox[1] - f.bbox[0];
      for (int d = 1; d < 3; ++d) {
        const float span = f.bbox[2 * d + 1] - f.bbox[2 * d];
        if (span > max_span) max_span = span;
      }
      float max_spread = -1.f;
      int cutfeat = 0;
      for (int d = 0; d < 3; ++d) {
        const float span = f.bbox[2 * d + 1] - f.bbox[2 * d];
        if (span > (1 - EPS) * max_span) {
          float mn, mx;
          min_max(ind, count, d, mn, mx);
          const float spread = mx - mn;
          if (spread > max_spread) { cutfeat = d; max_spread = spread; }
        }
      }
      const float split_val = (f.bbox[2 * cutfeat] + f.bbox[2 * cutfeat + 1]) / 2;
      float mn, mx;
      min_max(ind, count, cutfeat, mn, mx);
      float cutval;
      if (split_val < mn) cutval = mn;
      else if (split_val > mx) cutval = mx;
      else cutval = split_val;
      // planeSplit (:1016-1043)
      unsigned left = 0, right = count - 1, lim1, lim2;
      for (;;) {
        while (left <= right && get(ind[left], cutfeat) < cutval) ++left;
        while (right && left <= right && get(ind[right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        const unsigned t = ind[left]; ind[left] = ind[right]; ind[right] = t;
        ++left; --right;
      }
      lim1 = left;
      right = count - 1;
      for (;;) {
        while (left <= right && get(ind[left], cutfeat) <= cutval) ++left;
        while (right && left <= right && get(ind[right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        const unsigned t = ind[left]; ind[left] = ind[right]; ind[right] = t;
        ++left; --right;
      }
      lim2 = left;
      unsigned index;
      if (lim1 > count / 2) index = lim1;
      else if (lim2 < count / 2) index = lim2;
      else index = count / 2;
      f.idx = index; f.cutfeat = cutfeat; f.cutval = cutval;
      nd.a = cutfeat;
      for (int i = 0; i < 6; ++i) f.lbox[i] = f.bbox[i];
      f.lbox[2 * cutfeat + 1] = cutval;
      f.phase = 1;
      if (sp + 1 >= KT_DEPTH) { hdr[0] = 1; return; }
      KtFrame& c = st[sp + 1];
      c.left = f.left; c.right = f.left + index; c.phase = 0;
      for (int i = 0; i < 6; ++i) c.bbox[i] = f.lbox[i];
      ++sp;
      if (sp > maxdepth) maxdepth = sp;
    } else if (f.phase == 1) {  // child1 has returned: st[sp + 1] holds its frame (node index, tight box)
      KtFrame& c = st[sp + 1];
      nodes[f.node].child1 = c.node;
      for (int i = 0; i < 6; ++i) f.lbox[i] = c.bbox[i];
      for (int i = 0; i < 6; ++i) f.rbox[i] = f.bbox[i];
      f.rbox[2 * f.cutfeat] = f.cutval;
      f.phase = 2;
      c.left = f.left + f.idx; c.right = f.right; c.phase = 0;
      for (int i = 0; i < 6; ++i) c.bbox[i] = f.rbox[i];
      ++sp;
    } else {  // child2 has returned
      KtFrame& c = st[sp + 1];
      KtNode& nd = nodes[f.node];
      nd.child2 = c.node;
      for (int i = 0; i < 6; ++i) f.rbox[i] = c.bbox[i];
      nd.divlow = f.lbox[2 * f.cutfeat + 1];
      nd.divhigh = f.rbox[2 * f.cutfeat];
      for (int d = 0; d < 3; ++d) {
        f.bbox[2 * d] = fminf(f.lbox[2 * d], f.rbox[2 * d]);
        f.bbox[2 * d + 1] = fmaxf(f.lbox[2 * d + 1], f.rbox[2 * d + 1]);
      }
      --sp;
    }
  }
  // root_bbox after divideTree = the tight box of all points (st[0].bbox); findNeighbors uses it (:1045-1061)


  KtFrame& out = st[1];
  for (int i = 0; i < 6; ++i) out.bbox[i] = st[0].bbox[i];
  {
    float4* recs = reinterpret_cast<float4*>(ws + recs_off);
    for (int i = 0; i < n; ++i) recs[i] = make_float4(get(vind[i], 0), get(vind[i], 1), get(vind[i], 2), __int_as_float((int)vind[i]));
  }
  hdr[1] = st[0].node;
  hdr[2] = nnodes;
  hdr[3] = maxdepth;
}
// A grid of at most `b` workgroups walks the clouds.  nflag (pasnl_knn_batch_ref): only the clouds with flagged queries get a
// tree -- on tie-free batches every workgroup reads a few counters and returns, so the launch has to be CHEAP TO PLACE: a capped
// grid (the kernels of a forward running beside it hold the LDS these workgroups ask for; one workgroup per cloud of a large
// batch queued behind them and held the side stream for ~50 us per search, measured: +8 % on the classifier's step)
__global__ __launch_bounds__(64) void knn_tree_build_kernel(int b, int n, const float* __restrict__ pts_all, char* __restrict__ ws_all,
                                                           size_t stride, size_t recs_off, const int* __restrict__ nflag) {
  if (threadIdx.x != 0) return;
  for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x)
    if (!nflag || nflag[cloud] != 0) knn_tree_build_body(cloud, n, pts_all, ws_all, stride, recs_off);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tree, built by a WORKGROUP per cloud (n <= KTB_NMAX).  What makes the serial build slow is not its arithmetic
// but that one lane walks a chain of dependent global loads (60 ms for 16 clouds of 8192 points); what makes it look
// inherently serial is planeSplit's in-place Hoare partition.  Both yield:
//   * the activations of one tree LEVEL are independent (disjoint slices of vind): a wave per node, level by level, the
//     nodes of the next level queued in the workspace (breadth first instead of the reference's recursion -- the node
//     NUMBERS differ, the tree does not, and the search follows child pointers);
//   * computeMinMax is a min / max reduction (exact in any order);
//   * planeSplit (nanoflann.hpp:1016-1043) ends in a state that depends only on which elements satisfy the predicate:
//     with c of them, the i-th violator among the first c positions (ascending) has been swapped with the i-th satisfier
//     among the rest (DESCENDING), nothing else has moved, and the pointers meet at c (the `right &&` guard only ever stops
//     a scan that has nothing left to swap).  Ranks by ballot + prefix popcount, positions through a scratch slice, swaps in
//     parallel -- the same permutation as the loop, element for element (checked against a transcription of the loop on
//     200 000 random slices with ties, tests/test_knn_tree_partition.py, and against the serial build on the GPU);
//   * the tight box a child returns is the bounding box of its points, so divlow / divhigh (:956-957) are the maximum of
//     the left part / the minimum of the right part along the cut dimension -- two more reductions, no second traversal.
// The cut coordinate of a node's points sits in LDS next to vind and is permuted with it; the other coordinates are read
// from global memory (L2) for the candidate dimensions of middleSplit_ only.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ktb_wave_sync() {  // LDS operations of a wave execute in order; keep the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ void knn_tree_build_par_body(int cloud, int n, const float* __restrict__ pts_all, char* __restrict__ ws_all, size_t stride,
                                        size_t recs_off, int stop_level) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* vind = reinterpret_cast<unsigned*>(smem);                       // [n]
  float* vals = reinterpret_cast<float*>(vind + n);                          // [n] cut coordinate of vind[i] (current node)
  unsigned short* sc = reinterpret_cast<unsigned short*>(vals + n);          // [n] positions of misplaced elements
  float* part = reinterpret_cast<float*>(sc + ((n + 1) & ~1));               // [KTB_WAVES][6] root-box partials
  int* ctr = reinterpret_cast<int*>(part + KTB_WAVES * 6);                   // [0], [1]: queue lengths; [2]: nodes used; [3]: flag
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pts = pts_all + (size_t)cloud * n * 3;
  char* ws = ws_all + (size_t)cloud * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  unsigned* gvind = reinterpret_cast<unsigned*>(ws + kt_align(KT_HDR));
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  KtFrame* fr = reinterpret_cast<KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  const int qcap = n / (KT_LEAF + 1) + 2;
  KtWork* queue[2];
  queue[0] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(fr) + kt_align((size_t)KT_DEPTH * sizeof(KtFrame)));
  queue[1] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(queue[0]) + kt_align((size_t)qcap * sizeof(KtWork)));
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));  // lanes below this one

  // init_vind (:1318), computeBoundingBox (:1321-1346)
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += KTB_WAVES * 64) {
    vind[i] = (unsigned)i;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = pts[(size_t)i * 3 + d];
      lo[d] = v < lo[d] ? v : lo[d];
      hi[d] = v > hi[d] ? v : hi[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f32(lo[d]); hi[d] = wave_max_f32(hi[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { part[wave * 6 + 2 * d] = lo[d]; part[wave * 6 + 2 * d + 1] = hi[d]; }
  }
  if (tid < 4) ctr[tid] = 0;
  __syncthreads();
  float root[6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float l = part[2 * d], h = part[2 * d + 1];
    for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, part[w * 6 + 2 * d]); h = fmaxf(h, part[w * 6 + 2 * d + 1]); }
    root[2 * d] = l; root[2 * d + 1] = h;
  }
  if (tid == 0) {
    ctr[2] = 1;  // node 0 = the root
    if (n <= KT_LEAF) {
      nodes[0].child1 = nodes[0].child2 = -1; nodes[0].a = 0; nodes[0].divlow = __int_as_float(n); nodes[0].divhigh = 0.f;
    } else {
      KtWork w0; w0.node = 0; w0.left = 0; w0.right = (unsigned)n;
      for (int i = 0; i < 6; ++i) w0.box[i] = root[i];
      queue[0][0] = w0;
      ctr[0] = 1;
    }
    for (int i = 0; i < 6; ++i) fr[1].bbox[i] = root[i];  // root_bbox after divideTree = the tight box of all points (what the search reads)
  }
  __threadfence_block();
  __syncthreads();

  int cur = 0, level = 0;
  for (;;) {
    const int nq = ctr[cur];
    if (nq == 0) break;
    if (stop_level > 0 && level >= stop_level) {
      // hand the pending subtrees to knn_tree_build_deep_kernel (one workgroup each, records in LDS) -- unless one of them is
      // too large for its LDS (a very lopsided tree): then this kernel finishes the tree itself
      if (tid == 0) {
        int big = 0;
        for (int e = 0; e < nq; ++e) {
          const volatile KtWork* qe = queue[cur] + e;
          big |= (qe->right - qe->left) > (unsigned)KTB_LDS_NMAX_ ? 1 : 0;
        }
        ctr[3] |= big << 1;
      }
      __syncthreads();
      if ((ctr[3] & 2) == 0) break;
      stop_level = 0;
    }
    if (level + 2 >= KT_DEPTH) { if (tid == 0) ctr[3] = 1; break; }  // deeper than the search's stack: flagged, not built
    for (int e = wave; e < nq; e += KTB_WAVES) {
      const KtWork wk = kt_load_work(queue[cur] + e, lane);
      const unsigned left = wk.left, right = wk.right, count = right - left;
      // ---- middleSplit_ (:966-1005)
      const float EPS = 0.00001f;
      float max_span = wk.box[1] - wk.box[0];
      for (int d = 1; d < 3; ++d) {
        const float span = wk.box[2 * d + 1] - wk.box[2 * d];
        if (span > max_span) max_span = span;
      }
      float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
      int cutfeat = 0;
      for (int d = 0; d < 3; ++d) {
        const float span = wk.box[2 * d + 1] - wk.box[2 * d];
        if (span > (1 - EPS) * max_span) {
          float mn = INFINITY, mx = -INFINITY;  // computeMinMax (:898-907)
          for (unsigned p = lane; p < count; p += 64) {
            const float v = pts[(size_t)vind[left + p] * 3 + d];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
          }
          mn = wave_min_f32(mn); mx = wave_max_f32(mx);
          const float spread = mx - mn;
          if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn; mx_c = mx; }
        }
      }
      // (selects, not wk.box[2 * cutfeat]: a dynamically indexed local array lives in scratch memory)
      const float blo = cutfeat == 0 ? wk.box[0] : (cutfeat == 1 ? wk.box[2] : wk.box[4]);
      const float bhi = cutfeat == 0 ? wk.box[1] : (cutfeat == 1 ? wk.box[3] : wk.box[5]);
      const float split_val = (blo + bhi) / 2;
      float cutval;  // (the second computeMinMax of the reference, on cutfeat, returns mn_c / mx_c again)
      if (split_val < mn_c) cutval = mn_c;
      else if (split_val > mx_c) cutval = mx_c;
      else cutval = split_val;
      for (unsigned p = lane; p < count; p += 64) vals[left + p] = pts[(size_t)vind[left + p] * 3 + cutfeat];
      ktb_wave_sync();
      // ---- planeSplit (:1016-1043): two passes, each the parallel form of the Hoare loop (header)
      unsigned lim[2];
      unsigned lo_p = 0;
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        auto pred = [&](float v) { return pass == 0 ? v < cutval : v <= cutval; };
        unsigned cnt = 0;
        for (unsigned p0 = lo_p; p0 < count; p0 += 64) {
          const unsigned p = p0 + lane;
          cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p < count && pred(vals[left + p])));
        }
        const unsigned mid = lo_p + cnt;  // where the pointers meet
        unsigned nl = 0, nr = 0;
        for (unsigned p0 = lo_p; p0 < mid; p0 += 64) {  // violators among the first cnt positions, ascending
          const unsigned p = p0 + lane;
          const bool mis = p < mid && !pred(vals[left + p]);
          const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
          if (mis) sc[left + lo_p + nl + (unsigned)__builtin_popcountll(mk & lt_mask)] = (unsigned short)p;
          nl += (unsigned)__builtin_popcountll(mk);
        }
        for (unsigned q0 = 0; mid + q0 < count; q0 += 64) {  // satisfiers among the rest, descending
          const unsigned q = q0 + lane;
          const bool in = mid + q < count;
          const unsigned p = count - 1 - (in ? q : 0);
          const bool mis = in && pred(vals[left + p]);
          const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
          if (mis) sc[right - 1 - (nr + (unsigned)__builtin_popcountll(mk & lt_mask))] = (unsigned short)p;
          nr += (unsigned)__builtin_popcountll(mk);
        }
        ktb_wave_sync();
        for (unsigned i = lane; i < nl; i += 64) {  // nl == nr
          const unsigned a = left + sc[left + lo_p + i], b = left + sc[right - 1 - i];
          const unsigned ta = vind[a], tb = vind[b];
          const float va = vals[a], vb = vals[b];
          vind[a] = tb; vind[b] = ta;
          vals[a] = vb; vals[b] = va;
        }
        ktb_wave_sync();
        lim[pass] = mid;
        lo_p = mid;
      }
      unsigned index;
      if (lim[0] > count / 2) index = lim[0];
      else if (lim[1] < count / 2) index = lim[1];
      else index = count / 2;
      // ---- the children's tight boxes along cutfeat: divlow = max of the left part, divhigh = min of the right part (:956-957)
      float dl = -INFINITY, dh = INFINITY;
      for (unsigned p = lane; p < count; p += 64) {
        const float v = vals[left + p];
        if (p < index) dl = v > dl ? v : dl;
        else dh = v < dh ? v : dh;
      }
      dl = wave_max_f32(dl); dh = wave_min_f32(dh);
      if (lane == 0) {
        int child[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned cl = c == 0 ? left : left + index, cr = c == 0 ? left + index : right;
          const int id = atomicAdd(&ctr[2], 1);
          child[c] = id;
          if (cr - cl <= (unsigned)KT_LEAF) {  // leaf (:921-936)
            nodes[id].child1 = nodes[id].child2 = -1;
            nodes[id].a = (int)cl;
            nodes[id].divlow = __int_as_float((int)cr);
            nodes[id].divhigh = 0.f;
          } else {
            KtWork w;
            w.node = id; w.left = cl; w.right = cr;
            // left child: high = cutval (:946-947); right child: low = cutval (:951-952)
#pragma unroll
            for (int i = 0; i < 6; ++i) w.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : wk.box[i];
            queue[cur ^ 1][atomicAdd(&ctr[cur ^ 1], 1)] = w;
          }
        }
        KtNode nd;
        nd.child1 = child[0]; nd.child2 = child[1]; nd.a = cutfeat; nd.divlow = dl; nd.divhigh = dh;
        nodes[wk.node] = nd;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) ctr[cur] = 0;
    cur ^= 1;
    ++level;
    __syncthreads();
  }
  __syncthreads();
  float4* recs = reinterpret_cast<float4*>(ws + recs_off);
  for (int i = tid; i < n; i += KTB_WAVES * 64) {
    const unsigned v = vind[i];
    gvind[i] = v;
    recs[i] = make_float4(pts[(size_t)v * 3], pts[(size_t)v * 3 + 1], pts[(size_t)v * 3 + 2], __int_as_float((int)v));
  }
  if (tid == 0) { hdr[0] = ctr[3] & 1; hdr[1] = 0; hdr[2] = ctr[2]; hdr[3] = level; hdr[4] = (ctr[3] & 1) || stop_level == 0 ? 0 : ctr[cur]; hdr[5] = cur; hdr[6] = level; }
}
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_build_par_kernel(int b, int n, const float* __restrict__ pts_all,
                                                                           char* __restrict__ ws_all, size_t stride, size_t recs_off, int stop_level,
                                                                           const int* __restrict__ nflag) {
  for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {  // (a capped grid walks the clouds: knn_tree_build_kernel's note)
    if (nflag && nflag[cloud] == 0) continue;
    knn_tree_build_par_body(cloud, n, pts_all, ws_all, stride, recs_off, stop_level);
    __syncthreads();  // the next cloud re-uses the LDS
  }
}

// The same build with the POINTS in LDS (n <= KTB_LDS_NMAX): records {x, y, z, index} that move with the index list, so every
// pass over a node is a sequential LDS read.  In the kernel above a pass gathers pts[vind[i]] from global memory, one dependent
// round trip per 64 points.  16 x 8192: 556 -> 485 us -- what remains is one wave's ~9 passes over a large node at the top
// levels and ~3.5 us of fixed cost per node at the deep ones (EXPERIMENTS.md).
// One inner node of the build with the points in LDS (middleSplit_ + planeSplit + the children's tight bounds), by one wave:
// rec / sc positions `left .. right` are LDS positions.  -> cutfeat, cutval, index (the left child's size), divlow, divhigh
struct KtSplit { int cutfeat; float cutval; unsigned index; float dl, dh; };
// A node of at most 64 points (the thousands of nodes at the deep levels; ~3.5 us each through the general form below, whose ten
// passes are LDS round trips for one trip's worth of data): ONE POINT PER LANE, held in registers through both partition passes.
// The same closed form of planeSplit (header): with cnt satisfiers in [lo, count), the i-th violator among positions
// [lo, lo + cnt) (ascending) and the i-th satisfier at or behind lo + cnt (descending) exchange places -- ranks by ballot +
// popcount, the partner's lane through 64 bytes of scratch, the exchange itself by four ds_bpermute.  ~0.5 us.
__device__ __forceinline__ KtSplit ktb_split_node_small(float4* rec, unsigned short* sc, const KtWork& wk, const unsigned left,
                                                        const unsigned count, const int lane, const unsigned long long lt_mask) {
  const bool in = (unsigned)lane < count;
  float4 r = rec[left + (in ? lane : 0)];
  // ---- middleSplit_ (:966-1005)
  const float EPS = 0.00001f;
  float max_span = wk.box[1] - wk.box[0];
  for (int d = 1; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3], mx3[3];
  {
    const float c[3] = {r.x, r.y, r.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(in ? c[d] : INFINITY); mx3[d] = wave_max_f32(in ? c[d] : -INFINITY); }
  }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? wk.box[0] : (cutfeat == 1 ? wk.box[2] : wk.box[4]);
  const float bhi = cutfeat == 0 ? wk.box[1] : (cutfeat == 1 ? wk.box[3] : wk.box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  // ---- planeSplit (:1016-1043), both passes on registers
  unsigned lim[2];
  unsigned lo_p = 0;
  unsigned short* scr = sc + left;  // >= count entries of scratch: partner lanes by rank (violators from the front, satisfiers from the back)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float v = cutfeat == 0 ? r.x : (cutfeat == 1 ? r.y : r.z);
    const bool inr = in && (unsigned)lane >= lo_p;
    const bool sat = inr && (pass == 0 ? v < cutval : v <= cutval);
    const unsigned cnt = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sat));
    const unsigned mid = lo_p + cnt;
    const bool viol = inr && (unsigned)lane < mid && !sat;       // a violator among the first cnt positions
    const bool rsat = sat && (unsigned)lane >= mid;              // a satisfier among the rest
    const unsigned long long mv = __builtin_amdgcn_ballot_w64(viol), mr = __builtin_amdgcn_ballot_w64(rsat);
    if (mv != 0ull) {  // (uniform) popcount(mv) == popcount(mr)
      const unsigned vrank = (unsigned)__builtin_popcountll(mv & lt_mask);                                   // ascending
      const unsigned rrank = (unsigned)__builtin_popcountll(mr & ~lt_mask & ~(1ull << lane));                // descending: set bits above this lane
      if (viol) scr[vrank] = (unsigned short)lane;
      if (rsat) scr[count - 1 - rrank] = (unsigned short)lane;
      ktb_wave_sync();
      int partner = lane;
      if (viol) partner = scr[count - 1 - vrank];
      if (rsat) partner = scr[rrank];
      ktb_wave_sync();
      r.x = __shfl(r.x, partner); r.y = __shfl(r.y, partner); r.z = __shfl(r.z, partner); r.w = __shfl(r.w, partner);
    }
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  const float v = cutfeat == 0 ? r.x : (cutfeat == 1 ? r.y : r.z);
  const float dl = wave_max_f32(in && (unsigned)lane < index ? v : -INFINITY);
  const float dh = wave_min_f32(in && (unsigned)lane >= index ? v : INFINITY);
  if (in) rec[left + lane] = r;
  ktb_wave_sync();
  KtSplit o;
  o.cutfeat = cutfeat; o.cutval = cutval; o.index = index; o.dl = dl; o.dh = dh;
  return o;
}

__device__ __forceinline__ KtSplit ktb_split_node(float4* rec, unsigned short* sc, const KtWork& wk, const unsigned left,
                                                  const unsigned right, const int lane, const unsigned long long lt_mask) {
  const unsigned count = right - left;
  if (count <= 64u) return ktb_split_node_small(rec, sc, wk, left, count, lane, lt_mask);  // (wave-uniform)
  // ---- middleSplit_ (:966-1005)
  const float EPS = 0.00001f;
  float max_span = wk.box[1] - wk.box[0];
  for (int d = 1; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > max_span) max_span = span;
  }
  // computeMinMax (:898-907) of all three dimensions in ONE pass over the node's records (a 16-byte LDS read per point);
  // the reference evaluates only the dimensions whose span qualifies -- the selection below reads exactly those
  float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (unsigned p = lane; p < count; p += 64) {
    const float4 r = rec[left + p];
    const float c[3] = {r.x, r.y, r.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn3[d] = c[d] < mn3[d] ? c[d] : mn3[d];
      mx3[d] = c[d] > mx3[d] ? c[d] : mx3[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(mn3[d]); mx3[d] = wave_max_f32(mx3[d]); }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  // (selects, not wk.box[2 * cutfeat]: a dynamically indexed local array lives in scratch memory)
  const float blo = cutfeat == 0 ? wk.box[0] : (cutfeat == 1 ? wk.box[2] : wk.box[4]);
  const float bhi = cutfeat == 0 ? wk.box[1] : (cutfeat == 1 ? wk.box[3] : wk.box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;  // (the second computeMinMax of the reference, on cutfeat, returns mn_c / mx_c again)
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  const float* cutc = reinterpret_cast<const float*>(rec) + cutfeat;  // cutc[4 i] = the cut coordinate of record i
  // ---- planeSplit (:1016-1043): two passes, each the parallel form of the Hoare loop (header)
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    auto pred = [&](float v) { return pass == 0 ? v < cutval : v <= cutval; };
    unsigned cnt = 0;
    for (unsigned p0 = lo_p; p0 < count; p0 += 64) {
      const unsigned p = p0 + lane;
      cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p < count && pred(cutc[4 * (left + p)])));
    }
    const unsigned mid = lo_p + cnt;  // where the pointers meet
    unsigned nl = 0, nr = 0;
    for (unsigned p0 = lo_p; p0 < mid; p0 += 64) {  // violators among the first cnt positions, ascending
      const unsigned p = p0 + lane;
      const bool mis = p < mid && !pred(cutc[4 * (left + p)]);
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
      if (mis) sc[left + lo_p + nl + (unsigned)__builtin_popcountll(mk & lt_mask)] = (unsigned short)p;
      nl += (unsigned)__builtin_popcountll(mk);
    }
    for (unsigned q0 = 0; mid + q0 < count; q0 += 64) {  // satisfiers among the rest, descending
      const unsigned q = q0 + lane;
      const bool in = mid + q < count;
      const unsigned p = count - 1 - (in ? q : 0);
      const bool mis = in && pred(cutc[4 * (left + p)]);
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
      if (mis) sc[right - 1 - (nr + (unsigned)__builtin_popcountll(mk & lt_mask))] = (unsigned short)p;
      nr += (unsigned)__builtin_popcountll(mk);
    }
    ktb_wave_sync();
    for (unsigned i = lane; i < nl; i += 64) {  // nl == nr
      const unsigned a = left + sc[left + lo_p + i], b = left + sc[right - 1 - i];
      const float4 ra = rec[a], rb = rec[b];
      rec[a] = rb; rec[b] = ra;
    }
    ktb_wave_sync();
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  // ---- the children's tight boxes along cutfeat: divlow = max of the left part, divhigh = min of the right part (:956-957)
  float dl = -INFINITY, dh = INFINITY;
  for (unsigned p = lane; p < count; p += 64) {
    const float v = cutc[4 * (left + p)];
    if (p < index) dl = v > dl ? v : dl;
    else dh = v < dh ? v : dh;
  }
  dl = wave_max_f32(dl); dh = wave_min_f32(dh);
  KtSplit r;
  r.cutfeat = cutfeat; r.cutval = cutval; r.index = index; r.dl = dl; r.dh = dh;
  return r;
}

// A LARGE node split by the WHOLE workgroup (round 6).  One wave per node (above) walks a node of c points in c / 64 trips per pass
// and ten passes: the root of an 8192-point cloud alone took ~80 us of the first phase's 153, while fifteen waves waited for the
// level to end.  Here all KTB_WAVES waves work on the one node: min / max, the counts and the ranks of planeSplit's closed form
// (header of the parallel build) per wave CHUNK of consecutive positions -- chunk counts through LDS, violators ranked ascending
// from the chunks before, satisfiers descending from the chunks behind -- then the swaps and the children's tight bounds, all in
// strides of the workgroup.  The permutation is the same, element for element: the i-th violator among the first cnt positions
// (ascending) changes places with the i-th satisfier behind them (descending).  rec / sc: LDS or global memory (PosT: positions
// inside the node, 16 bits where a node has < 65536 points); red: >= 6 * KTB_WAVES words of LDS.
template <typename PosT>
__device__ __forceinline__ KtSplit ktb_split_node_wg(float4* rec, PosT* sc, float* red, const KtWork& wk, const unsigned left,
                                                     const unsigned right, const int tid) {
  constexpr int T = KTB_WAVES * 64;
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned count = right - left;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int* redi = reinterpret_cast<int*>(red);
  // ---- middleSplit_ (:966-1005); computeMinMax (:898-907) of all three dimensions in one pass
  const float EPS = 0.00001f;
  float max_span = wk.box[1] - wk.box[0];
  for (int d = 1; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (unsigned p = tid; p < count; p += T) {
    const float4 r = rec[left + p];
    const float c[3] = {r.x, r.y, r.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn3[d] = c[d] < mn3[d] ? c[d] : mn3[d];
      mx3[d] = c[d] > mx3[d] ? c[d] : mx3[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(mn3[d]); mx3[d] = wave_max_f32(mx3[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[wave * 6 + 2 * d] = mn3[d]; red[wave * 6 + 2 * d + 1] = mx3[d]; }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float l = red[2 * d], h = red[2 * d + 1];
    for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, red[w * 6 + 2 * d]); h = fmaxf(h, red[w * 6 + 2 * d + 1]); }
    mn3[d] = l; mx3[d] = h;
  }
  __syncthreads();  // (red is used again below)
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = wk.box[2 * d + 1] - wk.box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? wk.box[0] : (cutfeat == 1 ? wk.box[2] : wk.box[4]);
  const float bhi = cutfeat == 0 ? wk.box[1] : (cutfeat == 1 ? wk.box[3] : wk.box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  const float* cutc = reinterpret_cast<const float*>(rec) + cutfeat;  // cutc[4 i] = the cut coordinate of record i
  // ---- planeSplit (:1016-1043)
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    auto pred = [&](float v) { return pass == 0 ? v < cutval : v <= cutval; };
    // this wave's chunk of the positions [lo_p, count): a multiple of 64 positions, consecutive chunks for consecutive waves
    const unsigned span = count - lo_p;
    const unsigned chunk = ((span + T - 1) / T) * 64;
    const unsigned cs = min(count, lo_p + (unsigned)wave * chunk), ce = min(count, cs + chunk);
    unsigned c = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      c += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p < ce && pred(cutc[4 * (left + p)])));
    }
    if (lane == 0) redi[wave] = (int)c;
    __syncthreads();
    unsigned cnt = 0;
    for (int w = 0; w < KTB_WAVES; ++w) cnt += (unsigned)redi[w];
    const unsigned mid = lo_p + cnt;  // where the pointers meet
    __syncthreads();
    unsigned nv = 0, nr = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      const bool in = p < ce, sat = in && pred(cutc[4 * (left + p)]);
      nv += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in && p < mid && !sat));
      nr += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sat && p >= mid));
    }
    if (lane == 0) { redi[wave] = (int)nv; redi[KTB_WAVES + wave] = (int)nr; }
    __syncthreads();
    unsigned vbase = 0, rbase = 0, nl = 0;
    for (int w = 0; w < KTB_WAVES; ++w) {
      const unsigned v = (unsigned)redi[w], r = (unsigned)redi[KTB_WAVES + w];
      nl += v;
      if (w < wave) vbase += v;   // violators in the chunks before this one
      if (w > wave) rbase += r;   // satisfiers in the chunks behind this one
    }
    __syncthreads();
    unsigned runv = 0, runr = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      const bool in = p < ce, sat = in && pred(cutc[4 * (left + p)]);
      const bool viol = in && p < mid && !sat, rs = sat && p >= mid;
      const unsigned long long mv = __builtin_amdgcn_ballot_w64(viol), mr = __builtin_amdgcn_ballot_w64(rs);
      if (viol) sc[left + lo_p + vbase + runv + (unsigned)__builtin_popcountll(mv & lt_mask)] = (PosT)p;
      if (rs) {
        const unsigned asc = runr + (unsigned)__builtin_popcountll(mr & lt_mask);  // rank inside the chunk, ascending
        sc[right - 1 - (rbase + (nr - 1 - asc))] = (PosT)p;                        // descending over the whole node
      }
      runv += (unsigned)__builtin_popcountll(mv);
      runr += (unsigned)__builtin_popcountll(mr);
    }
    __syncthreads();
    for (unsigned i = tid; i < nl; i += T) {
      const unsigned a = left + (unsigned)sc[left + lo_p + i], b = left + (unsigned)sc[right - 1 - i];
      const float4 ra = rec[a], rb = rec[b];
      rec[a] = rb; rec[b] = ra;
    }
    __syncthreads();
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  // ---- the children's tight boxes along cutfeat: divlow = max of the left part, divhigh = min of the right part (:956-957)
  float dl = -INFINITY, dh = INFINITY;
  for (unsigned p = tid; p < count; p += T) {
    const float v = cutc[4 * (left + p)];
    if (p < index) dl = v > dl ? v : dl;
    else dh = v < dh ? v : dh;
  }
  dl = wave_max_f32(dl); dh = wave_min_f32(dh);
  if (lane == 0) { red[wave * 2] = dl; red[wave * 2 + 1] = dh; }
  __syncthreads();
  dl = red[0]; dh = red[1];
  for (int w = 1; w < KTB_WAVES; ++w) { dl = fmaxf(dl, red[w * 2]); dh = fminf(dh, red[w * 2 + 1]); }
  __syncthreads();
  KtSplit r;
  r.cutfeat = cutfeat; r.cutval = cutval; r.index = index; r.dl = dl; r.dh = dh;
  return r;
}
#ifndef KT_COOP_MIN_V
#define KT_COOP_MIN_V 4096  // measured (16 x 8192 self-kNN, 4 flagged clouds): 1024: 576 us, 2048: 518, 4096: 484, never: 505
#endif
constexpr unsigned KT_COOP_MIN = KT_COOP_MIN_V;
constexpr unsigned KT_COOP_FEW = 768;  // ... in a level of at most two nodes  // nodes above this many points are split by the whole workgroup, one after the other
// ... as a real CALL in the kernels that also hold the one-wave forms (inlined there, the three forms together need more than the 128
// registers a 1024-thread workgroup has: 140 bytes of scratch per lane, the deep kernel 132 -> 395 us); a large node is
// hundreds of trips: the call and the generic addressing of LDS do not show
__device__ __attribute__((noinline)) KtSplit ktb_split_node_wg_lds(float4* rec, unsigned short* sc, float* red, const float b0, const float b1,
                                                                   const float b2, const float b3, const float b4, const float b5,
                                                                   const unsigned left, const unsigned right, const int tid) {
  KtWork wk;
  wk.node = 0; wk.left = left; wk.right = right;
  wk.box[0] = b0; wk.box[1] = b1; wk.box[2] = b2; wk.box[3] = b3; wk.box[4] = b4; wk.box[5] = b5;
  return ktb_split_node_wg<unsigned short>(rec, sc, red, wk, left, right, tid);
}

__device__ void knn_tree_build_lds_body(int cloud, int n, const float* __restrict__ pts_all, char* __restrict__ ws_all, size_t stride,
                                        size_t recs_off, int stop_level) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* rec = reinterpret_cast<float4*>(smem);                             // [n] {x, y, z, index bits}: the points move with the index list
  unsigned short* sc = reinterpret_cast<unsigned short*>(rec + n);           // [n] positions of misplaced elements
  float* part = reinterpret_cast<float*>(sc + ((n + 1) & ~1));               // [KTB_WAVES][6] root-box partials
  int* ctr = reinterpret_cast<int*>(part + KTB_WAVES * 6);                   // [0], [1]: queue lengths; [2]: nodes used; [3]: flag; [4], [5]: large nodes in queue 0 / 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pts = pts_all + (size_t)cloud * n * 3;
  char* ws = ws_all + (size_t)cloud * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  unsigned* gvind = reinterpret_cast<unsigned*>(ws + kt_align(KT_HDR));
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  KtFrame* fr = reinterpret_cast<KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  const int qcap = n / (KT_LEAF + 1) + 2;
  KtWork* queue[2];
  queue[0] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(fr) + kt_align((size_t)KT_DEPTH * sizeof(KtFrame)));
  queue[1] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(queue[0]) + kt_align((size_t)qcap * sizeof(KtWork)));
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));  // lanes below this one

  // init_vind (:1318), computeBoundingBox (:1321-1346)
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += KTB_WAVES * 64) {
    const float c[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
    rec[i] = make_float4(c[0], c[1], c[2], __int_as_float(i));
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = c[d] < lo[d] ? c[d] : lo[d];
      hi[d] = c[d] > hi[d] ? c[d] : hi[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f32(lo[d]); hi[d] = wave_max_f32(hi[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { part[wave * 6 + 2 * d] = lo[d]; part[wave * 6 + 2 * d + 1] = hi[d]; }
  }
  if (tid < 6) ctr[tid] = 0;
  __syncthreads();
  float root[6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float l = part[2 * d], h = part[2 * d + 1];
    for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, part[w * 6 + 2 * d]); h = fmaxf(h, part[w * 6 + 2 * d + 1]); }
    root[2 * d] = l; root[2 * d + 1] = h;
  }
  if (tid == 0) {
    ctr[2] = 1;  // node 0 = the root
    if (n <= KT_LEAF) {
      nodes[0].child1 = nodes[0].child2 = -1; nodes[0].a = 0; nodes[0].divlow = __int_as_float(n); nodes[0].divhigh = 0.f;
    } else {
      KtWork w0; w0.node = 0; w0.left = 0; w0.right = (unsigned)n;
      for (int i = 0; i < 6; ++i) w0.box[i] = root[i];
      queue[0][0] = w0;
      ctr[0] = 1;
      ctr[4] = (unsigned)n > KT_COOP_MIN ? 1 : 0;
    }
    for (int i = 0; i < 6; ++i) fr[1].bbox[i] = root[i];  // root_bbox after divideTree = the tight box of all points (what the search reads)
  }
  __threadfence_block();
  __syncthreads();

  int cur = 0, level = 0;
  for (;;) {
    const int nq = ctr[cur];
    if (nq == 0) break;
    if (stop_level > 0 && level >= stop_level) break;  // the pending subtrees go to knn_tree_build_deep_kernel, one workgroup each
    if (level + 2 >= KT_DEPTH) { if (tid == 0) ctr[3] = 1; break; }  // deeper than the search's stack: flagged, not built
    // the children of a split node: ids, leaves, the next level's queue (one lane)
    auto emit = [&](const KtWork& wk, const KtSplit& sp_) {
      const unsigned left = wk.left, right = wk.right, index = sp_.index;
      const int cutfeat = sp_.cutfeat;
      const float cutval = sp_.cutval;
      int child[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned cl = c == 0 ? left : left + index, cr = c == 0 ? left + index : right;
        const int id = atomicAdd(&ctr[2], 1);
        child[c] = id;
        if (cr - cl <= (unsigned)KT_LEAF) {  // leaf (:921-936)
          nodes[id].child1 = nodes[id].child2 = -1;
          nodes[id].a = (int)cl;
          nodes[id].divlow = __int_as_float((int)cr);
          nodes[id].divhigh = 0.f;
        } else {
          KtWork w;
          w.node = id; w.left = cl; w.right = cr;
          // left child: high = cutval (:946-947); right child: low = cutval (:951-952)
#pragma unroll
          for (int i = 0; i < 6; ++i) w.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : wk.box[i];
          queue[cur ^ 1][atomicAdd(&ctr[cur ^ 1], 1)] = w;
          if (cr - cl > KT_COOP_MIN) atomicAdd(&ctr[4 + (cur ^ 1)], 1);
        }
      }
      KtNode nd;
      nd.child1 = child[0]; nd.child2 = child[1]; nd.a = cutfeat; nd.divlow = sp_.dl; nd.divhigh = sp_.dh;
      nodes[wk.node] = nd;
    };
    // a level of ONE or TWO nodes (the top of a tree, of a lopsided subtree) leaves fourteen waves idle while one walks a node in
    // count / 64 trips per pass: there the whole workgroup takes every node above KT_COOP_FEW points
    const unsigned coop_min = nq <= 2 ? KT_COOP_FEW : KT_COOP_MIN;
    if (ctr[4 + cur] != 0 || nq <= 2) {  // (uniform) the level's large nodes first, one after the other, by the whole workgroup
      for (int e = 0; e < nq; ++e) {
        const KtWork wk = kt_load_work(queue[cur] + e, lane);
        if (wk.right - wk.left <= coop_min) continue;
        const KtSplit sp_ = ktb_split_node_wg_lds(rec, sc, part, wk.box[0], wk.box[1], wk.box[2], wk.box[3], wk.box[4], wk.box[5], wk.left, wk.right, tid);
        if (tid == 0) emit(wk, sp_);
      }
    }
    for (int e = wave; e < nq; e += KTB_WAVES) {
      const KtWork wk = kt_load_work(queue[cur] + e, lane);
      const unsigned left = wk.left, right = wk.right;
      if (right - left > coop_min) continue;
      const KtSplit sp_ = ktb_split_node(rec, sc, wk, left, right, lane, lt_mask);
      if (lane == 0) emit(wk, sp_);
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) { ctr[cur] = 0; ctr[4 + cur] = 0; }
    cur ^= 1;
    ++level;
    __syncthreads();
  }
  __syncthreads();
  float4* recs = reinterpret_cast<float4*>(ws + recs_off);
  for (int i = tid; i < n; i += KTB_WAVES * 64) {  // the records are in leaf order already
    const float4 r = rec[i];
    gvind[i] = (unsigned)__float_as_int(r.w);
    recs[i] = r;
  }
  if (tid == 0) { hdr[0] = ctr[3]; hdr[1] = 0; hdr[2] = ctr[2]; hdr[3] = level; hdr[4] = ctr[3] ? 0 : ctr[cur]; hdr[5] = cur; hdr[6] = level; }
}
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_build_lds_kernel(int b, int n, const float* __restrict__ pts_all,
                                                                           char* __restrict__ ws_all, size_t stride, size_t recs_off, int stop_level,
                                                                           const int* __restrict__ nflag) {
  for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {  // (a capped grid walks the clouds: knn_tree_build_kernel's note)
    if (nflag && nflag[cloud] == 0) continue;
    knn_tree_build_lds_body(cloud, n, pts_all, ws_all, stride, recs_off, stop_level);
    __syncthreads();  // the next cloud re-uses the LDS
  }
}

// First phase for clouds that do not fit a workgroup's LDS (n > KTB_LDS_NMAX; round 6 -- before: a gathering build up to 10240
// points, 325 us for 8 lidar clouds, and ONE LANE beyond, ~1 s at 81920 points): the records live in the workspace, every node
// that is too large for the second phase (> KTD_LDSQ_MAX points) is split by the whole workgroup (ktb_split_node_wg on global
// memory), level by level; the rest is carried along.  What is left pending goes to knn_tree_build_deep_kernel, one workgroup per
// subtree with its records in LDS.  Scratch positions: 32 bits, in the index list's slot of the workspace (written last).
__device__ void knn_tree_build_big_body(int cloud, int n, const float* __restrict__ pts_all, char* __restrict__ ws_all, size_t stride,
                                        size_t recs_off) {
  __shared__ float part[KTB_WAVES * 6];
  __shared__ int ctr[8];  // [0], [1]: queue lengths; [2]: nodes used; [3]: flag; [4]: large nodes left
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pts = pts_all + (size_t)cloud * n * 3;
  char* ws = ws_all + (size_t)cloud * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  unsigned* sc = reinterpret_cast<unsigned*>(ws + kt_align(KT_HDR));  // [n] (the index list's slot)
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  KtFrame* fr = reinterpret_cast<KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  const int qcap = n / (KT_LEAF + 1) + 2;
  KtWork* queue[2];
  queue[0] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(fr) + kt_align((size_t)KT_DEPTH * sizeof(KtFrame)));
  queue[1] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(queue[0]) + kt_align((size_t)qcap * sizeof(KtWork)));
  float4* rec = reinterpret_cast<float4*>(ws + recs_off);
  // init_vind (:1318), computeBoundingBox (:1321-1346)
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += KTB_WAVES * 64) {
    const float c[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
    rec[i] = make_float4(c[0], c[1], c[2], __int_as_float(i));
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = c[d] < lo[d] ? c[d] : lo[d];
      hi[d] = c[d] > hi[d] ? c[d] : hi[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f32(lo[d]); hi[d] = wave_max_f32(hi[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { part[wave * 6 + 2 * d] = lo[d]; part[wave * 6 + 2 * d + 1] = hi[d]; }
  }
  if (tid < 8) ctr[tid] = 0;
  __syncthreads();
  if (tid == 0) {
    float root[6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float l = part[2 * d], h = part[2 * d + 1];
      for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, part[w * 6 + 2 * d]); h = fmaxf(h, part[w * 6 + 2 * d + 1]); }
      root[2 * d] = l; root[2 * d + 1] = h;
    }
    ctr[2] = 1;  // node 0 = the root
    KtWork w0; w0.node = 0; w0.left = 0; w0.right = (unsigned)n;
    for (int i = 0; i < 6; ++i) w0.box[i] = root[i];
    queue[0][0] = w0;
    ctr[0] = 1;
    ctr[4] = 1;  // n > KTB_LDS_NMAX > KTD_LDSQ_MAX
    for (int i = 0; i < 6; ++i) fr[1].bbox[i] = root[i];  // root_bbox after divideTree = the tight box of all points (what the search reads)
  }
  __threadfence_block();
  __syncthreads();
  int cur = 0, level = 0;
  // until every pending node fits a workgroup's LDS -- and for KTD_TOP levels at least: the second phase works a subtree off with
  // ONE workgroup, whose time is the thousands of small nodes at its deep levels (two subtrees of 5120 points: 309 us; sixteen of
  // 640: 40)
  while (ctr[4] != 0 || (level < KTD_TOP && ctr[cur] != 0)) {  // (uniform)
    const int nq = ctr[cur];
    if (level + 2 >= KT_DEPTH || ctr[2] + 2 * nq + 2 > KT_TOPIDS || 2 * nq > KTD_MAXWORK) { if (tid == 0) ctr[3] = 1; break; }  // flagged, not built
    __syncthreads();
    if (tid == 0) ctr[4] = 0;
    __syncthreads();
    for (int e = 0; e < nq; ++e) {
      const KtWork wk = kt_load_work(queue[cur] + e, lane);
      const unsigned left = wk.left, right = wk.right;
      if (level >= KTD_TOP && right - left <= (unsigned)KTD_LDSQ_MAX) {  // fits the second phase: carried to the next level as it is
        if (tid == 0) queue[cur ^ 1][atomicAdd(&ctr[cur ^ 1], 1)] = wk;
        continue;
      }
      const KtSplit sp_ = ktb_split_node_wg<unsigned>(rec, sc, part, wk, left, right, tid);
      if (tid == 0) {
        int child[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned cl = c == 0 ? left : left + sp_.index, cr = c == 0 ? left + sp_.index : right;
          const int id = atomicAdd(&ctr[2], 1);
          child[c] = id;
          if (cr - cl <= (unsigned)KT_LEAF) {  // leaf (:921-936)
            nodes[id].child1 = nodes[id].child2 = -1;
            nodes[id].a = (int)cl;
            nodes[id].divlow = __int_as_float((int)cr);
            nodes[id].divhigh = 0.f;
          } else {
            KtWork w;
            w.node = id; w.left = cl; w.right = cr;
#pragma unroll
            for (int i = 0; i < 6; ++i) w.box[i] = (i == 2 * sp_.cutfeat + 1 - c) ? sp_.cutval : wk.box[i];
            queue[cur ^ 1][atomicAdd(&ctr[cur ^ 1], 1)] = w;
            if (cr - cl > (unsigned)KTD_LDSQ_MAX) atomicAdd(&ctr[4], 1);
          }
        }
        KtNode nd;
        nd.child1 = child[0]; nd.child2 = child[1]; nd.a = sp_.cutfeat; nd.divlow = sp_.dl; nd.divhigh = sp_.dh;
        nodes[wk.node] = nd;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) ctr[cur] = 0;
    cur ^= 1;
    ++level;
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) { hdr[0] = ctr[3]; hdr[1] = 0; hdr[2] = ctr[2]; hdr[3] = level; hdr[4] = ctr[3] ? 0 : ctr[cur]; hdr[5] = cur; hdr[6] = level; }
}
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_build_big_kernel(int b, int n, const float* __restrict__ pts_all,
                                                                           char* __restrict__ ws_all, size_t stride, size_t recs_off,
                                                                           const int* __restrict__ nflag) {
  for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {
    if (nflag && nflag[cloud] == 0) continue;
    knn_tree_build_big_body(cloud, n, pts_all, ws_all, stride, recs_off);
    __syncthreads();
  }
}

// Second phase of the two-phase build: one workgroup per subtree the first phase left pending after KTD_TOP levels (<= 16 per
// cloud).  The deep levels are thousands of small nodes with ~3.5 us of fixed cost each; one workgroup per cloud worked them off
// 16 at a time on ONE CU -- here every subtree has its own workgroup (and CU), its records in LDS, its own node-id range
// (32 + 2 left ..: a subtree of c points has < 2 c nodes) and its level queues in LDS (or, for a subtree too large for that --
// then the only one of its size in the cloud -- in the workspace).
__device__ void knn_tree_build_deep_body(int cloud, int item, int n, char* __restrict__ ws_all, size_t stride, size_t recs_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* ws = ws_all + (size_t)cloud * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  if (item >= hdr[4]) return;
  unsigned* gvind = reinterpret_cast<unsigned*>(ws + kt_align(KT_HDR));
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  KtFrame* fr = reinterpret_cast<KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  const int qcap = n / (KT_LEAF + 1) + 2;
  KtWork* q1[2];
  q1[0] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(fr) + kt_align((size_t)KT_DEPTH * sizeof(KtFrame)));
  q1[1] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(q1[0]) + kt_align((size_t)qcap * sizeof(KtWork)));
  KtWork* g2[2];
  g2[0] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(q1[1]) + kt_align((size_t)qcap * sizeof(KtWork)));
  g2[1] = reinterpret_cast<KtWork*>(reinterpret_cast<char*>(g2[0]) + kt_align((size_t)(qcap + 3 * KTD_MAXWORK) * sizeof(KtWork)));
  float4* grecs = reinterpret_cast<float4*>(ws + recs_off);
  const KtWork w0 = q1[hdr[5]][item];  // written by the previous kernel
  const unsigned left0 = w0.left, count0 = w0.right - w0.left;

  float4* rec = reinterpret_cast<float4*>(smem);                                  // [count0]
  unsigned short* sc = reinterpret_cast<unsigned short*>(rec + count0);           // [count0]
  char* after = smem + (((size_t)count0 * 18 + 15) & ~(size_t)15);
  int* ctr = reinterpret_cast<int*>(after);                                       // [0], [1]: queue lengths; [2]: nodes used; [3]: flag; [4], [5]: large nodes in queue a / b
  float* red = reinterpret_cast<float*>(after + 32);                              // [KTB_WAVES][6] scratch of the workgroup-wide split
  const int lq = (int)(count0 / (KT_LEAF + 1)) + 2;
  const bool ldsq = count0 <= (unsigned)KTD_LDSQ_MAX;
  KtWork* const qa = ldsq ? reinterpret_cast<KtWork*>(after + 32 + KTB_WAVES * 6 * 4) : g2[0];  // (two named pointers: an indexed pair of pointers
  KtWork* const qb = ldsq ? qa + lq : g2[1];                                //  into different address spaces would live in scratch)
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (unsigned i = tid; i < count0; i += KTB_WAVES * 64) rec[i] = grecs[left0 + i];
  if (tid < 6) ctr[tid] = 0;
  __syncthreads();
  if (tid == 0) {
    KtWork w = w0;
    w.left = 0; w.right = count0;
    qa[0] = w;
    ctr[0] = 1;
    ctr[4] = count0 > KT_COOP_MIN ? 1 : 0;
  }
  __threadfence_block();
  __syncthreads();
  const int idbase = KT_TOPIDS + 2 * (int)left0;
  int cur = 0, level = hdr[6];  // the level the first phase stopped at
  for (;;) {
    const int nq = ctr[cur];
    if (nq == 0) break;
    if (level + 2 >= KT_DEPTH) { if (tid == 0) ctr[3] = 1; break; }
    auto emit = [&](const KtWork& wk, const KtSplit& sp_) {
      const unsigned left = wk.left, right = wk.right, index = sp_.index;
      const int cutfeat = sp_.cutfeat;
      const float cutval = sp_.cutval;
      int child[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned cl = c == 0 ? left : left + index, cr = c == 0 ? left + index : right;
        const int id = idbase + atomicAdd(&ctr[2], 1);
        child[c] = id;
        if (cr - cl <= (unsigned)KT_LEAF) {  // leaf (:921-936): positions in the CLOUD's leaf order
          nodes[id].child1 = nodes[id].child2 = -1;
          nodes[id].a = (int)(left0 + cl);
          nodes[id].divlow = __int_as_float((int)(left0 + cr));
          nodes[id].divhigh = 0.f;
        } else {
          KtWork w;
          w.node = id; w.left = cl; w.right = cr;
#pragma unroll
          for (int i = 0; i < 6; ++i) w.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : wk.box[i];
          (cur ? qa : qb)[atomicAdd(&ctr[cur ^ 1], 1)] = w;
          if (cr - cl > KT_COOP_MIN) atomicAdd(&ctr[4 + (cur ^ 1)], 1);
        }
      }
      KtNode nd;
      nd.child1 = child[0]; nd.child2 = child[1]; nd.a = cutfeat; nd.divlow = sp_.dl; nd.divhigh = sp_.dh;
      nodes[wk.node] = nd;
    };
    const unsigned coop_min = nq <= 2 ? KT_COOP_FEW : KT_COOP_MIN;  // (knn_tree_build_lds_body's note)
    if (ctr[4 + cur] != 0 || nq <= 2) {  // (uniform) large nodes: the whole workgroup, one node after the other
      for (int e = 0; e < nq; ++e) {
        const KtWork wk = kt_load_work((cur ? qb : qa) + e, lane);
        if (wk.right - wk.left <= coop_min) continue;
        const KtSplit sp_ = ktb_split_node_wg_lds(rec, sc, red, wk.box[0], wk.box[1], wk.box[2], wk.box[3], wk.box[4], wk.box[5], wk.left, wk.right, tid);
        if (tid == 0) emit(wk, sp_);
      }
    }
    for (int e = wave; e < nq; e += KTB_WAVES) {
      const KtWork wk = kt_load_work((cur ? qb : qa) + e, lane);
      const unsigned left = wk.left, right = wk.right;
      if (right - left > coop_min) continue;
      const KtSplit sp_ = ktb_split_node(rec, sc, wk, left, right, lane, lt_mask);
      if (lane == 0) emit(wk, sp_);
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) { ctr[cur] = 0; ctr[4 + cur] = 0; }
    cur ^= 1;
    ++level;
    __syncthreads();
  }
  __syncthreads();
  for (unsigned i = tid; i < count0; i += KTB_WAVES * 64) {
    const float4 r = rec[i];
    gvind[left0 + i] = (unsigned)__float_as_int(r.w);
    grecs[left0 + i] = r;
  }
  if (tid == 0) {
    if (ctr[3]) atomicExch(&hdr[0], 1);
    atomicMax(&hdr[3], level);
  }
}
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_build_deep_kernel(int b, int n, char* __restrict__ ws_all, size_t stride,
                                                                            size_t recs_off, const int* __restrict__ nflag) {
  // The pending subtrees of all clouds, numbered cloud after cloud, dealt round-robin to the workgroups of a (capped) grid.  (A
  // fixed (cloud, slot) -> workgroup map put subtree i of EVERY cloud on workgroup i: four flagged clouds worked their subtrees
  // off four at a time on 16 of the 64 workgroups, 172 us where one round takes 40.)
  for (int g = blockIdx.x;; g += gridDim.x) {
    int cloud = -1, item = 0, base = 0;
    for (int c = 0; c < b; ++c) {  // (b <= a few dozen header reads; uniform)
      if (nflag && nflag[c] == 0) continue;  // (the first phase did not run: the header is stale)
      const int cnt = reinterpret_cast<const int*>(ws_all + (size_t)c * stride)[4];
      if (g < base + cnt) { cloud = c; item = g - base; break; }
      base += cnt;
    }
    if (cloud < 0) return;
    knn_tree_build_deep_body(cloud, item, n, ws_all, stride, recs_off);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The search: one lane per query, nanoflann's searchLevel (:1351-1410) as a FLAT state machine.  Sixty-four lanes walk
// sixty-four different paths; written as the recursion reads (descend loop, leaf loop with an insertion loop inside, return
// loop) a wave executes every nested loop for its slowest lane while the others wait (measured: 2.6 ms for 512 queries per
// cloud).  Here every lane performs bounded micro-steps per iteration -- examine one leaf candidate, pop one frame, visit one
// node -- so an iteration costs the sum of three short code paths and all lanes advance.
//   * The result set is nanoflann's KNNResultSet (:67-135) in another representation: the k smallest (distance, arrival)
//     keys seen so far, UNSORTED, plus the largest of them.  addPoint's sorted insertion "behind the entries of equal
//     distance" keeps exactly those k keys in that order, and a candidate enters iff its distance is below the current
//     k-th (the reference tests the value it read at the leaf's entry, then its insertion drops what the fresh value would
//     have refused).  The new entry takes the slot of the old maximum; the new maximum is found per group of eight slots
//     (independent LDS reads; only the group that changed is rescanned) where the reference's shifting loop is a chain of
//     dependent ones.  The set is sorted once, at the end, by (distance, arrival).
//   * Result set and stack live in LDS (slot-major / lane-minor); every global read (the node a lane descends to, a leaf's
//     records in leaf order as the build left them) is requested one iteration before it is used.  A frame is three words:
//     word 0 = other child | feat << 28 | state << 30,  word 1 = mindistsq,  word 2 = cut (state 1) / the saved dists[feat] (2).
//   k <= 64 and n <= 65535 (16-bit arrival numbers and indices); frames beyond KT_LDS_DEPTH (pathological trees) in scratch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KT_LDS_DEPTH = 32;
constexpr float KT_FLT_MAX = 3.402823466e+38f;

template <typename IdxT>
__device__ void knn_tree_search_body(int bi, int blk, int n, int m, int k, const float* __restrict__ queries,
                                     const char* __restrict__ ws_all, size_t stride, size_t recs_off, IdxT* __restrict__ out,
                                     int* __restrict__ flag, const int* __restrict__ nflag, const int* __restrict__ flist) {
  // every query of the cloud, or (pasnl_knn_batch_ref) only the ones the canonical kernels flagged: entry `slot` of the cloud's list
  const int slot = blk * 64 + threadIdx.x;
  const int nq = nflag ? nflag[bi] : m;
  const bool live = slot < nq;
  const int j = nflag ? flist[(size_t)bi * m + (live ? slot : 0)] : slot;
  const char* ws = ws_all + (size_t)bi * stride;
  const int* hdr = reinterpret_cast<const int*>(ws);
  if (hdr[0] != 0) { if (slot == 0) atomicExch(flag, 1); return; }  // (the rows keep what was in them: the canonical order under _ref)
  const KtNode* nodes = reinterpret_cast<const KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
  const KtFrame* bst = reinterpret_cast<const KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
  const float4* recs = reinterpret_cast<const float4*>(ws + recs_off);
  const float* rootbox = bst[1].bbox;
  const float* qp = queries + ((size_t)bi * m + (live ? j : 0)) * 3;
  const float vec[3] = {qp[0], qp[1], qp[2]};
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* stk = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;                       // [KT_LDS_DEPTH][3][64]
  const int kp = (k + 7) & ~7;
  float* rd = reinterpret_cast<float*>(smem) + KT_LDS_DEPTH * 3 * 64 + threadIdx.x;      // [kp][64] distances
  uint32_t* rk = reinterpret_cast<uint32_t*>(rd) + (size_t)kp * 64;                      // [kp][64] arrival << 16 | index
  for (int s0 = k; s0 < kp; ++s0) { rd[s0 * 64] = 0.f; rk[s0 * 64] = 0u; }                // padding: key 0 never is the maximum
  uint32_t deep[(KT_DEPTH - KT_LDS_DEPTH) * 3];  // frames past the LDS part (never touched by sane trees)
  // (two address spaces: a reference that may point to either would turn every stack access into a flat load)
  auto frd = [&](int level, int word) -> uint32_t {
    if (level < KT_LDS_DEPTH) return stk[(level * 3 + word) * 64];
    return deep[(level - KT_LDS_DEPTH) * 3 + word];
  };
  auto fwr = [&](int level, int word, uint32_t v) {
    if (level < KT_LDS_DEPTH) stk[(level * 3 + word) * 64] = v;
    else deep[(level - KT_LDS_DEPTH) * 3 + word] = v;
  };
  // computeInitialDistances (:1045-1061)
  float dists[3] = {0.f, 0.f, 0.f};
  float distsq = 0.f;
  for (int d = 0; d < 3; ++d) {
    if (vec[d] < rootbox[2 * d]) { dists[d] = (vec[d] - rootbox[2 * d]) * (vec[d] - rootbox[2 * d]); distsq += dists[d]; }
    if (vec[d] > rootbox[2 * d + 1]) { dists[d] = (vec[d] - rootbox[2 * d + 1]) * (vec[d] - rootbox[2 * d + 1]); distsq += dists[d]; }
  }
  const float epsError = 1.f;  // 1 + SearchParams(10).eps, eps = 0
  enum { NODE = 0, LEAF = 1, POP = 2, DONE = 3 };
  int mode = live ? NODE : DONE;
  int node = hdr[1];
  float mindistsq = distsq;      // of the activation that is running
  int sp = 0;                    // frames below it
  int count = 0, maxpos = 0;     // entries in the set; the slot of its largest (distance, arrival) once it is full
  uint32_t arrivals = 0;         // candidates that entered so far
  float worst = KT_FLT_MAX;      // the k-th distance: dists[capacity - 1] = max until the set is full (:91-92)
  int li = 0, lend = 0;          // leaf cursor
  float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);  // the candidate at li (requested one step ahead)
  bool overflow = false, filled = false;
  unsigned long long gmax[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};  // per group of eight slots: its largest key ...
  int gpos[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                        // ... and that key's slot
  KtNode ndc = nodes[node];      // the node a descending lane is at (requested when the step that chose it ended)
  // One iteration: a candidate for the lanes inside a leaf, then one frame for the lanes that are returning (a lane that has
  // just finished its leaf among them), then one node for the lanes that are descending (a lane whose pop has just sent it
  // into a far child among them): the transitions leaf -> return -> descend cost no extra iteration.
  while (__any(mode != DONE)) {
    if (mode == LEAF) {
      const float4 c = rec;
      rec = recs[min(li + 1, n - 1)];  // the next candidate, whatever happens to this one
      // L2_Adaptor::evalMetric, dim 3: only the tail loop runs (:343-346): result += diff * diff, diff = query - point
      float dist = 0.f;
      { const float diff = vec[0] - c.x; dist += diff * diff; }
      { const float diff = vec[1] - c.y; dist += diff * diff; }
      { const float diff = vec[2] - c.z; dist += diff * diff; }
      if (dist < worst) {  // KNNResultSet::addPoint (:115-134), see the header
        const int slot = count < k ? count : maxpos;
        rd[slot * 64] = dist;
        rk[slot * 64] = (arrivals << 16) | (uint32_t)__float_as_int(c.w);
        ++arrivals;
        if (count < k) ++count;
        if (count == k) {
          // the largest (distance, arrival) of the full set, kept per group of eight slots (registers): the new entry took the
          // place of the old maximum, so only THAT group has changed (the first time the set is full every group is scanned)
          const int g_lo = filled ? (slot >> 3) : 0, g_hi = filled ? (slot >> 3) + 1 : (kp >> 3);
          for (int g = g_lo; g < g_hi; ++g) {
            uint32_t dk[8], ak[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dk[u] = __float_as_uint(rd[(g * 8 + u) * 64]); ak[u] = rk[(g * 8 + u) * 64]; }
            unsigned long long big = 0ull;
            int bp = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const unsigned long long key = ((unsigned long long)dk[u] << 32) | ak[u];
              if (key >= big) { big = key; bp = g * 8 + u; }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (t == g) { gmax[t] = big; gpos[t] = bp; }
          }
          filled = true;
          unsigned long long big = gmax[0];
          int bp = gpos[0];
#pragma unroll
          for (int t = 1; t < 8; ++t)
            if (gmax[t] >= big) { big = gmax[t]; bp = gpos[t]; }
          maxpos = bp;
          worst = __uint_as_float((uint32_t)(big >> 32));
        }
      }
      ++li;
      if (li >= lend) mode = POP;
    }
    if (mode == POP) {
      if (sp == 0) mode = DONE;
      else {
        const uint32_t w0 = frd(sp - 1, 0);
        const int feat = (int)((w0 >> 28) & 3u);
        if ((w0 >> 30) == 1u) {  // the near child is done (:1397-1405)
          const float fmind = __uint_as_float(frd(sp - 1, 1)), cut = __uint_as_float(frd(sp - 1, 2));
          const float dst = feat == 0 ? dists[0] : (feat == 1 ? dists[1] : dists[2]);
          const float mind = fmind + cut - dst;
          if (mind * epsError <= worst) {
            if (feat == 0) dists[0] = cut; else if (feat == 1) dists[1] = cut; else dists[2] = cut;
            fwr(sp - 1, 0, (w0 & 0x3FFFFFFFu) | (2u << 30));
            fwr(sp - 1, 2, __float_as_uint(dst));
            node = (int)(w0 & 0x0FFFFFFFu);
            ndc = nodes[node];
            mindistsq = mind;
            mode = NODE;
          } else {
            --sp;  // (dists[idx] = cut; ... dists[idx] = dst: unchanged)
          }
        } else {  // the far child is done: dists[idx] = dst (:1404)
          const float dst = __uint_as_float(frd(sp - 1, 2));
          if (feat == 0) dists[0] = dst; else if (feat == 1) dists[1] = dst; else dists[2] = dst;
          --sp;
        }
      }
    }
    if (mode == NODE) {
      const KtNode nd = ndc;
      if (nd.child1 < 0) {  // leaf (:1355-1369)
        li = nd.a;
        lend = __float_as_int(nd.divlow);
        rec = recs[li];
        mode = li < lend ? LEAF : POP;
      } else {
        const int idx = nd.a;
        const float val = idx == 0 ? vec[0] : (idx == 1 ? vec[1] : vec[2]);
        const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int best, other;
        float cut;
        if ((diff1 + diff2) < 0) { best = nd.child1; other = nd.child2; cut = (val - nd.divhigh) * (val - nd.divhigh); }
        else { best = nd.child2; other = nd.child1; cut = (val - nd.divlow) * (val - nd.divlow); }
        if (sp + 1 >= KT_DEPTH) { overflow = true; mode = DONE; }
        else {
          fwr(sp, 0, (uint32_t)other | ((uint32_t)idx << 28) | (1u << 30));
          fwr(sp, 1, __float_as_uint(mindistsq));
          fwr(sp, 2, __float_as_uint(cut));
          ++sp;
          node = best;  // (the near child inherits mindistsq, :1396)
          ndc = nodes[best];
        }
      }
    }
  }
  if (overflow) atomicExch(flag, 1);
  if (!live || overflow) return;  // (an overflowing search writes nothing: under _ref the row keeps the canonical order)
  // the set in the reference's order: ascending (distance, arrival).  k <= n: the set is full.
  IdxT* o = out + ((size_t)bi * m + j) * k;
  for (int s0 = k; s0 < kp; ++s0) { rd[s0 * 64] = __uint_as_float(0xFFFFFFFFu); rk[s0 * 64] = 0xFFFFFFFFu; }  // padding: never the minimum
  for (int r = 0; r < k; ++r) {
    unsigned long long best = ~0ull;
    int bp = 0;
    for (int s0 = 0; s0 < kp; s0 += 8) {
      uint32_t dk[8], ak[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { dk[u] = __float_as_uint(rd[(s0 + u) * 64]); ak[u] = rk[(s0 + u) * 64]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long key = ((unsigned long long)dk[u] << 32) | ak[u];
        if (key < best) { best = key; bp = s0 + u; }
      }
    }
    o[r] = (IdxT)(uint32_t)(best & 0xFFFFu);
    rd[bp * 64] = __uint_as_float(0xFFFFFFFFu);  // taken (no distance has these bits: not even NaN keys compare below ~0)
    rk[bp * 64] = 0xFFFFFFFFu;
  }
}
// one wave per workgroup; a capped grid walks the (cloud, block of 64 queries) pairs (knn_tree_build_kernel's note)
template <typename IdxT>
__global__ __launch_bounds__(64) void knn_tree_search_kernel(int b, int n, int m, int k, const float* __restrict__ queries,
                                                            const char* __restrict__ ws_all, size_t stride, size_t recs_off,
                                                            IdxT* __restrict__ out, int* __restrict__ flag,
                                                            const int* __restrict__ nflag, const int* __restrict__ flist) {
  const int mblk = (m + 63) / 64;
  for (int w = blockIdx.x; w < b * mblk; w += gridDim.x) {
    const int bi = w / mblk, blk = w - bi * mblk;
    if (blk * 64 >= (nflag ? nflag[bi] : m)) continue;
    knn_tree_search_body<IdxT>(bi, blk, n, m, k, queries, ws_all, stride, recs_off, out, flag, nflag, flist);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The search for FEW queries (pasnl_knn_batch_ref's flagged ones): ONE WAVE PER QUERY.  The lane-per-query kernel above is
// the throughput form -- a lone query walks ~250 dependent steps in one lane, 130 us for K = 32 in 1024 points, 0.5 ms in
// 8192, which is the whole price of a single chance tie in a batch.  Here the walk itself is wave-uniform (node, stack,
// running side distances: one copy, the stack in LDS), a leaf's <= 10 candidates are evaluated by ten lanes at once against the
// worst distance read at the leaf's entry (nanoflann.hpp:1357-1368 reads it once per leaf too), and the result set is
// nanoflann's sorted list itself, rank l in lane l: a candidate's place is the number of entries with distance <= its own (ballot
// + popcount: behind its equals, :115-134), the tail shifts by one lane (DPP).  Dependent steps: one per node and one per leaf.
// Tree: node / record accessors over global memory (the builds above) or LDS (knn_tree_small_kernel below).
// ---------------------------------------------------------------------------------------------------------------------
template <typename IdxT, int SLOTS, typename NodeP, typename RecP>
__device__ __forceinline__ bool knn_tree_search_wave(const float qx, const float qy, const float qz, const int k, const int root,
                                                     NodeP nodes, RecP recs, const float* rootbox, uint32_t* stk /* [KT_DEPTH * 3] */,
                                                     IdxT* __restrict__ orow, const int lane) {
  const float vec[3] = {qx, qy, qz};
  // computeInitialDistances (:1045-1061)
  float dists[3] = {0.f, 0.f, 0.f};
  float distsq = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (vec[d] < rootbox[2 * d]) { dists[d] = (vec[d] - rootbox[2 * d]) * (vec[d] - rootbox[2 * d]); distsq += dists[d]; }
    if (vec[d] > rootbox[2 * d + 1]) { dists[d] = (vec[d] - rootbox[2 * d + 1]) * (vec[d] - rootbox[2 * d + 1]); distsq += dists[d]; }
  }
  const float epsError = 1.f;
  // the sorted list: rank 64 s + lane in slot s of lane `lane` (K <= 64: one slot).  KNNResultSet::init: dists[capacity - 1] = max (:84-90)
  float ld[SLOTS];
  int li[SLOTS];
#pragma unroll
  for (int s0 = 0; s0 < SLOTS; ++s0) { ld[s0] = KT_FLT_MAX; li[s0] = 0; }
  const int ks = (k - 1) >> 6, kl = (k - 1) & 63;  // slot / lane of the K-th entry
  float worst = KT_FLT_MAX;
  int node = root, sp = 0;
  float mindistsq = distsq;
  bool descending = true;
  for (;;) {
    if (descending) {
      const KtNode nd = nodes[node];
      if (nd.child1 < 0) {  // leaf (:1355-1369): every point against the worst distance as it is NOW
        const int left = nd.a, right = __float_as_int(nd.divlow);
        for (int p0 = left; p0 < right; p0 += 64) {  // (<= KT_LEAF points: one trip)
          const bool in = p0 + lane < right;
          const float4 c = recs[in ? p0 + lane : left];
          float dist = 0.f;  // L2_Adaptor::evalMetric, dim 3: the tail loop (:343-346)
          { const float diff = vec[0] - c.x; dist += diff * diff; }
          { const float diff = vec[1] - c.y; dist += diff * diff; }
          { const float diff = vec[2] - c.z; dist += diff * diff; }
          unsigned long long mask = __builtin_amdgcn_ballot_w64(in && dist < worst);
          while (mask != 0ull) {
            const int src = (int)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            const float cd = readlane_f(dist, src);
            const int ci = __builtin_amdgcn_readlane(__float_as_int(c.w), src);
            // addPoint (:115-134): behind every entry with distance <= cd; beyond the capacity: dropped
            int pos = 0;
#pragma unroll
            for (int s0 = 0; s0 < SLOTS; ++s0) pos += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(s0 * 64 + lane < k && ld[s0] <= cd));
            if (pos < k) {
              float carry_d = 0.f;
              int carry_i = 0;
#pragma unroll
              for (int s0 = 0; s0 < SLOTS; ++s0) {  // ranks above pos move up by one; lane 63 of a slot carries into lane 0 of the next
                const float top_d = readlane_f(ld[s0], 63);
                const int top_i = __builtin_amdgcn_readlane(li[s0], 63);
                float sd = wave_shr1_f(ld[s0]);
                int si = wave_shr1_i(li[s0]);
                if (lane == 0) { sd = carry_d; si = carry_i; }
                const int r = s0 * 64 + lane;
                if (r > pos) { ld[s0] = sd; li[s0] = si; }
                if (r == pos) { ld[s0] = cd; li[s0] = ci; }
                carry_d = top_d; carry_i = top_i;
              }
            }
          }
        }
        {
          float t = KT_FLT_MAX;
#pragma unroll
          for (int s0 = 0; s0 < SLOTS; ++s0)
            if (s0 == ks) t = readlane_f(ld[s0], kl);
          worst = t;
        }
        descending = false;
      } else {
        const int idx = nd.a;
        const float val = idx == 0 ? vec[0] : (idx == 1 ? vec[1] : vec[2]);
        const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int best, other;
        float cut;
        if ((diff1 + diff2) < 0) { best = nd.child1; other = nd.child2; cut = (val - nd.divhigh) * (val - nd.divhigh); }
        else { best = nd.child2; other = nd.child1; cut = (val - nd.divlow) * (val - nd.divlow); }
        if (sp + 1 >= KT_DEPTH) return false;
        stk[sp * 3] = (uint32_t)other | ((uint32_t)idx << 28) | (1u << 30);
        stk[sp * 3 + 1] = __float_as_uint(mindistsq);
        stk[sp * 3 + 2] = __float_as_uint(cut);
        ++sp;
        node = best;  // (the near child inherits mindistsq, :1396)
      }
    } else {
      if (sp == 0) break;
      const uint32_t w0 = stk[(sp - 1) * 3];
      const int feat = (int)((w0 >> 28) & 3u);
      if ((w0 >> 30) == 1u) {  // the near child is done (:1397-1405)
        const float fmind = __uint_as_float(stk[(sp - 1) * 3 + 1]), cut = __uint_as_float(stk[(sp - 1) * 3 + 2]);
        const float dst = feat == 0 ? dists[0] : (feat == 1 ? dists[1] : dists[2]);
        const float mind = fmind + cut - dst;
        if (mind * epsError <= worst) {
          if (feat == 0) dists[0] = cut; else if (feat == 1) dists[1] = cut; else dists[2] = cut;
          stk[(sp - 1) * 3] = (w0 & 0x3FFFFFFFu) | (2u << 30);
          stk[(sp - 1) * 3 + 2] = __float_as_uint(dst);
          node = (int)(w0 & 0x0FFFFFFFu);
          mindistsq = mind;
          descending = true;
        } else {
          --sp;
        }
      } else {  // the far child is done: dists[idx] = dst (:1404)
        const float dst = __uint_as_float(stk[(sp - 1) * 3 + 2]);
        if (feat == 0) dists[0] = dst; else if (feat == 1) dists[1] = dst; else dists[2] = dst;
        --sp;
      }
    }
  }
#pragma unroll
  for (int s0 = 0; s0 < SLOTS; ++s0)
    if (s0 * 64 + lane < k) orow[s0 * 64 + lane] = (IdxT)li[s0];
  return true;
}

// waves of a capped grid walk the flagged queries of all clouds (global tree: the builds above)
constexpr int KTW_WAVES = 4;
template <typename IdxT, int SLOTS>
__global__ __launch_bounds__(KTW_WAVES * 64) void knn_tree_search_wave_kernel(int b, int n, int m, int k, const float* __restrict__ queries,
                                                                             const char* __restrict__ ws_all, size_t stride, size_t recs_off,
                                                                             IdxT* __restrict__ out, int* __restrict__ flag,
                                                                             const int* __restrict__ nflag, const int* __restrict__ flist) {
  __shared__ uint32_t stacks[KTW_WAVES][KT_DEPTH * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * KTW_WAVES + wave, nw = (long)gridDim.x * KTW_WAVES;
  long base = 0;  // flagged queries of the clouds before `bi`
  for (int bi = 0; bi < b; ++bi) {
    const int nq = nflag ? nflag[bi] : m;  // (no list: every query of every cloud -- the every-query mode beyond the lane search's limits)
    if (nq == 0) continue;
    const char* ws = ws_all + (size_t)bi * stride;
    const int* hdr = reinterpret_cast<const int*>(ws);
    const KtNode* nodes = reinterpret_cast<const KtNode*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4));
    const KtFrame* bst = reinterpret_cast<const KtFrame*>(ws + kt_align(KT_HDR) + kt_align((size_t)n * 4) + kt_align((size_t)KT_NNODES(n) * sizeof(KtNode)));
    const float4* recs = reinterpret_cast<const float4*>(ws + recs_off);
    // wave gw takes the entries e of this cloud with (base + e) % nw == gw
    long e = (gw - base % nw + nw) % nw;
    for (; e < nq; e += nw) {
      if (hdr[0] != 0) { if (lane == 0) atomicExch(flag, 1); break; }  // the tree is not complete: the rows keep the canonical order
      const int j = nflag ? flist[(size_t)bi * m + e] : (int)e;
      const float* qp = queries + ((size_t)bi * m + j) * 3;
      if (!knn_tree_search_wave<IdxT, SLOTS>(qp[0], qp[1], qp[2], k, hdr[1], nodes, recs, bst[1].bbox, stacks[wave],
                                      out + ((size_t)bi * m + j) * k, lane) && lane == 0)
        atomicExch(flag, 1);
    }
    base += nq;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small clouds (n <= KTS_NMAX: the classifier's, the deep levels of the segmentation models): tree AND searches of a flagged
// cloud in ONE workgroup, everything in LDS -- records, nodes, level queues, the searching waves' stacks.  One launch instead of
// three, no global round trip inside the build's level loop or the walk: a listed cloud costs 85 us here where build + deep +
// lane search took 184.  The build is knn_tree_build_lds_kernel's (same split code).  Its first stage (the kernel is further
// down, behind the tie paths): a cloud with a FEW listed queries never gets here -- ktp_resolve_cloud / ktp_descend_records put their
// runs of equal distances in arrival order from the tree nodes that separate the tied points alone (~12 us for a chance tie).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KTS_NMAX = 2048;
#define KTS_NNODES(n) (2 * (n) + 32)  // (one id range: the whole tree is built by one workgroup)
#ifdef PASNL_TUNING
// phase probe of the FIRST flagged cloud a workgroup takes (tools/knn_small_probe.py): s_memtime at [0] entry, [1] records + box in
// LDS, [2 + l] level l done (l < 20), [30] build done, [31] searches done
__device__ unsigned long long kts_probe[32];
#define KTS_MARK(i) do { if (tid == 0 && blockIdx.x == kts_first) kts_probe[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KTS_MARK(i) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------------------------------
// Tie paths (round 6): what the tree search DOES to a listed query's row, without the search.
//
// nanoflann's result set is a stable insertion sort cut at K (KNNResultSet::addPoint :115-134: a candidate goes behind every
// entry at the same distance; what falls off the end is gone), fed with the points in the order the walk reaches them; a
// subtree the walk prunes holds only points farther than the worst entry of that moment, which could not enter anyway.  So the
// row the reference returns is: all points sorted by (distance, ARRIVAL), first K -- the arrival order being the depth-first
// order of the whole tree with the query's near child first at every node, positions left to right inside a leaf.  The
// canonical row (distance, index) already has every entry whose distance is unique in its final place; only the entries of a
// run of EQUAL distances (and, for a run that reaches the K-th distance, the points outside the row at that distance) have
// to be put in arrival order.  Two points arrive in the order the tree decides at the node that separates them: the one in
// the query's near child first; never separated, the one at the lower position of their leaf first.
// Hence: one workgroup per listed query holds the cloud's records in LDS (the LDS build's layout), finds the tied points by
// one pass over the cloud (the same canonical fp32 distance, bit for bit), and splits ONLY the nodes that still hold two
// points of one run -- middleSplit_ + planeSplit by the whole workgroup (ktb_split_node_wg) or by one wave (<= 64 points),
// the same code as the builds, so these nodes are the reference tree's nodes.  Each tied point collects one bit per level
// (0: it went to the query's near child); (bits, final position) sorts a run into arrival order.  A chance tie between two
// unrelated points separates after ~2 splits: ~10 us for a 1024-point cloud where tree + search took 85, ~40 for an
// 8192-point one (225 + the search).  More than KTP_MAXQ listed queries in the batch, more than KTP_TMAX tied points for a
// query (lattices, padded clouds), more than KTP_MAXW splits: the cloud goes to the full build and search below (`nwork`).
// ---------------------------------------------------------------------------------------------------------------------
#ifdef PASNL_TUNING
// phase probe of workgroup 0 (tools/tie_path_probe.py): s_memtime at [0] entry, [1] the listed queries counted, [2] records + box in
// LDS, [3] the tied points listed, [4 + i] split i done (i < 24), [30] the row written
#define KTP_MARK(i) do { if (threadIdx.x == 0 && S->probe != 0) kts_probe[i] = __builtin_amdgcn_s_memtime(); } while (0)  // (S->probe: LDS; a flag in global memory costs a mark ~1 us)
__device__ unsigned long long ktp_wg_cycles[64];  // (knn_tie_path_kernel: cycles and final code of each listed cloud's workgroup, tools/tie_path_l2.py)
__device__ int ktp_paths[8];       // clouds by the form that took them: [0] sets only, [1] records moved, [2] tree + search, [3] return code 1 of the set form, [4] code 2
#define KTP_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&ktp_paths[i], 1); } while (0)
#else
#define KTP_MARK(i) do { } while (0)
#define KTP_COUNT(i) do { } while (0)
#endif
constexpr int KTP_MAXQ = 32;    // listed queries per batch this form takes (a workgroup each)
constexpr int KTP_TMAX = 64;    // tied points of one query (a lane each)
constexpr int KTP_MAXW = 32;    // nodes split for one cloud's listed queries
constexpr int KTP_DEPTH = 126;  // levels whose near / far bit fits the 128-bit key
constexpr int KTP_RED_WORDS = 384;  // [0, 96) min / max partials, [96, 352) the waves' ballots (ktp_split_node_1k: 2 x 16 x 64 bits, _8k: 128 x 64 bits), then 32 divlow / divhigh partials
struct KtpWork { unsigned left, right; float box[6]; int depth; };
constexpr int KTP_FEWQ = 16;    // listed queries of one cloud resolved together, at most (duplicated points list a dozen queries around them) ...
constexpr int KTP_DK_WORDS = 1024;  // ... and as many as their rows' distances fit here
constexpr int KTS_RED_WORDS = 384;  // (knn_tree_small_kernel: the same; kept apart to keep its LDS budget in view)
constexpr unsigned KTP_GUARD_ULPS = 8;  // see ktp_resolve_cloud: candidates this close above the K-th distance send the cloud to the real search
__host__ __device__ inline int ktp_max_queries(int k) { return KTP_DK_WORDS / k < KTP_FEWQ ? KTP_DK_WORDS / k : KTP_FEWQ; }
// a node of the descent WITHOUT records moved (ktp_resolve_cloud): the node's points are the cloud's points inside lo .. hi (bit d of
// inc: lo[d] belongs to the node, bit 3 + d: hi[d] does); box: the box handed down to it (what middleSplit_ reads)
struct KtiWork { float lo[3], hi[3], box[6]; int inc, depth; };
struct KtpShared {
  float dk[KTP_DK_WORDS];  // the canonical rows' distances (ascending), k per listed query
  float qxyz[KTP_FEWQ][4];
  int qj[KTP_FEWQ], qcnt[KTP_FEWQ];  // the query's number; points at or within KTP_GUARD_ULPS above its K-th distance
  int mem_idx[KTP_TMAX], mem_grp[KTP_TMAX], mem_pt[KTP_TMAX];  // a tied (query, point) per lane of wave 0: the point's index, its run
  unsigned long long mem_khi[KTP_TMAX], mem_klo[KTP_TMAX];      // (query << 8 | the run's first slot in the row), its point's slot below, its key
  // ktp_descend_records: the DISTINCT tied points (a duplicated pair is listed by a dozen queries): index, position, node, side at the
  // last split, and the points it shares a run with (bit per point slot)
  int pt_idx[KTP_TMAX], pt_pos[KTP_TMAX], pt_node[KTP_TMAX], pt_side[KTP_TMAX];
  unsigned long long pt_peer[KTP_TMAX];
  int cur_idx[KTP_TMAX], cur_m[KTP_TMAX];  // the tied points inside the node being split: index, point slot
  union { KtpWork work[KTP_MAXW]; KtiWork iwork[KTP_MAXW]; };
  KtSplit split;
  int nmem, nw, ncur, bad;
  int cloud, entry, probe, npts;
  int stage, cutfeat, adv;  // the set form's pass over the current node: -1 fresh, 0 min / max of all dimensions, 1 the cut at the box's middle, 2 at the points' edge
  float cutval;
  int wsum[KTB_WAVES];
};
__host__ __device__ constexpr size_t ktp_lds_bytes(int n) {  // (n = 0: the form that reads the cloud in global memory)
  return (size_t)n * 16 + (((size_t)n * 2 + 15) & ~(size_t)15) + KTP_RED_WORDS * 4 + ((sizeof(KtpShared) + 15) & ~(size_t)15);
}
__device__ __forceinline__ unsigned long long ktp_readlane_u64(unsigned long long v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
// A node of 65 .. 1024 points by the whole workgroup, ONE POINT PER THREAD, held in registers through both partition passes
// (ktb_split_node_small's scheme one level up; ktb_split_node_wg's general strides cost ~10 us a node in barriers and LDS round
// trips, most of a chance tie's whole resolution).  planeSplit's closed form (header of the parallel build): with cnt satisfiers
// in [lo, count), the i-th violator among positions [lo, lo + cnt) (ascending) and the i-th satisfier at or behind lo + cnt
// (descending) change places.  Every wave publishes the ballot of its satisfiers; from the 16 masks every thread derives cnt and
// the ranks (popcounts of masks), the partners meet through `sc`, the records change places in LDS and in the registers.
// red: KTP_RED_WORDS words.  Nine barriers.  Ends with a barrier.
__device__ __forceinline__ unsigned long long ktp_lanes_below(const int th) {  // lanes l of a wave with l < th
  return th <= 0 ? 0ull : (th >= 64 ? ~0ull : ((1ull << th) - 1ull));
}
__device__ __forceinline__ KtSplit ktp_split_node_1k(float4* rec, unsigned short* sc, float* red, const float* box, const unsigned left,
                                                     const unsigned count, const int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned right = left + count;
  const bool in = (unsigned)tid < count;
  float4 r = rec[left + (in ? (unsigned)tid : 0u)];
  // ---- middleSplit_ (:966-1005); computeMinMax (:898-907) of all three dimensions
  const float EPS = 0.00001f;
  float max_span = box[1] - box[0];
  for (int d = 1; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3], mx3[3];
  {
    const float c[3] = {r.x, r.y, r.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(in ? c[d] : INFINITY); mx3[d] = wave_max_f32(in ? c[d] : -INFINITY); }
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[wave * 6 + 2 * d] = mn3[d]; red[wave * 6 + 2 * d + 1] = mx3[d]; }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {  // (lane l takes wave l & 15's partial: every partial is in, some four times)
    mn3[d] = wave_min_f32(red[(lane & (KTB_WAVES - 1)) * 6 + 2 * d]);
    mx3[d] = wave_max_f32(red[(lane & (KTB_WAVES - 1)) * 6 + 2 * d + 1]);
  }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? box[0] : (cutfeat == 1 ? box[2] : box[4]);
  const float bhi = cutfeat == 0 ? box[1] : (cutfeat == 1 ? box[3] : box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  // ---- planeSplit (:1016-1043)
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(red + 96);
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float v = cutfeat == 0 ? r.x : (cutfeat == 1 ? r.y : r.z);
    const bool sat = in && (unsigned)tid >= lo_p && (pass == 0 ? v < cutval : v <= cutval);
    const unsigned long long mine = __builtin_amdgcn_ballot_w64(sat);
    if (lane == 0) masks[pass * KTB_WAVES + wave] = mine;
    __syncthreads();
    // lanes 0 .. 15 take one wave's mask each: counts, then prefix sums over the lanes
    const unsigned long long mw = lane < KTB_WAVES ? masks[pass * KTB_WAVES + lane] : 0ull;
    const int cincl = wave_inclusive_sum_i32((int)__builtin_popcountll(mw));
    const unsigned cnt = (unsigned)__builtin_amdgcn_readlane(cincl, 63);
    const unsigned mid = lo_p + cnt;  // where the pointers meet
    const unsigned long long below_l = ktp_lanes_below((int)mid - 64 * lane);
    const unsigned long long in_l = ktp_lanes_below((int)count - 64 * lane) & ~ktp_lanes_below((int)lo_p - 64 * lane);
    const int vc = lane < KTB_WAVES ? (int)__builtin_popcountll(in_l & below_l & ~mw) : 0;  // violators in front of mid
    const int rc = lane < KTB_WAVES ? (int)__builtin_popcountll(mw & ~below_l) : 0;         // satisfiers at or behind it
    const int vincl = wave_inclusive_sum_i32(vc), rincl = wave_inclusive_sum_i32(rc);
    const unsigned vbase = (unsigned)__builtin_amdgcn_readlane(vincl - vc, wave);            // violators before this wave's positions
    const unsigned rbase = (unsigned)(__builtin_amdgcn_readlane(rincl, 63) - __builtin_amdgcn_readlane(rincl, wave));  // satisfiers behind them
    const unsigned long long below = ktp_lanes_below((int)mid - 64 * wave);
    const unsigned long long inw = ktp_lanes_below((int)count - 64 * wave) & ~ktp_lanes_below((int)lo_p - 64 * wave);
    const unsigned long long myviol = inw & below & ~mine, myrs = mine & ~below;
    const bool is_v = ((myviol >> lane) & 1ull) != 0ull, is_r = ((myrs >> lane) & 1ull) != 0ull;
    unsigned rank = 0;  // of a violator: ascending; of a satisfier: descending
    if (is_v) {
      rank = vbase + (unsigned)__builtin_popcountll(myviol & lt_mask);
      sc[left + lo_p + rank] = (unsigned short)tid;
    }
    if (is_r) {
      rank = rbase + ((unsigned)__builtin_popcountll(myrs) - 1u - (unsigned)__builtin_popcountll(myrs & lt_mask));
      sc[right - 1 - rank] = (unsigned short)tid;
    }
    __syncthreads();
    unsigned partner = 0;
    float4 other = r;
    if (is_v) partner = left + (unsigned)sc[right - 1 - rank];
    if (is_r) partner = left + (unsigned)sc[left + lo_p + rank];
    if (is_v || is_r) other = rec[partner];
    __syncthreads();
    if (is_v || is_r) { rec[partner] = r; r = other; }
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  // ---- divlow = max of the left part, divhigh = min of the right part along cutfeat (:956-957)
  {
    const float v = cutfeat == 0 ? r.x : (cutfeat == 1 ? r.y : r.z);
    const float dl0 = wave_max_f32(in && (unsigned)tid < index ? v : -INFINITY), dh0 = wave_min_f32(in && (unsigned)tid >= index ? v : INFINITY);
    if (lane == 0) { red[160 + wave * 2] = dl0; red[160 + wave * 2 + 1] = dh0; }
  }
  __syncthreads();
  KtSplit o;
  o.cutfeat = cutfeat; o.cutval = cutval; o.index = index;
  o.dl = wave_max_f32(red[160 + (lane & (KTB_WAVES - 1)) * 2]);
  o.dh = wave_min_f32(red[160 + (lane & (KTB_WAVES - 1)) * 2 + 1]);
  __syncthreads();
  return o;
}
// The same for a node of 1025 .. 8192 points: position p = e * 1024 + tid, e < 8 -- a thread holds nothing but its violators' ranks; the
// cut coordinates are read from the records in LDS where needed.  The 64-position chunks c = e * 16 + wave are in position order:
// lanes l and 64 + l of every wave take the ballots of chunks l and 64 + l, prefix sums over the lanes give every chunk the
// violators before it and the satisfiers behind it.  A violator exchanges the two records itself (the pairs are disjoint).
// (ktb_split_node_wg's general strides: 31 / 23 / 20 us for 8192 / 4096 / 2048 points as a call; this form: see EXPERIMENTS.)
__device__ __forceinline__ KtSplit ktp_split_node_8k(float4* rec, unsigned short* sc, float* red, const float* box, const unsigned left,
                                                     const unsigned count, const int tid) {
  constexpr int T = KTB_WAVES * 64;
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned right = left + count;
  // ---- middleSplit_ (:966-1005); computeMinMax (:898-907) of all three dimensions
  const float EPS = 0.00001f;
  float max_span = box[1] - box[0];
  for (int d = 1; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (unsigned p = tid; p < count; p += T) {
    const float4 r = rec[left + p];
    const float c[3] = {r.x, r.y, r.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn3[d] = c[d] < mn3[d] ? c[d] : mn3[d]; mx3[d] = c[d] > mx3[d] ? c[d] : mx3[d]; }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(mn3[d]); mx3[d] = wave_max_f32(mx3[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[wave * 6 + 2 * d] = mn3[d]; red[wave * 6 + 2 * d + 1] = mx3[d]; }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    mn3[d] = wave_min_f32(red[(lane & (KTB_WAVES - 1)) * 6 + 2 * d]);
    mx3[d] = wave_max_f32(red[(lane & (KTB_WAVES - 1)) * 6 + 2 * d + 1]);
  }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float split_val = (box[2 * cutfeat] + box[2 * cutfeat + 1]) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  const float* cutc = reinterpret_cast<const float*>(rec) + cutfeat;  // cutc[4 i] = the cut coordinate of record i
  // ---- planeSplit (:1016-1043)
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(red + 96);  // [128]: chunk c = e * 16 + wave
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if ((unsigned)((e + 1) * T) <= lo_p || (unsigned)(e * T) >= count) {  // (uniform) no position of this stride takes part
        if (lane == 0) masks[e * KTB_WAVES + wave] = 0ull;
        continue;
      }
      const unsigned p = (unsigned)(e * T + tid);
      const bool in = p >= lo_p && p < count;
      const float v = cutc[4 * (left + (in ? p : 0u))];
      const unsigned long long mine = __builtin_amdgcn_ballot_w64(in && (pass == 0 ? v < cutval : v <= cutval));
      if (lane == 0) masks[e * KTB_WAVES + wave] = mine;
    }
    __syncthreads();
    const unsigned long long mA = masks[lane], mB = masks[64 + lane];
    const int cA = (int)__builtin_popcountll(mA), cB = (int)__builtin_popcountll(mB);
    const unsigned cnt = (unsigned)(__builtin_amdgcn_readlane(wave_inclusive_sum_i32(cA), 63) + __builtin_amdgcn_readlane(wave_inclusive_sum_i32(cB), 63));
    const unsigned mid = lo_p + cnt;  // where the pointers meet
    const unsigned long long belA = ktp_lanes_below((int)mid - 64 * lane), belB = ktp_lanes_below((int)mid - 64 * (64 + lane));
    const unsigned long long inA = ktp_lanes_below((int)count - 64 * lane) & ~ktp_lanes_below((int)lo_p - 64 * lane);
    const unsigned long long inB = ktp_lanes_below((int)count - 64 * (64 + lane)) & ~ktp_lanes_below((int)lo_p - 64 * (64 + lane));
    const int vA = (int)__builtin_popcountll(inA & belA & ~mA), vB = (int)__builtin_popcountll(inB & belB & ~mB);  // violators in front of mid
    const int rA = (int)__builtin_popcountll(mA & ~belA), rB = (int)__builtin_popcountll(mB & ~belB);              // satisfiers at or behind it
    const int viA = wave_inclusive_sum_i32(vA), viB = wave_inclusive_sum_i32(vB), riA = wave_inclusive_sum_i32(rA), riB = wave_inclusive_sum_i32(rB);
    const int vtotA = __builtin_amdgcn_readlane(viA, 63), rtotA = __builtin_amdgcn_readlane(riA, 63), rtot = rtotA + __builtin_amdgcn_readlane(riB, 63);
    unsigned vrank[8];
    unsigned isv = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      vrank[e] = 0;
      if ((unsigned)((e + 1) * T) <= lo_p || (unsigned)(e * T) >= count) continue;  // (uniform)
      const int c = e * KTB_WAVES + wave;  // (uniform)
      const unsigned p = (unsigned)(e * T + tid);
      const unsigned long long m = masks[c];
      const unsigned long long below = ktp_lanes_below((int)mid - 64 * c);
      const unsigned long long inw = ktp_lanes_below((int)count - 64 * c) & ~ktp_lanes_below((int)lo_p - 64 * c);
      const unsigned long long myviol = inw & below & ~m, myrs = m & ~below;
      const unsigned vbase = (unsigned)(e < 4 ? __builtin_amdgcn_readlane(viA - vA, c & 63) : vtotA + __builtin_amdgcn_readlane(viB - vB, c & 63));
      const unsigned rincl = (unsigned)(e < 4 ? __builtin_amdgcn_readlane(riA, c & 63) : rtotA + __builtin_amdgcn_readlane(riB, c & 63));
      const unsigned rbase = (unsigned)rtot - rincl;  // satisfiers behind this chunk
      if ((myviol >> lane) & 1ull) {
        vrank[e] = vbase + (unsigned)__builtin_popcountll(myviol & lt_mask);
        isv |= 1u << e;
        sc[left + lo_p + vrank[e]] = (unsigned short)p;
      }
      if ((myrs >> lane) & 1ull) {
        const unsigned rank = rbase + ((unsigned)__builtin_popcountll(myrs) - 1u - (unsigned)__builtin_popcountll(myrs & lt_mask));
        sc[right - 1 - rank] = (unsigned short)p;
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if ((isv >> e) & 1u) {
        const unsigned a = left + (unsigned)(e * T + tid), b = left + (unsigned)sc[right - 1 - vrank[e]];
        const float4 ra = rec[a], rb = rec[b];
        rec[a] = rb; rec[b] = ra;
      }
    }
    __syncthreads();
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  // ---- divlow = max of the left part, divhigh = min of the right part along cutfeat (:956-957)
  float dl = -INFINITY, dh = INFINITY;
  for (unsigned p = tid; p < count; p += T) {
    const float v = cutc[4 * (left + p)];
    if (p < index) dl = v > dl ? v : dl;
    else dh = v < dh ? v : dh;
  }
  dl = wave_max_f32(dl); dh = wave_min_f32(dh);
  if (lane == 0) { red[352 + wave * 2] = dl; red[352 + wave * 2 + 1] = dh; }
  __syncthreads();
  KtSplit o;
  o.cutfeat = cutfeat; o.cutval = cutval; o.index = index;
  o.dl = wave_max_f32(red[352 + (lane & (KTB_WAVES - 1)) * 2]);
  o.dh = wave_min_f32(red[352 + (lane & (KTB_WAVES - 1)) * 2 + 1]);
  __syncthreads();
  return o;
}
// The listed queries of one cloud with the records MOVED (ktp_list_tied has listed their tied points in S): the splits of the builds
// on the cloud's records, so that positions are known -- what the set form (ktp_descend_sets below) cannot tell: the reading order of a
// leaf that holds two points of a run (duplicated points: never separated), a split that falls among points exactly on the cut.
// All queries share the one descent: a run's arrival order differs between queries only in the near / far bits.
// rec: the cloud's records in INDEX order (position == index); rootbox: the tight box of the cloud (computeBoundingBox :1321-1346);
// out_c: the cloud's rows, the listed ones rewritten in place.  -> 0, or 1: not resolved here (rows untouched or partly in arrival
// order -- the caller hands the cloud to the full build).  Uniform.
template <typename IdxT>
__device__ __forceinline__ int ktp_descend_records(float4* rec, unsigned short* sc, float* red, KtpShared* S, const float* rootbox, const int n,
                                                   const int k, IdxT* __restrict__ out_c, const int tid) {
  constexpr int T = KTB_WAVES * 64;
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  if (tid == 0) { S->bad = 0; S->nw = 0; }
  __syncthreads();
  const int t = S->nmem;
  // wave 0, two views: lane = a tied (query, point) -- its key; lane = a distinct tied POINT -- position, node (state in LDS between
  // the steps: values held in registers across the splits cost more registers than a 1024-thread workgroup has)
  if (wave == 0) {
    const int m_idx = lane < t ? S->mem_idx[lane] : -1 - lane, m_grp = lane < t ? S->mem_grp[lane] : -1 - lane;
    int rep = lane;  // the first lane that lists the same point
    for (int o = t - 1; o >= 0; --o) if (__builtin_amdgcn_readlane(m_idx, o) == m_idx) rep = o;
    const unsigned long long firsts = __builtin_amdgcn_ballot_w64(lane < t && rep == lane);
    const int npts = (int)__builtin_popcountll(firsts);
    const int pslot = (int)__builtin_popcountll(firsts & ktp_lanes_below(rep));
    if (lane < t) {
      S->mem_pt[lane] = pslot; S->mem_khi[lane] = 0ull; S->mem_klo[lane] = 0ull;
      if (rep == lane) { S->pt_idx[pslot] = m_idx; S->pt_pos[pslot] = m_idx; S->pt_node[pslot] = 0; S->pt_peer[pslot] = 0ull; }
    }
    ktb_wave_sync();
    unsigned long long peers = 0ull;
    for (int o = 0; o < t; ++o) {
      const int ps = __builtin_amdgcn_readlane(pslot, o);
      if (o != lane && __builtin_amdgcn_readlane(m_grp, o) == m_grp) peers |= 1ull << ps;
    }
    if (lane < t && peers != 0ull) atomicOr(&S->pt_peer[pslot], peers);
    ktb_wave_sync();
    const unsigned long long pk = lane < npts ? S->pt_peer[lane] & ~(1ull << lane) : 0ull;
    if (n > KT_LEAF && __builtin_amdgcn_ballot_w64(pk != 0ull) != 0ull && lane == 0) {
      KtpWork w0;
      w0.left = 0u; w0.right = (unsigned)n; w0.depth = 0;
      for (int i = 0; i < 6; ++i) w0.box[i] = rootbox[i];
      S->work[0] = w0;
      S->nw = 1;
    }
    if (lane == 0) S->npts = npts;
  }
  __syncthreads();
  const int npts = S->npts;
  for (int cur = 0;; ++cur) {
    if (cur >= S->nw) break;  // (uniform: written before the last barrier)
    const unsigned xleft = S->work[cur].left, xright = S->work[cur].right, count = xright - xleft;
    if (wave == 0) {
      const bool mine = lane < npts && S->pt_node[lane < npts ? lane : 0] == cur;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(mine);
      if (mine) {
        const int s0 = (int)__builtin_popcountll(mm & lt_mask);
        S->cur_idx[s0] = S->pt_idx[lane];
        S->cur_m[s0] = lane;
      }
      if (lane == 0) S->ncur = (int)__builtin_popcountll(mm);
    }
    KtSplit sp;
    if (count > (unsigned)T) {  // (the top of a large cloud)
      sp = ktp_split_node_8k(rec, sc, red, S->work[cur].box, xleft, count, tid);
    } else if (count > 256u) {
      sp = ktp_split_node_1k(rec, sc, red, S->work[cur].box, xleft, count, tid);
    } else {  // one wave: 4-5 us for 65 .. 256 points (passes over LDS), ~1.5 for <= 64 (in registers); the workgroup's form: 7.5 whatever the size
      if (wave == 0) {
        KtWork wk;
        wk.node = 0; wk.left = xleft; wk.right = xright;
        for (int i = 0; i < 6; ++i) wk.box[i] = S->work[cur].box[i];
        const KtSplit s1 = ktb_split_node(rec, sc, wk, xleft, xright, lane, lt_mask);
        if (lane == 0) S->split = s1;
      }
      __syncthreads();
      sp = S->split;
    }
    // where the node's tied points are now
    const int ncur = S->ncur;
    for (unsigned p = xleft + tid; p < xright; p += T) {
      const int w = __float_as_int(rec[p].w);
      for (int s0 = 0; s0 < ncur; ++s0)
        if (S->cur_idx[s0] == w) S->pt_pos[S->cur_m[s0]] = (int)p;
    }
    __syncthreads();
    if (wave == 0) {
      const int depth = S->work[cur].depth;
      // the (query, point) view, first half: is its point in this node
      const int mpt = lane < t ? S->mem_pt[lane] : 0;
      const bool m_mine = lane < t && S->pt_node[mpt] == cur;
      ktb_wave_sync();
      // the point view: sides, and the children that still hold two points of a run
      {
        const int pl = lane < npts ? lane : 0;
        const bool mine = lane < npts && S->pt_node[pl] == cur;
        const unsigned long long pk = lane < npts ? S->pt_peer[pl] & ~(1ull << lane) : 0ull;
        const int side = (unsigned)S->pt_pos[pl] >= xleft + sp.index ? 1 : 0;
        if (mine) S->pt_side[pl] = side;
        int nw_reg = S->nw;
        bool fail = false;
        int node_next = -1;
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
          const unsigned cl = c == 0 ? xleft : xleft + sp.index, cr = c == 0 ? xleft + sp.index : xright;
          const bool in_c = mine && side == c;
          const unsigned long long cm = __builtin_amdgcn_ballot_w64(in_c);
          int e = -1;
          if (cr - cl > (unsigned)KT_LEAF && __builtin_amdgcn_ballot_w64(in_c && (pk & cm) != 0ull) != 0ull) {  // two of a run in an inner node: split it too
            if (nw_reg >= KTP_MAXW || depth + 1 >= KTP_DEPTH) fail = true;
            else {
              e = nw_reg++;
              if (lane == 0) {
                KtpWork w1;
                w1.left = cl; w1.right = cr; w1.depth = depth + 1;
                for (int i = 0; i < 6; ++i) w1.box[i] = (i == 2 * sp.cutfeat + 1 - c) ? sp.cutval : S->work[cur].box[i];
                S->work[e] = w1;
              }
            }
          }
          if (in_c) node_next = e;
        }
        if (mine) S->pt_node[pl] = node_next;
        if (lane == 0) { S->nw = nw_reg; if (fail) S->bad = 1; }
      }
      ktb_wave_sync();
      // the (query, point) view, second half.  searchLevel (:1380-1393): the child on the query's side of the gap first
      if (m_mine) {
        const float val = S->qxyz[S->mem_grp[lane] >> 8][sp.cutfeat];
        const float diff1 = val - sp.dl, diff2 = val - sp.dh;
        const int nearside = (diff1 + diff2) < 0 ? 0 : 1;
        if (S->pt_side[mpt] != nearside) {
          if (depth < 64) S->mem_khi[lane] |= 1ull << (63 - depth); else S->mem_klo[lane] |= 1ull << (127 - depth);
        }
      }
    }
    __syncthreads();
    if (cur < 24) KTP_MARK(4 + cur);
    if (S->bad != 0) return 1;
  }
  // a run in arrival order: (key, position) ascending, from the run's first slot on; what does not fit the row is dropped
  if (wave == 0) {
    const int ml = lane < t ? lane : 0;
    const int m_grp = lane < t ? S->mem_grp[ml] : -1 - lane, m_pos = S->pt_pos[S->mem_pt[ml]];
    const unsigned long long khi = S->mem_khi[ml], klo = S->mem_klo[ml];
    int rank = 0;
    for (int o = 0; o < t; ++o) {
      const int og = __builtin_amdgcn_readlane(m_grp, o), op = __builtin_amdgcn_readlane(m_pos, o);
      const unsigned long long oh = ktp_readlane_u64(khi, o), ol = ktp_readlane_u64(klo, o);
      const bool before = oh < khi || (oh == khi && (ol < klo || (ol == klo && op < m_pos)));
      rank += (o != lane && og == m_grp && before) ? 1 : 0;
    }
    const int slot = (m_grp & 255) + rank;
    if (lane < t && slot < k) out_c[(size_t)S->qj[m_grp >> 8] * k + slot] = (IdxT)S->mem_idx[ml];
  }
  __syncthreads();
  KTP_MARK(30);
  return 0;
}

// init_vind (:1318), computeBoundingBox (:1321-1346): the cloud's records {x, y, z, index} in index order and its tight box in
// LDS (part: 6 words per wave).  Barriers at both ends (the LDS may still be in use / is ready on return).
template <bool STORE = true>
__device__ __forceinline__ void kts_load_records(const float* __restrict__ pts, const int n, float4* rec, float* part, float* rootbox,
                                                 const int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += KTB_WAVES * 64) {
    const float c[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
    if (STORE) rec[i] = make_float4(c[0], c[1], c[2], __int_as_float(i));  // (false: the box only, the points stay in global memory)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = c[d] < lo[d] ? c[d] : lo[d];
      hi[d] = c[d] > hi[d] ? c[d] : hi[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f32(lo[d]); hi[d] = wave_max_f32(hi[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { part[wave * 6 + 2 * d] = lo[d]; part[wave * 6 + 2 * d + 1] = hi[d]; }
  }
  __syncthreads();
  if (wave == 0) {  // (lane l takes wave l & 15's partial)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float l = wave_min_f32(part[(lane & (KTB_WAVES - 1)) * 6 + 2 * d]), h = wave_max_f32(part[(lane & (KTB_WAVES - 1)) * 6 + 2 * d + 1]);
      if (lane == 0) { rootbox[2 * d] = l; rootbox[2 * d + 1] = h; }
    }
  }
  __syncthreads();
}
// The listed queries of ONE cloud together (nq <= ktp_max_queries(k)), WITHOUT moving a record: a node of the reference tree is a SET of points,
// and which child a point goes to is a comparison with the cut value -- positions decide only where points lie exactly on the cut
// and the split falls among them (index = count / 2 strictly between lim1 and lim2, :1037-1042), and in which order a leaf is
// read.  So a node is kept as the half-open box its ancestors' cuts left (KtiWork); a level is two passes of reductions over the
// cloud (middleSplit_'s min / max and count; planeSplit's lim1, lim2 and the children's divlow / divhigh as the extremes of
// the two sets) -- ~2 us where moving the records of a 1024-point node takes 7.  Two tied points that reach a leaf together, or a
// split that falls among equal coordinates: 2 = take the forms that move the records (ktp_resolve, a query at a time).
// rec: the records in index order, untouched.  -> 0 done, 1 not taken (the full build), 2 see above.  Uniform.
// GLOBAL: the points are read from the cloud in global memory (pts; rec unused) -- clouds of any size, no LDS but the bookkeeping's.
template <typename IdxT, bool GLOBAL = false>
__device__ __forceinline__ int ktp_resolve_cloud(const float4* rec, const float* __restrict__ pts, float* red, KtpShared* S, const float* rootbox,
                                                 const int n, const int k, const int nq, const int* __restrict__ flist_c,
                                                 const float* __restrict__ queries_c, IdxT* __restrict__ out_c, const int tid) {
  constexpr int T = KTB_WAVES * 64;
  const int lane = tid & 63, wave = tid >> 6;
  auto point = [&](const int p) -> float4 {
    if (GLOBAL) return make_float4(pts[(size_t)p * 3], pts[(size_t)p * 3 + 1], pts[(size_t)p * 3 + 2], 0.f);
    return rec[p];
  };
  if (tid < nq) {
    const int j = flist_c[tid];
    S->qj[tid] = j;
    S->qxyz[tid][0] = queries_c[(size_t)j * 3]; S->qxyz[tid][1] = queries_c[(size_t)j * 3 + 1]; S->qxyz[tid][2] = queries_c[(size_t)j * 3 + 2];
  }
  if (tid == 0) { S->nmem = 0; S->bad = 0; S->nw = 0; }
  if (tid < KTP_FEWQ) S->qcnt[tid] = 0;
  __syncthreads();
  for (int i = tid; i < nq * k; i += T) {
    const int q = i / k, s0 = i - q * k;
    const float4 r = point((int)out_c[(size_t)S->qj[q] * k + s0]);
    S->dk[q * k + s0] = dist2(S->qxyz[q][0], S->qxyz[q][1], S->qxyz[q][2], r.x, r.y, r.z);
  }
  __syncthreads();
  // the tied points of every query: at a distance its row holds twice, or at its K-th distance
  for (int p = tid; p < n; p += T) {
    const float4 r = point(p);
    for (int q = 0; q < nq; ++q) {
      const float* dk = S->dk + q * k;
      const float d = dist2(S->qxyz[q][0], S->qxyz[q][1], S->qxyz[q][2], r.x, r.y, r.z), rk = dk[k - 1];
      if (d <= __uint_as_float(__float_as_uint(rk) + KTP_GUARD_ULPS)) atomicAdd(&S->qcnt[q], 1);  // (about K of them)
      if (d <= rk) {
        int lo = 0, hi = k - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (dk[mid] < d) lo = mid + 1; else hi = mid;
        }
        if (dk[lo] != d) S->bad = 1;
        else if (d == rk || dk[lo + 1] == d) {
          const int slot = atomicAdd(&S->nmem, 1);
          if (slot < KTP_TMAX) { S->mem_idx[slot] = p; S->mem_grp[slot] = (q << 8) | lo; }
        }
      }
    }
  }
  __syncthreads();
  const int t = S->nmem;
  KTP_MARK(3);
  if (S->bad != 0 || t > KTP_TMAX) return 1;
  // More than K candidates at or within a few ulps above the K-th distance (a run that reaches beyond the row's end, a (K+1)-th
  // point next to the K-th): WHICH of them the reference keeps can hinge on the rounding of its pruning bound (mindistsq + cut_dist -
  // dists[idx], nanoflann.hpp:1396-1404, may exceed the distance of a point IN that subtree by an ulp: the subtree is skipped) -- not
  // a property of the tree's shape.  Only the real search reproduces that (found by tools/tie_path_fuzz.py: 1 listed query in 3.8 M)
  for (int q = 0; q < nq; ++q) if (S->qcnt[q] > k) return 1;  // (uniform)
  // The descent.  A level = ONE pass of the workgroup over the cloud (the node's count, lim1, lim2 by ballots; its extremes along the cut
  // dimension by wave reductions), the waves' partials through LDS, and wave 0's turn: the split, the tied points' sides, the children
  // that still hold two of a run.  Two barriers a level.  (Measured: a level of a 1024-point cloud 4.4 us; by wave 0 alone, no
  // barriers: 9.5 -- the pass is ~60 dependent instructions per 64 points.)
  // wave 0: one tied point per lane, in registers -- index, run (query << 8 | first slot), node, its coordinates, its query's, its key
  int m_idx = 0, m_grp = -1 - lane, m_node = -1, nw_reg = 0;
  float px = 0.f, py = 0.f, pz = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
  unsigned long long khi = 0ull, klo = 0ull;
  if (wave == 0) {
    if (lane < t) {
      m_idx = S->mem_idx[lane]; m_grp = S->mem_grp[lane]; m_node = 0;
      const float4 r = point(m_idx);
      px = r.x; py = r.y; pz = r.z;
      ux = S->qxyz[m_grp >> 8][0]; uy = S->qxyz[m_grp >> 8][1]; uz = S->qxyz[m_grp >> 8][2];
    }
    // twin: two points of a run at the same place (no cut separates them) -- or, in a cloud of thousands of points whose records are in
    // LDS, so close together that they most likely share a leaf of ten (closer than half a leaf's radius, estimated from the query's
    // own K-th distance: K points within r_K, so ten within ~r_K sqrt(10 / K)): the set form would walk ~10 levels at ~8 us to find
    // that out, the record-moving form is as fast there and needs no second start.  A guess that only chooses the form, not the result.
    const float near2 = !GLOBAL && n > KTS_NMAX && lane < t ? 2.5f * S->dk[(m_grp >> 8) * k + k - 1] / (float)k : 0.f;
    bool peer = false, twin = false, first = lane < t;
    for (int o = 0; o < t; ++o) {
      const bool same = o != lane && __builtin_amdgcn_readlane(m_grp, o) == m_grp;
      peer |= same;
      const float ex = readlane_f(px, o) - px, ey = readlane_f(py, o) - py, ez = readlane_f(pz, o) - pz;
      twin |= same && ((ex == 0.f && ey == 0.f && ez == 0.f) || (ex * ex + ey * ey) + ez * ez < near2);
      first = first && !(o < lane && __builtin_amdgcn_readlane(m_idx, o) == m_idx);
    }
    if (lane == 0) S->npts = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(first));  // distinct tied points (several queries list the same)
    if (n > KT_LEAF && __builtin_amdgcn_ballot_w64(lane < t && twin) != 0ull) {  // duplicated points: no cut separates them
      if (lane == 0) S->bad = 2;
    } else if (n > KT_LEAF && __builtin_amdgcn_ballot_w64(lane < t && peer) != 0ull) {
      nw_reg = 1;
      if (lane == 0) {
        KtiWork* w0 = &S->iwork[0];
        for (int d = 0; d < 3; ++d) { w0->lo[d] = -INFINITY; w0->hi[d] = INFINITY; }
        for (int i = 0; i < 6; ++i) w0->box[i] = rootbox[i];
        w0->inc = 63; w0->depth = 0;
        S->nw = 1;
      }
    }
    if (lane == 0) S->stage = -1;
  }
  __syncthreads();
  if (S->bad != 0) return S->bad;  // (uniform)
  int cur = 0, npass = 0;
  for (;;) {
    if (cur >= S->nw) break;  // (uniform: written before the last barrier)
    // (the node stays in LDS: an index that is not a constant -- box[2 * cutfeat] -- would send a copy in registers to scratch memory)
    const KtiWork* X = &S->iwork[cur];
    const int inc = X->inc;
    const float xlo[3] = {X->lo[0], X->lo[1], X->lo[2]}, xhi[3] = {X->hi[0], X->hi[1], X->hi[2]};
    // ---- middleSplit_ (:966-1005): the dimensions whose box side is (nearly) the longest may be cut
    const float EPS = 0.00001f;
    const float span[3] = {X->box[1] - X->box[0], X->box[3] - X->box[2], X->box[5] - X->box[4]};
    float max_span = span[0];
    for (int d = 1; d < 3; ++d) if (span[d] > max_span) max_span = span[d];
    int qual = 0;
    for (int d = 0; d < 3; ++d) qual |= span[d] > (1 - EPS) * max_span ? 1 << d : 0;
    int stage = S->stage, cutfeat = S->cutfeat;
    float cutval = S->cutval;
    if (stage < 0) {  // a fresh node
      if ((qual & (qual - 1)) == 0) {  // one candidate (a box has a longest side; ties between sides are the exception): no look at the spreads
        stage = 1;
        cutfeat = qual == 1 ? 0 : (qual == 2 ? 1 : 2);
        cutval = (X->box[2 * cutfeat] + X->box[2 * cutfeat + 1]) / 2;
      } else stage = 0;
    }
    // ---- one pass over the cloud: the node's points are the points inside lo .. hi.  stage 0: f = min x, y, z, max x, y, z
    // (computeMinMax :898-907); stages 1 (the cut at the box's middle), 2 (at the points' edge): ci = count, lim1 = points below the
    // cut, lim2 = at or below it; f[0] min, f[3] max along cutfeat, f[1] the least coordinate above the cut, f[4] the greatest below
    int ci[3] = {0, 0, 0};
    float f[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float4 nxt = point(wave * 64 + lane < n ? wave * 64 + lane : 0);  // (one trip ahead: a trip's ballots keep the loop from being pipelined)
    for (int p0 = wave * 64; p0 < n; p0 += T) {
      const int p = p0 + lane;
      const float4 r = nxt;
      if (p0 + T < n) nxt = point(p + T < n ? p + T : 0);
      const float c[3] = {r.x, r.y, r.z};
      bool in = p < n;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        in = in && (((inc >> d) & 1) ? c[d] >= xlo[d] : c[d] > xlo[d]);
        in = in && (((inc >> (3 + d)) & 1) ? c[d] <= xhi[d] : c[d] < xhi[d]);
      }
      if (stage == 0) {
        if (in) {
#pragma unroll
          for (int d = 0; d < 3; ++d) { f[d] = c[d] < f[d] ? c[d] : f[d]; f[3 + d] = c[d] > f[3 + d] ? c[d] : f[3 + d]; }
        }
      } else {
        const float v = cutfeat == 0 ? c[0] : (cutfeat == 1 ? c[1] : c[2]);
        ci[0] += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
        ci[1] += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in && v < cutval));
        ci[2] += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in && v <= cutval));
        if (in) {
          f[0] = v < f[0] ? v : f[0];
          f[3] = v > f[3] ? v : f[3];
          if (v > cutval) f[1] = v < f[1] ? v : f[1];
          if (v < cutval) f[4] = v > f[4] ? v : f[4];
        }
      }
    }
    if (stage == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { f[d] = wave_min_f32(f[d]); f[3 + d] = wave_max_f32(f[3 + d]); }
    } else {
      f[0] = wave_min_f32(f[0]); f[1] = wave_min_f32(f[1]); f[3] = wave_max_f32(f[3]); f[4] = wave_max_f32(f[4]);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) red[wave * 12 + i] = f[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) red[wave * 12 + 6 + i] = __int_as_float(ci[i]);
    }
    __syncthreads();
    if (wave == 0) {
      const int wl = lane & (KTB_WAVES - 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) { f[i] = wave_min_f32(red[wl * 12 + i]); f[3 + i] = wave_max_f32(red[wl * 12 + 3 + i]); }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        ci[i] = __builtin_amdgcn_readlane(wave_inclusive_sum_i32(lane < KTB_WAVES ? __float_as_int(red[lane * 12 + 6 + i]) : 0), 63);
      int adv = 0, fail = 0;
      if (stage == 0) {
        float max_spread = -1.f;
        int cf = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if ((qual >> d) & 1) {
            const float spread = f[3 + d] - f[d];
            if (spread > max_spread) { cf = d; max_spread = spread; }
          }
        }
        if (lane == 0) { S->stage = 1; S->cutfeat = cf; S->cutval = (X->box[2 * cf] + X->box[2 * cf + 1]) / 2; }
      } else if (stage == 1 && (cutval < f[0] || cutval > f[3])) {  // the box's middle misses the points: the cut goes to their edge (:996-1001)
        if (lane == 0) { S->stage = 2; S->cutfeat = cutfeat; S->cutval = cutval < f[0] ? f[0] : f[3]; }
      } else {
        // ---- planeSplit (:1016-1043) on the SET
        const unsigned count = (unsigned)ci[0], lim1 = (unsigned)ci[1], lim2 = (unsigned)ci[2], half = count / 2;
        unsigned index;
        int mode;  // 0: the left child is {v < cut}, 1: {v <= cut}, -1: the split falls among the points ON the cut
        if (lim1 > half) { index = lim1; mode = 0; }
        else if (lim2 < half) { index = lim2; mode = 1; }
        else { index = half; mode = half == lim1 ? 0 : (half == lim2 ? 1 : -1); }
        if (mode < 0) fail = 2;
        else {
          // divlow = the greatest coordinate of the left child, divhigh = the least of the right one (:956-957); points ON the cut, if any,
          // are the left child's greatest (mode 1) or the right child's least (mode 0)
          const bool on_cut = lim2 > lim1;
          const float dl = mode == 1 && on_cut ? cutval : f[4], dh = mode == 0 && on_cut ? cutval : f[1];
          const bool mine = m_node == cur;
          const int depth = X->depth;
          const float v = cutfeat == 0 ? px : (cutfeat == 1 ? py : pz);
          const int side = (mode == 0 ? v < cutval : v <= cutval) ? 0 : 1;
          // searchLevel (:1380-1393): the child on the query's side of the gap first
          const float val = cutfeat == 0 ? ux : (cutfeat == 1 ? uy : uz);
          const float diff1 = val - dl, diff2 = val - dh;
          const int nearside = (diff1 + diff2) < 0 ? 0 : 1;
          if (mine && side != nearside) {
            if (depth < 64) khi |= 1ull << (63 - depth); else klo |= 1ull << (127 - depth);
          }
          int node_next = -1;
#pragma unroll 1
          for (int c = 0; c < 2; ++c) {
            const unsigned cc = c == 0 ? index : count - index;
            const bool in_c = mine && side == c;
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(in_c);
            bool peer = false;
            for (int o = 0; o < t; ++o) peer |= (((cm >> o) & 1ull) != 0ull && o != lane && __builtin_amdgcn_readlane(m_grp, o) == m_grp);
            int e = -1;
            if (__builtin_amdgcn_ballot_w64(in_c && peer) != 0ull) {  // two of a run in this child
              if (cc <= (unsigned)KT_LEAF) fail = 2;  // ... a leaf: its reading order is a matter of positions
              else if (nw_reg >= KTP_MAXW || depth + 1 >= KTP_DEPTH) fail = 1;
              else {
                e = nw_reg++;
                if (lane < 14) reinterpret_cast<float*>(&S->iwork[e])[lane] = reinterpret_cast<const float*>(X)[lane];  // (a copy of the node ...
                ktb_wave_sync();
                if (lane == 0) {                                                                                        // ... with one face moved)
                  KtiWork* w1 = &S->iwork[e];
                  w1->depth = depth + 1;
                  w1->box[2 * cutfeat + 1 - c] = cutval;
                  if (c == 0) { w1->hi[cutfeat] = cutval; w1->inc = (inc & ~(8 << cutfeat)) | (mode == 1 ? (8 << cutfeat) : 0); }
                  else { w1->lo[cutfeat] = cutval; w1->inc = (inc & ~(1 << cutfeat)) | (mode == 0 ? (1 << cutfeat) : 0); }
                }
              }
            }
            if (in_c) node_next = e;
          }
          if (mine) m_node = node_next;
          adv = 1;
        }
        if (lane == 0) { S->stage = -1; S->nw = nw_reg; }
      }
      if (lane == 0) { S->adv = adv; if (fail != 0) S->bad = fail; }
    }
    __syncthreads();
    if (npass < 24) KTP_MARK(4 + npass);
    ++npass;
    if (S->bad != 0) return S->bad;
    cur += S->adv;
  }
  // every run in arrival order (no two of a run share a leaf below the root: the keys differ; a cloud that IS one leaf is read in index order)
  if (wave == 0) {
    int rank = 0;
    for (int o = 0; o < t; ++o) {
      const int og = __builtin_amdgcn_readlane(m_grp, o), op = __builtin_amdgcn_readlane(m_idx, o);
      const unsigned long long oh = ktp_readlane_u64(khi, o), ol = ktp_readlane_u64(klo, o);
      const bool before = oh < khi || (oh == khi && (ol < klo || (ol == klo && op < m_idx)));
      rank += (o != lane && og == m_grp && before) ? 1 : 0;
    }
    const int slot = (m_grp & 255) + rank;
    if (lane < t && slot < k) out_c[(size_t)S->qj[m_grp >> 8] * k + slot] = (IdxT)m_idx;
  }
  __syncthreads();
  KTP_MARK(30);
  return 0;
}

template <typename IdxT, bool GLOBAL>
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tie_path_kernel(int b, int n, int m, int k, const float* __restrict__ pts_all,
                                                                     const float* __restrict__ queries, IdxT* __restrict__ out,
                                                                     const int* __restrict__ nflag, const int* __restrict__ flist,
                                                                     int* __restrict__ nwork) {
  constexpr int T = KTB_WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // GLOBAL (clouds of more than KTB_LDS_NMAX points): no records in LDS, the set form reads the cloud; what it does not take goes on
  const size_t rec_bytes = GLOBAL ? 0 : (size_t)n * 16 + (((size_t)n * 2 + 15) & ~(size_t)15);
  float4* rec = reinterpret_cast<float4*>(smem);                                          // [n]
  unsigned short* sc = reinterpret_cast<unsigned short*>(rec + n);                        // [n]
  float* red = reinterpret_cast<float*>(smem + rec_bytes);                                // [KTP_RED_WORDS]
  KtpShared* S = reinterpret_cast<KtpShared*>(red + KTP_RED_WORDS);
  __shared__ float rootbox[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PASNL_TUNING
  if (tid == 0) S->probe = blockIdx.x == 0 ? 1 : 0;
  __syncthreads();
#endif
  KTP_MARK(0);
  // the batch's listed CLOUDS, numbered thread-major (every workgroup computes the same numbering)
  int cnt = 0;
  for (int c = tid; c < b; c += T) cnt += nflag[c] != 0 ? 1 : 0;
  const int incl = wave_inclusive_sum_i32(cnt);
  if (lane == 63) S->wsum[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < KTB_WAVES; ++w) {
    const int v = S->wsum[w];
    if (w < wave) base += v;
    total += v;
  }
  if (total > KTP_MAXQ) {  // (uniform) too many for a workgroup each: everything to the full builds
    for (int c = blockIdx.x * T + tid; c < b; c += gridDim.x * T) nwork[c] = nflag[c];
    return;
  }
  const int excl = base + incl - cnt;
  KTP_MARK(1);
  for (int g = blockIdx.x; g < total; g += gridDim.x) {
    __syncthreads();
    if (g >= excl && g < excl + cnt) {  // (one thread)
      int rem = g - excl;
      for (int c = tid; c < b; c += T) {
        if (nflag[c] == 0) continue;
        if (rem == 0) { S->cloud = c; break; }
        --rem;
      }
    }
    __syncthreads();
    const int cloud = S->cloud, nq = nflag[cloud];
#ifdef PASNL_TUNING
    const unsigned long long ktp_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (nq > ktp_max_queries(k)) {  // (uniform) a cloud of many ties: the full build
      if (tid == 0) atomicExch(&nwork[cloud], nq);
      continue;
    }
    const float* pts = pts_all + (size_t)cloud * n * 3;
    kts_load_records<!GLOBAL>(pts, n, rec, red, rootbox, tid);
    KTP_MARK(2);
    int rc = ktp_resolve_cloud<IdxT, GLOBAL>(rec, pts, red, S, rootbox, n, k, nq, flist + (size_t)cloud * m, queries + (size_t)cloud * m * 3,
                                             out + (size_t)cloud * m * k, tid);
    if (rc == 0) KTP_COUNT(0);
    if (rc == 1) KTP_COUNT(3);
    if (rc == 2) KTP_COUNT(4);
    if (rc == 2 && !GLOBAL) {  // two tied points in one leaf (duplicated points), or a split among equal coordinates: the form that moves the records
      rc = ktp_descend_records<IdxT>(rec, sc, red, S, rootbox, n, k, out + (size_t)cloud * m * k, tid);
      if (rc == 0) KTP_COUNT(1);
    }
    if (rc != 0 && tid == 0) atomicExch(&nwork[cloud], nq);  // every listed query of the cloud, the done ones too (rows are simply written again)
#ifdef PASNL_TUNING
    if (tid == 0 && g < 32) { ktp_wg_cycles[g] = __builtin_amdgcn_s_memtime() - ktp_t0; ktp_wg_cycles[32 + g] = (unsigned long long)(rc + 10 * S->npts + 1000 * S->nmem); }
#endif
  }
}

__host__ __device__ constexpr size_t kts_lds_bytes(int n) {
  const size_t lq = (size_t)(n / (KT_LEAF + 1)) + 2;
  return (size_t)n * 16 + (((size_t)n * 2 + 15) & ~(size_t)15) + (size_t)KTS_NNODES(n) * sizeof(KtNode) + 2 * lq * sizeof(KtWork) + 64 +
         KTB_WAVES * 6 * 4 + (size_t)KTB_WAVES * KT_DEPTH * 3 * 4 + KTS_RED_WORDS * 4 + ((sizeof(KtpShared) + 15) & ~(size_t)15);
}
static_assert(kts_lds_bytes(KTS_NMAX) <= 160 * 1024 && ktp_lds_bytes(KTB_LDS_NMAX) <= 160 * 1024, "a workgroup's LDS");
template <typename IdxT>
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_small_kernel(int b, int n, int m, int k, const float* __restrict__ pts_all,
                                                                       const float* __restrict__ queries, IdxT* __restrict__ out,
                                                                       int* __restrict__ flag, const int* __restrict__ nflag,
                                                                       const int* __restrict__ flist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* rec = reinterpret_cast<float4*>(smem);                                          // [n]
  unsigned short* sc = reinterpret_cast<unsigned short*>(rec + n);                        // [n]
  KtNode* nodes = reinterpret_cast<KtNode*>(smem + (size_t)n * 16 + (((size_t)n * 2 + 15) & ~(size_t)15));  // [KTS_NNODES(n)]
  const int lq = n / (KT_LEAF + 1) + 2;
  KtWork* const qa = reinterpret_cast<KtWork*>(nodes + KTS_NNODES(n));
  KtWork* const qb = qa + lq;
  int* ctr = reinterpret_cast<int*>(qb + lq);                                             // [0], [1] queue lengths, [2] nodes used, [3] flag; then the root box
  float* rootbox = reinterpret_cast<float*>(ctr + 4);                                     // [6] (+ padding to 64 bytes)
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ctr) + 64);              // [KTB_WAVES][6]
  uint32_t* stacks = reinterpret_cast<uint32_t*>(part + KTB_WAVES * 6);                   // [KTB_WAVES][KT_DEPTH * 3]
  float* red = reinterpret_cast<float*>(stacks + KTB_WAVES * KT_DEPTH * 3);               // [KTS_RED_WORDS] the set form's partials
  KtpShared* S = reinterpret_cast<KtpShared*>(red + KTS_RED_WORDS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#ifdef PASNL_TUNING
  // (the first listed cloud, found by the whole workgroup: one thread reading the counters one after the other cost every launch of
  // the tuning build ~30 us and misled a round of experiments)
  __shared__ int kts_first_s;
  if (tid == 0) kts_first_s = 0x7fffffff;
  __syncthreads();
  for (int c = tid; c < b; c += KTB_WAVES * 64) if (nflag[c] != 0) atomicMin(&kts_first_s, c);
  __syncthreads();
  const int kts_first = kts_first_s == 0x7fffffff ? -1 : kts_first_s % (int)gridDim.x;
#endif
#ifdef PASNL_TUNING
  if (tid == 0) S->probe = blockIdx.x == kts_first ? 1 : 0;
  __syncthreads();
#endif
  for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {
    const int nq = nflag[cloud];
    if (nq == 0) continue;  // (uniform)
    KTS_MARK(0);
    const float* pts = pts_all + (size_t)cloud * n * 3;
    // A FEW listed queries (chance ties): their runs of equal distances put in arrival order along the tree paths that separate them
    // (ktp_resolve_cloud: ~10 us where tree + search take 85); anything it does not take: the tree and the searches below
    bool resolved = false;
    if (nq <= ktp_max_queries(k)) {
      kts_load_records(pts, n, rec, part, rootbox, tid);
      const int rc = ktp_resolve_cloud<IdxT>(rec, pts, red, S, rootbox, n, k, nq, flist + (size_t)cloud * m, queries + (size_t)cloud * m * 3,
                                            out + (size_t)cloud * m * k, tid);
      resolved = rc == 0;
      if (rc == 0) KTP_COUNT(0);
      if (rc == 1) KTP_COUNT(3);
      if (rc == 2) KTP_COUNT(4);
      // 2: two tied points in one leaf -- duplicated points --, or a split among equal coordinates.  The form that moves the records walks
      // one chain of ~8 splits per leaf involved (~50 us), the tree + parallel searches below take 85-120: it pays for one or two
      // leaves -- at most four distinct tied points, however many queries list them (a duplicated pair lists a dozen)
      if (rc == 2 && S->npts <= 4) {
        resolved = ktp_descend_records<IdxT>(rec, sc, red, S, rootbox, n, k, out + (size_t)cloud * m * k, tid) == 0;
        if (resolved) KTP_COUNT(1);
      }
    }
    if (resolved) { KTS_MARK(31); continue; }  // (uniform)
    KTP_COUNT(2);
    kts_load_records(pts, n, rec, part, rootbox, tid);
    if (tid < 4) ctr[tid] = 0;
    __syncthreads();
    if (tid == 0) {
      ctr[2] = 1;  // node 0 = the root
      if (n <= KT_LEAF) {
        nodes[0].child1 = nodes[0].child2 = -1; nodes[0].a = 0; nodes[0].divlow = __int_as_float(n); nodes[0].divhigh = 0.f;
      } else {
        KtWork w0; w0.node = 0; w0.left = 0; w0.right = (unsigned)n;
        for (int i = 0; i < 6; ++i) w0.box[i] = rootbox[i];  // root_bbox after divideTree = the tight box of all points (what the search reads)
        qa[0] = w0;
        ctr[0] = 1;
      }
    }
    __syncthreads();
    KTS_MARK(1);
    int cur = 0, level = 0;
    for (;;) {
      const int nqn = ctr[cur];
      if (nqn == 0) break;
      if (level + 2 >= KT_DEPTH) { if (tid == 0) ctr[3] = 1; break; }
      for (int e = wave; e < nqn; e += KTB_WAVES) {
        const KtWork wk = kt_load_work((cur ? qb : qa) + e, lane);
        const unsigned left = wk.left, right = wk.right;
        const KtSplit sp_ = ktb_split_node(rec, sc, wk, left, right, lane, lt_mask);
        const int cutfeat = sp_.cutfeat;
        const float cutval = sp_.cutval;
        const unsigned index = sp_.index;
        if (lane == 0) {
          int child[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const unsigned cl = c == 0 ? left : left + index, cr = c == 0 ? left + index : right;
            const int id = atomicAdd(&ctr[2], 1);
            child[c] = id;
            if (cr - cl <= (unsigned)KT_LEAF) {  // leaf (:921-936)
              nodes[id].child1 = nodes[id].child2 = -1;
              nodes[id].a = (int)cl;
              nodes[id].divlow = __int_as_float((int)cr);
              nodes[id].divhigh = 0.f;
            } else {
              KtWork w;
              w.node = id; w.left = cl; w.right = cr;
#pragma unroll
              for (int i = 0; i < 6; ++i) w.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : wk.box[i];
              (cur ? qa : qb)[atomicAdd(&ctr[cur ^ 1], 1)] = w;
            }
          }
          KtNode nd;
          nd.child1 = child[0]; nd.child2 = child[1]; nd.a = cutfeat; nd.divlow = sp_.dl; nd.divhigh = sp_.dh;
          nodes[wk.node] = nd;
        }
      }
      __syncthreads();
      if (tid == 0) ctr[cur] = 0;
      cur ^= 1;
      ++level;
      __syncthreads();
      if (level < 20) KTS_MARK(1 + level);
    }
    __syncthreads();
    KTS_MARK(30);
    if (ctr[3] != 0) {  // deeper than the search's stack: flagged, the rows keep the canonical order
      if (tid == 0) atomicExch(flag, 1);
    } else {
      for (int e = wave; e < nq; e += KTB_WAVES) {
        const int j = flist[(size_t)cloud * m + e];
        const float* qp = queries + ((size_t)cloud * m + j) * 3;
        if (!knn_tree_search_wave<IdxT, 1>(qp[0], qp[1], qp[2], k, 0, nodes, rec, rootbox, stacks + wave * (KT_DEPTH * 3),
                                        out + ((size_t)cloud * m + j) * k, lane) && lane == 0)
          atomicExch(flag, 1);
      }
    }
    KTS_MARK(31);
    __syncthreads();  // the next cloud re-uses the LDS
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// A FEW listed queries in a cloud of thousands of points (pasnl_knn_batch_ref on the segmentation models' input levels: one
// chance tie per ~1e5 queries): the search visits ~60 of the tree's ~2000 nodes, yet the whole tree was built for it -- 390 us
// for an 8192-point cloud, 470 for a lidar-like 10240-point one, every step.  Here the tree is built ON DEMAND: one workgroup
// per listed cloud holds the index list, the cut coordinate of the node being split and a table of the nodes created so far in
// LDS; wave 0 runs nanoflann's search (knn_tree_search_wave's walk and result list) and stops at a node that has not been
// split yet; the whole workgroup then splits exactly that node (middleSplit_ + planeSplit in the closed form of
// ktb_split_node_wg, coordinates gathered from the cloud) and the walk resumes.  A node's split depends on its own slice of
// the index list alone, and that slice is what its ancestors' splits left there -- all of them on the walk's path, all done
// before: the nodes that exist are the reference tree's nodes, bit for bit; subtrees the search prunes are never touched.
// A batch with more listed queries than KTL_MAXQ goes to the full builds (`nwork` tells the later kernels what is left).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KTL_NMAX = 10240;     // index list + cut coordinates + scratch positions of a cloud in LDS: 10 bytes per point
constexpr int KTL_MAXNODES = 768;   // nodes the searches of one cloud may create (a search creates ~2 per level it descends)
constexpr int KTL_MAXQ = 32;        // listed queries per BATCH this form takes (a workgroup each, side by side)
struct KtlNode {                    // child1: >= 0 first child (the second follows it), -1 leaf, -2 not split yet
  unsigned left, right;
  int child1, cutfeat;
  float divlow, divhigh;
  float box[6];                     // the box handed down to this node (what middleSplit_ reads)
};
__host__ __device__ inline size_t ktl_lds_bytes(int n) {
  return (size_t)n * 4 + (size_t)n * 4 + (((size_t)n * 2 + 15) & ~(size_t)15) + (size_t)KTL_MAXNODES * sizeof(KtlNode) + KTB_WAVES * 6 * 4 + 64 +
         (size_t)KT_DEPTH * 3 * 4;
}

// split node `nid` (count > KT_LEAF) by the whole workgroup; creates its two children.  false: the node table is full
__device__ __forceinline__ bool ktl_expand(const float* __restrict__ pts, unsigned* vind, float* vals, unsigned short* sc, float* red,
                                           KtlNode* nodes, int* ctl, const int nid, const int tid) {
  constexpr int T = KTB_WAVES * 64;
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int* redi = reinterpret_cast<int*>(red);
  const unsigned left = nodes[nid].left, right = nodes[nid].right, count = right - left;
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = nodes[nid].box[i];
  if (ctl[0] + 2 > KTL_MAXNODES) return false;  // (uniform)
  // ---- middleSplit_ (:966-1005): computeMinMax of all three dimensions in one gathering pass
  const float EPS = 0.00001f;
  float max_span = box[1] - box[0];
#pragma unroll
  for (int d = 1; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (unsigned p = tid; p < count; p += T) {
    const float* q = pts + (size_t)vind[left + p] * 3;
    const float c[3] = {q[0], q[1], q[2]};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn3[d] = c[d] < mn3[d] ? c[d] : mn3[d];
      mx3[d] = c[d] > mx3[d] ? c[d] : mx3[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(mn3[d]); mx3[d] = wave_max_f32(mx3[d]); }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[wave * 6 + 2 * d] = mn3[d]; red[wave * 6 + 2 * d + 1] = mx3[d]; }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float l = red[2 * d], h = red[2 * d + 1];
    for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, red[w * 6 + 2 * d]); h = fmaxf(h, red[w * 6 + 2 * d + 1]); }
    mn3[d] = l; mx3[d] = h;
  }
  __syncthreads();
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? box[0] : (cutfeat == 1 ? box[2] : box[4]);
  const float bhi = cutfeat == 0 ? box[1] : (cutfeat == 1 ? box[3] : box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  for (unsigned p = tid; p < count; p += T) vals[left + p] = pts[(size_t)vind[left + p] * 3 + cutfeat];
  __syncthreads();
  // ---- planeSplit (:1016-1043): the closed form, ranks per wave chunk (ktb_split_node_wg)
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    auto pred = [&](float v) { return pass == 0 ? v < cutval : v <= cutval; };
    const unsigned span = count - lo_p;
    const unsigned chunk = ((span + T - 1) / T) * 64;
    const unsigned cs = min(count, lo_p + (unsigned)wave * chunk), ce = min(count, cs + chunk);
    unsigned c = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      c += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p < ce && pred(vals[left + (p < ce ? p : cs)])));
    }
    if (lane == 0) redi[wave] = (int)c;
    __syncthreads();
    unsigned cnt = 0;
    for (int w = 0; w < KTB_WAVES; ++w) cnt += (unsigned)redi[w];
    const unsigned mid = lo_p + cnt;
    __syncthreads();
    unsigned nv = 0, nr = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      const bool in = p < ce, sat = in && pred(vals[left + (in ? p : cs)]);
      nv += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in && p < mid && !sat));
      nr += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sat && p >= mid));
    }
    if (lane == 0) { redi[wave] = (int)nv; redi[KTB_WAVES + wave] = (int)nr; }
    __syncthreads();
    unsigned vbase = 0, rbase = 0, nl = 0;
    for (int w = 0; w < KTB_WAVES; ++w) {
      const unsigned v = (unsigned)redi[w], r = (unsigned)redi[KTB_WAVES + w];
      nl += v;
      if (w < wave) vbase += v;
      if (w > wave) rbase += r;
    }
    __syncthreads();
    unsigned runv = 0, runr = 0;
    for (unsigned p0 = cs; p0 < ce; p0 += 64) {
      const unsigned p = p0 + lane;
      const bool in = p < ce, sat = in && pred(vals[left + (in ? p : cs)]);
      const bool viol = in && p < mid && !sat, rs = sat && p >= mid;
      const unsigned long long mv = __builtin_amdgcn_ballot_w64(viol), mr = __builtin_amdgcn_ballot_w64(rs);
      if (viol) sc[left + lo_p + vbase + runv + (unsigned)__builtin_popcountll(mv & lt_mask)] = (unsigned short)p;
      if (rs) {
        const unsigned asc = runr + (unsigned)__builtin_popcountll(mr & lt_mask);
        sc[right - 1 - (rbase + (nr - 1 - asc))] = (unsigned short)p;
      }
      runv += (unsigned)__builtin_popcountll(mv);
      runr += (unsigned)__builtin_popcountll(mr);
    }
    __syncthreads();
    for (unsigned i = tid; i < nl; i += T) {
      const unsigned a = left + (unsigned)sc[left + lo_p + i], b = left + (unsigned)sc[right - 1 - i];
      const unsigned ta = vind[a], tb = vind[b];
      const float va = vals[a], vb = vals[b];
      vind[a] = tb; vind[b] = ta;
      vals[a] = vb; vals[b] = va;
    }
    __syncthreads();
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  float dl = -INFINITY, dh = INFINITY;  // divlow = max of the left part, divhigh = min of the right part (:956-957)
  for (unsigned p = tid; p < count; p += T) {
    const float v = vals[left + p];
    if (p < index) dl = v > dl ? v : dl;
    else dh = v < dh ? v : dh;
  }
  dl = wave_max_f32(dl); dh = wave_min_f32(dh);
  if (lane == 0) { red[wave * 2] = dl; red[wave * 2 + 1] = dh; }
  __syncthreads();
  if (tid == 0) {
    float l = red[0], h = red[1];
    for (int w = 1; w < KTB_WAVES; ++w) { l = fmaxf(l, red[w * 2]); h = fminf(h, red[w * 2 + 1]); }
    const int c1 = ctl[0];
    ctl[0] = c1 + 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      KtlNode ch;
      ch.left = c == 0 ? left : left + index;
      ch.right = c == 0 ? left + index : right;
      ch.child1 = ch.right - ch.left <= (unsigned)KT_LEAF ? -1 : -2;
      ch.cutfeat = 0; ch.divlow = 0.f; ch.divhigh = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) ch.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : box[i];  // left: high = cutval; right: low = cutval
      nodes[c1 + c] = ch;
    }
    nodes[nid].child1 = c1;
    nodes[nid].cutfeat = cutfeat;
    nodes[nid].divlow = l;
    nodes[nid].divhigh = h;
  }
  __syncthreads();
  return true;
}

// ... and for nodes of at most 64 points (most of what a search reaches): one point per lane, both partition passes in registers
// (ktb_split_node_small on the index list: the coordinates gathered once, the partner's lane through scratch, ds_bpermute)
__device__ __forceinline__ bool ktl_expand_small(const float* __restrict__ pts, unsigned* vind, unsigned short* sc, KtlNode* nodes,
                                                 int* ctl, const int nid, const int lane) {
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned left = nodes[nid].left, right = nodes[nid].right, count = right - left;
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = nodes[nid].box[i];
  if (ctl[0] + 2 > KTL_MAXNODES) return false;
  const bool in = (unsigned)lane < count;
  unsigned vi = vind[left + (in ? lane : 0)];
  float rx, ry, rz;
  { const float* q = pts + (size_t)vi * 3; rx = q[0]; ry = q[1]; rz = q[2]; }
  const float EPS = 0.00001f;
  float max_span = box[1] - box[0];
#pragma unroll
  for (int d = 1; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3], mx3[3];
  {
    const float c[3] = {rx, ry, rz};
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(in ? c[d] : INFINITY); mx3[d] = wave_max_f32(in ? c[d] : -INFINITY); }
  }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? box[0] : (cutfeat == 1 ? box[2] : box[4]);
  const float bhi = cutfeat == 0 ? box[1] : (cutfeat == 1 ? box[3] : box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  float v = cutfeat == 0 ? rx : (cutfeat == 1 ? ry : rz);  // the cut coordinate travels with the index
  unsigned lim[2];
  unsigned lo_p = 0;
  unsigned short* scr = sc + left;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const bool inr = in && (unsigned)lane >= lo_p;
    const bool sat = inr && (pass == 0 ? v < cutval : v <= cutval);
    const unsigned cnt = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sat));
    const unsigned mid = lo_p + cnt;
    const bool viol = inr && (unsigned)lane < mid && !sat;
    const bool rsat = sat && (unsigned)lane >= mid;
    const unsigned long long mv = __builtin_amdgcn_ballot_w64(viol), mr = __builtin_amdgcn_ballot_w64(rsat);
    if (mv != 0ull) {
      const unsigned vrank = (unsigned)__builtin_popcountll(mv & lt_mask);
      const unsigned rrank = (unsigned)__builtin_popcountll(mr & ~lt_mask & ~(1ull << lane));
      if (viol) scr[vrank] = (unsigned short)lane;
      if (rsat) scr[count - 1 - rrank] = (unsigned short)lane;
      ktb_wave_sync();
      int partner = lane;
      if (viol) partner = scr[count - 1 - vrank];
      if (rsat) partner = scr[rrank];
      ktb_wave_sync();
      vi = (unsigned)__shfl((int)vi, partner);
      v = __shfl(v, partner);
    }
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  const float dl = wave_max_f32(in && (unsigned)lane < index ? v : -INFINITY);
  const float dh = wave_min_f32(in && (unsigned)lane >= index ? v : INFINITY);
  if (in) vind[left + lane] = vi;
  if (lane == 0) {
    const int c1 = ctl[0];
    ctl[0] = c1 + 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      KtlNode ch;
      ch.left = c == 0 ? left : left + index;
      ch.right = c == 0 ? left + index : right;
      ch.child1 = ch.right - ch.left <= (unsigned)KT_LEAF ? -1 : -2;
      ch.cutfeat = 0; ch.divlow = 0.f; ch.divhigh = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) ch.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : box[i];
      nodes[c1 + c] = ch;
    }
    nodes[nid].child1 = c1;
    nodes[nid].cutfeat = cutfeat;
    nodes[nid].divlow = dl;
    nodes[nid].divhigh = dh;
  }
  ktb_wave_sync();
  return true;
}

// the same split by ONE wave (the searching one) for nodes of at most KTL_WAVE_MAX points: the ~50 small nodes a search reaches cost
// ~12 us each through the workgroup form (a dozen barriers of sixteen waves); the trips of one wave over <= 512 points are cheaper.
// The code of knn_tree_build_par_body's wave-per-node split on the (index list, cut coordinate) pair.
constexpr unsigned KTL_WAVE_MAX = 512;
__device__ __forceinline__ bool ktl_expand_wave(const float* __restrict__ pts, unsigned* vind, float* vals, unsigned short* sc,
                                                KtlNode* nodes, int* ctl, const int nid, const int lane) {
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned left = nodes[nid].left, right = nodes[nid].right, count = right - left;
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = nodes[nid].box[i];
  if (ctl[0] + 2 > KTL_MAXNODES) return false;
  const float EPS = 0.00001f;
  float max_span = box[1] - box[0];
#pragma unroll
  for (int d = 1; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > max_span) max_span = span;
  }
  float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (unsigned p = lane; p < count; p += 64) {
    const float* q = pts + (size_t)vind[left + p] * 3;
    const float c[3] = {q[0], q[1], q[2]};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn3[d] = c[d] < mn3[d] ? c[d] : mn3[d];
      mx3[d] = c[d] > mx3[d] ? c[d] : mx3[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) { mn3[d] = wave_min_f32(mn3[d]); mx3[d] = wave_max_f32(mx3[d]); }
  float max_spread = -1.f, mn_c = 0.f, mx_c = 0.f;
  int cutfeat = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = box[2 * d + 1] - box[2 * d];
    if (span > (1 - EPS) * max_span) {
      const float spread = mx3[d] - mn3[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; mn_c = mn3[d]; mx_c = mx3[d]; }
    }
  }
  const float blo = cutfeat == 0 ? box[0] : (cutfeat == 1 ? box[2] : box[4]);
  const float bhi = cutfeat == 0 ? box[1] : (cutfeat == 1 ? box[3] : box[5]);
  const float split_val = (blo + bhi) / 2;
  float cutval;
  if (split_val < mn_c) cutval = mn_c;
  else if (split_val > mx_c) cutval = mx_c;
  else cutval = split_val;
  for (unsigned p = lane; p < count; p += 64) vals[left + p] = pts[(size_t)vind[left + p] * 3 + cutfeat];
  ktb_wave_sync();
  unsigned lim[2];
  unsigned lo_p = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    auto pred = [&](float v) { return pass == 0 ? v < cutval : v <= cutval; };
    unsigned cnt = 0;
    for (unsigned p0 = lo_p; p0 < count; p0 += 64) {
      const unsigned p = p0 + lane;
      cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p < count && pred(vals[left + (p < count ? p : lo_p)])));
    }
    const unsigned mid = lo_p + cnt;
    unsigned nl = 0, nr = 0;
    for (unsigned p0 = lo_p; p0 < mid; p0 += 64) {  // violators among the first cnt positions, ascending
      const unsigned p = p0 + lane;
      const bool mis = p < mid && !pred(vals[left + (p < mid ? p : lo_p)]);
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
      if (mis) sc[left + lo_p + nl + (unsigned)__builtin_popcountll(mk & lt_mask)] = (unsigned short)p;
      nl += (unsigned)__builtin_popcountll(mk);
    }
    for (unsigned q0 = 0; mid + q0 < count; q0 += 64) {  // satisfiers among the rest, descending
      const unsigned q = q0 + lane;
      const bool in = mid + q < count;
      const unsigned p = count - 1 - (in ? q : 0);
      const bool mis = in && pred(vals[left + p]);
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(mis);
      if (mis) sc[right - 1 - (nr + (unsigned)__builtin_popcountll(mk & lt_mask))] = (unsigned short)p;
      nr += (unsigned)__builtin_popcountll(mk);
    }
    ktb_wave_sync();
    for (unsigned i = lane; i < nl; i += 64) {
      const unsigned a = left + sc[left + lo_p + i], b = left + sc[right - 1 - i];
      const unsigned ta = vind[a], tb = vind[b];
      const float va = vals[a], vb = vals[b];
      vind[a] = tb; vind[b] = ta;
      vals[a] = vb; vals[b] = va;
    }
    ktb_wave_sync();
    lim[pass] = mid;
    lo_p = mid;
  }
  unsigned index;
  if (lim[0] > count / 2) index = lim[0];
  else if (lim[1] < count / 2) index = lim[1];
  else index = count / 2;
  float dl = -INFINITY, dh = INFINITY;
  for (unsigned p = lane; p < count; p += 64) {
    const float v = vals[left + p];
    if (p < index) dl = v > dl ? v : dl;
    else dh = v < dh ? v : dh;
  }
  dl = wave_max_f32(dl); dh = wave_min_f32(dh);
  if (lane == 0) {
    const int c1 = ctl[0];
    ctl[0] = c1 + 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      KtlNode ch;
      ch.left = c == 0 ? left : left + index;
      ch.right = c == 0 ? left + index : right;
      ch.child1 = ch.right - ch.left <= (unsigned)KT_LEAF ? -1 : -2;
      ch.cutfeat = 0; ch.divlow = 0.f; ch.divhigh = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) ch.box[i] = (i == 2 * cutfeat + 1 - c) ? cutval : box[i];
      nodes[c1 + c] = ch;
    }
    nodes[nid].child1 = c1;
    nodes[nid].cutfeat = cutfeat;
    nodes[nid].divlow = dl;
    nodes[nid].divhigh = dh;
  }
  ktb_wave_sync();
  return true;
}

template <typename IdxT>
__global__ __launch_bounds__(KTB_WAVES * 64) void knn_tree_lazy_kernel(int b, int n, int m, int k, const float* __restrict__ pts_all,
                                                                      const float* __restrict__ queries, IdxT* __restrict__ out,
                                                                      int* __restrict__ flag, const int* __restrict__ nflag,
                                                                      const int* __restrict__ flist, int* __restrict__ nwork) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* vind = reinterpret_cast<unsigned*>(smem);                                   // [n]
  float* vals = reinterpret_cast<float*>(vind + n);                                      // [n] cut coordinate of the node being split
  unsigned short* sc = reinterpret_cast<unsigned short*>(vals + n);                      // [n]
  KtlNode* nodes = reinterpret_cast<KtlNode*>(smem + (size_t)n * 8 + (((size_t)n * 2 + 15) & ~(size_t)15));
  float* red = reinterpret_cast<float*>(nodes + KTL_MAXNODES);                           // [KTB_WAVES * 6]
  int* ctl = reinterpret_cast<int*>(red + KTB_WAVES * 6);                                // [0] nodes used, [1] command, [2] node to split, [3] failed
  uint32_t* stk = reinterpret_cast<uint32_t*>(ctl + 16);                                 // [KT_DEPTH * 3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ONE listed query per workgroup, each with its own on-demand tree (two queries of a cloud do not wait for each other; a shared
  // tree would make them take turns: measured 170 us per query on uniform 8192-point clouds against 225 for the full build of the
  // cloud) -- if the batch lists at most KTL_MAXQ queries in all; otherwise the full builds take everything (`nwork` = the counts).
  int total = 0;
  for (int c = 0; c < b; ++c) total += nflag[c];  // (b <= a few dozen; uniform)
  if (total > KTL_MAXQ) {
    for (int c = blockIdx.x * (KTB_WAVES * 64) + tid; c < b; c += gridDim.x * KTB_WAVES * 64) nwork[c] = nflag[c];
    return;
  }
  for (int g = blockIdx.x; g < total; g += gridDim.x) {
    int cloud = 0, first = 0;
    for (int c = 0, base = 0; c < b; ++c) {
      const int cnt = nflag[c];
      if (g < base + cnt) { cloud = c; first = g - base; break; }
      base += cnt;
    }
    const int nq = first + 1;  // (this workgroup: entry `first` of the cloud's list only)
    const float* pts = pts_all + (size_t)cloud * n * 3;
    // init_vind (:1318), computeBoundingBox (:1321-1346)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += KTB_WAVES * 64) {
      vind[i] = (unsigned)i;
      const float c[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        lo[d] = c[d] < lo[d] ? c[d] : lo[d];
        hi[d] = c[d] > hi[d] ? c[d] : hi[d];
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f32(lo[d]); hi[d] = wave_max_f32(hi[d]); }
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { red[wave * 6 + 2 * d] = lo[d]; red[wave * 6 + 2 * d + 1] = hi[d]; }
    }
    __syncthreads();
    if (tid == 0) {
      KtlNode root;
      root.left = 0; root.right = (unsigned)n; root.child1 = n <= KT_LEAF ? -1 : -2; root.cutfeat = 0; root.divlow = root.divhigh = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float l = red[2 * d], h = red[2 * d + 1];
        for (int w = 1; w < KTB_WAVES; ++w) { l = fminf(l, red[w * 6 + 2 * d]); h = fmaxf(h, red[w * 6 + 2 * d + 1]); }
        root.box[2 * d] = l; root.box[2 * d + 1] = h;
      }
      nodes[0] = root;
      ctl[0] = 1; ctl[3] = 0;
    }
    __syncthreads();
    float rootbox[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rootbox[i] = nodes[0].box[i];  // root_bbox = the tight box of all points (what the search reads)
    for (int e = first; e < nq; ++e) {
      const int j = flist[(size_t)cloud * m + e];
      const float* qp = queries + ((size_t)cloud * m + j) * 3;
      const float vec[3] = {qp[0], qp[1], qp[2]};
      // ---- wave 0: nanoflann's search (knn_tree_search_wave), resumable: it stops at a node that has not been split yet
      float dists[3] = {0.f, 0.f, 0.f};
      float distsq = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (vec[d] < rootbox[2 * d]) { dists[d] = (vec[d] - rootbox[2 * d]) * (vec[d] - rootbox[2 * d]); distsq += dists[d]; }
        if (vec[d] > rootbox[2 * d + 1]) { dists[d] = (vec[d] - rootbox[2 * d + 1]) * (vec[d] - rootbox[2 * d + 1]); distsq += dists[d]; }
      }
      float ld = KT_FLT_MAX, worst = KT_FLT_MAX, mindistsq = distsq;
      int li = 0, node = 0, sp = 0;
      bool descending = true, finished = false, failed = false;
      for (;;) {
        if (wave == 0 && !finished) {
          int want = -1;
          for (;;) {
            if (descending) {
              KtlNode nd = nodes[node];
              if (nd.child1 == -2) {  // not split yet
                if (nd.right - nd.left > KTL_WAVE_MAX) { want = node; break; }  // a large node: the workgroup does it, then this step runs again
                const bool ok = nd.right - nd.left <= 64u ? ktl_expand_small(pts, vind, sc, nodes, ctl, node, lane)
                                                          : ktl_expand_wave(pts, vind, vals, sc, nodes, ctl, node, lane);
                if (!ok) { want = -2; break; }  // (node table full)
                nd = nodes[node];
              }
              if (nd.child1 == -1) {                        // leaf (:1355-1369)
                const int left = (int)nd.left, right = (int)nd.right;
                const bool in = left + lane < right;
                const unsigned pi = vind[in ? left + lane : left];
                const float* c = pts + (size_t)pi * 3;
                float dist = 0.f;
                { const float diff = vec[0] - c[0]; dist += diff * diff; }
                { const float diff = vec[1] - c[1]; dist += diff * diff; }
                { const float diff = vec[2] - c[2]; dist += diff * diff; }
                unsigned long long mask = __builtin_amdgcn_ballot_w64(in && dist < worst);
                while (mask != 0ull) {
                  const int src = (int)__builtin_ctzll(mask);
                  mask &= mask - 1ull;
                  const float cd = readlane_f(dist, src);
                  const int ci = __builtin_amdgcn_readlane((int)pi, src);
                  const int pos = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < k && ld <= cd));
                  if (pos < k) {
                    const float sd = wave_shr1_f(ld);
                    const int si = wave_shr1_i(li);
                    if (lane > pos) { ld = sd; li = si; }
                    if (lane == pos) { ld = cd; li = ci; }
                  }
                }
                worst = readlane_f(ld, k - 1);
                descending = false;
              } else {
                const int idx = nd.cutfeat;
                const float val = idx == 0 ? vec[0] : (idx == 1 ? vec[1] : vec[2]);
                const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
                int best, other;
                float cut;
                if ((diff1 + diff2) < 0) { best = nd.child1; other = nd.child1 + 1; cut = (val - nd.divhigh) * (val - nd.divhigh); }
                else { best = nd.child1 + 1; other = nd.child1; cut = (val - nd.divlow) * (val - nd.divlow); }
                if (sp + 1 >= KT_DEPTH) { failed = true; finished = true; break; }
                stk[sp * 3] = (uint32_t)other | ((uint32_t)idx << 28) | (1u << 30);
                stk[sp * 3 + 1] = __float_as_uint(mindistsq);
                stk[sp * 3 + 2] = __float_as_uint(cut);
                ++sp;
                node = best;
              }
            } else {
              if (sp == 0) { finished = true; break; }
              const uint32_t w0 = stk[(sp - 1) * 3];
              const int feat = (int)((w0 >> 28) & 3u);
              if ((w0 >> 30) == 1u) {
                const float fmind = __uint_as_float(stk[(sp - 1) * 3 + 1]), cut = __uint_as_float(stk[(sp - 1) * 3 + 2]);
                const float dst = feat == 0 ? dists[0] : (feat == 1 ? dists[1] : dists[2]);
                const float mind = fmind + cut - dst;
                if (mind * 1.f <= worst) {
                  if (feat == 0) dists[0] = cut; else if (feat == 1) dists[1] = cut; else dists[2] = cut;
                  stk[(sp - 1) * 3] = (w0 & 0x3FFFFFFFu) | (2u << 30);
                  stk[(sp - 1) * 3 + 2] = __float_as_uint(dst);
                  node = (int)(w0 & 0x0FFFFFFFu);
                  mindistsq = mind;
                  descending = true;
                } else {
                  --sp;
                }
              } else {
                const float dst = __uint_as_float(stk[(sp - 1) * 3 + 2]);
                if (feat == 0) dists[0] = dst; else if (feat == 1) dists[1] = dst; else dists[2] = dst;
                --sp;
              }
            }
          }
          if (lane == 0) { ctl[1] = finished ? 0 : 1; ctl[2] = want; if (failed) ctl[3] = 1; if (want == -2) { ctl[3] = 2; ctl[1] = 0; } }
        }
        __syncthreads();
        const int cmd = ctl[1], want = ctl[2];
        if (cmd == 0) break;  // (uniform) the search is over (or the node table is full)
        const bool ok = ktl_expand(pts, vind, vals, sc, red, nodes, ctl, want, tid);  // (ends with a workgroup barrier)
        if (!ok) {  // the node table is full: this cloud goes to the full build after all
          if (tid == 0) { ctl[3] = 2; ctl[1] = 0; }
          __syncthreads();
          break;
        }
      }
      __syncthreads();
      const int bad = ctl[3];
      if (bad == 2) {  // (uniform) hand the cloud over: every listed query of it, the done ones too (their rows are simply written again)
        if (tid == 0) atomicExch(&nwork[cloud], nflag[cloud]);
        __syncthreads();
        break;
      }
      if (bad == 1) { if (tid == 0) { atomicExch(flag, 1); ctl[3] = 0; } }  // deeper than the stack: the row keeps the canonical order
      else if (wave == 0 && lane < k) out[((size_t)cloud * m + j) * k + lane] = (IdxT)li;
      __syncthreads();
    }
    __syncthreads();  // the next cloud re-uses the LDS
  }
}

}  // namespace pasnl

using namespace pasnl;
#ifdef PASNL_TUNING
extern "C" void pasnl_tuning_stamp(int slot, hipStream_t st);
#endif

// LDS of a subtree's workgroup: records + scratch positions (18 bytes per point) for a subtree of up to min(n, KTB_LDS_NMAX)
// points (the first phase keeps larger ones to itself), or those of KTD_LDSQ_MAX points plus their two level queues --
// whichever is larger
static int knn_tree_deep_launch(int b, int n, char* clouds, size_t stride, size_t recs_off, const int* nflag, hipStream_t st) {
  const int grid = nflag ? std::min(b * 16, 64) : b * 16;  // the pending subtrees are dealt to the grid; capped under _ref (knn_tree_build_kernel's note)
  using namespace pasnl;
  const size_t cmax = (size_t)(n < KTB_LDS_NMAX ? n : KTB_LDS_NMAX);
  const size_t c = cmax < (size_t)KTD_LDSQ_MAX ? cmax : (size_t)KTD_LDSQ_MAX;
  const size_t fixed = 32 + KTB_WAVES * 6 * 4;  // counters + the scratch of the workgroup-wide split
  const size_t with_q = ((c * 18 + 15) & ~(size_t)15) + fixed + 2 * (c / (KT_LEAF + 1) + 2) * sizeof(KtWork);
  const size_t without = ((cmax * 18 + 15) & ~(size_t)15) + fixed;
  const size_t lds2 = with_q > without ? with_q : without;
  if (lds2 > 160 * 1024) return PASNL_EUNSUPPORTED;
  if (lds2 > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(knn_tree_build_deep_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
    return PASNL_ELAUNCH;
  hipLaunchKernelGGL(knn_tree_build_deep_kernel, dim3(grid), dim3(KTB_WAVES * 64), lds2, st, b, n, clouds, stride, recs_off, nflag);
  return PASNL_OK;
}

// The workspace's first 256 bytes (the flag word and its padding) zeroed by a KERNEL: a hipMemsetAsync of them captured into a
// HIP graph wrote pointer-like garbage there from the second replay on (ROCm 7.2, measured: tools/dbg/tie_capture.py), and
// the reference tie order has to work inside a captured forward
__global__ void knn_tree_clear_kernel(int* __restrict__ flag) { flag[threadIdx.x] = 0; }

size_t pasnl::knn_tree_ws_bytes(int b, int n, int m, int k) {
  if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return 0;
  return 256 + (size_t)b * kt_cloud_bytes(n);  // (the search keeps its result sets in LDS)
}
extern "C" size_t pasnl_knn_tree_workspace_bytes(int b, int n, int m, int k) { return pasnl::knn_tree_ws_bytes(b, n, m, k); }

// Build + search.  only.nflag == nullptr: every cloud, every query.  Otherwise (pasnl_knn_batch_ref): the trees of the clouds with
// flagged queries and the rows of those queries; every kernel is launched whatever the counts are (they live on the device: no
// host synchronisation, capturable) and returns at once where there is nothing to do.
// depth_flag: set to 1 (never cleared here) when a tree or a search was deeper than KT_DEPTH; nullptr: the workspace's first word,
// cleared first (the contract of pasnl_knn_batch_tree).
int pasnl::knn_tree_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                           void* workspace, size_t workspace_bytes, pasnl::KnnTieFlags only, int* depth_flag, hipStream_t st) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k <= n, PASNL_EINVAL);
  if (b == 0 || m == 0) return PASNL_OK;
  PASNL_REQUIRE(support && queries && idx && workspace, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(k <= PASNL_KNN_MAX_K, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(workspace_bytes >= knn_tree_ws_bytes(b, n, m, k), PASNL_EWORKSPACE);
  char* base = static_cast<char*>(workspace);
  int* flag = depth_flag;
  if (!flag) {
    flag = reinterpret_cast<int*>(base);  // first word: set when a tree or a search was deeper than KT_DEPTH
    hipLaunchKernelGGL(knn_tree_clear_kernel, dim3(1), dim3(64), 0, st, flag);
  }
  const int* nflag = only.nflag;
  // under _ref the kernels are launched whatever the counts are: capped grids that walk the work (knn_tree_build_kernel's note)
  const int bgrid = nflag ? std::min(b, 32) : b;
  char* clouds = base + 256;
  const size_t stride = kt_cloud_bytes(n);
  const bool serial = tune_env("PASNL_KNN_TREE_SERIAL") != nullptr;  // (tuning build: the one-lane transcription, the checker of the parallel builds)
  const size_t recs_off = kt_recs_offset(n);
  if (!serial && (n > KTB_LDS_NMAX || tune_env("PASNL_KNN_TREE_BIG")) && !(n <= KTB_NMAX && tune_env("PASNL_KNN_TREE_GATHER"))) {
    // records in the workspace, large nodes by the whole workgroup, then one workgroup per LDS-sized subtree
    hipLaunchKernelGGL(knn_tree_build_big_kernel, dim3(bgrid), dim3(KTB_WAVES * 64), 0, st, b, n, support, clouds, stride, recs_off, nflag);
    const int rc = knn_tree_deep_launch(b, n, clouds, stride, recs_off, nflag, st);
    if (rc != PASNL_OK) return rc;
  } else if (serial) {
    hipLaunchKernelGGL(knn_tree_build_kernel, dim3(bgrid), dim3(64), 0, st, b, n, support, clouds, stride, recs_off, nflag);
  } else if (n <= KTB_LDS_NMAX && tune_env("PASNL_KNN_TREE_GATHER") == nullptr) {  // (tuning build: A/B against the gathering build)
    const size_t lds = (size_t)n * 16 + (size_t)((n + 1) & ~1) * 2 + (KTB_WAVES * 6 + 8) * 4;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(knn_tree_build_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return PASNL_ELAUNCH;
    const int two_phase = tune_env("PASNL_KNN_TREE_ONE_PHASE") == nullptr;  // (tuning build: A/B)
    hipLaunchKernelGGL(knn_tree_build_lds_kernel, dim3(bgrid), dim3(KTB_WAVES * 64), lds, st, b, n, support, clouds, stride, recs_off,
                       two_phase ? KTD_TOP : 0, nflag);
    if (two_phase) {
      const int rc = knn_tree_deep_launch(b, n, clouds, stride, recs_off, nflag, st);
      if (rc != PASNL_OK) return rc;
    }
  } else {
    const size_t lds = (size_t)n * 8 + (size_t)((n + 1) & ~1) * 2 + (KTB_WAVES * 6 + 4) * 4;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(knn_tree_build_par_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return PASNL_ELAUNCH;
    const int two_phase = tune_env("PASNL_KNN_TREE_ONE_PHASE") == nullptr;  // (tuning build: A/B)
    hipLaunchKernelGGL(knn_tree_build_par_kernel, dim3(bgrid), dim3(KTB_WAVES * 64), lds, st, b, n, support, clouds, stride, recs_off,
                       two_phase ? KTD_TOP : 0, nflag);
    if (two_phase) {
      const int rc = knn_tree_deep_launch(b, n, clouds, stride, recs_off, nflag, st);
      if (rc != PASNL_OK) return rc;
    }
  }
  // few queries (the flagged ones), or a shape beyond the lane-per-query kernel's packing (K > 64: a list of several registers per
  // lane; n > 65535: 16-bit arrival numbers and indices): one WAVE per query (knn_tree_search_wave_kernel)
  if (nflag || k > 64 || n > 65535) {
    const long waves = nflag ? 1024L : (long)b * m;
    const dim3 wgrid((unsigned)std::min((waves + KTW_WAVES - 1) / KTW_WAVES, 16384L));
#define PASNL_KT_WAVE(T, S)                                                                                                       \
    hipLaunchKernelGGL((knn_tree_search_wave_kernel<T, S>), wgrid, dim3(KTW_WAVES * 64), 0, st, b, n, m, k, queries, clouds, stride, \
                       recs_off, static_cast<T*>(idx), flag, nflag, only.flist)
#define PASNL_KT_WAVES(S) { if (idx_is_i64) PASNL_KT_WAVE(long long, S); else PASNL_KT_WAVE(int, S); }
    if (k <= 64) PASNL_KT_WAVES(1) else if (k <= 128) PASNL_KT_WAVES(2) else PASNL_KT_WAVES(4)
#undef PASNL_KT_WAVES
#undef PASNL_KT_WAVE
    return pasnl_launch_status();
  }
  const long sblocks = (long)((m + 63) / 64) * b;
  dim3 grid((unsigned)(nflag ? std::min(sblocks, 512L) : sblocks));
  const size_t lds = (size_t)KT_LDS_DEPTH * 3 * 64 * 4 + (size_t)((k + 7) & ~7) * 64 * 8;
#define PASNL_KT_SEARCH(T)                                                                                                        \
  {                                                                                                                               \
    auto kern = knn_tree_search_kernel<T>;                                                                                        \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                               (int)lds) != hipSuccess)                                                           \
      return PASNL_ELAUNCH;                                                                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, st, b, n, m, k, queries, clouds, stride, recs_off, static_cast<T*>(idx), flag,    \
                       nflag, only.flist);                                                                                        \
  }
  if (idx_is_i64) PASNL_KT_SEARCH(long long) else PASNL_KT_SEARCH(int)
#undef PASNL_KT_SEARCH
  return pasnl_launch_status();
}

extern "C" int pasnl_knn_batch_tree(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                    int idx_is_i64, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  return pasnl::knn_tree_launch(b, n, m, k, support, queries, idx, idx_is_i64, workspace, workspace_bytes,
                                pasnl::KnnTieFlags{nullptr, nullptr}, nullptr, pasnl_hip_stream(stream));
}

// ---------------------------------------------------------------------------------------------------------------------
// pasnl_knn_batch_ref: the reference's result -- nanoflann's order among equal distances included -- at the canonical
// kernels' price wherever distances are distinct.  (1) the flag counters are cleared; (2) the canonical search (grid-pruned
// or brute force, as pasnl_knn_batch_ws would choose) writes every row and lists the queries whose K-list contains two equal
// distances or ends on a tie (common.hpp knn_sorted_has_tie); (3) the KD-tree of every cloud with a listed query is built
// and (4) searched for exactly those queries, whose rows are overwritten.  A query that is not listed has a single possible
// answer under both orders, so the output is cpp_knn_batch's bit for bit (knn_.cxx:72-135, nanoflann.hpp:119-123).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void knn_ref_clear_kernel(int b, int* __restrict__ nflag, int* __restrict__ nwork) {  // nwork: 2 b ints (both hand-over counters)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b) { nflag[i] = 0; nwork[i] = 0; nwork[b + i] = 0; }
}

namespace {
struct RefLayout { size_t nflag, nwork, nwork2, flist, grid, tree, total; };
RefLayout ref_layout(int b, int n, int m, int k) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  RefLayout L;
  L.nflag = 0;
  L.nwork = al((size_t)b * 4);
  L.nwork2 = L.nwork + (size_t)b * 4;  // (right behind: one clearing pass, one aligned block)
  L.flist = L.nwork + al((size_t)b * 8);
  L.grid = L.flist + al((size_t)b * m * 4);
  L.tree = L.grid + al(k <= 64 ? pasnl::knn_grid_ws_bytes(b, n) : 0);
  L.total = L.tree + al(pasnl::knn_tree_ws_bytes(b, n, m, k));
  return L;
}
}  // namespace

extern "C" size_t pasnl_knn_batch_ref_workspace_bytes(int b, int n, int m, int k) {
  if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return 0;
  return ref_layout(b, n, m, k).total;
}

extern "C" int pasnl_knn_batch_ref(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                   int idx_is_i64, int* depth_flag, void* workspace, size_t workspace_bytes, int max_workgroups,
                                   pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0 && max_workgroups >= 0, PASNL_EINVAL);
  PASNL_REQUIRE(k <= n, PASNL_EINVAL);
  if (b == 0 || m == 0) return PASNL_OK;
  PASNL_REQUIRE(support && queries && idx && workspace && depth_flag, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(k <= PASNL_KNN_MAX_K, PASNL_EUNSUPPORTED);
  const RefLayout L = ref_layout(b, n, m, k);
  PASNL_REQUIRE(workspace_bytes >= L.total, PASNL_EWORKSPACE);
  hipStream_t st = pasnl_hip_stream(stream);
  char* base = static_cast<char*>(workspace);
  pasnl::KnnTieFlags flags{reinterpret_cast<int*>(base + L.nflag), reinterpret_cast<int*>(base + L.flist)};
#ifdef PASNL_TUNING
  const bool stamp = pasnl::tune_env("PASNL_STAMP_N") && atoi(pasnl::tune_env("PASNL_STAMP_N")) == n;
  if (stamp) pasnl_tuning_stamp(0, st);
#define PASNL_STAMP(i) do { if (stamp) pasnl_tuning_stamp(i, st); } while (0)
#else
#define PASNL_STAMP(i) do { } while (0)
#endif
  hipLaunchKernelGGL(knn_ref_clear_kernel, dim3((b + 255) / 256), dim3(256), 0, st, b, flags.nflag, reinterpret_cast<int*>(base + L.nwork));
  int rc = pasnl::knn_grid_launch(b, n, m, k, support, queries, idx, idx_is_i64, nullptr, base + L.grid, L.tree - L.grid,
                                  max_workgroups, flags, st);
  if (rc != PASNL_OK) return rc;
  PASNL_STAMP(1);
  if (pasnl::tune_env("PASNL_KNN_REF_NO_TREE")) return pasnl_launch_status();  // (tuning build: the canonical search + flags alone, A/B)
  if (const char* e = pasnl::tune_env("PASNL_KNN_REF_NO_TREE_ABOVE")) { if (n > atoi(e)) return pasnl_launch_status(); }  // (... for the large / the small
  if (const char* e = pasnl::tune_env("PASNL_KNN_REF_NO_TREE_BELOW")) { if (n < atoi(e)) return pasnl_launch_status(); }  //      searches of a model only)
  const bool small = n <= pasnl::KTS_NMAX && k <= 64;  // one kernel: the tie paths of a cloud's few listed queries, else its tree + searches
  if (!small && pasnl::tune_env("PASNL_KNN_REF_NO_TIE_PATH") == nullptr) {
    // a FEW listed queries (chance ties): the runs of equal distances put in the tree's arrival order along the tree paths that
    // separate them, a workgroup per listed cloud; `nwork` (b ints behind the lists' counters) = what is left to the builds below.
    // Clouds of more than KTB_LDS_NMAX points: the form that reads the cloud in global memory (no records in LDS).
    int* nwork = reinterpret_cast<int*>(base + L.nwork);
    const bool global = n > pasnl::KTB_LDS_NMAX;
    const size_t lds = pasnl::ktp_lds_bytes(global ? 0 : n);
#define PASNL_KTP(T, G)                                                                                                          \
    {                                                                                                                             \
      auto kern = pasnl::knn_tie_path_kernel<T, G>;                                                                               \
      if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                 (int)lds) != hipSuccess)                                                         \
        return PASNL_ELAUNCH;                                                                                                     \
      hipLaunchKernelGGL(kern, dim3(pasnl::KTP_MAXQ), dim3(pasnl::KTB_WAVES * 64), lds, st, b, n, m, k, support, queries,          \
                         static_cast<T*>(idx), flags.nflag, flags.flist, nwork);                                                  \
    }
    if (global) { if (idx_is_i64) PASNL_KTP(long long, true) else PASNL_KTP(int, true) }
    else { if (idx_is_i64) PASNL_KTP(long long, false) else PASNL_KTP(int, false) }
#undef PASNL_KTP
    flags.nflag = nwork;
    if (pasnl::tune_env("PASNL_KNN_REF_TIE_PATH_ONLY")) return pasnl_launch_status();  // (tuning build: timing without the builds' launches)
  }
  if (small) {  // (tree + searches of a listed cloud in one workgroup, all in LDS)
    const size_t lds = pasnl::kts_lds_bytes(n);
    const int grid = pasnl::tune_env("PASNL_KNN_REF_EMPTY_TREE") ? 0 : std::min(b, 64);
    if (grid == 0) return pasnl_launch_status();
#define PASNL_KTS(T)                                                                                                             \
    {                                                                                                                             \
      auto kern = pasnl::knn_tree_small_kernel<T>;                                                                                \
      if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                 (int)lds) != hipSuccess)                                                         \
        return PASNL_ELAUNCH;                                                                                                     \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(pasnl::KTB_WAVES * 64), lds, st, b, n, m, k, support, queries, static_cast<T*>(idx), \
                         depth_flag, flags.nflag, flags.flist);                                                                   \
    }
    if (idx_is_i64) PASNL_KTS(long long) else PASNL_KTS(int)
#undef PASNL_KTS
    PASNL_STAMP(2);
    return pasnl_launch_status();
  }
  if (n > pasnl::KTB_LDS_NMAX && n <= pasnl::KTL_NMAX && k <= 64) {
    // a FEW listed queries (the chance ties of the input levels) in clouds whose full build is the slow one (records in the
    // workspace: 455 us for a lidar-like 10240-point cloud): the tree on demand, along each search's path (177 us); `nwork` (b ints
    // behind the lists' counters) = what it leaves to the full builds below.  (Clouds up to 8192 points keep the full build: 225 us
    // with the records in LDS, against 170-300 per on-demand search on uniform clouds, whose searches reach ~100 nodes.)
    int* nwork = reinterpret_cast<int*>(base + L.nwork2);  // (its input: what the tie paths left)
    const size_t lds = pasnl::ktl_lds_bytes(n);
    const int grid = pasnl::KTL_MAXQ;  // one workgroup per listed query
#define PASNL_KTL(T)                                                                                                             \
    {                                                                                                                             \
      auto kern = pasnl::knn_tree_lazy_kernel<T>;                                                                                 \
      if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                 (int)lds) != hipSuccess)                                                         \
        return PASNL_ELAUNCH;                                                                                                     \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(pasnl::KTB_WAVES * 64), lds, st, b, n, m, k, support, queries, static_cast<T*>(idx), \
                         depth_flag, flags.nflag, flags.flist, nwork);                                                            \
    }
    if (idx_is_i64) PASNL_KTL(long long) else PASNL_KTL(int)
#undef PASNL_KTL
    flags.nflag = nwork;
  }
  return pasnl::knn_tree_launch(b, n, m, k, support, queries, idx, idx_is_i64, base + L.tree, L.total - L.tree, flags, depth_flag, st);
}

#ifdef PASNL_TUNING
// time stamps inside a captured step (tools/step_stamps.py): a one-thread kernel per stamp, wall_clock64() = the 100 MHz constant clock
__device__ unsigned long long pasnl_stamps[16];
__global__ void pasnl_stamp_kernel(int slot) { pasnl_stamps[slot] = wall_clock64(); }
extern "C" void pasnl_tuning_stamp(int slot, hipStream_t st) { hipLaunchKernelGGL(pasnl_stamp_kernel, dim3(1), dim3(1), 0, st, slot); }
extern "C" int pasnl_tuning_stamps_read(unsigned long long* host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(pasnl_stamps), sizeof(pasnl_stamps)) == hipSuccess ? 0 : -1;
}
extern "C" int pasnl_tie_path_wg_read(unsigned long long* host64) {
  return hipMemcpyFromSymbol(host64, HIP_SYMBOL(pasnl::ktp_wg_cycles), sizeof(pasnl::ktp_wg_cycles)) == hipSuccess ? 0 : -1;
}
extern "C" int pasnl_tie_paths_read(int* host8, int clear) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(pasnl::ktp_paths), sizeof(pasnl::ktp_paths)) != hipSuccess) return -1;
  if (clear) { int z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pasnl::ktp_paths), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int pasnl_knn_small_probe_read(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(pasnl::kts_probe), sizeof(pasnl::kts_probe)) == hipSuccess ? 0 : -1;
}
#endif
