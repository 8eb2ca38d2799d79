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
//   * knn_tree_build_kernel: ONE lane per cloud builds the tree serially into a caller workspace (the build is inherently a
//     sequence of in-place partitions; an optional exactness mode, not a fast path);
//   * knn_tree_search_kernel: one lane per query walks it with nanoflann's result-set insertion.
// Trees or searches deeper than KT_DEPTH levels (pathological, exponentially clustered data) raise a flag in the workspace,
// which the Python wrapper turns into PasnlUnsupported.
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
// workspace of one cloud: [flag, root, nodes used, depth] | vind[n] | nodes[2n] | build frames[KT_DEPTH]
static inline size_t kt_cloud_bytes(int n) {
  return kt_align_h(16) + kt_align_h((size_t)n * 4) + kt_align_h((size_t)2 * n * sizeof(KtNode)) + kt_align_h((size_t)KT_DEPTH * sizeof(KtFrame));
}

__global__ __launch_bounds__(64) void knn_tree_build_kernel(int n, const float* __restrict__ pts_all, char* __restrict__ ws_all,
                                                           size_t stride) {
  if (threadIdx.x != 0) return;
  const float* pts = pts_all + (size_t)blockIdx.x * n * 3;
  char* ws = ws_all + (size_t)blockIdx.x * stride;
  int* hdr = reinterpret_cast<int*>(ws);
  unsigned* vind = reinterpret_cast<unsigned*>(ws + kt_align(16));
  KtNode* nodes = reinterpret_cast<KtNode*>(ws + kt_align(16) + kt_align((size_t)n * 4));
  KtFrame* st = reinterpret_cast<KtFrame*>(ws + kt_align(16) + kt_align((size_t)n * 4) + kt_align((size_t)2 * n * sizeof(KtNode)));
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
  hdr[1] = st[0].node;
  hdr[2] = nnodes;
  hdr[3] = maxdepth;
}

struct KtSearchFrame {
  int node, other, feat, state;
  float mindistsq, cut, dst;
};

template <typename IdxT>
__global__ __launch_bounds__(64) void knn_tree_search_kernel(int n, int m, int k, const float* __restrict__ pts_all,
                                                            const float* __restrict__ queries, const char* __restrict__ ws_all,
                                                            size_t stride, float* __restrict__ rdist_all, int* __restrict__ ridx_all,
                                                            IdxT* __restrict__ out, int* __restrict__ flag) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= m) return;
  const int bi = blockIdx.y;
  const float* pts = pts_all + (size_t)bi * n * 3;
  const char* ws = ws_all + (size_t)bi * stride;
  const int* hdr = reinterpret_cast<const int*>(ws);
  if (hdr[0] != 0) { if (j == 0) atomicExch(flag, 1); return; }
  const unsigned* vind = reinterpret_cast<const unsigned*>(ws + kt_align(16));
  const KtNode* nodes = reinterpret_cast<const KtNode*>(ws + kt_align(16) + kt_align((size_t)n * 4));
  const KtFrame* bst = reinterpret_cast<const KtFrame*>(ws + kt_align(16) + kt_align((size_t)n * 4) + kt_align((size_t)2 * n * sizeof(KtNode)));
  const float* rootbox = bst[1].bbox;
  const float* qp = queries + ((size_t)bi * m + j) * 3;
  const float vec[3] = {qp[0], qp[1], qp[2]};
  float* rd = rdist_all + ((size_t)bi * m + j) * k;   // KNNResultSet: dists / indices, sorted, `count` valid entries
  int* ri = ridx_all + ((size_t)bi * m + j) * k;
  int count = 0;
  rd[k - 1] = 3.402823466e+38f;  // init(): dists[capacity-1] = max (:91-92)
  // computeInitialDistances (:1045-1061)
  float dists[3] = {0.f, 0.f, 0.f};
  float distsq = 0.f;
  for (int d = 0; d < 3; ++d) {
    if (vec[d] < rootbox[2 * d]) { dists[d] = (vec[d] - rootbox[2 * d]) * (vec[d] - rootbox[2 * d]); distsq += dists[d]; }
    if (vec[d] > rootbox[2 * d + 1]) { dists[d] = (vec[d] - rootbox[2 * d + 1]) * (vec[d] - rootbox[2 * d + 1]); distsq += dists[d]; }
  }
  const float epsError = 1.f;  // 1 + SearchParams(10).eps, eps = 0
  KtSearchFrame st[KT_DEPTH];
  int sp = 0;
  st[0].node = hdr[1]; st[0].mindistsq = distsq; st[0].state = 0;
  while (sp >= 0) {
    KtSearchFrame& f = st[sp];
    const KtNode nd = nodes[f.node];
    if (f.state == 0) {
      if (nd.child1 < 0) {  // leaf (:1355-1369): the worst distance is read ONCE, before the scan
        const float worst = rd[k - 1];
        const int left = nd.a, right = __float_as_int(nd.divlow);
        for (int i = left; i < right; ++i) {
          const unsigned index = vind[i];
          // L2_Adaptor::evalMetric, dim 3: only the tail loop runs (:343-346): result += diff * diff, diff = query - point
          float dist = 0.f;
          for (int d = 0; d < 3; ++d) {
            const float diff = vec[d] - pts[(size_t)index * 3 + d];
            dist += diff * diff;
          }
          if (dist < worst) {  // KNNResultSet::addPoint (:115-134): behind the entries of equal distance
            int p;
            for (p = count; p > 0; --p) {
              if (rd[p - 1] > dist) {
                if (p < k) { rd[p] = rd[p - 1]; ri[p] = ri[p - 1]; }
              } else break;
            }
            if (p < k) { rd[p] = dist; ri[p] = (int)index; }
            if (count < k) ++count;
          }
        }
        --sp;
        continue;
      }
      const int idx = nd.a;
      const float val = vec[idx];
      const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
      int best;
      if ((diff1 + diff2) < 0) { best = nd.child1; f.other = nd.child2; f.cut = (val - nd.divhigh) * (val - nd.divhigh); }
      else { best = nd.child2; f.other = nd.child1; f.cut = (val - nd.divlow) * (val - nd.divlow); }
      f.feat = idx;
      f.state = 1;
      if (sp + 1 >= KT_DEPTH) { atomicExch(flag, 1); return; }
      st[sp + 1].node = best; st[sp + 1].mindistsq = f.mindistsq; st[sp + 1].state = 0;
      ++sp;
    } else if (f.state == 1) {  // the near child is done (:1397-1405)
      const float dst = f.feat == 0 ? dists[0] : (f.feat == 1 ? dists[1] : dists[2]);
      const float mind = f.mindistsq + f.cut - dst;
      f.dst = dst;
      if (f.feat == 0) dists[0] = f.cut; else if (f.feat == 1) dists[1] = f.cut; else dists[2] = f.cut;
      if (mind * epsError <= rd[k - 1]) {
        f.state = 2;
        st[sp + 1].node = f.other; st[sp + 1].mindistsq = mind; st[sp + 1].state = 0;
        ++sp;
      } else {
        if (f.feat == 0) dists[0] = dst; else if (f.feat == 1) dists[1] = dst; else dists[2] = dst;
        --sp;
      }
    } else {
      if (f.feat == 0) dists[0] = f.dst; else if (f.feat == 1) dists[1] = f.dst; else dists[2] = f.dst;
      --sp;
    }
  }
  IdxT* o = out + ((size_t)bi * m + j) * k;
  for (int s = 0; s < k; ++s) o[s] = (IdxT)ri[s];
}

}  // namespace pasnl

using namespace pasnl;

extern "C" size_t pasnl_knn_tree_workspace_bytes(int b, int n, int m, int k) {
  if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return 0;
  return 256 + (size_t)b * kt_cloud_bytes(n) + kt_align_h((size_t)b * m * k * 4) * 2;
}

extern "C" int pasnl_knn_batch_tree(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                                    int idx_is_i64, void* workspace, size_t workspace_bytes, pasnl_stream_t stream) {
  PASNL_REQUIRE(b >= 0 && n > 0 && m >= 0 && k > 0, PASNL_EINVAL);
  PASNL_REQUIRE(k <= n, PASNL_EINVAL);
  if (b == 0 || m == 0) return PASNL_OK;
  PASNL_REQUIRE(support && queries && idx && workspace, PASNL_ENULL);
  PASNL_REQUIRE(b <= 65535, PASNL_EUNSUPPORTED);
  PASNL_REQUIRE(workspace_bytes >= pasnl_knn_tree_workspace_bytes(b, n, m, k), PASNL_EWORKSPACE);
  hipStream_t st = pasnl_hip_stream(stream);
  char* base = static_cast<char*>(workspace);
  int* flag = reinterpret_cast<int*>(base);  // first word: set when a tree or a search was deeper than KT_DEPTH
  if (hipMemsetAsync(flag, 0, 256, st) != hipSuccess) return PASNL_ELAUNCH;
  char* clouds = base + 256;
  const size_t stride = kt_cloud_bytes(n);
  float* rdist = reinterpret_cast<float*>(clouds + (size_t)b * stride);
  int* ridx = reinterpret_cast<int*>(reinterpret_cast<char*>(rdist) + kt_align_h((size_t)b * m * k * 4));
  hipLaunchKernelGGL(knn_tree_build_kernel, dim3(b), dim3(64), 0, st, n, support, clouds, stride);
  dim3 grid((m + 63) / 64, b);
  if (idx_is_i64)
    hipLaunchKernelGGL((knn_tree_search_kernel<long long>), grid, dim3(64), 0, st, n, m, k, support, queries, clouds, stride, rdist,
                       ridx, static_cast<long long*>(idx), flag);
  else
    hipLaunchKernelGGL((knn_tree_search_kernel<int>), grid, dim3(64), 0, st, n, m, k, support, queries, clouds, stride, rdist, ridx,
                       static_cast<int*>(idx), flag);
  return pasnl_launch_status();
}
