/*
 * pasnl.h -- C ABI of libpasnl_hip.so: the MI355X (gfx950) set-abstraction hot path of PointASNL.
 *
 * This is the drop-in boundary.  Every entry point replaces one C++ "Launcher" (or CPU loop) that the
 * reference's TensorFlow OpKernels call with raw pointers (citations are file:line in the reference
 * tree).  Conventions, identical for every function:
 *
 *   - all tensors are contiguous row-major, float32 / int32 (int64 only where stated);
 *   - pointers are DEVICE pointers owned by the caller (the Python host hands in torch storage);
 *     inputs are borrowed const, outputs are fully overwritten;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream); no implicit synchronisation, no allocation, no global state -> graph-capturable and
 *     thread-safe;
 *   - the return value is PASNL_OK (0) or a negative PASNL_E* code; nothing is enqueued on error.
 *     `pasnl_strerror` gives the text.  Shape rules mirror the reference's OP_REQUIRES checks.
 *   - b == 0 or an empty query/result dimension is a successful no-op.
 *
 * Arithmetic is the canonical fp32 of SURVEY.md Appendix A: IEEE round-to-nearest, no FMA
 * contraction, operations in the written order.  Index outputs are bit-exact against oracle/.
 */
#ifndef PASNL_H_
#define PASNL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PASNL_VERSION 100 /* 0.1.0 */

enum {
  PASNL_OK = 0,
  PASNL_EINVAL = -1,       /* bad dimension / attribute (the reference's InvalidArgument)   */
  PASNL_ENULL = -2,        /* a required pointer is NULL                                     */
  PASNL_EWORKSPACE = -3,   /* workspace smaller than pasnl_*_workspace_bytes                 */
  PASNL_ELAUNCH = -4,      /* hipGetLastError() != hipSuccess after the launch               */
  PASNL_EUNSUPPORTED = -5  /* valid request outside what the kernels cover (documented)      */
};

typedef void* pasnl_stream_t; /* hipStream_t */

int pasnl_version(void);
const char* pasnl_strerror(int code);
/* Number of HIP devices visible to the library (<0: HIP runtime error).  Used by the host to fail
 * loudly instead of falling back to a CPU path. */
int pasnl_device_count(void);

/* ------------------------------------------------------------------ sampling (tf_ops/sampling) */

/* Iterative farthest point sampling; idx[b,0] = 0.
 * replaces farthestpointsamplingLauncher  tf_sampling_g.cu:203-205 (kernel :105-170)
 * xyz (b,n,3) f32 -> idx (b,m) i32.  Tie rule: lowest (k mod 512, k) among maxima.
 * The reference needs a 32*n float temp (tf_sampling.cpp:115); this kernel keeps the running
 * distances in registers, so no workspace is required (m <= 0 -> PASNL_EINVAL as tf_sampling.cpp:99). */
int pasnl_farthest_point_sample(int b, int n, int m, const float* xyz, int* idx, pasnl_stream_t stream);

/* The same sampling, and the gather of the sampled coordinates in the same launch: new_xyz[b,j,:] = xyz[b, idx[b,j], :]
 * (bit-equal to pasnl_gather_point on idx).  Replaces the pair farthest_point_sample + gather_point every caller of the
 * reference runs back to back (pointasnl_util.py:33-49 sampling(), pointnet_util.py:44): the sampler has the picks in
 * LDS when it ends, so the second launch -- on the critical path of every set-abstraction layer -- disappears. */
int pasnl_farthest_point_sample_gather(int b, int n, int m, const float* xyz, int* idx, float* new_xyz,
                                       pasnl_stream_t stream);

/* out[b,j,:] = inp[b, idx[b,j], :] for 3-wide rows.
 * replaces gatherpointLauncher  tf_sampling_g.cu:206-208 (kernel :172-181) */
int pasnl_gather_point(int b, int n, int m, const float* inp, const int* idx, float* out, pasnl_stream_t stream);

/* inp_g (b,n,3) = scatter-add of out_g (b,m,3) through idx (b,m); inp_g is zeroed first.
 * replaces cudaMemset + scatteraddpointLauncher  tf_sampling.cpp:174-175, tf_sampling_g.cu:183-192 */
int pasnl_gather_point_grad(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, pasnl_stream_t stream);

/* Inverse-CDF sampling: out[b,j] = first r with cdf[b,r] >= inpr[b,j]*cdf[b,n-1]; `temp` (b*n floats)
 * receives the running sum.  replaces probsampleLauncher  tf_sampling_g.cu:198-201 (:7-104) */
int pasnl_prob_sample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out, pasnl_stream_t stream);

/* ------------------------------------------------------------------ grouping (tf_ops/grouping) */

/* First `nsample` points (ascending index) with max(sqrtf(d2),1e-20f) < radius, padded with the first
 * hit; pts_cnt = min(hits, nsample); zero-hit rows are all 0 (the reference leaves them undefined).
 * replaces queryBallPointLauncher  tf_grouping_g.cu:125-126 (kernel :3-36)
 * xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> idx (b,m,nsample) i32, pts_cnt (b,m) i32 */
int pasnl_query_ball_point(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                           int* idx, int* pts_cnt, pasnl_stream_t stream);

/* out[b,j,k,:] = points[b, idx[b,j,k], :], row length c.
 * replaces groupPointLauncher  tf_grouping_g.cu:133-134 (kernel :40-57).  With nsample == 1 this is the
 * C-wide row gather the models do through tf.gather_nd (pointasnl_util.py:43-49,63-71). */
int pasnl_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out,
                      pasnl_stream_t stream);

/* grad_points (b,n,c) = scatter-add of grad_out (b,m,nsample,c); zeroed first.
 * replaces cudaMemset + groupPointGradLauncher  tf_grouping.cpp:204-205, tf_grouping_g.cu:61-78 */
int pasnl_group_point_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                           float* grad_points, pasnl_stream_t stream);

/* Set-abstraction grouping, fused (pointasnl_util.py:63-74 + :248-249 + :258): for every query j and neighbour s
 *   new_point[b,j,s,:] = [ xyz[b,i,:] - new_xyz[b,j,:] | xyz[b,i,:] | feature[b,i,:] ],  i = idx[b,j,s]
 *   skip_max[b,j,:]    = max over s of new_point[b,j,s,:]
 * i.e. the two tf.gather_nd, the concat with xyz, the translation normalisation, the second concat and the
 * reduce_max of the skip connection in one pass; new_point is written once, nothing else touches HBM.
 * xyz (b,n,3), feature (b,n,c), idx (b,m,k) i32, new_xyz (b,m,3) -> new_point (b,m,k,6+c), skip_max (b,m,6+c). */
int pasnl_sa_group(int b, int n, int c, int m, int k, const float* xyz, const float* feature, const int* idx,
                   const float* new_xyz, float* new_point, float* skip_max, pasnl_stream_t stream);

/* k rounds of in-place selection sort per row of a (b,m,n) distance tensor; FULL (b,m,n) outputs, first k
 * columns meaningful, the tail is the swap residue.  replaces selectionSortLauncher
 * tf_grouping_g.cu:129-130 (kernel :83-123) */
int pasnl_select_top_k(int b, int n, int m, int k, const float* dist, int* outi, float* out, pasnl_stream_t stream);

/* Exact K nearest neighbours, ascending by (squared distance, index); the grouping search of all three
 * models.  replaces cpp_knn_batch / cpp_knn_batch_omp  utils/nearest_neighbors/knn_.cxx:72-135
 * (binding knn.pyx:71-109).  support (b,n,3), queries (b,m,3) -> idx (b,m,k), int32 when
 * idx_is_i64 == 0, int64 (the reference's `long`) otherwise.  dist2 (b,m,k) f32 is optional (NULL).
 * Requires 1 <= k <= n; k <= PASNL_KNN_MAX_K. */
#define PASNL_KNN_MAX_K 256
int pasnl_knn_batch(int b, int n, int m, int k, const float* support, const float* queries, void* idx,
                    int idx_is_i64, float* dist2, pasnl_stream_t stream);

/* Coverage-driven query selection + kNN.  replaces cpp_knn_batch_distance_pick(_omp)  knn_.cxx:136-266 (binding
 * knn.pyx:111-148): per cloud, nq times, among the points used least often so far take number (rnd % how many) in ascending
 * index order, output its k nearest neighbours (ascending (squared distance, index)) and its coordinates, raise the use
 * count of the neighbours by 1 and of the pick by 100.  pts (b,n,3) -> idx (b,nq,k) int64, queries (b,nq,3).  rnd (b,nq)
 * uint32 = the outputs of the caller's generator: the reference seeds one std::mt19937 with time(0) and walks the clouds in
 * order, so cloud i consumes outputs [i nq, (i+1) nq) (the Python mirror draws them with the same generator from an
 * explicit seed).  n <= 16384. */
int pasnl_knn_distance_pick(int b, int n, int nq, int k, const float* pts, const unsigned int* rnd, long long* idx,
                            float* queries, pasnl_stream_t stream);

/* The same search with a caller-provided workspace.  Clouds of PASNL_KNN_GRID_MIN_N <= n <= 16384 points and k <= 64 are
 * searched through a uniform grid built in the workspace (csrc/knn_grid.hip: counting sort by cell, expanding rings of
 * cells, acceptance only when no unexamined cell can hold a closer or tying point -> results bit-identical to
 * pasnl_knn_batch for every input); everything else is forwarded to pasnl_knn_batch.  pasnl_knn_workspace_bytes returns
 * the bytes the pair (b, n) needs (0: the grid is not used and workspace may be NULL).  The workspace is scratch: nothing
 * is kept between calls. */
#define PASNL_KNN_GRID_MIN_N 4096
size_t pasnl_knn_workspace_bytes(int b, int n);
int pasnl_knn_batch_ws(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                       float* dist2, void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* pasnl_knn_batch_ws as a BACKGROUND job: the query kernel runs on at most max_workgroups workgroups (a capped grid that walks
 * the queries) instead of one wave per query -- for a search enqueued on a side stream beside other work (a serving loop's
 * prefetch of the next batch's neighbour lists): a saturating grid of ~20 000 workgroups leaves the kernels of the other stream
 * waiting for free slots (measured: a 10-us kernel of the forward stretched to 225 us beside it).  Same results.  Falls back to
 * pasnl_knn_batch (uncapped) where the grid form does not apply. */
int pasnl_knn_batch_ws_bg(int b, int n, int m, int K, const float* support, const float* queries, void* idx, int idx_is_i64,
                          float* dist2, void* workspace, size_t workspace_bytes, int max_workgroups, pasnl_stream_t stream);

/* The same K nearest neighbours in the REFERENCE'S order among exactly equal distances (and with the reference's choice of
 * which of several tied candidates is the K-th): nanoflann keeps candidates of equal distance in the order its KD-tree visits
 * them (nanoflann.hpp:115-134, :1351-1410; tree: divideTree / middleSplit_ / planeSplit :916-1043, leaf size 10,
 * knn_.cxx:83).  Replaces cpp_knn_batch knn_.cxx:72-101 bit for bit, ties included, by rebuilding that tree and that search on
 * the GPU -- an exactness mode (a workgroup per cloud builds the tree level by level, a lane per query searches it: about
 * 1 ms for 16 clouds of 8192 points where the canonical kernels take 0.06), not a fast path; pasnl_knn_batch returns the
 * canonical (distance, index) order, identical whenever distances are distinct.  k <= n, k <= PASNL_KNN_MAX_K (k > 64 or
 * n > 65535: one wave per query instead of one lane; n > 10240: the tree from a one-lane build -- seconds at 1e5 points).  workspace:
 * pasnl_knn_tree_workspace_bytes(b, n, m, k) bytes; its first int32 is non-zero afterwards if a tree or a search was deeper
 * than 96 levels (pathological clustering: the result is then not valid). */
size_t pasnl_knn_tree_workspace_bytes(int b, int n, int m, int k);
int pasnl_knn_batch_tree(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                         void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* cpp_knn_batch's RESULT, ties included, at the canonical kernels' price -- what the Python mirror calls by default.
 * replaces cpp_knn_batch / cpp_knn_batch_omp  knn_.cxx:72-135 (result-set order: nanoflann.hpp:115-134).
 * nanoflann's list can differ from the (distance, index) list only where distances are EQUAL -- two of them inside a query's
 * K-list, or a candidate beyond the list at exactly the K-th distance.  So: the canonical search (pasnl_knn_batch_ws's choice
 * of kernel) writes every row and, from the sorted keys it already holds, lists the queries with such a tie.  A listed row
 * differs from nanoflann's only INSIDE its runs of equal distances, and two points of a run are reached by nanoflann's search in
 * the order decided at the tree node that separates them (the query's near child first, :1380-1393; positions left to right in a
 * leaf): for a FEW listed queries (<= 32 listed clouds, <= 16 queries and <= 64 tied points a cloud) only those nodes of the
 * reference tree are computed -- as point sets, one pass of reductions over the cloud per level, no tree built, no search run;
 * two tied points in one leaf (duplicated points): the records of a cloud of more than 2048 points are moved along that one
 * path.  Everything else: the KD-tree of the cloud is built and searched for its listed queries (pasnl_knn_batch_tree's
 * kernels).  All counts stay on the device: every kernel is launched and returns at once where nothing is listed -- no host
 * synchronisation, capturable.  Tie-free clouds pay two to five empty launches; output == pasnl_knn_batch_tree's bit for bit.
 * depth_flag (device int, required): set to 1 -- never cleared by the library -- if a listed query's tree or search was deeper
 * than 96 levels; such rows KEEP the canonical order (valid neighbours, canonical order among equals).
 * max_workgroups > 0: the grid-pruned canonical search as a background job (pasnl_knn_batch_ws_bg); 0: the usual grid.
 * k <= n, k <= PASNL_KNN_MAX_K.  Clouds of up to 2048 points (k <= 64): ONE kernel after the search -- the tie paths of a cloud's
 * <= 4 listed queries, else its tree and searches in one workgroup, all in LDS; larger ones: a tie-path kernel, then the builds
 * of pasnl_knn_batch_tree + one wave per listed query.
 * workspace: pasnl_knn_batch_ref_workspace_bytes(b, n, m, k); afterwards its first b int32 = the listed queries per cloud, and,
 * 256-byte aligned behind them, 2 b int32: what the tie paths / the on-demand tree left to the next stage (diagnostics). */
size_t pasnl_knn_batch_ref_workspace_bytes(int b, int n, int m, int k);
int pasnl_knn_batch_ref(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64,
                        int* depth_flag, void* workspace, size_t workspace_bytes, int max_workgroups, pasnl_stream_t stream);

/* ------------------------------------------------------- interpolation (tf_ops/3d_interpolation) */

/* Three nearest known points, squared distances ascending, lowest index first on ties.
 * replaces threenn_cpu  tf_interpolate.cpp:60-103
 * xyz1 (b,n,3) unknown, xyz2 (b,m,3) known -> dist (b,n,3) f32, idx (b,n,3) i32 */
int pasnl_three_nn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx,
                   pasnl_stream_t stream);

/* out[b,j,l] = (p[i1,l]*w1 + p[i2,l]*w2) + p[i3,l]*w3.
 * replaces threeinterpolate_cpu  tf_interpolate.cpp:107-127   (note the argument order b,m,c,n) */
int pasnl_three_interpolate(int b, int m, int c, int n, const float* points, const int* idx, const float* weight,
                            float* out, pasnl_stream_t stream);

/* grad_points (b,m,c) = scatter-add; zeroed first.
 * replaces memset + threeinterpolate_grad_cpu  tf_interpolate.cpp:258-259 (:131-153) */
int pasnl_three_interpolate_grad(int b, int n, int c, int m, const float* grad_out, const int* idx,
                                 const float* weight, float* grad_points, pasnl_stream_t stream);

/* Inverse-distance weights for three_interpolate: d=max(d,1e-10); w=(1/d)/sum(1/d), in that order.
 * replaces the four TF ops at pointasnl_util.py:308-311 / pointnet_util.py:212-215
 * dist (rows,3) -> weight (rows,3) */
int pasnl_three_weights(int rows, const float* dist, float* weight, pasnl_stream_t stream);

/* The head of a feature-propagation module in one launch (utils/pointnet_util.py:212-219 pointnet_fp_module;
 * utils/pointasnl_util.py:308-313 PointASNLDecodingLayer): the inverse-distance weights of pasnl_three_weights, the
 * interpolation of pasnl_three_interpolate and tf.concat([interpolated, points1], axis=2), same arithmetic, same bits:
 *   out (b,n,c2+c1) = [ sum_j w[b,i,j] points2[b, idx[b,i,j], :]  |  points1[b,i,:] ]
 * dist, idx (b,n,3) = pasnl_three_nn's outputs; points2 (b,m,c2); points1 (b,n,c1) or NULL with c1 == 0 (no concat).
 * Inference only (the differentiable path stays pasnl_three_interpolate + its gradient). */
int pasnl_fp_interpolate_cat(int b, int m, int c2, int n, int c1, const float* points2, const int* idx, const float* dist,
                             const float* points1, float* out, pasnl_stream_t stream);

/* ------------------------------------------------- PointASNL cells (utils/pointasnl_util.py) */

/* Fused Point-NonLocal attention core, mode 'dot' (pointasnl_util.py:197-212):
 *   out[b,i,:] = softmax_j( q[b,i,:] . k[b,j,:] / sqrt(cb) ) . v[b,j,:]
 * q (b,p,cb); kv (b,n,2*cb) with K = kv[...,:cb], V = kv[...,cb:] exactly as conv_kv produces them
 * (:193-194); out (b,p,cb).  The (b,p,n) attention map is never materialised.
 * variant: 0 = auto, 1 = vector-FMA kernel (cb <= 64), 2 = fp32 MFMA kernel, 3 = fp32 MFMA kernel with LDS-staged K/V
 * (the only MFMA form for cb = 128; kept selectable for A/B).  cb in {32, 64, 128} and kv / out 16-byte aligned;
 * anything else: PASNL_EUNSUPPORTED (the Python mirror then takes the op-by-op path on the vendor BLAS). */
int pasnl_nl_attention(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                       pasnl_stream_t stream);
/* pasnl_nl_attention with a scratch workspace: where b * ceil(p / 64) workgroups would leave CUs empty (cb = 32; the
 * SemanticKITTI model's layer 1_1: 8 x 1280 queries = 160 workgroups for 256 CUs) the KEYS are split over workgroups too
 * ("flash-decoding"): every part leaves its un-normalised (O, m, l) in the workspace and a second small kernel combines the
 * parts in ascending key order -- a fixed order: the result is a pure function of the inputs -- and normalises.  Same 1e-5
 * contract; not the bits of the one-workgroup form (another association of the running rescalings).
 * pasnl_nl_attention_workspace_bytes: the bytes that form needs for the shape, 0 where it is not used (workspace may be NULL). */
size_t pasnl_nl_attention_workspace_bytes(int b, int p, int n, int cb);
int pasnl_nl_attention_ws(int b, int p, int n, int cb, const float* q, const float* kv, float* out, int variant,
                          void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* Adaptive-Sampling micro self-attention over the first `as` neighbours of each query
 * (SampleWeights, pointasnl_util.py:136-146):  g groups, each q,k,v (as,cb):
 *   out[g,i,:] = softmax_j( q[g,i,:] . k[g,j,:] / sqrt(cb) ) . v[g,j,:]
 * q (g,as,cb); kv (g,as,2*cb) (K first, V second, :133-134); out (g,as,cb).  as <= 16, cb <= 256. */
int pasnl_as_attention(int g, int as, int cb, const float* q, const float* kv, float* out, pasnl_stream_t stream);

/* AdaptiveSampling tail (pointasnl_util.py:154-155,167-171): softmax over the neighbour axis of
 * logits (g,as,1+ch), then new_xyz[g,:] = sum_k w[g,k,0]*xyz[g,k,:] and
 * new_feature[g,c] = sum_k w[g,k,1+c]*feat[g,k,c].
 * xyz rows are taken from grouped_xyz (g,nsample,3) and feat from grouped_feature (g,nsample,ch):
 * only the first `as` of `nsample` neighbours are read (the :165-166 slices). */
int pasnl_as_reweight(int g, int as, int nsample, int ch, const float* logits, const float* grouped_xyz,
                      const float* grouped_feature, float* new_xyz, float* new_feature, pasnl_stream_t stream);

/* Set-abstraction "local cell", fused (pointasnl_util.py:264-274): per query group of k
 * neighbours, with x = new_point (groups,k,w) as produced by pasnl_sa_group (w = 6+C, columns 0..2 = centred xyz):
 *   H1 = relu(x W0 + b0) (k,c1);  H2 = relu(H1 W1 + b1) (k,c2);  G = relu(x[:,0:3] Ww + bw) (k,32);
 *   out[g] = H2^T G  flattened as (c2*32)   -- the input of the [1,c2] `after_conv` GEMM (:275-278).
 * W0 (w,c1), W1 (c1,c2), Ww (3,32) row-major with inference BN already folded in (tf_util.py).  Replaces two
 * conv2d, the weight-net conv2d, a transpose and a batched matmul of the reference graph; H1, H2 and G never
 * reach HBM.  k % 32 == 0; (c1,c2) in {(32,32),(64,64),(128,128)}; weights must fit 160 KiB of LDS. */
int pasnl_sa_local_cell(int groups, int k, int w, int c1, int c2, const float* x, const float* w0, const float* b0,
                        const float* w1, const float* b1, const float* ww, const float* bw, float* out,
                        pasnl_stream_t stream);

/* The same cell with the grouping fused in (pointasnl_util.py:63-74,248-249,258,264-274): row s of group (b,j) is
 * [xyz[i]-new_xyz[b,j] | xyz[i] | feature[i]], i = idx[b,j,s], gathered straight from the (b,n,3) / (b,n,c) tables
 * (L2-resident), so the (b,m,k,6+c) grouped tensor never exists in HBM.  Also returns the skip connection's
 * reduce_max over the k neighbours: skip_max (b,m,6+c) -- bit-equal to pasnl_sa_group's.
 * out (b*m, c2*32) as pasnl_sa_local_cell.  Replaces two tf.gather_nd, two concats, a subtraction, a reduce_max,
 * three conv2d, a transpose and a batched matmul of the reference graph.  Shape limits as above for c1 = c2 in {32, 64, 128}; also
 *   c1 = c2 = 16 (pointasnl_sem_seg_res.py:32: xyz-only rows c = 3, k = 32, new_xyz given): a 16x16x4-MFMA kernel, no padding;
 *   c1 = c2 in {256, 512} (pointasnl_sem_seg.py:34, pointasnl_sem_seg_res.py:46-51: k = 32, c % 16 == 0, feature 16-byte aligned,
 *   new_xyz given): one workgroup per group, weights streamed from L2; there w1 = b1 = NULL means the layer has a single
 *   convolution (mlp = [c, c]) and H2 = H1.
 * new_xyz == NULL: the centre of group (b,j) is its own neighbour 0, xyz[b, idx[b,j,0]] -- AdaptiveSampling with
 * as_neighbor == 0 (pointasnl_util.py:161-163), taken from the tile the kernel gathers anyway, so that the launch does not
 * wait for pasnl_take_neighbor0 (needs m <= n, else PASNL_EUNSUPPORTED). */
int pasnl_sa_cell(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                  const int* idx, const float* new_xyz, const float* w0, const float* b0, const float* w1,
                  const float* b1, const float* ww, const float* bw, float* out, float* skip_max,
                  pasnl_stream_t stream);

/* pasnl_sa_cell with new_xyz = NULL that ALSO writes what pasnl_take_neighbor0 would have: new_xyz (b,m,3) = the centres and
 * new_feature (b,m,3+c) = [centre | feature row of neighbour 0] (pointasnl_util.py:161-164) -- the wave that owns a group
 * holds that row's address anyway.  A set-abstraction layer without adaptive sampling then needs no gather launch between
 * its neighbour search and its cell.  c <= 128, else PASNL_EUNSUPPORTED. */
int pasnl_sa_cell_centre0(int b, int n, int c, int m, int k, int c1, int c2, const float* xyz, const float* feature,
                          const int* idx, const float* w0, const float* b0, const float* w1, const float* b1,
                          const float* ww, const float* bw, float* out, float* skip_max, float* new_xyz, float* new_feature,
                          pasnl_stream_t stream);

/* pasnl_sa_cell / pasnl_sa_cell_centre0 (new_xyz == NULL: the centres are neighbour 0 and new_xyz_out / new_feature_out are
 * written, else both are ignored) that ALSO get the feature rows of w0 (rows 6 .. 5 + c) and w1 in the matrix instruction's
 * operand order -- pasnl_mlp3_pack_weights(c, c1, w0 + 6 * c1, w0_features_packed) and pasnl_mlp3_pack_weights(c1, c2, w1,
 * w1_packed), once per variable -- for the kernels that stream their weights from L2 (one workgroup per group: c1 = c2 in
 * {256, 512}, and 128 with few groups or one convolution): 16-byte weight loads.  The row-major matrices are still required
 * (the first rows of w0, every other kernel); NULL packed pointers = the plain entry points; identical results. */
int pasnl_sa_cell_packed(int b, int n, int c, int npoint, int nsample, int c1, int c2, const float* xyz, const float* feature,
                         const int* idx, const float* new_xyz, const float* w0, const float* b0, const float* w1, const float* b1,
                         const float* ww, const float* bw, const float* w0_features_packed, const float* w1_packed, float* out,
                         float* skip_max, float* new_xyz_out, float* new_feature_out, pasnl_stream_t stream);

/* PointNet set-abstraction pooling (pointnet_util.py:137, tf.reduce_max(new_points, axis=[2], keep_dims=True)):
 * out[b,ch] = max over the n points of x (b,n,c).  The two group_all modules of pointasnl_cls pool 67 + 34 MB. */
int pasnl_max_pool_rows(int b, int n, int c, const float* x, float* out, pasnl_stream_t stream);

/* The same pooling into rows of a wider table: out[b * out_stride + ch], out_stride >= c -- the two pooled vectors of
 * pointasnl_cls land side by side in the (B, 1536) input of fc1 (models/pointasnl_cls.py:43-45: the tf.concat is free). */
int pasnl_max_pool_rows_strided(int b, int n, int c, const float* x, float* out, long out_stride, pasnl_stream_t stream);

/* The "group all" set-abstraction module in one kernel (csrc/mlp_pool.hip): pointnet_sa_module(..., group_all=True) of
 * utils/pointnet_util.py:87-137 as called at models/pointasnl_cls.py:39-40 -- the three 1x1 convolutions of `mlp` (BN folded
 * into w / bias by the caller, ReLU) over every point of a cloud and tf.reduce_max over the points:
 *   out[cloud * out_stride + ch] = max_i relu(relu(relu(x[cloud,i,:] w0 + b0) w1 + b1) w2 + b2)[ch]
 * x (b,n,k0) rows [xyz | points] (sample_and_group_all's concat, pointnet_util.py:79; any leading alignment columns the
 * caller added have zero rows in w0).  w0 (k0,c1), w1 (c1,c2), w2 (c2,c3) are handed over PACKED in the matrix instruction's
 * operand order -- pasnl_mlp3_pack_weights(k, n, w, packed) once per variable into pasnl_mlp3_packed_weights_bytes(k, n) bytes
 * (16-byte aligned): packed[((bt * 2 + h) * n + col) * 8 + u] = w[16 bt + 2 u + h][col], zero beyond k.  Covered: (c1,c2,c3) =
 * (128,256,512) and (256,512,1024), k0 % 4 == 0, x 16-byte aligned; otherwise PASNL_EUNSUPPORTED (the caller runs the layers
 * one by one).  workspace: pasnl_mlp3_max_pool_workspace_bytes(b, n, c3) bytes (maxima per tile of 32 points; no need to clear it). */
size_t pasnl_mlp3_packed_weights_bytes(int k, int n);
int pasnl_mlp3_pack_weights(int k, int n, const float* w, float* packed, pasnl_stream_t stream);
size_t pasnl_mlp3_max_pool_workspace_bytes(int b, int n, int c3);
int pasnl_mlp3_max_pool(int b, int n, int k0, int c1, int c2, int c3, const float* x, const float* w0, const float* b0,
                        const float* w1, const float* b1, const float* w2, const float* b2, float* out, long out_stride,
                        void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* ------------------------------------------------------------------ fp32-grade products on the bf16 matrix pipe (csrc/dense_bf16x3.hip)
 * An explicit MODE (the Python side's tf_util.DENSE_BF16X3; off by default, never used for the headline benchmark):
 * out (rows,n) = act(x (rows,kdim; row stride lda) . w + bias) for the long GEMMs behind tf_util.conv2d over a flattened
 * [nsample x channel] window (utils/pointasnl_util.py:275, 337; tf_util.py:120-185).  Every fp32 operand is split into three
 * bf16 terms and the six products of weight >= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16: about one more
 * rounding per product than an fp32 fmaf chain (inside the 1e-5 contract, not the same bits).
 *   pasnl_bf16x3_split_weights: w (kdim,n) fp32 -> wsplit (pasnl_bf16x3_weights_bytes(kdim, n) bytes, 16-byte aligned), once
 *   per layer; pasnl_dense_bf16x3: kdim % 32 == 0, n % 128 == 0, lda % 4 == 0, x 16-byte aligned, else PASNL_EUNSUPPORTED. */
size_t pasnl_bf16x3_weights_bytes(int kdim, int n);
int pasnl_bf16x3_split_weights(int kdim, int n, const float* w, void* wsplit, pasnl_stream_t stream);
int pasnl_dense_bf16x3(int rows, int kdim, int n, int lda, const float* x, const void* wsplit, const float* bias, int relu,
                       float* out, pasnl_stream_t stream);

/* ------------------------------------------------------------------ dense layers with few rows (csrc/dense.hip) */

/* out (rows,n) = act(x (rows,kdim) . w (kdim,n) + bias), rows <= 128, relu != 0 -> ReLU: the classifier head
 * (tf_util.fully_connected, tf_util.py:327-365, called at models/pointasnl_cls.py:46-50 with one row per cloud; BN folded
 * into w / bias by the caller).  A product this thin is cut along n AND along kdim over ~128 workgroups; the K slices meet in
 * `workspace` and are summed in slice order (bit-reproducible).  kdim % 8 == 0 and x 16-byte aligned, else PASNL_EUNSUPPORTED.
 * workspace: pasnl_dense_rows_workspace_bytes(rows, kdim, n) bytes of device memory, ZERO-FILLED once by the caller before
 * its first use (the kernel leaves its counters at zero); one workspace serves one stream at a time. */
size_t pasnl_dense_rows_workspace_bytes(int rows, int kdim, int n);
int pasnl_dense_rows(int rows, int kdim, int n, const float* x, const float* w, const float* bias, int relu, float* out,
                     void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* out (rows,n) = act(x (rows,kdim; row stride lda) . w (kdim,n) + bias) for THIN products with a long contraction -- the
 * [1,C] after_conv / decode_after_conv windows of the deep levels (pointasnl_util.py:277-280, :329-331 through tf_util.conv2d,
 * tf_util.py:120-185; BN folded by the caller): few 128 x 128 output tiles, kdim in the thousands.  128 x 128 tiles x K slices
 * on fp32 MFMA; the slices' partial tiles meet in `workspace` (pasnl_dense_splitk_workspace_bytes bytes, no initialisation
 * needed; one workspace serves one stream at a time) and are summed in slice order (bit-reproducible).
 * kdim % 16 == 0, lda % 4 == 0, n % 4 == 0, x / out / bias 16-byte aligned, else PASNL_EUNSUPPORTED. */
size_t pasnl_dense_splitk_workspace_bytes(int rows, int kdim, int n);
int pasnl_dense_splitk(int rows, int kdim, int n, int lda, const float* x, const float* w, const float* bias, int relu,
                       float* out, void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* Two projections of NARROW rows in one launch: out_i (rows_i, n_i) = x_i (rows_i, kdim_i) . w_i + bias_i, kdim_i <= 16,
 * n_i in {32, 64, 128, 256}, no activation: conv_kv and conv_query of the first layer's non-local cell, whose inputs are
 * coordinates (3 or 6 channels) -- pointasnl_util.py:186-193, two tf_util.conv2d with activation_fn=None.  rows_1 == 0
 * runs the first job alone.  w_i / bias_i / out_i 16-byte aligned, else PASNL_EUNSUPPORTED. */
int pasnl_narrow_project2(long rows0, int kdim0, int n0, const float* x0, const float* w0, const float* bias0, float* out0,
                          long rows1, int kdim1, int n1, const float* x1, const float* w1, const float* bias1, float* out1,
                          pasnl_stream_t stream);

/* Deterministic backward of gather_point / group_point / three_interpolate: the gradient row of a source point is
 * the sum of its contributions in ascending order of the forward output element -- the order of the reference's
 * sequential CPU loop (tf_interpolate.cpp:131-153) -- so results are bit-reproducible (the reference's GPU kernels,
 * tf_sampling_g.cu:183-192 and tf_grouping_g.cu:61-78, scatter with atomicAdd and are not).  No fp atomics; every
 * destination row is written (no memset needed).  Out-of-range indices are ignored.
 * workspace: device memory of pasnl_grad_workspace_bytes(b, targets_per_cloud, contributions_per_cloud) bytes, where
 * (targets, contributions) = (n, m) for gather_point, (n, m*nsample) for group_point, (m, 3*n) for three_interpolate.
 * Argument meaning as the atomic versions above.  targets_per_cloud <= 38400 (LDS histogram). */
size_t pasnl_grad_workspace_bytes(int b, int targets, long contributions);
int pasnl_gather_point_grad_det(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* workspace,
                                size_t workspace_bytes, pasnl_stream_t stream);
int pasnl_group_point_grad_det(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                               float* grad_points, void* workspace, size_t workspace_bytes, pasnl_stream_t stream);
int pasnl_three_interpolate_grad_det(int b, int n, int c, int m, const float* grad_out, const int* idx, const float* weight,
                                     float* grad_points, void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* Tail of a set-abstraction layer, fused (pointasnl_util.py:258-261 skip connection, :213-216 back-projection of the
 * non-local cell, :282-290 the two adds and the aggregation layer):
 *   out = relu( ( after + relu(skip_max ws + bs) + relu(att wb + bb) ) wagg + bagg )
 * after (rows,c) = the after_conv output; skip_max (rows,w) = pasnl_sa_cell's skip maxima; att (rows,cb) = pasnl_nl_attention's
 * output (cb == 0 and att/wb/bb NULL for a layer without non-local cell); ws (w,c), wb (cb,c), wagg (c,c) and the biases are
 * the BN-folded `skip`, `conv_back_project`, `aggregation` layers.  c % 32 == 0 and c <= 512, else PASNL_EUNSUPPORTED (the
 * Python mirror then runs the three GEMMs and two adds). */
int pasnl_sa_tail(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att, const float* ws,
                  const float* bs, const float* wb, const float* bb, const float* wagg, const float* bagg, float* out,
                  pasnl_stream_t stream);

/* pasnl_sa_tail with the residual connection of the `_res` model in its epilogue (models/pointasnl_sem_seg_res.py:37,42,47,52:
 * l1_2_points += l1_1_points ...):  out = pasnl_sa_tail(...) + residual, residual (rows,c). */
int pasnl_sa_tail_res(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                      const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                      const float* bagg, const float* residual, float* out, pasnl_stream_t stream);

/* pasnl_sa_tail / _res / _cat with the three weight matrices PACKED in the matrix instruction's operand order (the kernel then
 * reads them in 16-byte pieces: 3-5 us per launch): pasnl_sa_tail_pack_weights(k, c, w, packed) once per variable into
 * pasnl_sa_tail_packed_weights_bytes(k, c) bytes (16-byte aligned): packed[((chunk * 2 + h) * c + col) * 16 + t] =
 * w[32 chunk + 2 t + h][col], zero beyond k.  residual (rows,c) or NULL; out_cat + new_xyz or NULL (see pasnl_sa_tail_cat). */
size_t pasnl_sa_tail_packed_weights_bytes(int k, int c);
int pasnl_sa_tail_pack_weights(int k, int c, const float* w, float* packed, pasnl_stream_t stream);
int pasnl_sa_tail_packed(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                         const float* ws_packed, const float* bs, const float* wb_packed, const float* bb, const float* wagg_packed,
                         const float* bagg, const float* residual, const float* new_xyz, float* out_cat, float* out,
                         pasnl_stream_t stream);

/* pasnl_sa_tail that writes its rows a second time as out_cat (rows, c + 4) = [0 | new_xyz (rows,3) | out]: the
 * tf.concat([xyz, points]) the next group_all module starts with (pointnet_util.py:77-80, sample_and_group_all), one
 * zero column in front so that the rows stay 16-byte aligned (the consumer's weights get a zero row in front). */
int pasnl_sa_tail_cat(int rows, int w, int cb, int c, const float* after, const float* skip_max, const float* att,
                      const float* ws, const float* bs, const float* wb, const float* bb, const float* wagg,
                      const float* bagg, float* out, const float* new_xyz, float* out_cat, pasnl_stream_t stream);

/* Decoder local cell (PointASNLDecodingLayer, pointasnl_util.py:323-331): per point p of the dense level with its k
 * nearest neighbours i_s = idx[b,p,s] (self-kNN on xyz):
 *   F = [xyz[i_s] | feature[i_s]] (k,3+c);  G = relu((xyz[i_s]-xyz[p]) Ww + bw) (k,32);  out[b,p] = F^T G  (3+c,32)
 * = the input of the [1,3+c] `decode_after_conv` GEMM.  feature = the three_interpolate output (b,n,c); Ww (3,32), bw
 * (32) = decode_weight_net/wconv0 with inference BN folded.  Replaces two tf.gather_nd, a concat, a subtraction, a
 * conv2d, a transpose and a batched matmul; no grouped tensor in HBM.  k in {16, 32}. */
int pasnl_decode_cell(int b, int n, int c, int k, const float* xyz, const float* feature, const int* idx, const float* ww,
                      const float* bw, float* out, pasnl_stream_t stream);

/* The same cell with its (3+c)*32 output values per point in a TILED order, for a consumer that contracts all of them (the
 * `decode_after_conv` GEMM with its weight rows permuted to match): 4 full-width 1-KiB stores per 32-channel tile instead
 * of 16 256-byte ones and, for c % 128 == 0, one 16-byte feature load per neighbour and 128 features instead of four 4-byte
 * ones -- the plain kernel is bound by the vector-memory instructions it issues, not by bytes.  Position q of a point holds
 *     q < 96:                                        channel q / 32 (a coordinate), j = q % 32          (reference order)
 *     q = 96 + T*1024 + (2g+h)*128 + 4m + i:         channel 3 + 32 V (T / V) + V m + T % V,   j = 8g + 4h + i
 * with V = pasnl_decode_cell_tiled_v4(c, feature) ? 4 : 1, g < 4, h < 2, m < 32, i < 4.  k == 16 and c % 32 == 0, else
 * PASNL_EUNSUPPORTED. */
int pasnl_decode_cell_tiled(int b, int n, int c, int k, const float* xyz, const float* feature, const int* idx,
                            const float* ww, const float* bw, float* out, pasnl_stream_t stream);
int pasnl_decode_cell_tiled_v4(int c, const float* feature);

/* AdaptiveSampling with as_neighbor == 0 (pointasnl_util.py:161-164): new_xyz (b,m,3) = xyz[idx[b,j,0]] and
 * new_feature (b,m,3+c) = [xyz | feature][idx[b,j,0]], idx (b,m,k) the neighbour indices of the layer. */
int pasnl_take_neighbor0(int b, int n, int c, int m, int k, const float* xyz, const float* feature, const int* idx,
                         float* new_xyz, float* new_feature, pasnl_stream_t stream);

/* AdaptiveSampling without the grouped tensors (pointasnl_util.py:121-171):
 *   pasnl_as_gather: x (b,m,as,6+c) = [xyz[i_s]-xyz[i_0] | xyz[i_s] | feature[i_s]], i_s = idx[b,j,s], s < as -- the
 *     input of conv_kv_ds / conv_query_ds (concat(normalized_xyz, shift_group_points)); idx (b,m,k) with k >= as.
 *   pasnl_as_attention_qkv: as pasnl_as_attention on ONE tensor kvq (g,as,3*cb) = [K | V | Q] per row (the output of
 *     a single GEMM with the conv_kv_ds and conv_query_ds weights side by side).
 *   pasnl_as_reweight_x: as pasnl_as_reweight, reading coordinates and (xyz | feature) rows (ch = 3+c channels) from x. */
int pasnl_as_gather(int b, int n, int c, int m, int k, int as, const float* xyz, const float* feature, const int* idx, float* out,
                    pasnl_stream_t stream);
int pasnl_as_attention_qkv(int g, int as, int cb, const float* kvq, float* out, pasnl_stream_t stream);
/*   pasnl_as_attention_proj: the same attention with the projections fused in, for narrow inputs (w <= 15, cb = 32 or 64):
 *   x (g,as,w) = the rows pasnl_as_gather produces, wkvq (w,3*cb) / bkvq (3*cb) = the BN-folded [conv_kv_ds | conv_query_ds]
 *   weights (pointasnl_util.py:126-135); K, V, Q = x.wkvq + bkvq never reach memory. */
int pasnl_as_attention_proj(int g, int as, int cb, int w, const float* x, const float* wkvq, const float* bkvq, float* out,
                            pasnl_stream_t stream);
/*   pasnl_as_cell_narrow: the whole AdaptiveSampling cell of a narrow layer after the gather (pointasnl_util.py:112-173): the
 *   projections and the attention as above, then mlp2 (cb -> 32 -> 1+ch; wa (cb,32), ba (32), wb (32,1+ch), bb (1+ch), BN
 *   folded), the softmax over the neighbours and the re-weighted sums.  x (g,as,w) with w = 3 + ch <= 15 ->
 *   new_xyz (g,3), new_feature (g,ch). */
int pasnl_as_cell_narrow(int g, int as, int cb, int w, int ch, const float* x, const float* wkvq, const float* bkvq,
                         const float* wa, const float* ba, const float* wb, const float* bb, float* new_xyz, float* new_feature,
                         pasnl_stream_t stream);
/*   pasnl_as_cell_wide: the same cell for wide layers, after their projection GEMM: kvq (g,as,3*cb) = [K | V | Q] rows
 *   (any cb <= 144: the reference's widths are (3 + c) / 2 = 33, 65, ...), x (g,as,w) the gathered rows (for the re-weighted sums) -> new_xyz (g,3), new_feature (g,ch). */
int pasnl_as_cell_wide(int g, int as, int cb, int w, int ch, const float* kvq, const float* x, const float* wa, const float* ba,
                       const float* wb, const float* bb, float* new_xyz, float* new_feature, pasnl_stream_t stream);

/* pasnl_as_cell_wide on projection rows that are WIDER than 3 cb: kvq (g*as, ld), ld >= 3 cb, columns [K | V | Q | unused].
 * The reference's bottleneck widths are (3 + c) / 2 = 33, 65, ...: a GEMM with N = 195 runs at 38 TF where N = 224 runs at
 * 82 (measured, 98304 x 134 inputs), so the Python mirror pads the projection's weights with zero columns. */
int pasnl_as_cell_wide_ld(int g, int as, int cb, int w, int ch, const float* kvq, int ld, const float* x, const float* wa,
                          const float* ba, const float* wb, const float* bb, float* new_xyz, float* new_feature,
                          pasnl_stream_t stream);
int pasnl_as_reweight_x(int g, int as, int ch, const float* logits, const float* x, float* new_xyz, float* new_feature,
                        pasnl_stream_t stream);

/* Voxel-grid subsampling (utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:4-106), the input
 * stage of the ScanNet / SemanticKITTI "grid" pipelines: points (n,3) [+ features (n,fdim)] [+ classes (n,ldim)] ->
 * one row per occupied voxel of edge sample_dl: barycentre, feature mean, majority label.  Voxel keys, fp32 sums in
 * input order and the barycentre / mean arithmetic follow the reference bit for bit; rows come out in ASCENDING VOXEL
 * KEY (the reference emits unordered_map iteration order) and a label tie goes to the smallest label (the reference:
 * first maximum in hash order).  Outputs are sized for n rows; out_count (device int) receives the number of voxels.
 * workspace: pasnl_grid_subsample_workspace_bytes(n) bytes of device memory.  No host synchronisation. */
size_t pasnl_grid_subsample_workspace_bytes(long n);
int pasnl_grid_subsample(long n, int fdim, int ldim, const float* points, const float* features, const int* classes,
                         float sample_dl, float* out_points, float* out_features, int* out_classes, int* out_count,
                         void* workspace, size_t workspace_bytes, pasnl_stream_t stream);

/* The search inside `crop_pc` (SemanticKITTI/semantic_kitti_dataset_grid.py:265-272): around ONE centre per crop, the k nearest
 * points of a scan -- sklearn `KDTree.query(center, k=num_point+buffer)` (:271) -- or every point within a radius --
 * `query_radius(center, r=in_radius)` (:269).  b crops; crop c searches the n points at points + c*scan_stride*3
 * (scan_stride = 0: every crop searches the same scan) around centres[c,:].  Ranking key: the squared distance
 * ((dx*dx)+(dy*dy))+(dz*dz) evaluated in DOUBLE on the float32 coordinates (what sklearn computes on its float64 copy of
 * the data), so the selected set is sklearn's; a tie at the k-th distance goes to the lowest indices (sklearn: tree visit
 * order).  radius > 0: the radius form, d2 <= radius*radius inclusive (k ignored); otherwise k[c] (device ints; NULL: kcap
 * for every crop) nearest, clamped to [0, min(n, kcap)].
 * -> out_idx (b,kcap) i32: the selected indices in ASCENDING INDEX order (crop_pc shuffles them at once, :274; sort by
 * out_d2 for KDTree.query's order), entries behind the count are not written; out_d2 (b,kcap) f64 their squared
 * distances (NULL: not wanted); out_count (b) i32: the number selected (radius form: the TRUE number within the radius,
 * which may exceed kcap -- only the first kcap by index are stored).
 * An exact radix selection on the 63 key bits (no tree), 8 launches, no host synchronisation.
 * workspace: pasnl_knn_crop_workspace_bytes(b, n) bytes (8 n per crop for the keys + histograms). */
size_t pasnl_knn_crop_workspace_bytes(int b, long n);
int pasnl_knn_crop(int b, long n, long scan_stride, const float* points, const float* centres, const int* k, int kcap,
                   double radius, int* out_idx, double* out_d2, int* out_count, void* workspace, size_t workspace_bytes,
                   pasnl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PASNL_H_ */
