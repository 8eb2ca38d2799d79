// extern "C" doorway to the reference's OWN grid_subsampling (compiled from
// /root/reference/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp and cpp_utils/cloud/cloud.cpp
// where they lie; see oracle/Makefile).  Test infrastructure only; output goes to oracle/_ref/.
#include <cstring>
#include "grid_subsampling/grid_subsampling.h"  // found through -I/root/reference/utils/cpp_wrappers/cpp_subsampling

extern "C" int ref_grid_subsample(long n, int fdim, int ldim, const float* pts, const float* feats, const int* cls, float dl,
                                  float* out_pts, float* out_feats, int* out_cls) {
  std::vector<PointXYZ> original(n), sub;
  for (long i = 0; i < n; ++i) original[i] = PointXYZ(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]);
  std::vector<float> of(feats, feats + (size_t)n * fdim), sf;
  std::vector<int> oc(cls, cls + (size_t)n * ldim), sc;
  grid_subsampling(original, sub, of, sf, oc, sc, dl, 0);  // grid_subsampling.cpp:4-106
  for (size_t v = 0; v < sub.size(); ++v) { out_pts[v * 3] = sub[v].x; out_pts[v * 3 + 1] = sub[v].y; out_pts[v * 3 + 2] = sub[v].z; }
  if (fdim) std::memcpy(out_feats, sf.data(), sf.size() * sizeof(float));
  if (ldim) std::memcpy(out_cls, sc.data(), sc.size() * sizeof(int));
  return (int)sub.size();  // rows are in unordered_map iteration order
}
