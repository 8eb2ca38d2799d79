// extern "C" doorway to the reference's CUDA launchers, whose two .cu files are compiled UNCHANGED by hipcc
// (-x hip -include hip/hip_runtime.h) from /root/reference/tf_ops/{sampling,grouping} where they lie.
// GPU-box-only secondary oracle; never linked into the product.
#include <hip/hip_runtime.h>

// declarations as in tf_sampling.cpp:65,94,125,150 and tf_grouping.cpp:66,108,142,173 (C++ linkage)
void probsampleLauncher(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out);
void farthestpointsamplingLauncher(int b, int n, int m, const float* inp, float* temp, int* out);
void gatherpointLauncher(int b, int n, int m, const float* inp, const int* idx, float* out);
void scatteraddpointLauncher(int b, int n, int m, const float* out_g, const int* idx, float* inp_g);
void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2, int* idx,
                            int* pts_cnt);
void selectionSortLauncher(int b, int n, int m, int k, const float* dist, int* outi, float* out);
void groupPointLauncher(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out);
void groupPointGradLauncher(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                            float* grad_points);

extern "C" {
int ref_sync() { return (int)hipDeviceSynchronize(); }
void ref_prob_sample(int b, int n, int m, const float* p, const float* r, float* temp, int* out) {
  probsampleLauncher(b, n, m, p, r, temp, out);
}
void ref_fps(int b, int n, int m, const float* inp, float* temp, int* out) { farthestpointsamplingLauncher(b, n, m, inp, temp, out); }
void ref_gather_point(int b, int n, int m, const float* inp, const int* idx, float* out) { gatherpointLauncher(b, n, m, inp, idx, out); }
void ref_gather_point_grad(int b, int n, int m, const float* og, const int* idx, float* ig) {
  scatteraddpointLauncher(b, n, m, og, idx, ig);
}
void ref_query_ball_point(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2, int* idx,
                          int* cnt) {
  queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, cnt);
}
void ref_select_top_k(int b, int n, int m, int k, const float* dist, int* outi, float* out) {
  selectionSortLauncher(b, n, m, k, dist, outi, out);
}
void ref_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out) {
  groupPointLauncher(b, n, c, m, nsample, points, idx, out);
}
void ref_group_point_grad(int b, int n, int c, int m, int nsample, const float* go, const int* idx, float* gp) {
  groupPointGradLauncher(b, n, c, m, nsample, go, idx, gp);
}
}
