// extern "C" doorway to the reference's OWN kNN (compiled from /root/reference/utils/nearest_neighbors/knn_.cxx
// where it lies; see oracle/Makefile).  Test infrastructure only; output goes to oracle/_ref/.
#include <cstddef>
#include <ctime>
#include "knn_.h"  // found through -I/root/reference/utils/nearest_neighbors

// cpp_knn_batch_distance_pick seeds its std::mt19937 with time(0) (knn_.cxx:141).  Inside THIS library `time` is the
// function below (hidden visibility: bound at link time, invisible outside), so that a test can fix the seed and compare
// the reference's picks with the port's.  Nothing else in the reference's kNN code calls time().
static long ref_knn_fixed_time = 0;
extern "C" __attribute__((visibility("hidden"))) time_t time(time_t* t) {
  if (t) *t = (time_t)ref_knn_fixed_time;
  return (time_t)ref_knn_fixed_time;
}

extern "C" {
void ref_knn_batch(const float* pts, size_t b, size_t n, size_t dim, const float* queries, size_t m, size_t k, long* out) {
  cpp_knn_batch(pts, b, n, dim, queries, m, k, out);  // knn_.cxx:72-101
}
void ref_knn_batch_omp(const float* pts, size_t b, size_t n, size_t dim, const float* queries, size_t m, size_t k,
                       long* out) {
  cpp_knn_batch_omp(pts, b, n, dim, queries, m, k, out);  // knn_.cxx:104-135
}
void ref_knn(const float* pts, size_t n, size_t dim, const float* queries, size_t m, size_t k, long* out) {
  cpp_knn(pts, n, dim, queries, m, k, out);  // knn_.cxx:22-43
}
void ref_knn_batch_distance_pick(long seed, const float* pts, size_t b, size_t n, size_t dim, float* queries, size_t nq, size_t k,
                                 long* out) {
  ref_knn_fixed_time = seed;
  cpp_knn_batch_distance_pick(pts, b, n, dim, queries, nq, k, out);  // knn_.cxx:136-200
}
}
