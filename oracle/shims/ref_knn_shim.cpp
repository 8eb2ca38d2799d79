// extern "C" doorway to the reference's OWN kNN (compiled from /root/reference/utils/nearest_neighbors/knn_.cxx
// where it lies; see oracle/Makefile).  Test infrastructure only; output goes to oracle/_ref/.
#include <cstddef>
#include "knn_.h"  // found through -I/root/reference/utils/nearest_neighbors

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
}
