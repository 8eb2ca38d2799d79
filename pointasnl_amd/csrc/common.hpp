// Shared device/host helpers for libpasnl_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pasnl.h"

#define PASNL_WAVE 64

#define PASNL_REQUIRE(cond, code) \
  do {                            \
    if (!(cond)) return (code);   \
  } while (0)

// Launch epilogue: the reference launchers never check errors; this ABI does.
static inline int pasnl_launch_status() { return hipGetLastError() == hipSuccess ? PASNL_OK : PASNL_ELAUNCH; }

static inline hipStream_t pasnl_hip_stream(pasnl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

namespace pasnl {

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// DPP controls (GFX9 encoding).
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHR1 = 0x138;

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t keep, uint32_t src) {
  // lanes whose DPP source is invalid or masked keep `keep`
  return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}

// Wave-wide maximum of an unsigned 64-bit key; result is wave-uniform (taken from lane 63).
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#define PASNL_STEP(CTRL, RM)                                            \
  {                                                                     \
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);                \
    uint32_t olo = dpp_u32<CTRL, RM>(lo, lo);                           \
    uint32_t ohi = dpp_u32<CTRL, RM>(hi, hi);                           \
    uint64_t o = ((uint64_t)ohi << 32) | olo;                           \
    v = o > v ? o : v;                                                  \
  }
  PASNL_STEP(DPP_ROW_SHR1, 0xf)
  PASNL_STEP(DPP_ROW_SHR2, 0xf)
  PASNL_STEP(DPP_ROW_SHR4, 0xf)
  PASNL_STEP(DPP_ROW_SHR8, 0xf)
  PASNL_STEP(DPP_ROW_BCAST15, 0xa)
  PASNL_STEP(DPP_ROW_BCAST31, 0xc)
#undef PASNL_STEP
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// value of lane-1 (lane 0 keeps its own); the in-wave "shift the sorted list up by one" primitive.
__device__ __forceinline__ float wave_shr1_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_WAVE_SHR1, 0xf, 0xf, false));
}
__device__ __forceinline__ int wave_shr1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }

// Wave-wide min / max of a float and inclusive prefix sum of an int by DPP (rows of 16: shr 1, 2, 4, 8, then the two row
// broadcasts): register-file moves of a few cycles each where __shfl_xor / __shfl_up are ds_bpermute round trips.
// wave_min_f32 / wave_max_f32: the result is wave-uniform (taken from lane 63).
#define PASNL_DPP_F32(OP, CTRL, RM, IDENT)                                                                          \
  v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(IDENT), __float_as_int(v), CTRL, RM, 0xf, false)));
__device__ __forceinline__ float wave_min_f32(float v) {
  const float inf = __builtin_inff();
  PASNL_DPP_F32(fminf, DPP_ROW_SHR1, 0xf, inf) PASNL_DPP_F32(fminf, DPP_ROW_SHR2, 0xf, inf) PASNL_DPP_F32(fminf, DPP_ROW_SHR4, 0xf, inf)
  PASNL_DPP_F32(fminf, DPP_ROW_SHR8, 0xf, inf) PASNL_DPP_F32(fminf, DPP_ROW_BCAST15, 0xa, inf) PASNL_DPP_F32(fminf, DPP_ROW_BCAST31, 0xc, inf)
  return readlane_f(v, 63);
}
__device__ __forceinline__ float wave_max_f32(float v) {
  const float ninf = -__builtin_inff();
  PASNL_DPP_F32(fmaxf, DPP_ROW_SHR1, 0xf, ninf) PASNL_DPP_F32(fmaxf, DPP_ROW_SHR2, 0xf, ninf) PASNL_DPP_F32(fmaxf, DPP_ROW_SHR4, 0xf, ninf)
  PASNL_DPP_F32(fmaxf, DPP_ROW_SHR8, 0xf, ninf) PASNL_DPP_F32(fmaxf, DPP_ROW_BCAST15, 0xa, ninf) PASNL_DPP_F32(fmaxf, DPP_ROW_BCAST31, 0xc, ninf)
  return readlane_f(v, 63);
}
#undef PASNL_DPP_F32
__device__ __forceinline__ int wave_inclusive_sum_i32(int v) {
#define PASNL_DPP_ADD(CTRL, RM) v += __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xf, false);
  PASNL_DPP_ADD(DPP_ROW_SHR1, 0xf) PASNL_DPP_ADD(DPP_ROW_SHR2, 0xf) PASNL_DPP_ADD(DPP_ROW_SHR4, 0xf) PASNL_DPP_ADD(DPP_ROW_SHR8, 0xf)
  PASNL_DPP_ADD(DPP_ROW_BCAST15, 0xa) PASNL_DPP_ADD(DPP_ROW_BCAST31, 0xc)
#undef PASNL_DPP_ADD
  return v;
}

// Canonical squared distance: ((dx*dx)+(dy*dy))+(dz*dz), no contraction.
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

// The same canonical distance from one point to QW queries at once, written on 2-vectors so that the compiler emits
// v_pk_add/mul_f32 (two queries per instruction; each component is the identical IEEE operation sequence, and
// (p-q)^2 == (q-p)^2 bit for bit).
typedef float pasnl_f32x2 __attribute__((ext_vector_type(2)));
template <int QW>
__device__ __forceinline__ void dist2_multi(const float (&qx)[QW], const float (&qy)[QW], const float (&qz)[QW], float x, float y,
                                            float z, float (&d)[QW]) {
  static_assert(QW % 2 == 0 || QW == 1, "queries are processed in pairs");
  if constexpr (QW == 1) {
    d[0] = dist2(qx[0], qy[0], qz[0], x, y, z);
  } else {
#pragma unroll
    for (int i = 0; i < QW; i += 2) {
      const pasnl_f32x2 dx = pasnl_f32x2{qx[i], qx[i + 1]} - x, dy = pasnl_f32x2{qy[i], qy[i + 1]} - y,
                        dz = pasnl_f32x2{qz[i], qz[i + 1]} - z;
      const pasnl_f32x2 r = (dx * dx + dy * dy) + dz * dz;
      d[i] = r[0];
      d[i + 1] = r[1];
    }
  }
}

// ---- wave-wide bitonic sort (used by the kNN kernels)
template <typename T>
__device__ __forceinline__ T shfl_xor_any(T v, int j);
// partner lane ^ j.  Inside a row of 16 lanes the exchange is one or two DPP moves (quad permutes for j = 1, 2; for j = 4, 8 the
// banks that read "from the right" and the banks that read "from the left" are two row shifts under complementary bank
// masks) -- a register-file operation of a few cycles where __shfl_xor is a ds_bpermute round trip through the LDS crossbar
// (~100 cycles with its wait, and a bitonic sort of 64 keys has 18 such steps of 21).  j is a constant after unrolling.
__device__ __forceinline__ uint32_t shfl_xor_u32(uint32_t v, int j) {
  const int x = (int)v;
  switch (j) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);      // banks 0, 2 <- lane + 4 (row_shl:4)
      return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);   // banks 1, 3 <- lane - 4 (row_shr:4)
    }
    case 8: {
      const int t = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xf, 0x3, false);      // banks 0, 1 <- lane + 8 (row_shl:8)
      return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x118, 0xf, 0xc, false);   // banks 2, 3 <- lane - 8 (row_shr:8)
    }
    default: return (uint32_t)__shfl_xor(x, j);
  }
}
template <>
__device__ __forceinline__ uint32_t shfl_xor_any<uint32_t>(uint32_t v, int j) { return shfl_xor_u32(v, j); }
template <>
__device__ __forceinline__ unsigned long long shfl_xor_any<unsigned long long>(unsigned long long v, int j) {
  uint32_t lo = shfl_xor_u32((uint32_t)v, j), hi = shfl_xor_u32((uint32_t)(v >> 32), j);
  return ((unsigned long long)hi << 32) | lo;
}

// ascending bitonic sort of 64*NREG keys; element e lives in register e/64 of lane e%64
template <int NREG, typename KeyT>
__device__ __forceinline__ void wave_bitonic_sort(KeyT (&v)[NREG], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * NREG; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j >= 64) {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int pr = r ^ (j >> 6);
          if (pr > r) {
            const bool up = ((r * 64) & k) == 0;
            KeyT a = v[r], b = v[pr];
            KeyT lo = a < b ? a : b, hi = a < b ? b : a;
            v[r] = up ? lo : hi;
            v[pr] = up ? hi : lo;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          KeyT mine = v[r];
          KeyT other = shfl_xor_any<KeyT>(mine, j);
          const bool up = (((r * 64 + lane) & k) == 0);
          const bool lower = (lane & j) == 0;
          const bool take_min = up == lower;
          KeyT lo = mine < other ? mine : other, hi = mine < other ? other : mine;
          v[r] = take_min ? lo : hi;
        }
      }
    }
  }
}

// ---- kNN in the reference's order (pasnl_knn_batch_ref): which queries need the KD-tree search at all.
// nanoflann's result can differ from the canonical (distance, index) list only where distances are EQUAL: inside the K-list
// (its order among equals is the tree's visit order) or at its end (which of several candidates at the K-th distance is kept).
// A query with neither has ONE possible answer, and the canonical kernels have already written it.
struct KnnTieFlags {
  int* nflag;   // [b] flagged queries per cloud (zeroed before the launch); nullptr: no flagging
  int* flist;   // [b][m] their numbers, in arrival order (the order does not matter: rows are independent)
};
// sorted ascending keys (distance bits << 32 | index): rank l in lane l of key0, rank 64 + l in lane l of key1 (absent: ~0).
// True (wave-uniform) if two of the first k distances are equal or the (k+1)-th candidate ties with the k-th.  1 <= k <= 64.
__device__ __forceinline__ bool knn_sorted_has_tie(unsigned long long key0, unsigned long long key1, int k, int lane) {
  const uint32_t d = (uint32_t)(key0 >> 32);
  const uint32_t prev = (uint32_t)wave_shr1_i((int)d);  // lane l: rank l - 1 (lane 0: its own, excluded below)
  const uint32_t dnext = k < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)d, k & 63)
                                : (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key1 >> 32), 0);
  const uint32_t dlast = (uint32_t)__builtin_amdgcn_readlane((int)d, (k - 1) & 63);
  return __builtin_amdgcn_ballot_w64(lane >= 1 && lane < k && d == prev) != 0ull || dnext == dlast;
}
__device__ __forceinline__ void knn_flag_query(const KnnTieFlags f, int bi, int m, int j, int lane) {
  if (lane == 0) f.flist[(size_t)bi * m + atomicAdd(&f.nflag[bi], 1)] = j;
}

// Tuning / A-B switches read from the environment exist ONLY in the diagnostic build (make tuning ->
// libpasnl_hip_tuning.so, never loaded by the package): the product library reads no environment variable and
// keeps no global state (include/pasnl.h), so a launch is a pure function of its arguments.
#ifdef PASNL_TUNING
inline const char* tune_env(const char* name) { return getenv(name); }
#else
constexpr const char* tune_env(const char*) { return nullptr; }
#endif


// internal launchers behind pasnl_knn_batch / _ws / _tree / _ref (host side; defined in grouping.hip, knn_grid.hip, knn_tree.hip)
int knn_brute_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64, float* dist2,
                     KnnTieFlags flags, hipStream_t st);
size_t knn_grid_ws_bytes(int b, int n);
int knn_grid_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64, float* dist2,
                    void* workspace, size_t workspace_bytes, int max_workgroups, KnnTieFlags flags, hipStream_t st);
size_t knn_tree_ws_bytes(int b, int n, int m, int k);
int knn_tree_launch(int b, int n, int m, int k, const float* support, const float* queries, void* idx, int idx_is_i64, void* workspace,
                    size_t workspace_bytes, KnnTieFlags only, int* depth_flag, hipStream_t st);
}  // namespace pasnl
