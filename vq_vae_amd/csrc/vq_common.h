// Shared definitions of the gfx950 vector-quantizer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vq_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The search kernel sweeps the codebook in stages of 64 codes (two 32-row MFMA tiles).
constexpr int VQ_STAGE_CODES = 64;
// Rows (latent vectors) per workgroup of the search kernel: 4 waves x 2 column tiles x 32.
constexpr int VQ_WG_ROWS = 256;
// A row whose squared norm exceeds this (or is NaN) leaves the MFMA path and is searched by
// the exact kernel, which implements torch.argmin's NaN rule.  Below it every distance of
// the row is finite, so a plain '<' scan and the NaN-aware scan agree.
constexpr float VQ_NORM_LIMIT = 1.0e30f;
// Workgroups of the exact (fix-up) kernel when it walks the list of flagged rows.
constexpr int VQ_FIX_WGS = 1024;

__host__ __device__ inline int vq_kpad(int K) { return (K + VQ_STAGE_CODES - 1) / VQ_STAGE_CODES * VQ_STAGE_CODES; }
// One stage of the codebook image: 64 codes x D floats in MFMA-fragment order, then the
// 64 squared norms of the stage padded to one 1 KiB LDS-DMA piece.
__host__ __device__ inline size_t vq_stage_floats(int D) { return (size_t)VQ_STAGE_CODES * D + 256; }

__host__ __device__ inline bool vq_fast_dim(int D) { return D == 16 || D == 32 || D == 64 || D == 128 || D == 256; }

// Control words shared by the kernels of one step (device memory, 16 ints).
enum { VQ_CTL_NF_COUNT = 0, VQ_CTL_CB_FLAG = 1, VQ_CTL_BAD_INDEX = 2, VQ_CTL_WORDS = 16 };

static inline int vq_hip_status(hipError_t e) { return e == hipSuccess ? VQ_OK : VQ_ERR_HIP_BASE + (int)e; }

#define VQ_CHECK_HIP(expr)                         \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return vq_hip_status(_e); \
  } while (0)

// torch.argmin's replacement rule: NaN is below everything and the first NaN stays.
__device__ __forceinline__ bool vq_argmin_takes(float d, float best) {
  return (d < best) || (d != d && best == best);
}

// Lexicographic merge of two (distance, index) candidates under that rule.
__device__ __forceinline__ bool vq_merge_takes(float d, int i, float best, int bi) {
  if (vq_argmin_takes(d, best)) return true;
  if (vq_argmin_takes(best, d)) return false;
  // equal distances, or both NaN
  return i < bi;
}

// Internal (not in the public header): the pairwise variant of the MFMA search, z := E.
extern "C" int vq_pairwise_search_launch(const float* E, const float* image, const float* e2, int K, int D,
                                         float* nn_dist, int64_t* nn_idx, int32_t* idx32, double* sum_partial,
                                         float* mm_partial, int32_t* nf_rows, int32_t* control, void* stream);
