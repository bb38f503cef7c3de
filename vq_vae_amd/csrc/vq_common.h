// Shared definitions of the gfx950 vector-quantizer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vq_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
// loss partials of rows that several workgroups of the exact kernel share (one slot per row)
constexpr int VQ_NEAR_SLOTS = VQ_FIX_WGS / 2;
// filtered search: with at most this many rows left open by the first level, the second level is skipped and
// the per-row kernel takes them directly
constexpr int VQ_FILTER_FEW_ROWS = 32;

__host__ __device__ inline int vq_kpad(int K) { return (K + VQ_STAGE_CODES - 1) / VQ_STAGE_CODES * VQ_STAGE_CODES; }
// One stage of the codebook image: 64 codes x D floats in MFMA-fragment order, then the
// 64 squared norms of the stage padded to one 1 KiB LDS-DMA piece.
__host__ __device__ constexpr size_t vq_stage_floats(int D) { return (size_t)VQ_STAGE_CODES * D + 256; }

__host__ __device__ inline bool vq_fast_dim(int D) { return D == 16 || D == 32 || D == 64 || D == 128 || D == 256; }
// Channel count the matrix-core kernels run a D-channel problem at: D itself when it is one of the five
// above, otherwise the next one up with the extra channels zero (0 beyond 256: no matrix-core path).
// Zero channels change nothing: fma(0, 0, acc) = acc bit for bit (a chain that starts at +0 never holds -0).
__host__ __device__ inline int vq_pad_dim(int D) {
  return D <= 0 ? 0 : D <= 16 ? 16 : D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : D <= 256 ? 256 : 0;
}

// Control words shared by the kernels of one step (device memory, 16 ints).
enum { VQ_CTL_NF_COUNT = 0, VQ_CTL_CB_FLAG = 1, VQ_CTL_BAD_INDEX = 2, VQ_CTL_ZMAX = 3, VQ_CTL_OVF_COUNT = 4, VQ_CTL_OVF2_COUNT = 5, VQ_CTL_NEAR_COUNT = 6, VQ_CTL_WORDS = 16 };

static inline int vq_hip_status(hipError_t e) { return e == hipSuccess ? VQ_OK : VQ_ERR_HIP_BASE + (int)e; }

#define VQ_CHECK_HIP(expr)                         \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return vq_hip_status(_e); \
  } while (0)

// ---- kernels whose dynamic LDS exceeds the 64 KiB default ----------------------------------
// Such a kernel needs hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device before its
// first launch.  Every launcher goes through vq_lds_ready<kernel, bytes>(): the kernel is entered
// in a process-wide table when the library is loaded (static registration), the attribute is set
// on the first launch on a device -- or up front for all kernels by vq_init(device), which keeps
// the first call of every entry point free of this host-side work.
void vq_lds_register(const void* kern, int bytes);
int vq_lds_ensure(const void* kern, int bytes, int* dev_out);
struct VqLdsRegistrar {
  VqLdsRegistrar(const void* k, int b) { vq_lds_register(k, b); }
};
template <auto Kern, int Bytes>
struct VqLdsReg {
  static inline VqLdsRegistrar r{reinterpret_cast<const void*>(Kern), Bytes};
};
// Returns a status; *dev_out (may be null) receives the current device ordinal.
template <auto Kern, int Bytes>
inline int vq_lds_ready(int* dev_out = nullptr) {
  (void)&VqLdsReg<Kern, Bytes>::r;  // odr-use: instantiates the registration
  return vq_lds_ensure(reinterpret_cast<const void*>(Kern), Bytes, dev_out);
}

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

// max |z| as the bit pattern of the absolute value (monotone for non-NaN; NaN patterns are
// above infinity's, so a NaN anywhere makes the maximum read "not finite").  Result in
// control word VQ_CTL_ZMAX.
static __global__ __launch_bounds__(256) void vq_absmax_kernel(const float* __restrict__ z, int64_t numel,
                                                               int32_t* control) {
  uint32_t m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // 16-byte loads, four in flight per lane, over the 16-byte-aligned body; the few elements before
  // and after it singly
  const int64_t head = (((uintptr_t)z & 15) == 0) ? 0 : (int64_t)((16 - ((uintptr_t)z & 15)) / 4);
  const int64_t h = head < numel ? head : numel;
  const int64_t nvec = (numel - h) / 4;
  const f32x4* zv = reinterpret_cast<const f32x4*>(z + h);
  int64_t i = tid;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = zv[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t b = __float_as_uint(v[u][q]) & 0x7fffffffu;
        m = b > m ? b : m;
      }
  }
  for (; i < nvec; i += stride) {
    const f32x4 v = zv[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t b = __float_as_uint(v[q]) & 0x7fffffffu;
      m = b > m ? b : m;
    }
  }
  for (int64_t e = tid; e < h + (numel - h - 4 * nvec); e += stride) {
    const int64_t at = e < h ? e : h + 4 * nvec + (e - h);
    const uint32_t b = __float_as_uint(z[at]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned int*>(&control[VQ_CTL_ZMAX]), m);
}

static inline int vq_launch_absmax(const float* z, int64_t numel, int32_t* control, hipStream_t s) {
  VQ_CHECK_HIP(hipMemsetAsync(&control[VQ_CTL_ZMAX], 0, 4, s));
  int64_t blocks = (numel + 256 * 32 - 1) / (256 * 32);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(vq_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, numel, control);
  return vq_hip_status(hipGetLastError());
}

// Internal (not in the public header): the pairwise variant of the MFMA search, z := E.
extern "C" int vq_pairwise_search_launch(const float* E, const float* image, const float* e2, int K, int D,
                                         int64_t row_base, int64_t n_rows, float* nn_dist, int64_t* nn_idx,
                                         int32_t* idx32, double* sum_partial, float* mm_partial, int32_t* nf_rows,
                                         int32_t* control, void* stream);

// Internal (not in the public header): knobs of the persistent short-sweep search (vq_search.hip).
// LDS-DMA of 1 KiB pieces (16 bytes per lane) through a buffer descriptor: piece I of a run lies at
// "descriptor + soff + I KiB + lane x 16" in memory and at "l + I KiB + lane x 16" in LDS; four consecutive pieces
// share the scalar offset and M0 and differ in the instruction's immediate only, so that a DMA instruction needs
// no vector instruction beside it.  `count` (wave-uniform) cuts the run short.
template <int IMM>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* l, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, IMM, 0);
}
template <int I, int N>
__device__ __forceinline__ void dma_run(__amdgpu_buffer_rsrc_t r, float* l, uint32_t voff, uint32_t soff, int count) {
  if constexpr (I < N) {
    if (I < count) dma16<(I % 4) * 1024>(r, l + (I / 4) * 1024, voff, soff + (I / 4) * 4096);
    dma_run<I + 1, N>(r, l, voff, soff, count);
  }
}
extern "C" int vq_internal_tune(int what, int value);

// Internal (not in the public header): bf16 filter + exact verification (vq_filter.hip).
bool vq_filter_supported(int K, int D);
extern "C" int vq_filter_verify_launch(const float* z, int nchw, int64_t N, int D, int64_t HW, const float* E,
                                       const float* e2, const void* fimage, int K, float T, int64_t code_base,
                                       int64_t* idx64, int32_t* idx32, float* dmin, int32_t* nf_rows,
                                       int32_t** open_rows_out, int32_t** hard_rows_out, int32_t* control, void* work,
                                       void* stream);
