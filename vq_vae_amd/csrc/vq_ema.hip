// EMA codebook update for gfx950: code histogram, embedding sums, cluster-size update with
// the persisted Laplace smoothing, codebook renormalisation, forward scalars, backward.
//
// Replaces vqvae/ema_vector_quantizer.py:61-92, :105-106 of the reference.  The reference
// builds the statistics from dense N x K one-hot products; here they come from the index
// vector: a bincount and a segmented sum.  All of it is HBM-bound byte moving, so the
// kernels are organised around coalesced streams and LDS-private accumulation, not MFMA.
#include "vq_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// Histogram.  LDS-private bins per workgroup (K <= 16384), one global integer atomic per
// non-empty bin at the end; integer sums are exact, so the result does not depend on order.
// ---------------------------------------------------------------------------------------
constexpr int HIST_LDS_BINS = 16384;

template <typename IndexT>
__global__ __launch_bounds__(256) void vq_hist_lds_kernel(const IndexT* __restrict__ idx, int64_t M, int K,
                                                          uint32_t* __restrict__ counts, int32_t* control) {
  extern __shared__ uint32_t bins[];
  for (int k = threadIdx.x; k < K; k += 256) bins[k] = 0;
  __syncthreads();
  int bad = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += stride) {
    const int64_t k = (int64_t)idx[i];
    if (k >= 0 && k < K) atomicAdd(&bins[k], 1u);
    else ++bad;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    const uint32_t c = bins[k];
    if (c) atomicAdd(&counts[k], c);
  }
  if (bad && control) atomicAdd(&control[VQ_CTL_BAD_INDEX], bad);
}

template <typename IndexT>
__global__ __launch_bounds__(256) void vq_hist_global_kernel(const IndexT* __restrict__ idx, int64_t M, int K,
                                                             uint32_t* __restrict__ counts, int32_t* control) {
  int bad = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += stride) {
    const int64_t k = (int64_t)idx[i];
    if (k >= 0 && k < K) atomicAdd(&counts[k], 1u);
    else ++bad;
  }
  if (bad && control) atomicAdd(&control[VQ_CTL_BAD_INDEX], bad);
}

// ---------------------------------------------------------------------------------------
// Embedding sums (the scatter-add).  A workgroup owns 4 channels x one slab of rows x a
// range of up to 8192 codes; wave w accumulates channel c0+w into its own K-float LDS
// array with ds_add_f32.  With the NCHW layout a channel of 64 consecutive rows is 256
// contiguous bytes, so z streams in whole lines and is read exactly once.  No two waves
// ever touch the same LDS word and each wave walks its rows in a fixed order, so the sums
// are reproducible.  The slab is flushed as [k][4 channels] 16-byte pieces.
// ---------------------------------------------------------------------------------------
constexpr int SCAT_KR = 8192;     // codes per pass (4 x 8192 floats = 128 KiB of LDS)
constexpr int SCAT_MIN_ROWS = 4096;

struct ScatterPlan { int nslab; int64_t rows_per_slab; };

ScatterPlan scatter_plan(int64_t N, int K, int D) {
  const int slices = (D + 3) / 4;
  const int kr = (K + SCAT_KR - 1) / SCAT_KR;
  int64_t want = (512 + (int64_t)slices * kr - 1) / ((int64_t)slices * kr);  // >= 512 workgroups in all
  const int64_t maxslab = (N + SCAT_MIN_ROWS - 1) / SCAT_MIN_ROWS;
  if (want > maxslab) want = maxslab;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  // rows per slab: multiple of 256 so wave-sized runs stay aligned
  int64_t rows = (N + want - 1) / want;
  rows = (rows + 255) / 256 * 256;
  const int nslab = (int)((N + rows - 1) / rows);
  return {nslab, rows};
}

__global__ __launch_bounds__(256) void vq_scatter_kernel(const float* __restrict__ z, int nchw, int64_t N, int D,
                                                         int64_t HW, const int32_t* __restrict__ idx32, int K,
                                                         int64_t rows_per_slab, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [4][kr_len]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 4;
  const int slab = blockIdx.y;
  const int k0 = blockIdx.z * SCAT_KR;
  const int kr_len = (K - k0) < SCAT_KR ? (K - k0) : SCAT_KR;
  for (int i = threadIdx.x; i < 4 * kr_len; i += 256) acc[i] = 0.0f;
  __syncthreads();

  const int c = c0 + wave;
  const int64_t n_begin = (int64_t)slab * rows_per_slab;
  const int64_t n_end = (n_begin + rows_per_slab) < N ? (n_begin + rows_per_slab) : N;
  float* mine = acc + (size_t)wave * kr_len;
  if (c < D) {
    constexpr int U = 8;  // independent loads in flight per lane
    for (int64_t n0 = n_begin + lane; n0 < n_end; n0 += 64 * U) {
      float v[U];
      int k[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t n = n0 + 64 * u;
        const bool in = n < n_end;
        const int64_t nn = in ? n : n_begin;
        int64_t off;
        if (nchw) { const int64_t b = nn / HW, hw = nn - b * HW; off = (b * D + c) * HW + hw; }
        else off = nn * D + c;
        v[u] = z[off];
        k[u] = in ? idx32[nn] - k0 : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k[u] >= 0 && k[u] < kr_len) atomicAdd(&mine[k[u]], v[u]);
    }
  }
  __syncthreads();
  // flush: slab[k][c0..c0+3]
  float* out = slabs + (size_t)slab * K * D;
  for (int kk = threadIdx.x; kk < kr_len; kk += 256) {
    float* dst = out + (size_t)(k0 + kk) * D + c0;
    if (c0 + 3 < D && (D & 3) == 0) {
      *reinterpret_cast<f32x4*>(dst) = f32x4{acc[kk], acc[kr_len + kk], acc[2 * kr_len + kk], acc[3 * kr_len + kk]};
    } else {
      for (int w = 0; w < 4; ++w)
        if (c0 + w < D) dst[w] = acc[(size_t)w * kr_len + kk];
    }
  }
}

__global__ __launch_bounds__(256) void vq_reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, int64_t KD,
                                                              float* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= KD) return;
  float s = slabs[i];
  for (int t = 1; t < nslab; ++t) s += slabs[(size_t)t * KD + i];
  dw[i] = s;
}

// ---------------------------------------------------------------------------------------
// One-workgroup kernels over K-sized vectors (fixed reduction order => reproducible).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (t < off) sh[t] += sh[t + off];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(1024) void vq_step_scalars_kernel(const float* __restrict__ loss_partial, int64_t n_partial,
                                                               int64_t N_total, int D, float beta,
                                                               const uint32_t* __restrict__ counts, int K,
                                                               float* loss_out, float* perplexity_out, double* sq_sum_out) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  if (loss_partial && (loss_out || sq_sum_out)) {
    double s = 0.0;
    for (int64_t i = t; i < n_partial; i += 1024) s += (double)loss_partial[i];
    const double tot = block_sum_1024(s, sh);
    if (t == 0) {
      if (sq_sum_out) *sq_sum_out = tot;
      // beta * mean((q - z)^2)  (ref :92): mean in fp32, then the fp32 product
      if (loss_out) *loss_out = beta * (float)(tot / ((double)N_total * (double)D));
    }
  }
  if (counts && perplexity_out) {
    double s = 0.0;
    const float nf = (float)N_total;
    for (int k = t; k < K; k += 1024) {
      const float p = (float)counts[k] / nf;           // mean over rows of the one-hot column (ref :105)
      s += (double)(p * logf(p + 1e-10f));     // ref :106
    }
    const double tot = block_sum_1024(s, sh);
    if (t == 0) *perplexity_out = expf(-(float)tot);
  }
}

__global__ __launch_bounds__(1024) void vq_cluster_update_kernel(const uint32_t* __restrict__ counts,
                                                                 const int64_t* __restrict__ counts64, int K, float decay,
                                                                 float omd, float eps, float k_eps,
                                                                 float* __restrict__ usage, float* __restrict__ cs) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int k = t; k < K; k += 1024) {
    const float c = counts64 ? (float)counts64[k] : (float)counts[k];
    usage[k] = usage[k] + c;                       // ref :67
    const float v = cs[k] * decay + omd * c;       // ref :71-72
    cs[k] = v;
    s += (double)v;
  }
  const float n = (float)block_sum_1024(s, sh);    // ref :75
  const float denom = n + k_eps;
  for (int k = t; k < K; k += 1024) cs[k] = (cs[k] + eps) / denom * n;  // ref :76-80, stored back
}

__global__ __launch_bounds__(256) void vq_codebook_update_kernel(const float* __restrict__ slabs, int nslab, int K, int D,
                                                                 float decay, float omd, const float* __restrict__ cs,
                                                                 float* __restrict__ ema_w, float* __restrict__ E) {
  const int64_t KD = (int64_t)K * D;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= KD) return;
  float dw = slabs[i];
  for (int t = 1; t < nslab; ++t) dw += slabs[(size_t)t * KD + i];
  const float w = ema_w[i] * decay + omd * dw;     // ref :85
  ema_w[i] = w;
  E[i] = w / cs[i / D];                            // ref :89
}

__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ grad_q, const float* __restrict__ grad_loss,
                                                          const float* __restrict__ z, const float* __restrict__ q_st,
                                                          int64_t numel, float coef, float* __restrict__ grad_z) {
  const float g = grad_loss ? *grad_loss * coef : 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < numel; i += stride) {
    if (i + 3 < numel) {
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + i);
      const f32x4 qv = *reinterpret_cast<const f32x4*>(q_st + i);
      f32x4 o = (zv - qv) * g;
      if (grad_q) o += *reinterpret_cast<const f32x4*>(grad_q + i);
      *reinterpret_cast<f32x4*>(grad_z + i) = o;
    } else {
      for (int64_t e = i; e < numel; ++e) grad_z[e] = (grad_q ? grad_q[e] : 0.0f) + g * (z[e] - q_st[e]);
    }
  }
}

__global__ __launch_bounds__(256) void vq_onehot_kernel(const int64_t* __restrict__ idx, int64_t N, int K,
                                                        float* __restrict__ enc) {
  // one workgroup per row chunk; every element is written exactly once (no memset pass)
  const int64_t total4 = ((int64_t)N * K) / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool vec = (K & 3) == 0;
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
      const int64_t e = i * 4;
      const int64_t n = e / K;
      const int k = (int)(e - n * K);
      const int64_t w = idx[n];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (w >= k && w < k + 4) v[(int)(w - k)] = 1.0f;
      *reinterpret_cast<f32x4*>(enc + e) = v;
    }
  } else {
    const int64_t total = (int64_t)N * K;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
      const int64_t n = e / K;
      enc[e] = (idx[n] == e - n * K) ? 1.0f : 0.0f;
    }
  }
}

}  // namespace

extern "C" {

int vq_scatter_slabs(int64_t N, int K, int D) {
  if (N <= 0 || K <= 0 || D <= 0) return 0;
  return scatter_plan(N, K, D).nslab;
}

int vq_code_histogram(const void* idx, int idx_is_64, int64_t M, int K, uint32_t* counts, int accumulate,
                      int32_t* control, void* stream) {
  if (!idx || !counts || M < 0 || K <= 0) return VQ_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) VQ_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)K * 4, s));
  if (M == 0) return VQ_OK;
  int64_t blocks = (M + 256 * 16 - 1) / (256 * 16);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (K <= HIST_LDS_BINS) {
    const size_t lds = (size_t)K * 4;
    if (idx_is_64)
      hipLaunchKernelGGL(vq_hist_lds_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), lds, s,
                         static_cast<const int64_t*>(idx), M, K, counts, control);
    else
      hipLaunchKernelGGL(vq_hist_lds_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), lds, s,
                         static_cast<const int32_t*>(idx), M, K, counts, control);
  } else {
    if (idx_is_64)
      hipLaunchKernelGGL(vq_hist_global_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                         static_cast<const int64_t*>(idx), M, K, counts, control);
    else
      hipLaunchKernelGGL(vq_hist_global_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                         static_cast<const int32_t*>(idx), M, K, counts, control);
  }
  return vq_hip_status(hipGetLastError());
}

int vq_ema_scatter(const float* z, int layout, int64_t N, int D, int64_t HW, const int32_t* idx32, int K,
                   float* slabs, void* stream) {
  if (!z || !idx32 || !slabs || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  const ScatterPlan p = scatter_plan(N, K, D);
  const int kr = (K + SCAT_KR - 1) / SCAT_KR;
  const int kr_len = K < SCAT_KR ? K : SCAT_KR;
  const size_t lds = (size_t)4 * kr_len * 4;
  static bool attr_done[64] = {};
  int dev = 0;
  VQ_CHECK_HIP(hipGetDevice(&dev));
  if (!attr_done[dev & 63]) {
    VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vq_scatter_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SCAT_KR * 4));
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL(vq_scatter_kernel, dim3((D + 3) / 4, p.nslab, kr), dim3(256), lds, (hipStream_t)stream, z,
                     layout == VQ_LAYOUT_NCHW ? 1 : 0, N, D, layout == VQ_LAYOUT_NCHW ? HW : 1, idx32, K,
                     p.rows_per_slab, slabs);
  return vq_hip_status(hipGetLastError());
}

int vq_reduce_slabs(const float* slabs, int nslab, int K, int D, float* dw, void* stream) {
  if (!slabs || !dw || nslab <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  const int64_t KD = (int64_t)K * D;
  hipLaunchKernelGGL(vq_reduce_slabs_kernel, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     slabs, nslab, KD, dw);
  return vq_hip_status(hipGetLastError());
}

int vq_step_scalars(const float* loss_partial, int64_t n_partial, int64_t N_total, int D, float beta,
                    const uint32_t* counts, int K, float* loss_out, float* perplexity_out, double* sq_sum_out,
                    void* stream) {
  if (N_total <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (counts && K <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_step_scalars_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_partial, n_partial,
                     N_total, D, beta, counts, K, loss_out, perplexity_out, sq_sum_out);
  return vq_hip_status(hipGetLastError());
}

int vq_ema_cluster_update(const uint32_t* counts, const int64_t* counts64, int K, float decay,
                          float one_minus_decay, float eps, float k_eps, float* usage, float* cluster_size,
                          void* stream) {
  if ((!counts && !counts64) || !usage || !cluster_size || K <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_cluster_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, counts64, K,
                     decay, one_minus_decay, eps, k_eps, usage, cluster_size);
  return vq_hip_status(hipGetLastError());
}

int vq_ema_codebook_update(const float* slabs, int nslab, int K, int D, float decay, float one_minus_decay,
                           const float* cluster_size, float* ema_w, float* E, void* stream) {
  if (!slabs || !cluster_size || !ema_w || !E || nslab <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  const int64_t KD = (int64_t)K * D;
  hipLaunchKernelGGL(vq_codebook_update_kernel, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, slabs, nslab, K, D, decay, one_minus_decay, cluster_size, ema_w, E);
  return vq_hip_status(hipGetLastError());
}

int vq_backward(const float* grad_q, const float* grad_loss, const float* z, const float* q_st, int64_t numel,
                float coef, float* grad_z, void* stream) {
  if (!z || !q_st || !grad_z || numel <= 0) return VQ_ERR_INVALID_ARG;
  int64_t blocks = (numel / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(vq_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_q, grad_loss,
                     z, q_st, numel, coef, grad_z);
  return vq_hip_status(hipGetLastError());
}

int vq_onehot(const int64_t* idx64, int64_t N, int K, float* enc, void* stream) {
  if (!idx64 || !enc || N <= 0 || K <= 0) return VQ_ERR_INVALID_ARG;
  int64_t blocks = ((int64_t)N * K / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(vq_onehot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx64, N, K, enc);
  return vq_hip_status(hipGetLastError());
}

}  // extern "C"
