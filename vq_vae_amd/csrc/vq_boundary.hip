// The quantizer's output side when a 1x1 convolution follows it (gfx950).
//
// In the reference the quantized latents go straight into `post_quant_conv`, a 1x1 convolution
// (vqvae/custom_vqgan.py:51, :72, called at :95 on the output of :81).  Every quantized row is a
// codebook row, so  conv(q)[n] = W q[n] + b = (E W^T + b)[idx[n]] : the convolution of N rows is
// the convolution of the K codebook rows (a K x D x D' product, 1/512 of the work at N = 4 Mi,
// K = 8192) followed by a gather.  The kernels here are that gather, fused with the commitment
// loss of vqvae/ema_vector_quantizer.py:92, and the matching backward, which reads the codebook
// row instead of a stored N x D tensor of quantized latents.
#include "vq_common.h"

namespace {

__device__ __forceinline__ float bwave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// out[n, :] = P[idx[n], :] in the layout of z (NCHW with DP channels, or (N, DP) rows), and the
// loss partial of the workgroup's 256 rows: sum over rows and channels of (E[idx[n], c] - z[n, c])^2,
// accumulated exactly as vq_gather_st_kernel does.  Rows whose index is outside [0, K) are skipped.
__global__ __launch_bounds__(256) void vq_gather_project_kernel(const float* __restrict__ z, int nchw, int64_t N,
                                                                int D, int64_t HW, const float* __restrict__ E,
                                                                int K, const int64_t* __restrict__ idx,
                                                                const float* __restrict__ P, int DP,
                                                                float* __restrict__ out,
                                                                float* __restrict__ loss_partial) {
  __shared__ float red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * VQ_WG_ROWS;
  float part = 0.0f;
  if (nchw) {
    // thread = row: the z loads and the stores of a wave are contiguous along hw; the two code rows
    // come through the caches, four 16-byte reads per 64-byte sector
    const int64_t n = row0 + t;
    const int64_t k = n < N ? idx[n] : -1;
    if (k >= 0 && k < K) {
      const int64_t b = n / HW, hw = n - b * HW;
      const int64_t zoff = b * D * HW + hw;
      const int64_t ooff = b * DP * HW + hw;
      if ((D & 15) == 0 && (DP & 15) == 0) {
        const f32x4* e = reinterpret_cast<const f32x4*>(E + (size_t)k * D);
        for (int c16 = 0; c16 < D / 16; ++c16) {
          f32x4 ev[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) ev[q] = e[4 * c16 + q];
          float zv[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) zv[i] = __builtin_nontemporal_load(&z[zoff + (int64_t)(16 * c16 + i) * HW]);   // streams: kept from displacing the table rows in L2
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float diff = ev[i >> 2][i & 3] - zv[i];
            part = __builtin_fmaf(diff, diff, part);
          }
        }
        const f32x4* p = reinterpret_cast<const f32x4*>(P + (size_t)k * DP);
        for (int c16 = 0; c16 < DP / 16; ++c16) {
          f32x4 pv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pv[q] = p[4 * c16 + q];
#pragma unroll
          for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(pv[i >> 2][i & 3], &out[ooff + (int64_t)(16 * c16 + i) * HW]);
        }
      } else {
        const float* es = E + (size_t)k * D;
        for (int c = 0; c < D; ++c) {
          const float diff = es[c] - z[zoff + (int64_t)c * HW];
          part = __builtin_fmaf(diff, diff, part);
        }
        const float* ps = P + (size_t)k * DP;
        for (int c = 0; c < DP; ++c) out[ooff + (int64_t)c * HW] = ps[c];
      }
    }
  } else if (((D | DP) & 3) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out) |
                                      reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(P)) & 15) == 0) {
    // rows: a wave takes rows wave, wave + 4, ... of the block, four at a time, 16 bytes per lane
    // (one wave instruction = a whole row at 256 channels); latents and outputs non-temporal
    const int D4 = D >> 2, DP4 = DP >> 2;
    for (int r = wave; r < VQ_WG_ROWS; r += 16) {
      int64_t n[4], k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        n[u] = row0 + r + 4 * u;
        k[u] = n[u] < N ? idx[n[u]] : -1;
        if (k[u] >= K) k[u] = -1;
      }
      for (int c4 = lane; c4 < D4; c4 += 64) {
        f32x4 zv[4], ev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u] >= 0) {
            zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + n[u] * D) + c4);
            ev[u] = reinterpret_cast<const f32x4*>(E + (size_t)k[u] * D)[c4];
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u] >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float diff = ev[u][i] - zv[u][i];
              part = __builtin_fmaf(diff, diff, part);
            }
          }
      }
      for (int c4 = lane; c4 < DP4; c4 += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u] >= 0)
            __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(P + (size_t)k[u] * DP)[c4],
                                        reinterpret_cast<f32x4*>(out + n[u] * DP) + c4);
      }
    }
  } else {
    // a wave walks one row at a time, lanes along the channels
    for (int r = wave; r < VQ_WG_ROWS; r += 4) {
      const int64_t n = row0 + r;
      if (n >= N) break;
      const int64_t k = idx[n];
      if (k < 0 || k >= K) continue;
      const float* e = E + (size_t)k * D;
      for (int c = lane; c < D; c += 64) {
        const float diff = e[c] - z[n * D + c];
        part = __builtin_fmaf(diff, diff, part);
      }
      const float* p = P + (size_t)k * DP;
      for (int c = lane; c < DP; c += 64) out[n * DP + c] = p[c];
    }
  }
  const float ws = bwave_sum(part);
  if (lane == 0) red[wave] = ws;
  __syncthreads();
  if (t == 0 && loss_partial) loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grad_z[n, c] = grad_q[n, c] + g * (z[n, c] - E[idx[n], c]),  g = *grad_loss * coef: the backward of
// the straight-through output plus the commitment term, with the quantized row read from the
// (saved, pre-update) codebook.  A row whose index is outside [0, K) passes grad_q through.
__global__ __launch_bounds__(256) void vq_backward_gather_kernel(const float* __restrict__ grad_q,
                                                                 const float* __restrict__ grad_loss,
                                                                 const float* __restrict__ z, int nchw, int64_t N,
                                                                 int D, int64_t HW, const float* __restrict__ E,
                                                                 int K, const int64_t* __restrict__ idx, float coef,
                                                                 float* __restrict__ grad_z) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * VQ_WG_ROWS;
  const float g = grad_loss ? *grad_loss * coef : 0.0f;
  if (nchw) {
    const int64_t n = row0 + t;
    if (n >= N) return;
    const int64_t k = idx[n];
    const bool ok = k >= 0 && k < K;
    const float* es = E + (size_t)(ok ? k : 0) * D;
    const float gg = ok ? g : 0.0f;
    const int64_t b = n / HW, hw = n - b * HW;
    const int64_t off0 = b * D * HW + hw;
    if ((D & 15) == 0) {
      const f32x4* e = reinterpret_cast<const f32x4*>(es);
      for (int c16 = 0; c16 < D / 16; ++c16) {
        f32x4 ev[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ev[q] = e[4 * c16 + q];
        float zv[16], gv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t o = off0 + (int64_t)(16 * c16 + i) * HW;
          zv[i] = __builtin_nontemporal_load(&z[o]);
          gv[i] = grad_q ? __builtin_nontemporal_load(&grad_q[o]) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
          __builtin_nontemporal_store(gv[i] + gg * (zv[i] - ev[i >> 2][i & 3]), &grad_z[off0 + (int64_t)(16 * c16 + i) * HW]);
      }
    } else {
      for (int c = 0; c < D; ++c) {
        const int64_t o = off0 + (int64_t)c * HW;
        grad_z[o] = (grad_q ? grad_q[o] : 0.0f) + gg * (z[o] - es[c]);
      }
    }
  } else if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(grad_z) | reinterpret_cast<uintptr_t>(E) |
                               reinterpret_cast<uintptr_t>(grad_q)) & 15) == 0) {
    // rows, 16 bytes per lane, four rows of the block in flight per wave (see vq_gather_project_kernel)
    const int D4 = D >> 2;
    for (int r = wave; r < VQ_WG_ROWS; r += 16) {
      int64_t n[4], k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        n[u] = row0 + r + 4 * u;
        k[u] = n[u] < N ? idx[n[u]] : -2;          // -2: no such row; -1 or beyond K: a row that passes grad_q through
        if (k[u] >= K) k[u] = -1;
      }
      for (int c4 = lane; c4 < D4; c4 += 64) {
        f32x4 zv[4], ev[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u] != -2) {
            zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + n[u] * D) + c4);
            gv[u] = grad_q ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grad_q + n[u] * D) + c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            ev[u] = reinterpret_cast<const f32x4*>(E + (size_t)(k[u] >= 0 ? k[u] : 0) * D)[c4];
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u] != -2) {
            const float gg = k[u] >= 0 ? g : 0.0f;
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = gv[u][i] + gg * (zv[u][i] - ev[u][i]);
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(grad_z + n[u] * D) + c4);
          }
      }
    }
  } else {
    for (int r = wave; r < VQ_WG_ROWS; r += 4) {
      const int64_t n = row0 + r;
      if (n >= N) break;
      const int64_t k = idx[n];
      const bool ok = k >= 0 && k < K;
      const float* e = E + (size_t)(ok ? k : 0) * D;
      const float gg = ok ? g : 0.0f;
      for (int c = lane; c < D; c += 64) {
        const int64_t o = n * D + c;
        grad_z[o] = (grad_q ? grad_q[o] : 0.0f) + gg * (z[o] - e[c]);
      }
    }
  }
}

}  // namespace

extern "C" {

int vq_gather_project(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, int K,
                      const int64_t* idx64, const float* P, int DP, float* out, float* loss_partial, void* stream) {
  if (!z || !E || !idx64 || !P || !out || N <= 0 || K <= 0 || D <= 0 || DP <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  if ((((uintptr_t)E | (uintptr_t)P) & 15) != 0) return VQ_ERR_INVALID_ARG;
  const int64_t blocks = vq_search_blocks(N);
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  hipLaunchKernelGGL(vq_gather_project_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                     nchw ? 1 : 0, N, D, nchw ? HW : 1, E, K, idx64, P, DP, out, loss_partial);
  return vq_hip_status(hipGetLastError());
}

int vq_backward_gather(const float* grad_q, const float* grad_loss, const float* z, int layout, int64_t N, int D,
                       int64_t HW, const float* E, int K, const int64_t* idx64, float coef, float* grad_z,
                       void* stream) {
  if (!z || !E || !idx64 || !grad_z || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  if (((uintptr_t)E & 15) != 0) return VQ_ERR_INVALID_ARG;
  const int64_t blocks = vq_search_blocks(N);
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  hipLaunchKernelGGL(vq_backward_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_q,
                     grad_loss, z, nchw ? 1 : 0, N, D, nchw ? HW : 1, E, K, idx64, coef, grad_z);
  return vq_hip_status(hipGetLastError());
}

}  // extern "C"
