// EMA embedding sums for row-major latents by counting sort (gfx950).
//
// Replaces vqvae/ema_vector_quantizer.py:84 (encodings.t() @ flat_input) for z stored as (N, D)
// rows -- which is also what a channels_last (B, D, H, W) tensor is in memory.  A row's D values
// are contiguous there, so the cheapest way to add the rows of one code is to read each row whole,
// once, in the order of its code:
//
//   1. ss_hist_kernel    per-workgroup histograms of the code indices (LDS bins, one chunk of rows
//                        per workgroup), written as a [chunks][K] table
//   2. ss_colscan_kernel the table becomes, per code, the exclusive prefix over the chunks; the
//                        per-code totals come out beside it
//      ss_offsets_kernel exclusive prefix of the totals over the codes: where each code's rows start
//   3. ss_rank_kernel    every row gets its position in code order (LDS cursors seeded from 2.) and
//                        leaves its row number and code there
//   4. ss_sum_kernel     the sorted positions are cut into equal pieces; a group of D/4 lanes reads
//                        the rows of its piece whole (one float4 per lane, eight rows in flight),
//                        accumulates them in registers and adds a finished run of one code to that
//                        code's accumulator row with 64-bit integer atomics
//   5. ss_finish_kernel  accumulators -> float sums
//
// The sums are kept in 64-bit fixed point (scale 2^F from max|z| and the batch's row count, as in
// vq_scatter_kernel), so neither the order of the rows inside a code (step 3 uses atomics) nor
// the way the positions are cut into pieces changes a single bit of the result.  A batch holding
// NaN or infinity takes float accumulation instead, which propagates them like the reference's
// dense product.  Work is balanced over the sorted positions, not over the codes: a batch whose
// rows all pick one code still covers the chip.
#include "vq_common.h"

namespace {

constexpr int SS_THREADS = 512;
constexpr int SS_MAX_K = 16384;      // LDS bins of the histogram / rank workgroups (64 KiB)
constexpr int SS_MIN_CHUNK = 8192;   // rows per histogram workgroup, at least
constexpr int SS_MAX_CHUNKS = 1024;

struct SortPlan {
  int chunk_rows;   // rows per histogram / rank workgroup
  int chunks;
  int piece_rows;   // sorted positions per lane group of the sum kernel (multiple of 8)
};

SortPlan sort_plan(int64_t N) {
  int64_t chunk = (N + SS_MAX_CHUNKS - 1) / SS_MAX_CHUNKS;
  if (chunk < SS_MIN_CHUNK) chunk = SS_MIN_CHUNK;
  chunk = (chunk + 511) / 512 * 512;
  const int chunks = (int)((N + chunk - 1) / chunk);
  // pieces: about 16 waves of work per CU when the batch allows, 8..128 rows each
  int64_t piece = N / (256 * 16 * 4);
  piece = (piece + 7) / 8 * 8;
  if (piece < 8) piece = 8;
  if (piece > 128) piece = 128;
  return {(int)chunk, chunks, (int)piece};
}

__host__ __device__ inline size_t ss_align(size_t b) { return (b + 255) / 256 * 256; }

struct SortWork {
  unsigned long long* acc;  // [K][4][D/4] fixed-point (or float) accumulators
  uint32_t* perm;           // [N] row number at each sorted position
  int32_t* skey;            // [N] code (relative to code_base) at each sorted position
  uint32_t* hist;           // [chunks][K]
  uint32_t* offsets;        // [K + 1]
  uint32_t* totals;         // [K]
};

size_t sort_work_bytes(int64_t N, int K, int D) {
  const SortPlan p = sort_plan(N);
  return ss_align((size_t)K * D * 8) + 2 * ss_align((size_t)N * 4) + ss_align((size_t)p.chunks * K * 4) +
         ss_align((size_t)(K + 1) * 4) + ss_align((size_t)K * 4);
}

SortWork carve(void* work, int64_t N, int K, int D) {
  const SortPlan p = sort_plan(N);
  char* b = static_cast<char*>(work);
  SortWork w;
  w.acc = reinterpret_cast<unsigned long long*>(b); b += ss_align((size_t)K * D * 8);
  w.perm = reinterpret_cast<uint32_t*>(b); b += ss_align((size_t)N * 4);
  w.skey = reinterpret_cast<int32_t*>(b); b += ss_align((size_t)N * 4);
  w.hist = reinterpret_cast<uint32_t*>(b); b += ss_align((size_t)p.chunks * K * 4);
  w.offsets = reinterpret_cast<uint32_t*>(b); b += ss_align((size_t)(K + 1) * 4);
  w.totals = reinterpret_cast<uint32_t*>(b);
  return w;
}

__global__ __launch_bounds__(SS_THREADS) void ss_hist_kernel(const int32_t* __restrict__ idx32, int N, int K,
                                                             int code_base, int chunk_rows,
                                                             uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t bins[];
  for (int k = threadIdx.x; k < K; k += SS_THREADS) bins[k] = 0;
  __syncthreads();
  const int n0 = blockIdx.x * chunk_rows;
  const int n1 = (N - n0) < chunk_rows ? N : n0 + chunk_rows;
  for (int n = n0 + threadIdx.x; n < n1; n += SS_THREADS) {
    const int kk = idx32[n] - code_base;
    if ((unsigned)kk < (unsigned)K) atomicAdd(&bins[kk], 1u);
  }
  __syncthreads();
  uint32_t* out = hist + (size_t)blockIdx.x * K;
  for (int k = threadIdx.x; k < K; k += SS_THREADS) out[k] = bins[k];
}

// hist[c][k] <- sum of hist[c'][k] over c' < c;  totals[k] <- sum over all chunks
__global__ __launch_bounds__(256) void ss_colscan_kernel(uint32_t* __restrict__ hist, int chunks, int K,
                                                         uint32_t* __restrict__ totals) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  uint32_t run = 0;
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = hist[(size_t)(c + u) * K + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      hist[(size_t)(c + u) * K + k] = run;
      run += v[u];
    }
  }
  for (; c < chunks; ++c) {
    const uint32_t v = hist[(size_t)c * K + k];
    hist[(size_t)c * K + k] = run;
    run += v;
  }
  totals[k] = run;
}

// offsets[k] = sum of totals[k'] over k' < k, offsets[K] = all of them (one workgroup, K <= 16384)
__global__ __launch_bounds__(1024) void ss_offsets_kernel(const uint32_t* __restrict__ totals, int K,
                                                          uint32_t* __restrict__ offsets) {
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const int per = (K + 1023) / 1024;
  const int k0 = t * per;
  uint32_t s = 0;
  for (int i = 0; i < per; ++i) s += (k0 + i < K) ? totals[k0 + i] : 0u;
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the per-thread sums
    const uint32_t add = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  uint32_t run = part[t] - s;
  for (int i = 0; i < per; ++i) {
    if (k0 + i < K) {
      offsets[k0 + i] = run;
      run += totals[k0 + i];
    }
  }
  if (t == 1023) offsets[K] = part[1023];
}

__global__ __launch_bounds__(SS_THREADS) void ss_rank_kernel(const int32_t* __restrict__ idx32, int N, int K,
                                                             int code_base, int chunk_rows,
                                                             const uint32_t* __restrict__ hist,
                                                             const uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ perm, int32_t* __restrict__ skey) {
  extern __shared__ uint32_t cursor[];
  const uint32_t* mine = hist + (size_t)blockIdx.x * K;
  for (int k = threadIdx.x; k < K; k += SS_THREADS) cursor[k] = offsets[k] + mine[k];
  __syncthreads();
  const int n0 = blockIdx.x * chunk_rows;
  const int n1 = (N - n0) < chunk_rows ? N : n0 + chunk_rows;
  for (int n = n0 + threadIdx.x; n < n1; n += SS_THREADS) {
    const int kk = idx32[n] - code_base;
    if ((unsigned)kk < (unsigned)K) {
      const uint32_t pos = atomicAdd(&cursor[kk], 1u);
      perm[pos] = (uint32_t)n;
      skey[pos] = kk;
    }
  }
}

// The fixed-point scale of a batch: max|z| < 2^e, at most 2^r rows => |sum| * 2^F < 2^62.
__device__ __forceinline__ bool ss_scale(const int32_t* control, int N, double& scale, double& inv_scale) {
  const uint32_t zmax_bits = (uint32_t)control[VQ_CTL_ZMAX];
  const bool fixed = zmax_bits != 0 && zmax_bits < 0x7f800000u;   // 0 = unknown / all zero: float sums are exact enough
  int F = 0;
  if (fixed) {
    int e;
    (void)frexpf(__uint_as_float(zmax_bits), &e);
    const int r = 32 - __clz((unsigned)(N > 1 ? N - 1 : 1));
    F = 62 - e - r;
  }
  scale = ldexp(1.0, F);
  inv_scale = ldexp(1.0, -F);
  return fixed;
}

// LPR = lanes per row = D / 4.  A wave holds 64 / LPR independent lane groups.
template <int LPR>
__global__ __launch_bounds__(256) void ss_sum_kernel(const float* __restrict__ z, int N, const uint32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ skey,
                                                     const uint32_t* __restrict__ offsets, int K, int piece_rows,
                                                     const int32_t* __restrict__ control,
                                                     unsigned long long* __restrict__ acc) {
  constexpr int D = 4 * LPR;
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPR;
  const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
  const int64_t M = (int64_t)offsets[K];          // rows that have a code in range
  const int64_t p0 = gid * piece_rows;
  if (p0 >= M) return;
  const int64_t p1 = (M - p0) < piece_rows ? M : p0 + piece_rows;
  double scale, inv_scale;
  const bool fixed = ss_scale(control, N, scale, inv_scale);
  float* accf = reinterpret_cast<float*>(acc);

  long long a[4] = {0, 0, 0, 0};
  float f[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = -1;

  auto flush = [&]() {
    const size_t base = (size_t)cur * 4 * LPR + gl;   // acc[code][i][lane of the row]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (fixed) atomicAdd(&acc[base + (size_t)i * LPR], (unsigned long long)a[i]);
      else atomicAdd(&accf[base + (size_t)i * LPR], f[i]);
      a[i] = 0;
      f[i] = 0.f;
    }
  };

  uint32_t rn[8];
  int rk[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int64_t q = p0 + u < p1 ? p0 + u : p1 - 1;
    rn[u] = perm[q];
    rk[u] = skey[q];
  }
  for (int64_t p = p0; p < p1; p += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(z + (size_t)rn[u] * D)[gl];
    int ck[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ck[u] = rk[u];
    if (p + 8 < p1) {   // row numbers of the next eight positions, in flight under this batch's adds
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t q = p + 8 + u < p1 ? p + 8 + u : p1 - 1;
        rn[u] = perm[q];
        rk[u] = skey[q];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p + u < p1) {
        if (ck[u] != cur) {
          if (cur >= 0) flush();
          cur = ck[u];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (fixed) a[i] += __double2ll_rn((double)v[u][i] * scale);
          else f[i] += v[u][i];
        }
      }
    }
  }
  if (cur >= 0) flush();
}

__global__ __launch_bounds__(256) void ss_finish_kernel(const unsigned long long* __restrict__ acc, int N, int K, int D,
                                                        const int32_t* __restrict__ control, float* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * D) return;
  double scale, inv_scale;
  const bool fixed = ss_scale(control, N, scale, inv_scale);
  const int LPR = D / 4;
  const int64_t k = i / D;
  const int c = (int)(i - k * D);
  const size_t src = ((size_t)k * 4 + (c & 3)) * LPR + (c >> 2);
  dw[i] = fixed ? (float)((double)(long long)acc[src] * inv_scale) : reinterpret_cast<const float*>(acc)[src];
}

bool sorted_supported(int K, int D) {
  if (K <= 0 || K > SS_MAX_K || D < 4 || D > 256 || (D & 3)) return false;
  const int lpr = D / 4;
  return (lpr & (lpr - 1)) == 0;   // D/4 lanes per row: a power of two up to 64
}

template <int LPR>
void launch_sum(const float* z, int N, const SortWork& w, int K, int piece_rows, const int32_t* control, hipStream_t s) {
  constexpr int G = 64 / LPR;
  const int64_t groups = ((int64_t)N + piece_rows - 1) / piece_rows;
  const int64_t blocks = (groups + 4 * G - 1) / (4 * G);
  hipLaunchKernelGGL(ss_sum_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s, z, N, w.perm, w.skey, w.offsets, K,
                     piece_rows, control, w.acc);
}

}  // namespace

extern "C" {

int vq_scatter_sorted_supported(int K, int D) { return sorted_supported(K, D) ? 1 : 0; }

size_t vq_scatter_sorted_work_bytes(int64_t N, int K, int D) {
  if (N <= 0 || !sorted_supported(K, D)) return 0;
  return sort_work_bytes(N, K, D);
}

int vq_ema_scatter_sorted(const float* z, int64_t N, int D, const int32_t* idx32, int K, int64_t code_base,
                          int zmax_valid, int32_t* control, float* dw, void* work, void* stream) {
  if (!z || !idx32 || !control || !dw || !work || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (!sorted_supported(K, D)) return VQ_ERR_UNSUPPORTED;
  if (N >= ((int64_t)1 << 31) || code_base < 0 || code_base + K >= ((int64_t)1 << 31)) return VQ_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(work) & 255)) return VQ_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!zmax_valid) {
    int rc = vq_launch_absmax(z, N * D, control, s);
    if (rc != VQ_OK) return rc;
  }
  const SortPlan p = sort_plan(N);
  const SortWork w = carve(work, N, K, D);
  int arc = vq_lds_ready<ss_hist_kernel, SS_MAX_K * 4>();
  if (arc == VQ_OK) arc = vq_lds_ready<ss_rank_kernel, SS_MAX_K * 4>();
  if (arc != VQ_OK) return arc;
  VQ_CHECK_HIP(hipMemsetAsync(w.acc, 0, (size_t)K * D * 8, s));
  hipLaunchKernelGGL(ss_hist_kernel, dim3(p.chunks), dim3(SS_THREADS), (size_t)K * 4, s, idx32, (int)N, K,
                     (int)code_base, p.chunk_rows, w.hist);
  hipLaunchKernelGGL(ss_colscan_kernel, dim3((K + 255) / 256), dim3(256), 0, s, w.hist, p.chunks, K, w.totals);
  hipLaunchKernelGGL(ss_offsets_kernel, dim3(1), dim3(1024), 0, s, w.totals, K, w.offsets);
  hipLaunchKernelGGL(ss_rank_kernel, dim3(p.chunks), dim3(SS_THREADS), (size_t)K * 4, s, idx32, (int)N, K,
                     (int)code_base, p.chunk_rows, w.hist, w.offsets, w.perm, w.skey);
  VQ_CHECK_HIP(hipGetLastError());
  switch (D / 4) {
    case 1: launch_sum<1>(z, (int)N, w, K, p.piece_rows, control, s); break;
    case 2: launch_sum<2>(z, (int)N, w, K, p.piece_rows, control, s); break;
    case 4: launch_sum<4>(z, (int)N, w, K, p.piece_rows, control, s); break;
    case 8: launch_sum<8>(z, (int)N, w, K, p.piece_rows, control, s); break;
    case 16: launch_sum<16>(z, (int)N, w, K, p.piece_rows, control, s); break;
    case 32: launch_sum<32>(z, (int)N, w, K, p.piece_rows, control, s); break;
    default: launch_sum<64>(z, (int)N, w, K, p.piece_rows, control, s); break;
  }
  VQ_CHECK_HIP(hipGetLastError());
  const int64_t KD = (int64_t)K * D;
  hipLaunchKernelGGL(ss_finish_kernel, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0, s, w.acc, (int)N, K, D, control,
                     dw);
  return vq_hip_status(hipGetLastError());
}

}  // extern "C"
