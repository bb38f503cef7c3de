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
// Embedding sums (the scatter-add).  A workgroup of 8 waves owns 2 channels x one slab of
// rows x a range of up to 8192 codes and accumulates into LDS.  With the NCHW layout a
// channel of 64 consecutive rows is 256 contiguous bytes, so z streams in whole lines and is
// read exactly once.
//
// LDS float atomics (ds_add_f32) run at well under one lane per clock on gfx950, while the
// integer ones (ds_add_u32/u64) run at full bank rate, so the sums are kept in 64-bit
// fixed point: every value is scaled by 2^F (F chosen from max|z| and the batch's row count
// so that no sum can overflow), rounded once to an integer, and added with ds_add_u64.
// Integer addition commutes exactly, so the result is independent of the order in which
// waves and lanes arrive -- bit-reproducible -- and its error (one rounding of at most
// max|z| * 2^-40 per element) is far below an fp32 summation's.  A batch that contains a
// NaN or an infinity takes the float-atomic path instead, which propagates them into the
// sums exactly as the reference's dense product does.
// ---------------------------------------------------------------------------------------
constexpr int SCAT_KR = 8192;     // codes per pass (2 x 8192 x 8 B = 128 KiB of LDS)
constexpr int SCAT_MIN_ROWS = 4096;
constexpr int SCAT_THREADS = 512;

struct ScatterPlan { int nslab; int64_t rows_per_slab; };

// Workgroups a launch aims at: one per CU (a workgroup's accumulators take 128 KiB of LDS, so a CU holds one anyway).
// Every slab costs each of its workgroups an LDS clear and a flush of its K x 2 sums, whatever the batch: with 512
// workgroups (4 slabs at D = 256) a 64 Ki-row batch took 78 us, with 256 (2 slabs) 53 us; 4 Mi rows 0.93 ms either way.
int g_scatter_wgs = 256;   // (vq_internal_scatter_wgs changes it: experiments)

ScatterPlan scatter_plan(int64_t N, int K, int D) {
  const int slices = (D + 1) / 2;
  const int kr = (K + SCAT_KR - 1) / SCAT_KR;
  int64_t want = (g_scatter_wgs + (int64_t)slices * kr - 1) / ((int64_t)slices * kr);  // that many workgroups in all
  const int64_t maxslab = (N + SCAT_MIN_ROWS - 1) / SCAT_MIN_ROWS;
  if (want > maxslab) want = maxslab;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  // rows per slab: multiple of 1024 so that wave-sized runs stay aligned
  int64_t rows = (N + want - 1) / want;
  rows = (rows + 1023) / 1024 * 1024;
  const int nslab = (int)((N + rows - 1) / rows);
  return {nslab, rows};
}

// Division of a row index (< 2^31) by the invariant HW: q = umulhi(n, M) >> sh, exact for every
// n < 2^31 when M = floor(2^(31+s)/d) + 1 and sh = s - 1, s = ceil(log2 d)  (d >= 2).
struct FastDiv { uint32_t d, M, sh; };

FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{d, 0u, 0u};
  if (d <= 1) return f;
  uint32_t s = 0;
  while (((uint64_t)1 << s) < d) ++s;
  f.M = (uint32_t)((((uint64_t)1 << (31 + s)) / d) + 1);
  f.sh = s - 1;
  return f;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.M) >> f.sh);
}

__device__ __forceinline__ void scat_add(unsigned long long* acc_i, float* acc_f, bool fixed, double scale, int k,
                                         int kr_len, float v) {
  if ((unsigned)k < (unsigned)kr_len) {
    if (fixed) atomicAdd(&acc_i[k], (unsigned long long)__double2ll_rn((double)v * scale));
    else atomicAdd(&acc_f[k], v);
  }
}

// MODE 0: row-major z.  MODE 1: NCHW, any HW.  MODE 2: NCHW with HW % 4 == 0 -- four consecutive
// rows of one channel are 16 contiguous bytes, so every lane streams float4 / int4.
template <int MODE>
__global__ __launch_bounds__(SCAT_THREADS) void vq_scatter_kernel(const float* __restrict__ z, int N, int D,
                                                                  FastDiv hwdiv, const int32_t* __restrict__ idx32,
                                                                  int K, int code_base, int rows_per_slab,
                                                                  const int32_t* __restrict__ control,
                                                                  float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long acc_i[];  // [2][kr_len] int64, or floats
  float* acc_f = reinterpret_cast<float*>(acc_i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 2;
  const int slab = blockIdx.y;
  const int k0 = blockIdx.z * SCAT_KR;
  const int kr_len = (K - k0) < SCAT_KR ? (K - k0) : SCAT_KR;
  for (int i = threadIdx.x; i < 2 * kr_len; i += SCAT_THREADS) acc_i[i] = 0ull;

  // fixed-point scale: max|z| < 2^e, at most 2^r rows  =>  |sum| * 2^F < 2^62
  const uint32_t zmax_bits = (uint32_t)control[VQ_CTL_ZMAX];
  // 0 = unknown (or an all-zero batch): the float path is always valid
  const bool fixed = zmax_bits != 0 && zmax_bits < 0x7f800000u;
  int F = 0;
  if (fixed) {
    int e;
    (void)frexpf(__uint_as_float(zmax_bits), &e);  // zmax = m * 2^e, m in [0.5, 1)
    // (r from the batch's row count, not from the slab's: the scale, and with it every bit of a
    // slab's sums, does not depend on the slab plan)
    const int r = 32 - __clz((unsigned)(N > 1 ? N - 1 : 1));
    F = 62 - e - r;
  }
  const double scale = ldexp(1.0, F), inv_scale = ldexp(1.0, -F);
  __syncthreads();

  const int chsel = wave >> 2, sub = wave & 3;   // 4 waves per channel share the rows of the slab
  const int c = c0 + chsel;
  const int n_begin = slab * rows_per_slab;
  const int n_end = (N - n_begin) < rows_per_slab ? N : n_begin + rows_per_slab;
  unsigned long long* mine_i = acc_i + (size_t)chsel * kr_len;
  float* mine_f = acc_f + (size_t)chsel * kr_len;
  const int kbase = code_base + k0;
  const int HW = (int)hwdiv.d;
  if (c < D) {
    if (MODE == 2) {
      constexpr int U = 8;  // 8 x 16 B of z and of indices in flight per lane, 8 waves per CU (4: 1.02 ms, 8: 0.91 ms, 16: 0.95 ms at 4 Mi rows)
      for (int n0 = n_begin + 4 * lane + 256 * sub; n0 < n_end; n0 += 1024 * U) {
        f32x4 v[U];
        int4 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n = n0 + 1024 * u;
          const bool in = n < n_end;             // n_end, n_begin and n are multiples of 4 here
          const int nn = in ? n : n_begin;
          const uint32_t b = fastdiv((uint32_t)nn, hwdiv);
          const size_t off = ((size_t)b * D + c) * HW + (nn - (int)b * HW);
          v[u] = *reinterpret_cast<const f32x4*>(z + off);
          k[u] = *reinterpret_cast<const int4*>(idx32 + nn);
          if (!in) k[u] = int4{kbase - 1, kbase - 1, kbase - 1, kbase - 1};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          scat_add(mine_i, mine_f, fixed, scale, k[u].x - kbase, kr_len, v[u][0]);
          scat_add(mine_i, mine_f, fixed, scale, k[u].y - kbase, kr_len, v[u][1]);
          scat_add(mine_i, mine_f, fixed, scale, k[u].z - kbase, kr_len, v[u][2]);
          scat_add(mine_i, mine_f, fixed, scale, k[u].w - kbase, kr_len, v[u][3]);
        }
      }
    } else {
      constexpr int U = 8;
      for (int n0 = n_begin + lane + 64 * sub; n0 < n_end; n0 += 256 * U) {
        float v[U];
        int k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n = n0 + 256 * u;
          const bool in = n < n_end;
          const int nn = in ? n : n_begin;
          size_t off;
          if (MODE == 1) {
            const uint32_t b = fastdiv((uint32_t)nn, hwdiv);
            off = ((size_t)b * D + c) * HW + (nn - (int)b * HW);
          } else {
            off = (size_t)nn * D + c;
          }
          v[u] = z[off];
          k[u] = in ? idx32[nn] - kbase : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) scat_add(mine_i, mine_f, fixed, scale, k[u], kr_len, v[u]);
      }
    }
  }
  __syncthreads();
  // flush: slab[k][c0], slab[k][c0+1]
  float* out = slabs + (size_t)slab * K * D;
  for (int kk = threadIdx.x; kk < kr_len; kk += SCAT_THREADS) {
    float s0, s1;
    if (fixed) {
      s0 = (float)((double)(long long)acc_i[kk] * inv_scale);
      s1 = (float)((double)(long long)acc_i[kr_len + kk] * inv_scale);
    } else {
      s0 = acc_f[kk];
      s1 = acc_f[kr_len + kk];
    }
    float* dst = out + (size_t)(k0 + kk) * D + c0;
    dst[0] = s0;
    if (c0 + 1 < D) dst[1] = s1;
  }
}

// ---------------------------------------------------------------------------------------
// Row-major z ((N, D), which is also what a channels_last tensor is in memory), D % 4 == 0.
// There a row's D values are contiguous, so the decomposition turns around: a workgroup owns a
// range of srow_codes(D) codes with ALL channels in LDS (32 KiB of sums + 8 KiB of row lists: the
// kernel lives on memory latency, so several workgroups share a CU) and one slab of rows.  Its
// waves scan the slab's indices 512 rows at a time and list the rows whose code falls in the
// range; a listed row is read whole -- one coalesced float4 per lane -- and added to the code's
// LDS row by conflict-free ds_add_u64 (lane l owns channels 4l .. 4l+3).  z is read exactly once
// over all workgroups; the indices are re-read once per code range, from L2.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int srow_codes(int D) { return 4096 / D > 0 ? 4096 / D : 1; }   // x D x 8 B = 32 KiB

__global__ __launch_bounds__(SCAT_THREADS) void vq_scatter_rows_kernel(const float* __restrict__ z, int N, int D,
                                                                       const int32_t* __restrict__ idx32, int K,
                                                                       int code_base, int rows_per_slab,
                                                                       const int32_t* __restrict__ control,
                                                                       float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long racc_i[];  // [codes][D] int64, or floats
  float* racc_f = reinterpret_cast<float*>(racc_i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KR = srow_codes(D);
  const int k0 = blockIdx.x * KR;
  const int kr_len = (K - k0) < KR ? (K - k0) : KR;
  const int slab = blockIdx.y;
  for (int i = threadIdx.x; i < kr_len * D; i += SCAT_THREADS) racc_i[i] = 0ull;

  const uint32_t zmax_bits = (uint32_t)control[VQ_CTL_ZMAX];
  const bool fixed = zmax_bits != 0 && zmax_bits < 0x7f800000u;   // see vq_scatter_kernel
  int F = 0;
  if (fixed) {
    int e;
    (void)frexpf(__uint_as_float(zmax_bits), &e);
    const int r = 32 - __clz((unsigned)(N > 1 ? N - 1 : 1));
    F = 62 - e - r;
  }
  const double scale = ldexp(1.0, F), inv_scale = ldexp(1.0, -F);
  __syncthreads();

  const int n_begin = slab * rows_per_slab;
  const int n_end = (N - n_begin) < rows_per_slab ? N : n_begin + rows_per_slab;
  const int kbase = code_base + k0;
  const int D4 = D >> 2;
  // Matches are sparse (a range holds srow_codes of K codes), and a row fetched on the spot would
  // cost the wave a full memory latency each.  So a wave first collects (row, code) pairs in a
  // list of its own in LDS and drains it eight rows at a time, all eight loads in flight.
  volatile int2* list = reinterpret_cast<volatile int2*>(racc_i + (size_t)KR * D) + wave * 128;
  int cnt = 0;   // wave-uniform

  auto drain = [&]() {
    for (int e0 = 0; e0 < cnt; e0 += 8) {
      int rown[8], rowk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u < cnt ? e0 + u : e0;
        rown[u] = __builtin_amdgcn_readfirstlane(list[e].x);
        rowk[u] = __builtin_amdgcn_readfirstlane(list[e].y);
      }
      for (int c4 = lane; c4 < D4; c4 += 64) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(z + (size_t)rown[u] * D)[c4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (e0 + u < cnt) {
            const int base = rowk[u] * D + 4 * c4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (fixed) atomicAdd(&racc_i[base + i], (unsigned long long)__double2ll_rn((double)v[u][i] * scale));
              else atomicAdd(&racc_f[base + i], v[u][i]);
            }
          }
        }
      }
    }
    cnt = 0;
  };

  for (int g0 = n_begin + wave * 512; g0 < n_end; g0 += 8 * 512) {
    int kk[8];   // this wave's 512 rows: eight groups of 64, lane = row within the group
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int n = g0 + 64 * q + lane;
      kk[q] = n < n_end ? idx32[n] - kbase : -1;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool m = (unsigned)kk[q] < (unsigned)kr_len;
      const uint64_t mask = __ballot(m);
      if (mask != 0) {
        if (m) {
          const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
          list[pos].x = g0 + 64 * q + lane;
          list[pos].y = kk[q];
        }
        cnt += __builtin_popcountll(mask);
        if (cnt >= 64) drain();
      }
    }
  }
  drain();
  __syncthreads();
  // flush: slab[k0 + kk][:] , coalesced float4
  float* out = slabs + ((size_t)slab * K + k0) * D;
  for (int i = threadIdx.x; i < kr_len * D4; i += SCAT_THREADS) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = fixed ? (float)((double)(long long)racc_i[4 * i + j] * inv_scale) : racc_f[4 * i + j];
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// slabs of the row-major kernel: enough workgroups to fill the chip, and enough slabs that a batch
// whose rows all pick one code still spreads over many CUs
ScatterPlan scatter_plan_rows(int64_t N, int K, int D) {
  const int ranges = (K + srow_codes(D) - 1) / srow_codes(D);
  int64_t want = (4096 + ranges - 1) / ranges;
  if (want < 16) want = 16;
  const int64_t maxslab = (N + SCAT_MIN_ROWS - 1) / SCAT_MIN_ROWS;
  if (want > maxslab) want = maxslab;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  int64_t rows = (N + want - 1) / want;
  rows = (rows + 4095) / 4096 * 4096;      // a workgroup's eight waves take 4096 rows per round
  return {(int)((N + rows - 1) / rows), rows};
}

bool scatter_rows_ok(int layout, int64_t HW, int D, const float* z) {
  const bool rowmajor = layout == VQ_LAYOUT_ND || HW <= 1;
  return rowmajor && (D % 4 == 0) && D <= 2048 && ((reinterpret_cast<uintptr_t>(z) & 15) == 0);
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

// End-of-step housekeeping, done by whichever kernel runs last in a step so that the next step needs no
// memsets: the control words are copied to words 8..15 (where the host can still read the step's counters)
// and zeroed, and the histogram bins of the NEXT step (the other of two alternating buffers) are zeroed.
__device__ __forceinline__ void vq_step_cleanup(int32_t* control, uint32_t* counts_next, int K, int64_t tid,
                                                int64_t nthreads) {
  if (control != nullptr && tid < 8) {
    const int32_t v = control[tid];
    control[8 + tid] = v;
    control[tid] = 0;
  }
  if (counts_next != nullptr)
    for (int64_t k = tid; k < K; k += nthreads) counts_next[k] = 0u;
}

// vq_step_scalars and vq_ema_cluster_update in one launch (both are single-workgroup kernels over K-sized
// vectors; the arithmetic and its order are those of the two kernels).  usage == nullptr: scalars only.
__global__ __launch_bounds__(1024) void vq_step_finish_kernel(const float* __restrict__ loss_partial, int64_t n_partial,
                                                              int64_t N_total, int D, float beta,
                                                              const uint32_t* __restrict__ counts, int K, float decay,
                                                              float omd, float eps, float k_eps, float* __restrict__ usage,
                                                              float* __restrict__ cs, float* loss_out,
                                                              float* perplexity_out, int32_t* cleanup_control,
                                                              uint32_t* cleanup_counts) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  if (loss_partial && loss_out) {
    double s = 0.0;
    for (int64_t i = t; i < n_partial; i += 1024) s += (double)loss_partial[i];
    const double tot = block_sum_1024(s, sh);
    if (t == 0) *loss_out = beta * (float)(tot / ((double)N_total * (double)D));   // ref :92
  }
  if (counts && perplexity_out) {
    double s = 0.0;
    const float nf = (float)N_total;
    for (int k = t; k < K; k += 1024) {
      const float p = (float)counts[k] / nf;           // ref :105
      s += (double)(p * logf(p + 1e-10f));             // ref :106
    }
    const double tot = block_sum_1024(s, sh);
    if (t == 0) *perplexity_out = expf(-(float)tot);
  }
  if (usage != nullptr) {
    double s = 0.0;
    for (int k = t; k < K; k += 1024) {
      const float c = (float)counts[k];
      usage[k] = usage[k] + c;                       // ref :67
      const float v = cs[k] * decay + omd * c;       // ref :71-72
      cs[k] = v;
      s += (double)v;
    }
    const float n = (float)block_sum_1024(s, sh);    // ref :75
    const float denom = n + k_eps;
    for (int k = t; k < K; k += 1024) cs[k] = (cs[k] + eps) / denom * n;  // ref :76-80, stored back
  }
  vq_step_cleanup(cleanup_control, cleanup_counts, K, t, 1024);
}

// ---- data-parallel exchange: pack before the all-reduce, finish from the reduced buffer after it ----
// buf = [dw (K*D) | counts mod 4096 (K) | counts div 4096 (K) | rows mod 4096 | rows div 4096], all fp32.
// Both halves of a count sum to integers below 2^24 for up to 4096 ranks of fewer than 2^24 rows each, and
// fp32 adds such integers exactly, so ONE fp32 SUM all-reduce carries the embedding sums, the exact code
// counts and the exact global row count.
__global__ __launch_bounds__(256) void vq_pack_stats_kernel(const float* __restrict__ slabs, int nslab, int64_t KD,
                                                            const uint32_t* __restrict__ counts, int K, int64_t n_local,
                                                            float* __restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < KD) {
    float v = slabs[i];
    for (int t = 1; t < nslab; ++t) v += slabs[(size_t)t * KD + i];   // slab order, as vq_reduce_slabs
    buf[i] = v;
  } else if (i < KD + K) {
    buf[i] = (float)(counts[i - KD] & 4095u);
  } else if (i < KD + 2 * (int64_t)K) {
    buf[i] = (float)(counts[i - KD - K] >> 12);
  } else if (i == KD + 2 * (int64_t)K) {
    buf[i] = (float)(n_local & 4095);
    buf[i + 1] = (float)(n_local >> 12);
  }
}

// After the all-reduce: loss of the local batch, perplexity and cluster sizes from the reduced counts
// (the arithmetic of vq_step_finish, with counts and the row total read from the buffer's tail).
__global__ __launch_bounds__(1024) void vq_step_finish_reduced_kernel(
    const float* __restrict__ loss_partial, int64_t n_partial, int64_t N_local, int D, float beta,
    const float* __restrict__ tail, int K, float decay, float omd, float eps, float k_eps, float* __restrict__ usage,
    float* __restrict__ cs, float* loss_out, float* perplexity_out, int64_t* __restrict__ counts64_out,
    int64_t* n_total_out, int32_t* cleanup_control, uint32_t* cleanup_counts) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  if (loss_partial && loss_out) {
    double s = 0.0;
    for (int64_t i = t; i < n_partial; i += 1024) s += (double)loss_partial[i];
    const double tot = block_sum_1024(s, sh);
    if (t == 0) *loss_out = beta * (float)(tot / ((double)N_local * (double)D));   // ref :92, local-batch mean
  }
  const int64_t n_total = (int64_t)tail[2 * K] + ((int64_t)tail[2 * K + 1] << 12);
  if (t == 0 && n_total_out) *n_total_out = n_total;
  auto count_of = [&](int k) -> int64_t { return (int64_t)tail[k] + ((int64_t)tail[K + k] << 12); };
  if (perplexity_out) {
    double s = 0.0;
    const float nf = (float)n_total;
    for (int k = t; k < K; k += 1024) {
      const float p = (float)count_of(k) / nf;         // ref :105 over the global batch
      s += (double)(p * logf(p + 1e-10f));             // ref :106
    }
    const double tot = block_sum_1024(s, sh);
    if (t == 0) *perplexity_out = expf(-(float)tot);
  }
  if (counts64_out)
    for (int k = t; k < K; k += 1024) counts64_out[k] = count_of(k);
  if (usage != nullptr) {
    double s = 0.0;
    for (int k = t; k < K; k += 1024) {
      const float c = (float)count_of(k);
      usage[k] = usage[k] + c;                       // ref :67
      const float v = cs[k] * decay + omd * c;       // ref :71-72
      cs[k] = v;
      s += (double)v;
    }
    const float n = (float)block_sum_1024(s, sh);    // ref :75
    const float denom = n + k_eps;
    for (int k = t; k < K; k += 1024) cs[k] = (cs[k] + eps) / denom * n;  // ref :76-80, stored back
  }
  vq_step_cleanup(cleanup_control, cleanup_counts, K, t, 1024);
}

__global__ __launch_bounds__(256) void vq_codebook_update_end_kernel(const float* __restrict__ slabs, int nslab, int K,
                                                                     int D, float decay, float omd,
                                                                     const float* __restrict__ cs,
                                                                     float* __restrict__ ema_w, float* __restrict__ E,
                                                                     int32_t* cleanup_control, uint32_t* cleanup_counts) {
  const int64_t KD = (int64_t)K * D;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  vq_step_cleanup(cleanup_control, cleanup_counts, K, i, (int64_t)gridDim.x * 256);
  if (i >= KD) return;
  float dw = slabs[i];
  for (int t = 1; t < nslab; ++t) dw += slabs[(size_t)t * KD + i];
  const float w = ema_w[i] * decay + omd * dw;     // ref :85
  ema_w[i] = w;
  E[i] = w / cs[i / D];                            // ref :89
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


// ---------------------------------------------------------------------------------------
// Re-seeding of codes (maintenance).  Replaces the per-code Python loops of
// vqvae/ema_vector_quantizer.py:185-196, :262-274 and the row initialisation of :131-148: a
// target code becomes a uniformly drawn donor code plus N(0, noise^2), its EMA statistics the
// donor's times `scale`, its usage count zero.  Random numbers: Philox4x32-10, counter =
// (target position, channel group), key = seed -- the same seed gives the same result on any
// launch geometry.  Two kernels (stage, commit) so that a donor that is itself a target is read
// before anything is written.
// ---------------------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// two uniform 32-bit words -> two independent N(0, 1) values (Box-Muller; u1 in (0, 1])
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-08f;   // (k + 1) / 2^24
  const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;            // k / 2^24
  const float r = __builtin_sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  n0 = r * cs;
  n1 = r * sn;
}

__global__ __launch_bounds__(256) void vq_reseed_stage_kernel(const float* __restrict__ E, const float* __restrict__ ema_w,
                                                              const float* __restrict__ cs, int D,
                                                              const int64_t* __restrict__ donors, int n_donors,
                                                              float noise, float scale, uint32_t seed_lo, uint32_t seed_hi,
                                                              float* __restrict__ stage) {
  // stage row layout: [D new code][D new ema_w][1 new cluster size]
  const int t = blockIdx.x;
  const Philox4 pick = philox4x32_10((uint32_t)t, 0xffffffffu, 0u, 0u, seed_lo, seed_hi);
  const int64_t donor = donors[(int)(((uint64_t)pick.x * (uint64_t)n_donors) >> 32)];
  float* out = stage + (size_t)t * (2 * D + 1);
  for (int c4 = threadIdx.x; c4 < (D + 3) / 4; c4 += 256) {
    const Philox4 r = philox4x32_10((uint32_t)t, (uint32_t)c4, 0u, 0u, seed_lo, seed_hi);
    float nz[4];
    box_muller(r.x, r.y, nz[0], nz[1]);
    box_muller(r.z, r.w, nz[2], nz[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 4 * c4 + i;
      if (c < D) {
        out[c] = E[donor * D + c] + nz[i] * noise;
        out[D + c] = ema_w[donor * D + c] * scale;
      }
    }
  }
  if (threadIdx.x == 0) out[2 * D] = cs[donor] * scale;
}

__global__ __launch_bounds__(256) void vq_reseed_commit_kernel(const float* __restrict__ stage, int D,
                                                               const int64_t* __restrict__ targets, float* __restrict__ E,
                                                               float* __restrict__ ema_w, float* __restrict__ cs,
                                                               float* __restrict__ usage) {
  const int t = blockIdx.x;
  const int64_t k = targets[t];
  const float* in = stage + (size_t)t * (2 * D + 1);
  for (int c = threadIdx.x; c < D; c += 256) {
    E[k * D + c] = in[c];
    ema_w[k * D + c] = in[D + c];
  }
  if (threadIdx.x == 0) {
    cs[k] = in[2 * D];
    usage[k] = 0.0f;
  }
}

}  // namespace

extern "C" {

int vq_scatter_slabs(int64_t N, int K, int D) {
  if (N <= 0 || K <= 0 || D <= 0) return 0;
  return scatter_plan(N, K, D).nslab;
}

int vq_internal_scatter_wgs(int v) { const int old = g_scatter_wgs; g_scatter_wgs = v; return old; }

int vq_scatter_slabs_for(int64_t N, int K, int D, int layout, int64_t HW) {
  if (N <= 0 || K <= 0 || D <= 0) return 0;
  // (a row-major z that turns out not to be 16-byte aligned takes the generic kernel, which writes
  // fewer slabs and zero-fills the rest)
  if (scatter_rows_ok(layout, HW, D, nullptr)) return scatter_plan_rows(N, K, D).nslab;
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
                   int64_t code_base, int zmax_valid, int32_t* control, float* slabs, int nslab_given, void* stream) {
  if (!z || !idx32 || !slabs || !control || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  if (N >= ((int64_t)1 << 31) || code_base < 0 || code_base + K >= ((int64_t)1 << 31)) return VQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (!zmax_valid) {
    int rc = vq_launch_absmax(z, N * D, control, s);
    if (rc != VQ_OK) return rc;
  }
  if (scatter_rows_ok(layout, HW, D, z) && nslab_given != 0) {
    const ScatterPlan rp = scatter_plan_rows(N, K, D);
    if (rp.nslab == nslab_given) {
      const size_t rlds = (size_t)srow_codes(D) * D * 8 + 8 * 128 * 8;   // sums + the waves' row lists
      const dim3 rgrid((K + srow_codes(D) - 1) / srow_codes(D), rp.nslab);
      hipLaunchKernelGGL(vq_scatter_rows_kernel, rgrid, dim3(SCAT_THREADS), rlds, s, z, (int)N, D, idx32, K,
                         (int)code_base, (int)rp.rows_per_slab, control, slabs);
      return vq_hip_status(hipGetLastError());
    }
  }
  const ScatterPlan p = scatter_plan(N, K, D);
  if (nslab_given != 0 && nslab_given < p.nslab) return VQ_ERR_INVALID_ARG;
  if (nslab_given > p.nslab)   // (a row-major z that is not 16-byte aligned: the slabs this kernel does not write are zero)
    VQ_CHECK_HIP(hipMemsetAsync(slabs + (size_t)p.nslab * K * D, 0, (size_t)(nslab_given - p.nslab) * K * D * 4, s));
  const int kr = (K + SCAT_KR - 1) / SCAT_KR;
  const int kr_len = K < SCAT_KR ? K : SCAT_KR;
  const size_t lds = (size_t)2 * kr_len * 8;
  const bool nchw = layout == VQ_LAYOUT_NCHW && HW > 1;
  const bool vec4 = nchw && (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(z) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(idx32) & 15) == 0);
  const int mode = vec4 ? 2 : (nchw ? 1 : 0);
  const int arc = mode == 2 ? vq_lds_ready<vq_scatter_kernel<2>, 2 * SCAT_KR * 8>()
                : mode == 1 ? vq_lds_ready<vq_scatter_kernel<1>, 2 * SCAT_KR * 8>()
                            : vq_lds_ready<vq_scatter_kernel<0>, 2 * SCAT_KR * 8>();
  if (arc != VQ_OK) return arc;
  const FastDiv fd = make_fastdiv(nchw ? (uint32_t)HW : 1u);
  const dim3 grid((D + 1) / 2, p.nslab, kr);
  if (mode == 2)
    hipLaunchKernelGGL(vq_scatter_kernel<2>, grid, dim3(SCAT_THREADS), lds, s, z, (int)N, D, fd, idx32, K,
                       (int)code_base, (int)p.rows_per_slab, control, slabs);
  else if (mode == 1)
    hipLaunchKernelGGL(vq_scatter_kernel<1>, grid, dim3(SCAT_THREADS), lds, s, z, (int)N, D, fd, idx32, K,
                       (int)code_base, (int)p.rows_per_slab, control, slabs);
  else
    hipLaunchKernelGGL(vq_scatter_kernel<0>, grid, dim3(SCAT_THREADS), lds, s, z, (int)N, D, fd, idx32, K,
                       (int)code_base, (int)p.rows_per_slab, control, slabs);
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

int vq_step_finish(const float* loss_partial, int64_t n_partial, int64_t N_total, int D, float beta,
                   const uint32_t* counts, int K, float decay, float one_minus_decay, float eps, float k_eps,
                   float* usage, float* cluster_size, float* loss_out, float* perplexity_out,
                   int32_t* cleanup_control, uint32_t* cleanup_counts, void* stream) {
  if (N_total <= 0 || D <= 0 || K <= 0 || !counts) return VQ_ERR_INVALID_ARG;
  if ((usage == nullptr) != (cluster_size == nullptr)) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_step_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_partial, n_partial,
                     N_total, D, beta, counts, K, decay, one_minus_decay, eps, k_eps, usage, cluster_size, loss_out,
                     perplexity_out, cleanup_control, cleanup_counts);
  return vq_hip_status(hipGetLastError());
}

int vq_ema_codebook_update_end(const float* slabs, int nslab, int K, int D, float decay, float one_minus_decay,
                               const float* cluster_size, float* ema_w, float* E, int32_t* cleanup_control,
                               uint32_t* cleanup_counts, void* stream) {
  if (!slabs || !cluster_size || !ema_w || !E || nslab <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  const int64_t KD = (int64_t)K * D;
  hipLaunchKernelGGL(vq_codebook_update_end_kernel, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, slabs, nslab, K, D, decay, one_minus_decay, cluster_size, ema_w, E,
                     cleanup_control, cleanup_counts);
  return vq_hip_status(hipGetLastError());
}

size_t vq_packed_stats_floats(int K, int D, int with_sums) {
  if (K <= 0 || D <= 0) return 0;
  return (size_t)(with_sums ? (int64_t)K * D : 0) + 2 * (size_t)K + 2;
}

int vq_pack_stats(const float* slabs, int nslab, int K, int D, const uint32_t* counts, int64_t n_local, float* buf,
                  void* stream) {
  if (!counts || !buf || K <= 0 || D <= 0 || n_local < 0) return VQ_ERR_INVALID_ARG;
  if (slabs != nullptr && nslab <= 0) return VQ_ERR_INVALID_ARG;
  if (n_local >= ((int64_t)1 << 24)) return VQ_ERR_UNSUPPORTED;     // a count's upper half must stay below 2^12
  const int64_t KD = slabs ? (int64_t)K * D : 0;
  const int64_t total = KD + 2 * (int64_t)K + 2;
  hipLaunchKernelGGL(vq_pack_stats_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs,
                     nslab, KD, counts, K, n_local, buf);
  return vq_hip_status(hipGetLastError());
}

int vq_step_finish_reduced(const float* loss_partial, int64_t n_partial, int64_t N_local, int D, float beta,
                           const float* reduced_tail, int K, float decay, float one_minus_decay, float eps, float k_eps,
                           float* usage, float* cluster_size, float* loss_out, float* perplexity_out,
                           int64_t* counts64_out, int64_t* n_total_out, int32_t* cleanup_control,
                           uint32_t* cleanup_counts, void* stream) {
  if (!reduced_tail || K <= 0 || D <= 0 || N_local < 0) return VQ_ERR_INVALID_ARG;
  if ((usage == nullptr) != (cluster_size == nullptr)) return VQ_ERR_INVALID_ARG;
  if (loss_partial && loss_out && N_local == 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_step_finish_reduced_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_partial, n_partial,
                     N_local, D, beta, reduced_tail, K, decay, one_minus_decay, eps, k_eps, usage, cluster_size, loss_out,
                     perplexity_out, counts64_out, n_total_out, cleanup_control, cleanup_counts);
  return vq_hip_status(hipGetLastError());
}

int vq_quantize_step(const vq_step_args* p, void* stream) {
  if (!p || !p->z || !p->E || !p->idx64 || !p->q_out || !p->loss_out || !p->perplexity_out || !p->image || !p->e2 ||
      !p->control || !p->idx32 || !p->nf_rows || !p->loss_partial || !p->counts || !p->counts_next)
    return VQ_ERR_INVALID_ARG;
  if (p->N <= 0 || p->K <= 0 || p->D <= 0) return VQ_ERR_INVALID_ARG;
  if (p->training && (!p->ema_w || !p->cluster_size || !p->usage || !p->slabs || p->nslab <= 0)) return VQ_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (!p->control_clean) {
    rc = vq_control_reset(p->control, stream);
    if (rc != VQ_OK) return rc;
    VQ_CHECK_HIP(hipMemsetAsync(p->counts, 0, (size_t)p->K * 4, s));
  }
  rc = vq_codebook_prep(p->E, p->K, p->D, p->image, p->e2, p->control, stream);
  if (rc != VQ_OK) return rc;
  rc = vq_search(p->z, p->layout, p->N, p->D, p->HW, p->E, p->image, p->e2, p->K, p->temperature, 0, p->idx64, p->idx32,
                 p->dmin, p->q_out, p->loss_partial, p->nf_rows, p->control, p->split_ws, stream);
  if (rc != VQ_OK) return rc;
  rc = vq_code_histogram(p->idx32, 0, p->N, p->K, p->counts, 1, p->control, stream);
  if (rc != VQ_OK) return rc;
  const int64_t n_partial = (int64_t)(vq_loss_partial_bytes(p->N) / 4);
  if (p->training) {
    int nslab = p->nslab;
    if (p->sort_work != nullptr) {
      rc = vq_ema_scatter_sorted(p->z, p->N, p->D, p->idx32, p->K, 0, 1, p->control, p->slabs, p->sort_work, stream);
      nslab = 1;
    } else {
      rc = vq_ema_scatter(p->z, p->layout, p->N, p->D, p->HW, p->idx32, p->K, 0, 1, p->control, p->slabs, nslab, stream);
    }
    if (rc != VQ_OK) return rc;
    rc = vq_step_finish(p->loss_partial, n_partial, p->N, p->D, p->beta, p->counts, p->K, p->decay, p->one_minus_decay,
                        p->eps, p->k_eps, p->usage, p->cluster_size, p->loss_out, p->perplexity_out, nullptr, nullptr,
                        stream);
    if (rc != VQ_OK) return rc;
    return vq_ema_codebook_update_end(p->slabs, nslab, p->K, p->D, p->decay, p->one_minus_decay, p->cluster_size,
                                      p->ema_w, p->E, p->control, p->counts_next, stream);
  }
  return vq_step_finish(p->loss_partial, n_partial, p->N, p->D, p->beta, p->counts, p->K, p->decay, p->one_minus_decay,
                        p->eps, p->k_eps, nullptr, nullptr, p->loss_out, p->perplexity_out, p->control, p->counts_next,
                        stream);
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

size_t vq_reseed_stage_bytes(int n_targets, int D) {
  return n_targets > 0 && D > 0 ? (size_t)n_targets * (2 * (size_t)D + 1) * 4 : 256;
}

int vq_reseed_codes(float* E, float* ema_w, float* cluster_size, float* usage, int K, int D, const int64_t* targets,
                    int n_targets, const int64_t* donors, int n_donors, float noise, float scale, uint64_t seed,
                    float* stage, void* stream) {
  if (!E || !ema_w || !cluster_size || !usage || !targets || !donors || !stage) return VQ_ERR_INVALID_ARG;
  if (K <= 0 || D <= 0 || n_targets < 0 || n_donors <= 0) return VQ_ERR_INVALID_ARG;
  if (n_targets == 0) return VQ_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(vq_reseed_stage_kernel, dim3(n_targets), dim3(256), 0, s, E, ema_w, cluster_size, D, donors,
                     n_donors, noise, scale, (uint32_t)seed, (uint32_t)(seed >> 32), stage);
  hipLaunchKernelGGL(vq_reseed_commit_kernel, dim3(n_targets), dim3(256), 0, s, stage, D, targets, E, ema_w,
                     cluster_size, usage);
  return vq_hip_status(hipGetLastError());
}

}  // extern "C"
