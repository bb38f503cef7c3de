// Codebook statistics for gfx950: cluster-size stats, index (histogram) stats, pairwise
// codebook distances, and the (distance, index) packing of the row-sharded search.
//
// Replaces vqvae/ema_vector_quantizer.py:205-232, vqvae/codebook_utilization_check.py:83-108
// and vqvae/vqgan_trainer.py:594-612 of the reference.  The K-sized reductions run in one
// workgroup with a fixed tree, so results are reproducible; nothing here syncs with the host.
#include "vq_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
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

__global__ __launch_bounds__(1024) void vq_cluster_stats_kernel(const float* __restrict__ cs, const float* __restrict__ usage,
                                                                int K, double* __restrict__ out) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  double act = 0.0, tot = 0.0, use = 0.0;
  for (int k = t; k < K; k += 1024) {
    act += cs[k] > 0.0f ? 1.0 : 0.0;
    tot += (double)cs[k];
    use += (double)usage[k];
  }
  act = block_sum(act, sh);
  tot = block_sum(tot, sh);
  use = block_sum(use, sh);
  double ent = 0.0;
  if (act > 0.0) {
    const float totf = (float)tot;
    double e = 0.0;
    for (int k = t; k < K; k += 1024) {
      const float p = cs[k] / totf;
      e -= (double)(p * log2f(p + 1e-10f));
    }
    ent = block_sum(e, sh);
  }
  if (t == 0) {
    const float entf = (float)ent;
    out[0] = act;
    out[1] = act / (double)K;
    out[2] = act > 0.0 ? (double)entf : 0.0;
    out[3] = act > 0.0 ? (double)entf / (double)log2f((float)K) : 0.0;
    out[4] = (double)(float)use;
  }
}

__global__ __launch_bounds__(1024) void vq_index_stats_kernel(const uint32_t* __restrict__ counts, int K, int64_t M,
                                                              double* __restrict__ out64, float* __restrict__ out32) {
  __shared__ double sh[1024];
  const int t = threadIdx.x;
  double uniq = 0.0, total = 0.0;
  for (int k = t; k < K; k += 1024) {
    uniq += counts[k] > 0 ? 1.0 : 0.0;
    total += (double)counts[k];
  }
  uniq = block_sum(uniq, sh);
  total = block_sum(total, sh);
  if (out64) {
    // float64, like the NumPy code: p = counts / len(all_indices)
    double e = 0.0, ps = 0.0;
    for (int k = t; k < K; k += 1024)
      if (counts[k] > 0) {
        const double p = (double)counts[k] / (double)M;
        e -= p * log2(p + 1e-10);
        ps += p;
      }
    const double ent = block_sum(e, sh);
    const double psum = block_sum(ps, sh);
    double g = 0.0;
    const double mean = uniq > 0.0 ? psum / uniq : 1.0;
    for (int k = t; k < K; k += 1024)
      if (counts[k] > 0) {
        const double r = ((double)counts[k] / (double)M) / mean;
        g += r * r;
      }
    const double gs = block_sum(g, sh);
    if (t == 0) {
      out64[0] = uniq;
      out64[1] = uniq / (double)K;
      out64[2] = ent;
      out64[3] = ent / log2((double)K);
      out64[4] = uniq > 0.0 ? 1.0 - gs / uniq : 0.0;
    }
  }
  if (out32) {
    // fp32, like the torch code: probs = counts / sum(counts)
    const float totf = (float)total;
    double e = 0.0;
    for (int k = t; k < K; k += 1024)
      if (counts[k] > 0) {
        const float p = (float)counts[k] / totf;
        if (p > 0.0f) e -= (double)(p * log2f(p + 1e-10f));
      }
    const double ent = block_sum(e, sh);
    if (t == 0) {
      out32[0] = (float)uniq;
      out32[1] = (float)uniq / (float)K;
      out32[2] = (float)ent;
      out32[3] = (float)ent / log2f((float)K);
    }
  }
}

__global__ __launch_bounds__(1024) void vq_pairwise_finish_kernel(const double* __restrict__ sum_partial,
                                                                  const float* __restrict__ mm_partial, int64_t nblocks,
                                                                  int K, int want_mean,
                                                                  const int32_t* __restrict__ control,
                                                                  double* __restrict__ out) {
  __shared__ double sh[1024];
  __shared__ float shmn[1024], shmx[1024];
  const int t = threadIdx.x;
  if (control[VQ_CTL_CB_FLAG] != 0 || control[VQ_CTL_NF_COUNT] != 0) {
    if (t == 0) { const double nan = __builtin_nan(""); out[0] = nan; out[1] = nan; out[2] = nan; }
    return;
  }
  double s = 0.0;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  for (int64_t i = t; i < nblocks; i += 1024) {
    s += sum_partial[i];
    mn = mm_partial[2 * i] < mn ? mm_partial[2 * i] : mn;
    mx = mm_partial[2 * i + 1] > mx ? mm_partial[2 * i + 1] : mx;
  }
  const double tot = block_sum(s, sh);
  shmn[t] = mn; shmx[t] = mx;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (t < off) {
      shmn[t] = shmn[t + off] < shmn[t] ? shmn[t + off] : shmn[t];
      shmx[t] = shmx[t + off] > shmx[t] ? shmx[t + off] : shmx[t];
    }
    __syncthreads();
  }
  if (t == 0) {
    out[0] = (double)shmn[0];
    out[1] = !want_mean ? tot : (K > 1 ? tot / ((double)K * (double)(K - 1)) : 0.0);
    out[2] = (double)shmx[0];
  }
}

__device__ __forceinline__ uint32_t ordered_key(float d) {
  if (d != d) return 0u;  // NaN sorts first, as torch.argmin treats it
  uint32_t u = __float_as_uint(d);
  if ((u << 1) == 0u) u = 0u;  // -0.0 and +0.0 compare equal in torch.argmin: one key for both
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void vq_pack_kernel(const float* __restrict__ dmin, const int64_t* __restrict__ idx,
                                                      int64_t N, int64_t* __restrict__ packed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // top bit flipped: the SIGNED order of the key is the intended order (int64 MIN all-reduce)
  if (i < N)
    packed[i] = (int64_t)((((uint64_t)ordered_key(dmin[i]) << 32) | (uint64_t)(uint32_t)idx[i]) ^ 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void vq_unpack_kernel(const int64_t* __restrict__ packed, int64_t N,
                                                        float* __restrict__ dmin, int64_t* __restrict__ idx64,
                                                        int32_t* __restrict__ idx32) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const uint64_t p = (uint64_t)packed[i] ^ 0x8000000000000000ull;
  const uint32_t key = (uint32_t)(p >> 32);
  const uint32_t id = (uint32_t)p;
  if (dmin) {
    float d;
    if (key == 0u) d = __builtin_nanf("");
    else d = __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
    dmin[i] = d;
  }
  if (idx64) idx64[i] = (int64_t)id;
  if (idx32) idx32[i] = (int32_t)id;
}

// acc[n] = min(acc[n], other[n]) over packed (distance, index) keys: the single-device form of the
// MIN all-reduce that merges codebook-row shards.
__global__ __launch_bounds__(256) void vq_packed_min_kernel(int64_t* __restrict__ acc, const int64_t* __restrict__ other,
                                                            int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    const int64_t a = acc[n], b = other[n];
    acc[n] = b < a ? b : a;
  }
}

}  // namespace

extern "C" {

int vq_cluster_stats(const float* cluster_size, const float* usage, int K, double* out, void* stream) {
  if (!cluster_size || !usage || !out || K <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_cluster_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cluster_size, usage, K, out);
  return vq_hip_status(hipGetLastError());
}

int vq_index_stats(const uint32_t* counts, int K, int64_t M, double* out64, float* out32, void* stream) {
  if (!counts || K <= 0 || M <= 0 || (!out64 && !out32)) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_index_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, K, M, out64, out32);
  return vq_hip_status(hipGetLastError());
}

static int pairwise_common(const float* E, const float* image, const float* e2, int K, int D, int64_t row_base,
                           int64_t n_rows, float* nn_dist, int64_t* nn_idx, double* sum_partial, float* mm_partial,
                           double* out, int want_mean, int32_t* control, void* stream) {
  if (!E || !image || !e2 || !nn_dist || !nn_idx || !sum_partial || !mm_partial || !out || !control || K <= 0)
    return VQ_ERR_INVALID_ARG;
  if (row_base < 0 || n_rows <= 0 || row_base + n_rows > K) return VQ_ERR_INVALID_ARG;
  if (!vq_fast_dim(D)) return VQ_ERR_UNSUPPORTED;
  const int64_t blocks = vq_search_blocks(n_rows);
  // rows with a non-finite norm are appended behind the min/max partials (scratch of n_rows ints)
  int32_t* nf_rows = reinterpret_cast<int32_t*>(mm_partial + 2 * blocks);
  int rc = vq_pairwise_search_launch(E, image, e2, K, D, row_base, n_rows, nn_dist, nn_idx, nullptr, sum_partial,
                                     mm_partial, nf_rows, control, stream);
  if (rc != VQ_OK) return rc;
  hipLaunchKernelGGL(vq_pairwise_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sum_partial, mm_partial,
                     blocks, K, want_mean, control, out);
  return vq_hip_status(hipGetLastError());
}

int vq_pairwise_stats(const float* E, const float* image, const float* e2, int K, int D, float* nn_dist,
                      int64_t* nn_idx, double* sum_partial, float* mm_partial, double* out, int32_t* control,
                      void* stream) {
  return pairwise_common(E, image, e2, K, D, 0, K, nn_dist, nn_idx, sum_partial, mm_partial, out, 1, control, stream);
}

int vq_pairwise_rows(const float* E, const float* image, const float* e2, int K, int D, int64_t row_base,
                     int64_t n_rows, float* nn_dist, int64_t* nn_idx, double* sum_partial, float* mm_partial,
                     double* out, int32_t* control, void* stream) {
  return pairwise_common(E, image, e2, K, D, row_base, n_rows, nn_dist, nn_idx, sum_partial, mm_partial, out, 0,
                         control, stream);
}

// ---- vq_pairwise_stats without caller-provided scratch ------------------------------------
// The partial sums, the min/max partials and the list of non-finite rows live in a file-scope
// device array instead of caller memory.  Every GPU holds its own copy of the array at its own
// address for the life of the process; the address is looked up once per device.
#define VQ_PW_STATIC_BLOCKS ((VQ_PAIRWISE_STATIC_MAX_K + VQ_WG_ROWS - 1) / VQ_WG_ROWS)
#define VQ_PW_STATIC_SUM_BYTES (((size_t)VQ_PW_STATIC_BLOCKS * 8 + 255) / 256 * 256)
#define VQ_PW_STATIC_MM_BYTES ((2 * (size_t)VQ_PW_STATIC_BLOCKS + VQ_PAIRWISE_STATIC_MAX_K) * 4)
static __device__ __align__(256) unsigned char vq_pw_scratch[VQ_PW_STATIC_SUM_BYTES + VQ_PW_STATIC_MM_BYTES];
static void* vq_pw_scratch_ptr[64] = {};  // device address of vq_pw_scratch, per device ordinal

// Address of this device's copy of the array; `dev` must be the current device.
static int pw_scratch_lookup(int dev) {
  if (dev < 0 || dev >= 64) return VQ_ERR_UNSUPPORTED;
  if (vq_pw_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    VQ_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(vq_pw_scratch)));
    if (p == nullptr) return VQ_ERR_HIP_BASE + (int)hipErrorInvalidSymbol;
    vq_pw_scratch_ptr[dev] = p;
  }
  return VQ_OK;
}

// Device a stream belongs to (the current device for the NULL stream).
static int stream_device(hipStream_t s, int* dev) {
  int cur = 0;
  VQ_CHECK_HIP(hipGetDevice(&cur));
  *dev = cur;
  if (s != nullptr) {
    hipDevice_t d;
    if (hipStreamGetDevice(s, &d) == hipSuccess) *dev = (int)d;
    else (void)hipGetLastError();  // e.g. a special stream handle: keep the current device
  }
  return VQ_OK;
}

int vq_pairwise_stats_static(const float* E, const float* image, const float* e2, int K, int D, float* nn_dist,
                             int64_t* nn_idx, double* out, int32_t* control, void* stream) {
  if (!E || !image || !e2 || !nn_dist || !nn_idx || !out || !control || K <= 0) return VQ_ERR_INVALID_ARG;
  if (K > VQ_PAIRWISE_STATIC_MAX_K) return VQ_ERR_TOO_LARGE;
  if (!vq_fast_dim(D)) return VQ_ERR_UNSUPPORTED;
  // the array of the GPU the stream (and so the kernels) run on, not of whatever device is current
  int dev = 0, cur = 0;
  int rc = stream_device((hipStream_t)stream, &dev);
  if (rc != VQ_OK) return rc;
  if (dev < 0 || dev >= 64) return VQ_ERR_UNSUPPORTED;
  if (vq_pw_scratch_ptr[dev] == nullptr) {
    VQ_CHECK_HIP(hipGetDevice(&cur));
    if (cur != dev) VQ_CHECK_HIP(hipSetDevice(dev));
    rc = pw_scratch_lookup(dev);
    if (cur != dev) (void)hipSetDevice(cur);
    if (rc != VQ_OK) return rc;
  }
  // Nothing in the array has to start at zero: every workgroup of the sweep overwrites its own
  // partials, the finish kernel reads exactly those, and the row list is indexed by a counter in
  // `control`, which the caller resets.
  unsigned char* base = static_cast<unsigned char*>(vq_pw_scratch_ptr[dev]);
  double* sum_partial = reinterpret_cast<double*>(base);
  float* mm_partial = reinterpret_cast<float*>(base + VQ_PW_STATIC_SUM_BYTES);
  return vq_pairwise_stats(E, image, e2, K, D, nn_dist, nn_idx, sum_partial, mm_partial, out, control, stream);
}

int vq_pack_dist_index(const float* dmin, const int64_t* idx64, int64_t N, int64_t* packed, void* stream) {
  if (!dmin || !idx64 || !packed || N <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_pack_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dmin, idx64,
                     N, packed);
  return vq_hip_status(hipGetLastError());
}

int vq_packed_min(int64_t* acc, const int64_t* other, int64_t N, void* stream) {
  if (!acc || !other || N <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_packed_min_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc,
                     other, N);
  return vq_hip_status(hipGetLastError());
}

int vq_unpack_dist_index(const int64_t* packed, int64_t N, float* dmin, int64_t* idx64, int32_t* idx32,
                         void* stream) {
  if (!packed || N <= 0) return VQ_ERR_INVALID_ARG;
  hipLaunchKernelGGL(vq_unpack_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed, N,
                     dmin, idx64, idx32);
  return vq_hip_status(hipGetLastError());
}

int vq_version(void) { return 130; }

int vq_init(int device) {
  int count = 0, cur = 0;
  VQ_CHECK_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count || device >= 64) return VQ_ERR_INVALID_ARG;
  VQ_CHECK_HIP(hipGetDevice(&cur));
  if (cur != device) VQ_CHECK_HIP(hipSetDevice(device));
  int rc = vq_lds_ensure(nullptr, 0, nullptr);   // every registered kernel, on the now-current device
  if (rc == VQ_OK) rc = pw_scratch_lookup(device);
  if (cur != device) (void)hipSetDevice(cur);
  return rc;
}

const char* vq_status_string(int status) {
  if (status == VQ_OK) return "ok";
  if (status == VQ_ERR_INVALID_ARG) return "invalid argument";
  if (status == VQ_ERR_UNSUPPORTED) return "unsupported shape";
  if (status == VQ_ERR_TOO_LARGE) return "size beyond the fixed maximum of this entry point";
  if (status >= VQ_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(status - VQ_ERR_HIP_BASE));
  return "unknown status";
}

}  // extern "C"

// ---- table of the kernels that need more dynamic LDS than the default (see vq_common.h) ----------
// Zero-initialised storage, so registration from other files' static initialisers works whatever
// the order in which the files are initialised.
namespace {
struct LdsEntry { const void* kern; int bytes; bool done[64]; };
constexpr int VQ_LDS_MAX = 512;
LdsEntry g_lds[VQ_LDS_MAX];
int g_lds_n = 0;

LdsEntry* lds_find(const void* kern) {
  for (int i = 0; i < g_lds_n; ++i)
    if (g_lds[i].kern == kern) return &g_lds[i];
  return nullptr;
}
}  // namespace

void vq_lds_register(const void* kern, int bytes) {
  LdsEntry* e = lds_find(kern);
  if (e == nullptr && g_lds_n < VQ_LDS_MAX) {
    e = &g_lds[g_lds_n++];
    e->kern = kern;
  }
  if (e != nullptr && bytes > e->bytes) e->bytes = bytes;
}

extern "C" int vq_large_lds_kernels(void) { return g_lds_n; }

// kern == nullptr: every registered kernel (vq_init).
int vq_lds_ensure(const void* kern, int bytes, int* dev_out) {
  int dev = 0;
  VQ_CHECK_HIP(hipGetDevice(&dev));
  if (dev_out) *dev_out = dev;
  if (dev < 0 || dev >= 64) return VQ_ERR_UNSUPPORTED;
  if (kern == nullptr) {
    for (int i = 0; i < g_lds_n; ++i) {
      if (g_lds[i].done[dev]) continue;
      VQ_CHECK_HIP(hipFuncSetAttribute(g_lds[i].kern, hipFuncAttributeMaxDynamicSharedMemorySize, g_lds[i].bytes));
      g_lds[i].done[dev] = true;
    }
    return VQ_OK;
  }
  LdsEntry* e = lds_find(kern);
  if (e == nullptr) {            // not registered at load time (should not happen): register now
    vq_lds_register(kern, bytes);
    e = lds_find(kern);
    if (e == nullptr) return vq_hip_status(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  }
  if (!e->done[dev]) {
    VQ_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, e->bytes));
    e->done[dev] = true;
  }
  return VQ_OK;
}
