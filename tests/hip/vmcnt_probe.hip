// Test-only probe (not part of the product library): do a wave's vector memory operations retire
// in issue order on gfx950 when LDS-DMA loads and global stores are mixed?
//
// The convolution kernels wait for an LDS-DMA with `s_waitcnt vmcnt(n)`, n = the number of store
// instructions the wave issued behind it.  That is exact only if a younger store never leaves the
// counter before an older load.  Each wave here repeats: clear an LDS slot, issue one DMA from a cold
// address (a miss all the way to HBM), issue NS stores to a line that stays in cache, wait for
// vmcnt(NS + SLACK), read the slot.  SLACK = 0 is the kernels' wait: a stale slot would mean a store
// overtook the load.  SLACK = 1 does not wait for the DMA at all and shows that the probe sees a
// late DMA when there is one.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NS, int SLACK>
__global__ __launch_bounds__(64) void vmcnt_probe_kernel(const uint32_t* __restrict__ src, uint64_t src_words,
                                                         uint32_t* __restrict__ hot, int iters,
                                                         unsigned int* __restrict__ stale) {
  __shared__ __attribute__((aligned(16))) uint32_t slot[256];
  const int lane = threadIdx.x;
  unsigned int bad = 0;
  uint64_t h = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345u;
  uint32_t* mine = hot + (size_t)blockIdx.x * 64;
  for (int it = 0; it < iters; ++it) {
    h = h * 6364136223846793005ull + 1442695040888963407ull;
    const uint64_t piece = (h >> 20) % (src_words / 256);          // a 1 KiB piece somewhere in the buffer
    const uint32_t* g = src + piece * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[lane * 4 + q] = 0xdeadbeefu;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      asm volatile("global_store_dword %0, %1, off" ::"v"(mine + lane), "v"((uint32_t)(it + s)) : "memory");
    }
    constexpr int N = NS + SLACK;
    __builtin_amdgcn_s_waitcnt(((N & 15) | ((N >> 4) << 14)) | 0x0F70);
    uint32_t v[4];
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<uint4*>(v)) : "v"((uint32_t)(lane * 16)) : "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) bad += v[q] != (uint32_t)((piece * 256 + lane * 4 + q) * 2654435761u) ? 1u : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (bad) atomicAdd(stale, bad);
}

__global__ __launch_bounds__(256) void vmcnt_probe_fill_kernel(uint32_t* __restrict__ src, uint64_t words) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) src[i] = (uint32_t)(i * 2654435761u);
}

extern "C" int vmcnt_probe_fill(uint32_t* src, uint64_t words, void* stream) {
  hipLaunchKernelGGL(vmcnt_probe_fill_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, src, words);
  return (int)hipGetLastError();
}

// mode 0: 16 stores, the kernels' wait; mode 1: 16 stores, one short of waiting for the DMA;
// mode 2: 56 operations behind the DMA (the largest count the kernels use)
extern "C" int vmcnt_probe(int mode, const uint32_t* src, uint64_t src_words, uint32_t* hot, int blocks, int iters,
                           unsigned int* stale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (src_words < 256 || blocks <= 0 || iters <= 0) return -1;
  if (mode == 0) hipLaunchKernelGGL((vmcnt_probe_kernel<16, 0>), dim3(blocks), dim3(64), 0, s, src, src_words, hot, iters, stale);
  else if (mode == 1) hipLaunchKernelGGL((vmcnt_probe_kernel<16, 1>), dim3(blocks), dim3(64), 0, s, src, src_words, hot, iters, stale);
  else if (mode == 2) hipLaunchKernelGGL((vmcnt_probe_kernel<56, 0>), dim3(blocks), dim3(64), 0, s, src, src_words, hot, iters, stale);
  else return -1;
  return (int)hipGetLastError();
}
