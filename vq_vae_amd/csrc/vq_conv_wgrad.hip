// Weight and bias gradients of the 1x1 convolutions either side of the quantizer (gfx950).
//
// The reference trains `quant_conv` and `post_quant_conv` (vqvae/custom_vqgan.py:50-51, called at
// :66 and :72/:95) through autograd; for a 1x1 convolution  y[n, o] = sum_i W[o, i] x[n, i] + b[o]
// the parameter gradients are
//     dW[o, i] = sum_n gy[n, o] x[n, i]          db[o] = sum_n gy[n, o]
// a product whose long dimension is the batch: C_out x C_in outputs, each a sum over all N rows.
//
// Design (both operands stream, the outputs stay in registers):
//   * A workgroup of 8 waves owns a 256 x 256 block of dW for a contiguous range of rows; wave w
//     holds the 64 x 128 sub-block (w >> 1, w & 1) as 2 x 4 accumulator tiles of
//     v_mfma_f32_32x32x2_f32 -- 128 registers per lane, never spilled, written once at the end.
//   * The rows stream through LDS in stages of 32: 32 KiB of gy and 32 KiB of x per stage, copied
//     by LDS-DMA (global_load_lds_dwordx4) into one of two stage buffers while the MFMAs of the
//     other run.  The DMA writes LDS linearly, so the swizzle that makes the fragment reads
//     conflict-free is applied on the source side: each lane fetches the 16 bytes that belong in
//     the slot the hardware will write for it.
//   * The product is a sum over rows, so the order of rows inside a stage is free as long as both
//     operands use the same one; in NCHW a lane's four k-steps are one ds_read_b128 of four
//     consecutive pixels of its channel.
//   * Workgroup g leaves its partial block in the workspace; a second kernel adds the partials of
//     an output in the fixed order g = 0, 1, ... -- no atomics, the result is the same bits on
//     every run.
#include "vq_common.h"

namespace {

constexpr int WG_RS = 32;                          // rows per stage
constexpr int WG_CH = 256;                         // channels of one operand block
constexpr int WG_OP_FLOATS = WG_CH * WG_RS;        // 8192 floats = 32 KiB
constexpr int WG_STAGE_FLOATS = 2 * WG_OP_FLOATS;  // gy block, then x block
constexpr int WG_PART_FLOATS = WG_CH * WG_CH + WG_CH;  // a partial dW block and its db
constexpr int WG_MAX_SPLIT = 256;                  // workgroups along the rows (one per CU)

struct WgradArgs {
  const float* gy;
  const float* x;
  int64_t N;
  int64_t HW;
  int64_t nstages;      // ceil(N / 32)
  int64_t per_wg;       // stages per workgroup
  int Co, Ci;
  int tiles_i;          // 256-channel blocks along C_in
  float* part;          // [split][tiles][WG_PART_FLOATS]
};

// One operand block of one stage.  NCHW: slot (16 bytes) c*8 + (q ^ ((c >> 1) & 7)) holds pixels
// [4q, 4q + 4) of channel c.  Rows: float r*256 + (c ^ (32 * (r & 1))) holds channel c of row r.
template <bool NCHW>
__device__ __forceinline__ void wgrad_stage_copy(const float* src, int C, int ch0, int64_t HW, int64_t N,
                                                 int64_t stage, float* lds_op, int wave, int lane) {
  char* dst = reinterpret_cast<char*>(lds_op);
  if (NCHW) {
    const int64_t row0 = stage * WG_RS;
    const int64_t img = row0 / HW, hw0 = row0 - img * HW;
    const float* base = src + img * C * HW + hw0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = wave + 8 * k;            // 1 KiB piece: channels [8p, 8p + 8)
      const int c = 8 * p + (lane >> 3);
      const int q = (lane & 7) ^ ((c >> 1) & 7);
      int cg = ch0 + c;
      cg = cg < C ? cg : C - 1;
      const float* g = base + (int64_t)cg * HW + 4 * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = wave + 8 * k;            // 1 KiB piece: row r of the stage
      int64_t n = stage * WG_RS + r;
      n = n < N ? n : N - 1;
      int c = ch0 + 4 * (lane ^ (8 * (r & 1)));
      c = c + 4 <= C ? c : C - 4;
      const float* g = src + n * C + c;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + r * 1024), 16, 0, 0);
    }
  }
}

template <bool NCHW>
__global__ __launch_bounds__(512, 1) void vq_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int rg = wave >> 1, cg = wave & 1;   // 64 output channels, 128 input channels
  const int tile = blockIdx.y;
  const int o0 = (tile / a.tiles_i) * WG_CH, i0 = (tile % a.tiles_i) * WG_CH;

  const int64_t s_begin = (int64_t)blockIdx.x * a.per_wg;
  int64_t s_end = s_begin + a.per_wg;
  if (s_end > a.nstages) s_end = a.nstages;

  f32x16 acc[2][4];
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  if (s_begin < s_end) {
    wgrad_stage_copy<NCHW>(a.gy, a.Co, o0, a.HW, a.N, s_begin, lds, wave, lane);
    wgrad_stage_copy<NCHW>(a.x, a.Ci, i0, a.HW, a.N, s_begin, lds + WG_OP_FLOATS, wave, lane);
  }
  // The barriers that publish a stage must see this wave's LDS-DMA landed: vmcnt(0), spelled out
  // rather than left to the fence the compiler happens to emit for __syncthreads().
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  int cur = 0;
  for (int64_t st = s_begin; st < s_end; ++st) {
    float* buf = lds + cur * WG_STAGE_FLOATS;
    if (!NCHW && (st + 1) * WG_RS > a.N) {
      // the last stage of a batch that is not a multiple of 32 rows: rows past the end were copied
      // from row N-1; they must add nothing
      const int valid = (int)(a.N - st * WG_RS);
      for (int e = threadIdx.x; e < (WG_RS - valid) * WG_CH; e += 512) {
        buf[valid * WG_CH + e] = 0.0f;
        buf[WG_OP_FLOATS + valid * WG_CH + e] = 0.0f;
      }
      __syncthreads();
    }
    if (st + 1 < s_end) {
      float* nxt = lds + (cur ^ 1) * WG_STAGE_FLOATS;
      wgrad_stage_copy<NCHW>(a.gy, a.Co, o0, a.HW, a.N, st + 1, nxt, wave, lane);
      wgrad_stage_copy<NCHW>(a.x, a.Ci, i0, a.HW, a.N, st + 1, nxt + WG_OP_FLOATS, wave, lane);
    }
    const float* A = buf;
    const float* B = buf + WG_OP_FLOATS;
    if (NCHW) {
      const int sw = (i >> 1) & 7;
      const f32x4* A4 = reinterpret_cast<const f32x4*>(A) + (64 * rg + i) * 8;
      const f32x4* B4 = reinterpret_cast<const f32x4*>(B) + (128 * cg + i) * 8;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int q = (2 * m + h) ^ sw;
        f32x4 av[2], bv[4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) av[ta] = A4[ta * 256 + q];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) bv[tb] = B4[tb * 256 + q];
        if (cg == 0) {
#pragma unroll
          for (int ta = 0; ta < 2; ++ta) bsum[ta] += (av[ta][0] + av[ta][1]) + (av[ta][2] + av[ta][3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
              acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta][r], bv[tb][r], acc[ta][tb], 0, 0, 0);
      }
    } else {
      // row 2s + h of the stage; channel c sits at float c ^ 32h of the row, so tile t of a lane in
      // half h is at 32 (t ^ h): even tiles from one base, odd tiles from the other
      const float* Ar = A + h * WG_CH + 64 * rg + i;
      const float* Br = B + h * WG_CH + 128 * cg + i;
      const float* Ae = Ar + 32 * h;
      const float* Ao = Ar + 32 - 32 * h;
      const float* Be = Br + 32 * h;
      const float* Bo = Br + 32 - 32 * h;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float av[2][4], bv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = 4 * m + r;
          av[0][r] = Ae[s * 2 * WG_CH];
          av[1][r] = Ao[s * 2 * WG_CH];
          bv[0][r] = Be[s * 2 * WG_CH];
          bv[1][r] = Bo[s * 2 * WG_CH];
          bv[2][r] = Be[s * 2 * WG_CH + 64];
          bv[3][r] = Bo[s * 2 * WG_CH + 64];
        }
        if (cg == 0) {
#pragma unroll
          for (int ta = 0; ta < 2; ++ta) bsum[ta] += (av[ta][0] + av[ta][1]) + (av[ta][2] + av[ta][3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
              acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta][r], bv[tb][r], acc[ta][tb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA of stage st + 1 (see above)
    __syncthreads();
    cur ^= 1;
  }

  // partial block: register r of a tile is output channel (r & 3) + 8 (r >> 2) + 4 h, lane column i
  float* part = a.part + ((size_t)blockIdx.x * gridDim.y + tile) * WG_PART_FLOATS;
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 64 * rg + 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * h;
        part[o * WG_CH + 128 * cg + 32 * tb + i] = acc[ta][tb][r];
      }
  if (cg == 0) {
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      const float s = bsum[ta] + __shfl_xor(bsum[ta], 32);
      if (h == 0) part[WG_CH * WG_CH + 64 * rg + 32 * ta + i] = s;
    }
  }
}

// dW[o, i] (and db[o]) = the partials of the workgroups, added in workgroup order.
__global__ __launch_bounds__(256) void vq_wgrad_reduce_kernel(const float* __restrict__ part, int split, int tiles,
                                                              int tiles_i, int Co, int Ci, float* __restrict__ dW,
                                                              float* __restrict__ db) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nW = (int64_t)Co * Ci;
  if (e < nW) {
    const int o = (int)(e / Ci), i = (int)(e - (int64_t)o * Ci);
    const int tile = (o / WG_CH) * tiles_i + i / WG_CH;
    const float* p = part + (size_t)tile * WG_PART_FLOATS + (o % WG_CH) * WG_CH + (i % WG_CH);
    float s = 0.0f;
    for (int g = 0; g < split; ++g) s += p[(size_t)g * tiles * WG_PART_FLOATS];
    dW[e] = s;
  } else if (e < nW + Co && db) {
    const int o = (int)(e - nW);
    const int tile = (o / WG_CH) * tiles_i;
    const float* p = part + (size_t)tile * WG_PART_FLOATS + WG_CH * WG_CH + (o % WG_CH);
    float s = 0.0f;
    for (int g = 0; g < split; ++g) s += p[(size_t)g * tiles * WG_PART_FLOATS];
    db[o] = s;
  }
}

struct WgradPlan {
  int64_t nstages, per_wg;
  int split, tiles_o, tiles_i;
};

WgradPlan wgrad_plan(int64_t N, int Co, int Ci) {
  WgradPlan p;
  p.nstages = (N + WG_RS - 1) / WG_RS;
  p.tiles_o = (Co + WG_CH - 1) / WG_CH;
  p.tiles_i = (Ci + WG_CH - 1) / WG_CH;
  int64_t split = WG_MAX_SPLIT / (p.tiles_o * p.tiles_i);
  if (split < 1) split = 1;
  if (split > p.nstages) split = p.nstages;
  p.per_wg = (p.nstages + split - 1) / split;
  p.split = (int)((p.nstages + p.per_wg - 1) / p.per_wg);
  return p;
}

}  // namespace

extern "C" {

int vq_conv1x1_wgrad_supported(int layout, int64_t N, int C_out, int C_in, int64_t HW) {
  if (N <= 0 || C_out <= 0 || C_in <= 0) return 0;
  if (C_out > 4 * WG_CH || C_in > 4 * WG_CH) return 0;
  if (layout == VQ_LAYOUT_NCHW) return HW > 0 && N % HW == 0 && HW % WG_RS == 0;
  if (layout == VQ_LAYOUT_ND) return C_out % 4 == 0 && C_in % 4 == 0;
  return 0;
}

int64_t vq_conv1x1_wgrad_work_bytes(int64_t N, int C_out, int C_in) {
  if (N <= 0 || C_out <= 0 || C_in <= 0) return 0;
  const WgradPlan p = wgrad_plan(N, C_out, C_in);
  return (int64_t)p.split * p.tiles_o * p.tiles_i * WG_PART_FLOATS * 4;
}

int vq_conv1x1_wgrad(const float* gy, const float* x, int layout, int64_t N, int C_out, int C_in, int64_t HW,
                     float* dW, float* db, void* work, void* stream) {
  if (!gy || !x || !dW || !work) return VQ_ERR_INVALID_ARG;
  if (!vq_conv1x1_wgrad_supported(layout, N, C_out, C_in, HW)) return VQ_ERR_UNSUPPORTED;
  if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)work) & 15) != 0) return VQ_ERR_INVALID_ARG;
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  const WgradPlan p = wgrad_plan(N, C_out, C_in);
  WgradArgs a;
  a.gy = gy;
  a.x = x;
  a.N = N;
  a.HW = nchw ? HW : 1;
  a.nstages = p.nstages;
  a.per_wg = p.per_wg;
  a.Co = C_out;
  a.Ci = C_in;
  a.tiles_i = p.tiles_i;
  a.part = static_cast<float*>(work);
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds_bytes = 2 * WG_STAGE_FLOATS * 4;
  const int arc = nchw ? vq_lds_ready<vq_wgrad_kernel<true>, lds_bytes>() : vq_lds_ready<vq_wgrad_kernel<false>, lds_bytes>();
  if (arc != VQ_OK) return arc;
  const int tiles = p.tiles_o * p.tiles_i;
  const dim3 grid((unsigned)p.split, (unsigned)tiles), block(512);
  if (nchw) hipLaunchKernelGGL(vq_wgrad_kernel<true>, grid, block, lds_bytes, s, a);
  else hipLaunchKernelGGL(vq_wgrad_kernel<false>, grid, block, lds_bytes, s, a);
  VQ_CHECK_HIP(hipGetLastError());
  const int64_t outs = (int64_t)C_out * C_in + C_out;
  hipLaunchKernelGGL(vq_wgrad_reduce_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, s,
                     static_cast<const float*>(work), p.split, tiles, p.tiles_i, C_out, C_in, dW, db);
  return vq_hip_status(hipGetLastError());
}

}  // extern "C"
