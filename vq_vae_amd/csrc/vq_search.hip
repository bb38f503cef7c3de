// Nearest-codebook search for gfx950 (MI355X).
//
// Replaces vqvae/ema_vector_quantizer.py:35-58, :92-98 of the reference: distances
// ||z||^2 + ||e||^2 - 2 z.e, optional temperature, argmin with lowest-index ties, the
// gather of the winning code, the straight-through output and the commitment-loss sum.
// The N x K distance matrix and the N x K one-hot matrix never exist in memory.
//
// Design (z-stationary, MFMA-bound):
//   * A workgroup of 8 waves owns 256 rows of z.  Each wave keeps its 32 rows (one 32-row
//     MFMA column tile; a template parameter allows 4 waves with two tiles each) in registers
//     for the whole sweep -- the 512 KiB register file of a CU is the largest on-chip store,
//     so z is read from HBM exactly once and never touches LDS.
//   * The codebook streams through LDS in stages of 64 codes.  vq_codebook_prep stores it
//     in "image" order, i.e. exactly the bytes a wave's lanes want for the A operand of
//     v_mfma_f32_32x32x2_f32, so a stage is a linear 65 KiB global->LDS DMA copy
//     (global_load_lds_dwordx4) and an A fragment for four k-steps is one conflict-free
//     ds_read_b128 per lane.  Two stage buffers: the DMA of stage s+1 flies under the
//     MFMAs of stage s.
//   * MFMA orientation: codes on the 32 rows (A), latents on the 32 columns (B).  A lane
//     then holds 16 distances of ONE latent, so the argmin is a register scan plus one
//     cross-half exchange at the very end.
//   * fp32 MFMA is bit-for-bit an fmaf chain over the channels in increasing order, which
//     is what the CPU oracle computes; indices are therefore compared bit-exactly.
//   * The scan of a finished accumulator tile runs while the MFMAs of the next one are in
//     flight (two accumulator sets alternate), and the SIMD's second wave fills what is left.
#include "vq_common.h"

namespace {

struct SearchArgs {
  const float* z;
  const float* E;       // row-major codebook, for the gather
  const float* image;   // staged codebook
  const float* e2;      // per-code squared norms (exact kernel)
  int64_t N;
  int64_t HW;
  int K;
  int Dreal;            // channels of z and E in memory (the kernels' template D when no padding is involved)
  int nstages;
  float T;
  int64_t code_base;
  int64_t* idx64;
  int32_t* idx32;
  float* dmin;
  float* q_out;
  float* loss_partial;
  int64_t loss_slots;   // entries of loss_partial the search itself owns (unused ones are zeroed)
  int32_t* nf_rows;
  int32_t* control;
  // pairwise mode (z := E): skip the pair (n, n); per-workgroup partials of sum/min/max
  double* pw_sum;
  float* pw_minmax;
  int64_t pw_row_base;  // pairwise over a range of rows: row n of z is code pw_row_base + n
  // codebook split for small batches: blockIdx.y sweeps stages [y*sps, y*sps + sps) and leaves
  // its per-row winner in part_d / part_i [ksplit][N]; a merge kernel picks the overall winner
  int ksplit;
  int stages_per_split;
  float* part_d;
  int32_t* part_i;
  // filtered search, fall-back role: run only if the first filter pass left more than N/8 rows
  // unsettled (control[VQ_CTL_OVF_COUNT]); rows the verification already handed to the exact
  // kernel (marked -2 in idx32) are skipped without a new list entry
  int fallback_role;
  // deferred temperature: rows with a code close above the minimum are listed from the END of nf_rows backwards
  // (control[VQ_CTL_NEAR_COUNT] entries) and redone one per workgroup with every distance divided; a channel
  // count that the per-row routine does not take (near_ok == 0) sends them to the front list instead
  int near_ok;
  float* near_loss;     // VQ_NEAR_SLOTS loss partials for such rows (zeroed by the sweep, which counts them as unused slots)
};

// The filtered search gives up on a batch whose first pass left more than an eighth of the rows
// unsettled (codes too close together for the bf16 filter): the fp32 kernel then redoes all rows.
__device__ __forceinline__ bool vq_filter_gave_up(const int32_t* control, int64_t N) {
  return (int64_t)control[VQ_CTL_OVF_COUNT] * 8 > N;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Issue the LDS-DMA copy of one codebook stage.  The image is lane-linear, so piece p of
// the stage (1 KiB) goes to LDS offset p*1024; lane l supplies bytes [16l, 16l+16).
template <int D, int NW>
__device__ __forceinline__ void stage_copy(const float* image, int stage, float* lds_buf, int wave, int lane) {
  // With 8 waves only waves 0-3 issue the DMA.  Waves w and w+4 share a SIMD: while the first
  // one spends ~1.5k cycles issuing its 16-17 pieces, its partner already runs MFMAs, and for
  // the rest of the stage the two stay out of phase -- one wave's scan of a finished tile then
  // falls into the other's pure MFMA stretch instead of colliding with the partner's own scan.
  constexpr int NL = NW == 8 ? 4 : NW;  // loader waves
  constexpr int PIECES = D / 4 + 1;
  if (wave >= NL) return;
  const char* src = reinterpret_cast<const char*>(image) + (size_t)stage * vq_stage_floats(D) * 4 + lane * 16;
  char* dst = reinterpret_cast<char*>(lds_buf);
#pragma unroll
  for (int i = 0; i < (PIECES + NL - 1) / NL; ++i) {
    const int p = i * NL + wave;
    if (p < PIECES) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024),
          (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  }
}

// The same copy addressed through a buffer descriptor over the image: a wave takes a contiguous run of the
// stage's pieces, so that a piece is "descriptor + scalar offset + immediate (0 / 1 / 2 / 3 KiB) + lane x 16" on
// the memory side and "M0 + the same immediate + lane x 16" on the LDS side -- a DMA instruction needs no vector
// instruction beside it (the pointer form spends a 64-bit vector add per piece, and the piece addresses, hoisted
// out of the stage loop, live in scalar registers that spill), only two scalar adds per four pieces.  Vector
// instructions are what a loader wave must not spend: they issue in the gaps of its SIMD partner's MFMAs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* image, int nstages, int D) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(image), 0, (int)((size_t)nstages * vq_stage_floats(D) * 4), 0x00020000);
}
template <int D, int NW>
__device__ __forceinline__ void stage_copy_buf(__amdgpu_buffer_rsrc_t r, int stage, float* lds_buf, int wave, int lane) {
  constexpr int NL = NW == 8 ? 4 : NW;  // loader waves (see stage_copy)
  constexpr int PIECES = D / 4 + 1;
  constexpr int PER = (PIECES + NL - 1) / NL;
  if (wave >= NL) return;
  const int first = wave * PER;
  const int count = PIECES - first < PER ? PIECES - first : PER;
  dma_run<0, PER>(r, lds_buf + first * 256, (uint32_t)lane * 16u, (uint32_t)stage * (uint32_t)(vq_stage_floats(D) * 4) + (uint32_t)first * 1024u,
                  count);
}

// Scan one finished 32-code x 32-latent accumulator tile.  Register r of the tile is code
// row (r&3) + 8*(r>>2) + 4*(lane>>5); the 4*(lane>>5) part is added once after the sweep.
//
// TEMP: how a temperature T != 1 (finite, > 0) enters.
//   VQ_T_NONE     T = 1.
//   VQ_T_LITERAL  every distance is divided by T before it is compared, as the reference writes it.
//   VQ_T_DEFERRED distances are compared undivided, and `sec` keeps the value that the running minimum
//                 displaced last.  Division by T > 0 with one rounding is monotone, so the smallest quotient is
//                 that of the smallest distance m, and the reference's winner is the first code whose quotient
//                 equals it.  That is the first code at distance m unless some code before it lies so close
//                 above m that the two quotients round to the same number -- and the first such code of a scan
//                 in index order was a running minimum when it was met (everything before it is farther), so
//                 it or a later, smaller running minimum is what `sec` holds at the end.  The caller tests
//                 sec against m (vq_t_near) and sends the few rows that fail to the exact kernel.
enum { VQ_T_NONE = 0, VQ_T_LITERAL = 1, VQ_T_DEFERRED = 2 };

template <int TEMP, bool PAIRWISE>
__device__ __forceinline__ void scan_tile(const f32x16& c, const f32x4 (&e2v)[4], float z2, int code0, float T,
                                          int self_code, float& best, int& bidx, float& sec, double& psum, float& pmin,
                                          float& pmax) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = z2 + e2v[q][r];
      float d = __builtin_fmaf(-2.0f, c[4 * q + r], t);
      if (TEMP == VQ_T_LITERAL) d = d / T;
      const int code = code0 + 8 * q + r;
      if (PAIRWISE) {
        const bool real = (e2v[q][r] < __builtin_inff()) && (code != self_code);
        if (real) {
          psum += (double)d;
          pmin = d < pmin ? d : pmin;
          pmax = d > pmax ? d : pmax;
        } else {
          d = __builtin_inff();
        }
      }
      const bool tk = d < best;
      if (TEMP == VQ_T_DEFERRED) sec = tk ? best : sec;
      best = tk ? d : best;
      bidx = tk ? code : bidx;
    }
  }
}

// Deferred temperature: two finished scans of disjoint code sets (best_a, sec_a) and (best_b, sec_b) become
// one -- the smaller of the two minima is the minimum, and the closest value above it that either scan knows of
// is the other minimum (when it differs) or one of the two displaced values.
__device__ __forceinline__ float vq_t_merge_sec(float best_a, float sec_a, float best_b, float sec_b) {
  const float lo = best_a < best_b ? best_a : best_b;
  const float hi = best_a < best_b ? best_b : best_a;
  float sec = sec_a < sec_b ? sec_a : sec_b;
  if (hi > lo && hi < sec) sec = hi;
  return sec;
}

// Deferred temperature: may a code at distance sec > m have the same quotient by T as one at distance m?
// With u = 2^-24, a quotient of normal size is rounded by at most a factor 1 +- u, so two distances more than
// a factor (1 + u) / (1 - u) < 1 + 2^-22 apart keep their order strictly (for m < 0 the same bound holds on
// the absolute errors); gap and 2^-22 |m| are both exact in fp32.  Below normal size the quotients' spacing is
// absolute (2^-149), and a quotient that overflows equals every other that does.  q = m / T as computed.
__device__ __forceinline__ bool vq_t_near(float m, float sec, float T, float q) {
  const float gap = sec - m;                       // exact when the two are close
  const float am = m < 0.0f ? -m : m;
  const float aq = q < 0.0f ? -q : q;
  return gap <= 2.384185791015625e-07f * am || gap <= T * 7.888609052210118e-31f || !(aq < __builtin_inff());
}

// CT = 32-latent column tiles per wave.  CT = 2: 4 waves, one per SIMD, 512 registers each.
// CT = 1: 8 waves, two per SIMD, 256 registers each -- the partner wave's MFMAs cover a wave's
// scan, DMA issue and LDS waits.
// PAD: z and E have a.Dreal < D channels in memory; the image carries zeros in channels Dreal .. D-1, the
// tile's registers for those channels are zero, and the tail takes the winning code's values from the image.
template <int D, int CT, bool NCHW, int TEMP, bool PAIRWISE, bool PAD = false>
__global__ __launch_bounds__(512 / CT, 2 / CT) void vq_search_mfma_kernel(SearchArgs a) {
  constexpr int NW = 8 / CT;  // waves per workgroup
  constexpr int S = D / 2;  // k-steps of the 32x32x2 MFMA
  constexpr int G = D / 8;  // ds_read_b128 groups (4 k-steps each)
  constexpr int STAGE_F = VQ_STAGE_CODES * D + 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * VQ_WG_ROWS + wave * (32 * CT);

  if (a.fallback_role && !vq_filter_gave_up(a.control, a.N)) return;

  const bool split = a.ksplit > 1;
  const int st0 = blockIdx.y * a.stages_per_split;                 // first stage of this block
  const int nst = (a.nstages - st0) < a.stages_per_split ? (a.nstages - st0) : a.stages_per_split;
  const float* image = a.image + (size_t)st0 * STAGE_F;

  // A non-finite codebook sends every row to the exact kernel.
  if (a.control[VQ_CTL_CB_FLAG] != 0) {
    if (threadIdx.x == 0 && a.loss_partial && !split)
      for (int64_t i = blockIdx.x; i < a.loss_slots || i == blockIdx.x; i += gridDim.x) a.loss_partial[i] = 0.0f;
    if (split && h == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int64_t n = row0 + ct * 32 + j;
        if (n < a.N) a.part_i[(size_t)blockIdx.y * a.N + n] = -1;
      }
    }
    if (h == 0 && blockIdx.y == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int64_t n = row0 + ct * 32 + j;
        if (n < a.N) a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
      }
    }
    return;
  }

  // Prologue DMA of stage 0 goes out first; it lands while z is being loaded.
  const __amdgpu_buffer_rsrc_t img = image_rsrc(image, nst, D);
  stage_copy_buf<D, NW>(img, 0, lds, wave, lane);

  // ---- z tile into registers, in B-operand order: zr[ct][s] = z[row][2s + h] -----------
  float zr[CT][S];
  float z2[CT];
  int64_t zoff[CT];            // element offset of this lane's first channel of the row
  int64_t zstride = 2 * a.HW;  // NCHW: elements between channels 2s+h and 2s+2+h
  uint32_t zmax_bits = 0;      // max |z| of this lane's values (bit pattern), for the EMA scatter's scaling
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int64_t n = row0 + ct * 32 + j;
    if (n >= a.N) n = a.N - 1;  // clamped duplicate: loads stay in bounds, outputs are masked
    float p = 0.0f;
    if (PAD) {
      // element (channel c) of row n sits at zoff + c * cs; channels beyond Dreal read as zero
      const int64_t cs = NCHW ? a.HW : 1;
      if (NCHW) {
        const int64_t b = n / a.HW, hw = n - b * a.HW;
        zoff[ct] = b * a.Dreal * a.HW + hw;
      } else {
        zoff[ct] = n * a.Dreal;
      }
      const float* src = a.z + zoff[ct];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int c = 2 * s + h;
        const float v = src[(int64_t)(c < a.Dreal ? c : a.Dreal - 1) * cs];
        zr[ct][s] = c < a.Dreal ? v : 0.0f;
      }
    } else if (NCHW) {
      const int64_t b = n / a.HW, hw = n - b * a.HW;
      zoff[ct] = (b * D + h) * a.HW + hw;
      const float* src = a.z + zoff[ct];
#pragma unroll
      for (int s = 0; s < S; ++s) zr[ct][s] = src[(int64_t)s * zstride];
    } else {
      zoff[ct] = n * D;
      const f32x4* src = reinterpret_cast<const f32x4*>(a.z + zoff[ct]);
#pragma unroll
      for (int m = 0; m < D / 4; ++m) {
        const f32x4 v = src[m];
        zr[ct][2 * m] = h ? v[1] : v[0];
        zr[ct][2 * m + 1] = h ? v[3] : v[2];
      }
    }
    uint32_t am = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      p = __builtin_fmaf(zr[ct][s], zr[ct][s], p);
      const uint32_t ab = __float_as_uint(zr[ct][s]) & 0x7fffffffu;
      am = ab > am ? ab : am;
    }
    zmax_bits = am > zmax_bits ? am : zmax_bits;
    z2[ct] = p + __shfl_xor(p, 32);  // even-channel chain + odd-channel chain
    // Cut the address expressions of the loads off from those of the stores after the
    // sweep: otherwise the compiler keeps 128 store addresses alive through the MFMA loop.
    asm volatile("" : "+v"(zoff[ct]));
  }
  asm volatile("" : "+s"(zstride));
  if (!PAIRWISE && blockIdx.y == 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)zmax_bits, off);
      zmax_bits = o > zmax_bits ? o : zmax_bits;
    }
    if (lane == 0 && zmax_bits) atomicMax(reinterpret_cast<unsigned int*>(&a.control[VQ_CTL_ZMAX]), zmax_bits);
  }

  float best[CT];
  float sec[CT];               // deferred temperature: the value the running minimum displaced last
  int bidx[CT];
  double psum[CT];
  float pmin[CT], pmax[CT];
  int self_code[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    best[ct] = __builtin_inff();
    sec[ct] = __builtin_inff();
    bidx[ct] = 0;
    psum[ct] = 0.0;
    pmin[ct] = __builtin_inff();
    pmax[ct] = -__builtin_inff();
    // pairwise: the row index is the code index; remove the 4*h part the scan leaves out
    self_code[ct] = PAIRWISE ? (int)(a.pw_row_base + row0 + 32 * ct + j) - 4 * h : -1;
  }

  // Accumulator set B (second 32 codes of a stage) is scanned one stage late, under the
  // MFMAs of the next stage's first half.  Before the first stage it holds zeros and its
  // norms are +inf, so that first scan changes nothing.
  f32x16 cA[CT], cB[CT];
  f32x4 e2B[4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) cB[ct][i] = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) e2B[q] = f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the stage-0 DMA (and its z loads)
  __syncthreads();                     // stage 0 has landed

  int cur = 0;
  for (int st = 0; st < nst; ++st) {
    float* buf = lds + cur * STAGE_F;
    if (st + 1 < nst) stage_copy_buf<D, NW>(img, st + 1, lds + (cur ^ 1) * STAGE_F, wave, lane);

    const f32x4* A = reinterpret_cast<const f32x4*>(buf) + lane;
    const f32x4* e2s = reinterpret_cast<const f32x4*>(buf + VQ_STAGE_CODES * D) + h;

    // ---- first half: codes [64 st, 64 st + 32) into set A; scan set B of stage st-1 -----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) cA[ct][i] = 0.0f;
    {
      f32x4 af = A[0];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 an = A[(g + 1 < G ? g + 1 : g) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            cA[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], zr[ct][4 * g + q], cA[ct], 0, 0, 0);
        af = an;
      }
    }
    {
      const int code0 = (st0 + st - 1) * VQ_STAGE_CODES + 32;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        scan_tile<TEMP, PAIRWISE>(cB[ct], e2B, z2[ct], code0, a.T, self_code[ct], best[ct], bidx[ct], sec[ct], psum[ct],
                                  pmin[ct], pmax[ct]);
    }

    // ---- second half: codes [64 st + 32, 64 st + 64) into set B; scan set A --------------
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) cB[ct][i] = 0.0f;
    {
      f32x4 af = A[G * 64];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 an = A[(G + (g + 1 < G ? g + 1 : g)) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            cB[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], zr[ct][4 * g + q], cB[ct], 0, 0, 0);
        af = an;
      }
    }
    {
      f32x4 e2A[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e2A[q] = e2s[2 * q];
      const int code0 = (st0 + st) * VQ_STAGE_CODES;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        scan_tile<TEMP, PAIRWISE>(cA[ct], e2A, z2[ct], code0, a.T, self_code[ct], best[ct], bidx[ct], sec[ct], psum[ct],
                                  pmin[ct], pmax[ct]);
    }
    // norms of this stage's second half, kept for the deferred scan
#pragma unroll
    for (int q = 0; q < 4; ++q) e2B[q] = e2s[8 + 2 * q];

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the DMA of stage st+1
    __syncthreads();                     // every wave is done with buf; stage st+1 has landed
    cur ^= 1;
  }
  {
    const int code0 = (st0 + nst - 1) * VQ_STAGE_CODES + 32;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      scan_tile<TEMP, PAIRWISE>(cB[ct], e2B, z2[ct], code0, a.T, self_code[ct], best[ct], bidx[ct], sec[ct], psum[ct],
                                pmin[ct], pmax[ct]);
  }

  // ---- merge the two lane halves (rows +0..3 vs +4..7 of every 8), write the winners ----
  float lp = 0.0f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int bi = bidx[ct] + 4 * h;
    float b = best[ct];
    const float ob = __shfl_xor(b, 32);
    const int oi = __shfl_xor(bi, 32);
    const bool take = (ob < b) || (ob == b && oi < bi);
    bool near = false;
    if (TEMP == VQ_T_DEFERRED) {
      const float sc = vq_t_merge_sec(b, sec[ct], ob, __shfl_xor(sec[ct], 32));
      const float m = take ? ob : b;
      const float qd = m / a.T;
      near = vq_t_near(m, sc, a.T, qd);
      b = qd; // the winning distance as the reference reports it
    } else {
      b = take ? ob : b;
    }
    bi = take ? oi : bi;

    const int64_t n = row0 + ct * 32 + j;
    const bool valid = n < a.N;
    const bool finite = z2[ct] <= VQ_NORM_LIMIT && !near &&  // false for NaN
                        !(a.fallback_role && valid && a.idx32[n] == -2);
    const bool ok = valid && finite;
    if (h == 0) {
      if (split) {
        if (valid) {
          a.part_d[(size_t)blockIdx.y * a.N + n] = b;
          a.part_i[(size_t)blockIdx.y * a.N + n] = ok ? (int32_t)(bi + a.code_base) : -1;
          if (!ok && blockIdx.y == 0) a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
        }
      } else if (ok) {
        if (a.idx64) a.idx64[n] = (int64_t)bi + a.code_base;
        if (a.idx32) a.idx32[n] = (int32_t)(bi + a.code_base);
        if (a.dmin) a.dmin[n] = b;
      } else if (valid && !a.fallback_role) {
        if (near && a.near_ok && z2[ct] <= VQ_NORM_LIMIT) {
          a.nf_rows[a.N - 1 - atomicAdd(&a.control[VQ_CTL_NEAR_COUNT], 1)] = (int32_t)n;
          a.idx32[n] = 0;                   // the exact kernel's workgroups count themselves in here
          if (a.idx64) a.idx64[n] = -1;     // ... and keep the smallest (quotient, index) key here
        } else {
          a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
        }
      }
    }
    if (PAIRWISE) {
      // reduce sum/min/max of this latent's distances over the two halves
      psum[ct] += __shfl_xor(psum[ct], 32);
      const float omn = __shfl_xor(pmin[ct], 32), omx = __shfl_xor(pmax[ct], 32);
      pmin[ct] = omn < pmin[ct] ? omn : pmin[ct];
      pmax[ct] = omx > pmax[ct] ? omx : pmax[ct];
      if (!valid || h != 0) { psum[ct] = 0.0; pmin[ct] = __builtin_inff(); pmax[ct] = -__builtin_inff(); }
    }

    // ---- gather E[bi], straight-through output, squared-difference sum -------------------
    if (PAD && a.q_out != nullptr) {
      // the winning code's values in this lane's channel order, from the image (in global memory): piece
      // g of the code's 32-row tile holds for lane (i, h) the channels 8 g + h + {0, 2, 4, 6}
      const int wc = ok ? bi : 0;
      const f32x4* irow = reinterpret_cast<const f32x4*>(a.image + (size_t)(wc >> 6) * STAGE_F + (size_t)((wc >> 5) & 1) * 32 * D) +
                          (h * 32 + (wc & 31));
      const int64_t cs = NCHW ? a.HW : 1;
      float* obase = a.q_out + zoff[ct];
      float part = 0.0f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 v = irow[g * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 8 * g + 2 * q + h;
          const float d0 = v[q] - zr[ct][4 * g + q];
          part = __builtin_fmaf(d0, d0, part);      // zero channels add nothing
          if (ok && c < a.Dreal) obase[(int64_t)c * cs] = zr[ct][4 * g + q] + d0;
        }
      }
      lp += ok ? part : 0.0f;
    } else if (a.q_out != nullptr) {
      const f32x4* erow = reinterpret_cast<const f32x4*>(a.E + (size_t)(ok ? bi : 0) * D);
      float* obase = a.q_out + zoff[ct];
      float part = 0.0f;
#pragma unroll
      for (int m = 0; m < D / 4; ++m) {
        const f32x4 v = erow[m];
        const float q0 = h ? v[1] : v[0];
        const float q1 = h ? v[3] : v[2];
        const float d0 = q0 - zr[ct][2 * m];
        const float d1 = q1 - zr[ct][2 * m + 1];
        part = __builtin_fmaf(d0, d0, part);
        part = __builtin_fmaf(d1, d1, part);
        if (ok) {
          if (NCHW) {
            obase[(int64_t)(2 * m) * zstride] = zr[ct][2 * m] + d0;
            obase[(int64_t)(2 * m + 1) * zstride] = zr[ct][2 * m + 1] + d1;
          } else {
            obase[4 * m + h] = zr[ct][2 * m] + d0;
            obase[4 * m + 2 + h] = zr[ct][2 * m + 1] + d1;
          }
        }
      }
      lp += ok ? part : 0.0f;
    }
  }

  // ---- per-workgroup partials (fixed reduction order: reproducible) ---------------------
  float* red = lds + 2 * STAGE_F;  // 64 floats past the stage buffers
  if (a.q_out != nullptr && a.loss_partial != nullptr) {
    const float ws = wave_sum(lp);
    if (lane == 0) red[wave] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot;
      if (NW == 4) tot = (red[0] + red[1]) + (red[2] + red[3]);
      else tot = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
      a.loss_partial[blockIdx.x] = tot;
      for (int64_t i = (int64_t)blockIdx.x + gridDim.x; i < a.loss_slots; i += gridDim.x) a.loss_partial[i] = 0.0f;
    }
  }
  if (PAIRWISE) {
    double sd = 0.0;
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      sd += psum[ct];
      mn = pmin[ct] < mn ? pmin[ct] : mn;
      mx = pmax[ct] > mx ? pmax[ct] : mx;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      sd += __shfl_xor(sd, off);
      const float o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off);
      mn = o1 < mn ? o1 : mn;
      mx = o2 > mx ? o2 : mx;
    }
    double* redd = reinterpret_cast<double*>(red + 8);   // 8 doubles
    float* rmn = red + 24;                               // 8 floats
    float* rmx = red + 32;                               // 8 floats
    if (lane == 0) { redd[wave] = sd; rmn[wave] = mn; rmx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      float m0 = rmn[0], m1 = rmx[0];
      for (int w = 0; w < NW; ++w) {
        tot += redd[w];
        m0 = rmn[w] < m0 ? rmn[w] : m0;
        m1 = rmx[w] > m1 ? rmx[w] : m1;
      }
      a.pw_sum[blockIdx.x] = tot;
      a.pw_minmax[2 * blockIdx.x] = m0;
      a.pw_minmax[2 * blockIdx.x + 1] = m1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same search for SHORT sweeps (small codebooks), where the loads of a row tile before the
// sweep and the gather + stores after it weigh as much as the sweep itself: a persistent kernel
// whose waves walk over many row tiles and whose two wave groups run half a period apart.
//
//   * One workgroup of 8 waves per CU for the whole launch.  Wave g of the grid takes the 32-row
//     tiles g, g + G, g + 2G, ... (G = waves in the grid).
//   * The workgroup moves in lockstep STEPS, one barrier each; step k has codebook stage
//     k mod nstages in front of every wave.  RESIDENT: the whole image sits in LDS and stays
//     there.  Otherwise the stages cycle through two buffers by LDS-DMA without end, stage
//     after stage, wrapping from the last to the first.
//   * A wave's period is X memory steps + nstages sweep steps.  Memory step m handles the m-th
//     slice of the channels: it writes that slice of the outputs of the tile the wave has just
//     finished (the winning code's values from the image, straight-through values,
//     squared-difference sum), loads the same slice of its next tile into the registers this
//     frees, and fetches the winning code's values for the next slice.  A slice is small enough
//     (a few dozen vector-memory instructions) that the wave issues it without waiting for
//     memory -- a wave can have 63 such instructions outstanding -- so a memory step is short
//     and never holds up the barrier its computing partner is heading for.  Then the wave
//     sweeps: nstages steps starting at WHATEVER stage is in front of it, wrapping around the
//     codebook's end.
//   * Waves 0-3 and waves 4-7 (w and w + 4 share a SIMD) run half a period apart, so while one
//     wave of a SIMD is in its memory steps its partner is mid-sweep and has the matrix pipe to
//     itself: loads, stores and MFMAs overlap instead of taking turns.
//   * A sweep that starts at stage s0 > 0 sees the codes [64 s0, K) before the codes [0, 64 s0).
//     Within each of the two runs the scan visits codes in increasing order, so '<' keeps the
//     lowest index; the winner of the first run is set aside at the wrap and the two are merged
//     with '<=' in favour of the second run, whose indices are all lower.  Same winner, same
//     distance bits as the straight sweep.
//   * LDS-DMA is issued by the waves of the group that is NOT in a memory step (its only
//     outstanding vector-memory operations are then the DMA itself, so the vmcnt(0) before the
//     barrier waits for nothing else); the barrier is a bare s_barrier, memory-step waves pass
//     it with their loads and stores still in flight.
// ---------------------------------------------------------------------------------------
// Memory steps per period.  Image resident: 1 (the winning code's values come from LDS, the step never
// waits for memory).  Stages cycling: 2 -- the values come from the image in global memory / L2 and are
// fetched one step ahead, and half a tile's values are what fits beside the tile in the register file.
// Measured at 1 Mi latents, 256 channels (K = 512 cycling: 4 steps 2.33 ms, 2 steps 2.25 ms; K = 64
// resident: 2 steps 0.54 ms, 1 step 0.50 ms): every instruction of a memory step issues between two of
// the partner's MFMAs, so a step costs the partner little but lasts long, and the fewer steps a wave
// spends outside its sweep, the fewer stages its partner sweeps alone.
// cache policy of the latents' loads and the outputs' stores in NCHW (the aux operand of the buffer instructions;
// bit 1 = non-temporal): they pass through once, and left to the default they push the codebook image out of L2.
// Not for row-major latents: there a wave instruction touches 32 bytes of each of 32 rows, and without L2 to gather
// such pieces into lines the same kernel ran 1.18 instead of 0.70 ms at K = 64, D = 256
#ifndef VQ_CYC_STREAM
#define VQ_CYC_STREAM 2   // measured at K = 8192, D = 256, 4 Mi rows: 20.0 -> 16.5 GB read past L2 per launch; K = 512: 8.94 -> 8.81 ms
#endif
#ifndef VQ_CYC_XR
#define VQ_CYC_XR 1
#endif
#ifndef VQ_CYC_XS
#define VQ_CYC_XS 2
#endif
__host__ __device__ constexpr int cyc_mem_steps(int D, bool resident) {
  return resident ? VQ_CYC_XR : VQ_CYC_XS;
}
constexpr int VQ_CYC_WAVES = 2048;   // most waves a grid of this kernel has (256 CUs x 8): one loss partial each

template <int D>
__device__ __forceinline__ void cyc_stage_copy(const float* image, int stage, float* lds_buf, int li, int nl, int lane) {
  constexpr int PIECES = D / 4 + 1;
  const char* src = reinterpret_cast<const char*>(image) + (size_t)stage * vq_stage_floats(D) * 4 + lane * 16;
  char* dst = reinterpret_cast<char*>(lds_buf);
  for (int p = li; p < PIECES; p += nl) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
  }
}

template <int M> struct CycStep { static constexpr int value = M; };

// Buffer descriptor over memory starting at p: loads and stores through it are addressed as
// "descriptor base (scalar) + scalar byte offset + this lane's 32-bit byte offset", so an access needs
// no vector address arithmetic at all -- vector instructions are what a memory step must not spend,
// they run on the same pipe as the partner wave's fp32 MFMAs.
// `bytes`: what lies between p and the end of the tensor -- an access beyond it reads zero / writes nothing.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cyc_rsrc(const void* p, int64_t bytes) {
  const uint32_t range = bytes > 0xffffffffll ? 0xffffffffu : (bytes < 0 ? 0u : (uint32_t)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)range, 0x00020000);
}

// Row-major latents: lane (j, 0) holds channels c .. c + 3 of its row and lane (j, 1) channels c + 4 .. c + 7
// (one 16-byte access each); the matrix core wants the even channels in the lower lane half and the odd
// ones in the upper.  Two half-wave swaps turn {c, c+1, c+2, c+3 | c+4, ..., c+7} into
// {c, c+2, c+4, c+6 | c+1, c+3, c+5, c+7}; the same two swaps turn it back.
__device__ __forceinline__ void half_interleave(f32x4& v) {
  // swap: the upper half of the first register with the lower half of the second
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
  auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
  // r[0] = (c | c+1), r[1] = (c+4 | c+5), t[0] = (c+2 | c+3), t[1] = (c+6 | c+7)
  v[0] = __uint_as_float(r[0]); v[1] = __uint_as_float(t[0]); v[2] = __uint_as_float(r[1]); v[3] = __uint_as_float(t[1]);
}
__device__ __forceinline__ void half_deinterleave(f32x4& v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
  auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
  // from (c | c+1), (c+4 | c+5): r[0] = (c | c+4), r[1] = (c+1 | c+5); likewise t for c+2, c+3, c+6, c+7
  v[0] = __uint_as_float(r[0]); v[1] = __uint_as_float(r[1]); v[2] = __uint_as_float(t[0]); v[3] = __uint_as_float(t[1]);
}

// Addressing.  NCHW (H*W a multiple of 32, so a tile lies inside one image): 4-byte accesses, lane
// (j, h) at pixel j of channel 2s + h -- 128-byte runs along the pixels.  Rows: the two lanes of a row
// read 16 bytes each (8 consecutive channels between them) and exchange halves so that the even
// channels end up in the lower lane half and the odd ones in the upper.  Either way the tile's base is a
// scalar and the lane's offset from it a constant (cyc_rsrc).
template <int D, bool NCHW, bool TEMP, bool RESIDENT>
__global__ __launch_bounds__(512, 2) void vq_search_cyclic_kernel(SearchArgs a, int64_t ntiles) {
  constexpr int TMODE = TEMP ? VQ_T_DEFERRED : VQ_T_NONE;
  constexpr bool VEC = !NCHW;
  constexpr int S = D / 2;
  constexpr int G = D / 8;      // channel groups: 8 channels = 4 k-steps = one 16-byte piece of a code's image row
  constexpr int X = cyc_mem_steps(D, RESIDENT);
  constexpr int U = G / X;      // channel groups per memory step
  constexpr bool PREFETCH = !RESIDENT;
  constexpr int STAGE_F = VQ_STAGE_CODES * D + 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int gi = wave >> 2;                        // group: SIMD partners are waves w and w + 4
  const int nst = a.nstages;
  const int P = nst + X;                           // a wave's period in steps
  const int off = gi ? P / 2 : 0;                  // group 1 runs half a period behind
  const int64_t GW = (int64_t)gridDim.x * 8;       // waves in the grid
  const int64_t g = (int64_t)blockIdx.x * 8 + wave;
  const int cnt = g < ntiles ? (int)((ntiles - g + GW - 1) / GW) : 0;                    // tiles of this wave
  const int cnt_max = (int)((ntiles - (int64_t)blockIdx.x * 8 + GW - 1) / GW);          // ... of wave 0 of this workgroup
  const int kend = P / 2 + cnt_max * P + X;        // steps: up to group 1's last slice of its last tile

  if (a.control[VQ_CTL_CB_FLAG] != 0) {
    // a non-finite codebook sends every row to the exact kernel
    for (int i = 0; i < cnt; ++i) {
      const int64_t n = (g + (int64_t)i * GW) * 32 + j;
      if (h == 0 && n < a.N) a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
    }
    if (a.loss_partial && lane == 0)
      for (int64_t i = g; i < a.loss_slots; i += GW) a.loss_partial[i] = 0.0f;
    return;
  }

  // ---- prologue: the resident image, or stage 0 of the cycle (issued by group 1, which idles first)
  if (RESIDENT) {
    for (int st = 0; st < nst; ++st) cyc_stage_copy<D>(a.image, st, lds + (size_t)st * STAGE_F, wave, 8, lane);
  } else if (gi == 1) {
    cyc_stage_copy<D>(a.image, 0, lds, wave & 3, 4, lane);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  float zr[S];
  f32x4 gcur[PREFETCH ? U : 1];   // the winning code's values for the slice the next memory step writes
  float z2 = 0.0f;
  float best = 0.0f, best_hi = 0.0f;
  float sec = 0.0f, sec_hi = 0.0f;   // deferred temperature: the value the running minimum displaced last
  int bidx = 0, bidx_hi = 0;
  int win = 0;               // winner of the finished tile (both lane halves hold it)
  bool okf = false;          // ... and whether this lane's row is a real, finite row
  float lp = 0.0f;           // this lane's squared-difference sum over all its tiles
  uint32_t zmax_bits = 0;
  int64_t zo_ld = 0, zo_st = 0;   // element offset of this lane's first channel: tile being loaded / written

  int s = 0;                 // stage in front of the waves: k mod nst
  int p0 = 0;                // k mod P: group 0's phase, which decides who issues the DMA
  int it = 0;                // tile iteration of this wave

  // The winning code's values in this lane's channel order, from the image: piece gq of the code's
  // 32-row tile holds for lane (i, h) the channels 8 gq + h + {0, 2, 4, 6}.
  // (resident: a pointer into LDS; cycling: the lane's byte offset into the image in global memory,
  // read through a buffer descriptor with the piece's offset as the scalar part)
  auto code_off = [&]() __attribute__((always_inline)) -> uint32_t {
    const int wc = okf ? win : 0;
    return (uint32_t)(((wc >> 6) * STAGE_F + ((wc >> 5) & 1) * 32 * D) * 4 + (h * 32 + (wc & 31)) * 16);
  };
  const __amdgpu_buffer_rsrc_t irsrc = cyc_rsrc(a.image, (int64_t)nst * STAGE_F * 4);
  auto code_piece = [&](uint32_t coff, int gq) __attribute__((always_inline)) -> f32x4 {
    if (RESIDENT) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + coff + gq * 1024);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, coff, gq * 1024, 0));
  };

  // this lane's byte offset from the tile's base
  const uint32_t loff = NCHW ? (uint32_t)(h * (uint32_t)a.HW + j) * 4u : (uint32_t)(j * D + 4 * h) * 4u;
  // loads of a slice are issued in one memory step and brought into the matrix core's lane order in the
  // next step (or at the start of the sweep), when they have long landed
  auto finish_loads = [&](auto mtag) __attribute__((always_inline)) {
    constexpr int M = decltype(mtag)::value;
    if (!VEC) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int gq = M * U + u;
      f32x4 v = {zr[4 * gq], zr[4 * gq + 1], zr[4 * gq + 2], zr[4 * gq + 3]};
      half_interleave(v);
#pragma unroll
      for (int q = 0; q < 4; ++q) zr[4 * gq + q] = v[q];
      if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto mem_step = [&](auto mtag) __attribute__((always_inline)) {
    constexpr int M = decltype(mtag)::value;
    const bool have_prev = it >= 1 && it <= cnt && a.q_out != nullptr;
    const bool have_next = it < cnt;
    const uint32_t coff = code_off();
    if (M == 0) {
      if (PREFETCH && have_prev) {
#pragma unroll
        for (int u = 0; u < U; ++u) gcur[u] = code_piece(coff, u);
      }
      zo_st = zo_ld;
      if (it < cnt) {
        const uint32_t tile = (uint32_t)(g + (int64_t)it * GW);
        if (NCHW) {
          const uint32_t tpi = (uint32_t)(a.HW >> 5);           // tiles per image
          const uint32_t img = tile / tpi;
          zo_ld = (int64_t)img * D * a.HW + (int64_t)(tile - img * tpi) * 32;
        } else {
          // (rows past the end of the last tile: their loads fall into the next rows of the buffer or are
          // clipped by the descriptor's range; their outputs are never written)
          zo_ld = (int64_t)tile * 32 * D;
        }
      }
    }
    if (M >= 1 && have_next) finish_loads(CycStep<(M + X - 1) % X>{});
    // bytes between this lane's consecutive values (NCHW) or consecutive 16-byte pieces (rows)
    uint32_t bstride = NCHW ? (uint32_t)(2 * a.HW * 4) : 32u;
    // opaque: keeps the channel offsets from being computed once, outside the step loop, and held in
    // scalar registers for the whole launch
    if (NCHW) asm volatile("" : "+s"(bstride));
    if (have_prev) {
      const int64_t nprev = (g + (int64_t)(it - 1) * GW) * 32 + j;
      const bool valid = nprev < a.N;
      const __amdgpu_buffer_rsrc_t orsrc = cyc_rsrc(a.q_out + zo_st, (a.N * D - zo_st) * 4);
      float part = 0.0f;
      if (VEC) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int gq = M * U + u;
          const f32x4 ev = PREFETCH ? gcur[u] : code_piece(coff, gq);
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d0 = ev[q] - zr[4 * gq + q];
            part = __builtin_fmaf(d0, d0, part);
            o[q] = zr[4 * gq + q] + d0;
          }
          half_deinterleave(o);
          // a row is written whole: rows that are not finite get their values from the exact kernel
          // afterwards; rows past the end are not written
          if (valid) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, loff, gq * bstride, 0);
        }
      } else if (okf) {   // one predicated region: rows past the end and non-finite rows write nothing
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int gq = M * U + u;
          const f32x4 ev = PREFETCH ? gcur[u] : code_piece(coff, gq);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d0 = ev[q] - zr[4 * gq + q];
            part = __builtin_fmaf(d0, d0, part);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, zr[4 * gq + q] + d0), orsrc, loff,
                                                  (4 * gq + q) * bstride, VQ_CYC_STREAM);
          }
        }
      }
      lp += okf ? part : 0.0f;
    }
    if (have_next) {
      const __amdgpu_buffer_rsrc_t zrsrc = cyc_rsrc(a.z + zo_ld, (a.N * D - zo_ld) * 4);
      if (VEC) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int gq = M * U + u;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrsrc, loff, gq * bstride, 0));
#pragma unroll
          for (int q = 0; q < 4; ++q) zr[4 * gq + q] = v[q];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            zr[4 * (M * U + u) + q] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(zrsrc, loff, (4 * (M * U + u) + q) * bstride, VQ_CYC_STREAM));
      }
    }
    if (PREFETCH && M + 1 < X && have_prev) {
#pragma unroll
      for (int u = 0; u < U; ++u) gcur[u] = code_piece(coff, (M + 1) * U + u);
    }
  };

  // Every wave runs the same number of steps (kend), one barrier each.  step_begin issues the DMA
  // of the next stage if this wave's group is the one to do so and names the buffer of the stage
  // in front; step_end is the barrier.
  int k = 0;
  bool issuer = false;
  const float* buf = lds;
  auto dma_issue = [&]() __attribute__((always_inline)) {
    if (!RESIDENT && issuer && k + 1 < kend)
      stage_copy_buf<D, 8>(irsrc, s + 1 == nst ? 0 : s + 1, lds + (size_t)((k + 1) & 1) * STAGE_F, wave & 3, lane);
  };
  // defer_dma: the caller issues the DMA itself, later in the step (a sweeping wave does so between its
  // two halves: at the start of a step the memory-step waves send their loads and stores, and a DMA
  // instruction queued behind that burst would hold up the MFMAs behind it in this wave's stream)
  auto step_begin = [&](bool defer_dma) __attribute__((always_inline)) {
    if (!RESIDENT) {
      issuer = gi == (p0 < X ? 1 : 0);
      buf = lds + (size_t)(k & 1) * STAGE_F;
      if (!defer_dma) dma_issue();
    } else {
      buf = lds + (size_t)s * STAGE_F;
    }
  };
  auto step_end = [&]() __attribute__((always_inline)) {
    if (!RESIDENT && issuer) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA of the next stage has landed
    __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0): this wave is done reading buf
    __builtin_amdgcn_s_barrier();
    ++k;
    if (++s == nst) s = 0;
    if (++p0 == P) p0 = 0;
  };

  for (int i = 0; i < off; ++i) { step_begin(false); step_end(); }   // group 1: half a period behind

  for (it = 0; it <= cnt_max; ++it) {
    // ================= memory steps: slice by slice, the outputs of tile it-1 and the loads of tile it
    // A memory step is a few hundred vector and scalar instructions beside the partner's MFMAs.  At equal
    // priority the arbiter keeps offering the slot to the wave whose next MFMA is waiting for the pipe,
    // and this wave's instructions trickle out at about one per MFMA; with priority they issue back to
    // back, and the partner still gets the one slot per 64 cycles its MFMAs need.
    __builtin_amdgcn_s_setprio(3);
    step_begin(false); if (it <= cnt) mem_step(CycStep<0>{}); step_end();
    if (X >= 2) { step_begin(false); if (it <= cnt) mem_step(CycStep<1 % X>{}); step_end(); }
    if (X >= 4) {
      step_begin(false); if (it <= cnt) mem_step(CycStep<2 % X>{}); step_end();
      step_begin(false); if (it <= cnt) mem_step(CycStep<3 % X>{}); step_end();
    }
    __builtin_amdgcn_s_setprio(0);
    if (it == cnt_max) break;

    // ================= sweep steps: the stage in front against this wave's tile ================
    for (int c = 0; c < nst; ++c) {
      step_begin(it < cnt);
      if (it < cnt) {
      if (c == 0) {
        finish_loads(CycStep<X - 1>{});   // the last slice of the tile's loads
        float p = 0.0f;
        uint32_t am = 0;
#pragma unroll
        for (int sI = 0; sI < S; ++sI) {
          p = __builtin_fmaf(zr[sI], zr[sI], p);
          const uint32_t ab = __float_as_uint(zr[sI]) & 0x7fffffffu;
          am = ab > am ? ab : am;
        }
        zmax_bits = am > zmax_bits ? am : zmax_bits;
        z2 = p + __shfl_xor(p, 32);  // even-channel chain + odd-channel chain
        best = __builtin_inff(); bidx = 0;
        best_hi = __builtin_inff(); bidx_hi = 0;
        sec = __builtin_inff(); sec_hi = __builtin_inff();
      } else if (s == 0) {
        // wrapped around the end of the codebook: what was seen so far has the higher indices
        best_hi = best; bidx_hi = bidx; sec_hi = sec;
        best = __builtin_inff(); bidx = 0; sec = __builtin_inff();
      }
      f32x16 cA, cB;
      const f32x4* A = reinterpret_cast<const f32x4*>(buf) + lane;
      const f32x4* e2s = reinterpret_cast<const f32x4*>(buf + VQ_STAGE_CODES * D) + h;
      double pd = 0.0; float pf0 = 0.0f, pf1 = 0.0f;   // unused outputs of scan_tile

      // first half: codes [64 s, 64 s + 32) into set A
#pragma unroll
      for (int i = 0; i < 16; ++i) cA[i] = 0.0f;
      {
        f32x4 af = A[0];
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          const f32x4 an = A[(gq + 1 < G ? gq + 1 : gq) * 64];
#pragma unroll
          for (int q = 0; q < 4; ++q) cA = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], zr[4 * gq + q], cA, 0, 0, 0);
          af = an;
        }
      }
      dma_issue();   // (deferred from the start of the step)
      // second half: codes [64 s + 32, 64 s + 64) into set B; the scan of set A falls under its MFMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) cB[i] = 0.0f;
      {
        f32x4 af = A[G * 64];
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          const f32x4 an = A[(G + (gq + 1 < G ? gq + 1 : gq)) * 64];
#pragma unroll
          for (int q = 0; q < 4; ++q) cB = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], zr[4 * gq + q], cB, 0, 0, 0);
          af = an;
        }
      }
      {
        f32x4 e2A[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e2A[q] = e2s[2 * q];
        scan_tile<TMODE, false>(cA, e2A, z2, s * VQ_STAGE_CODES, a.T, -1, best, bidx, sec, pd, pf0, pf1);
      }
      {
        f32x4 e2B[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e2B[q] = e2s[8 + 2 * q];
        scan_tile<TMODE, false>(cB, e2B, z2, s * VQ_STAGE_CODES + 32, a.T, -1, best, bidx, sec, pd, pf0, pf1);
      }

      if (c == nst - 1) {
        // ---- the tile is swept: merge of the two runs and of the two lane halves ----------------
        if (TEMP) sec = vq_t_merge_sec(best, sec, best_hi, sec_hi);
        if (!(best <= best_hi)) { best = best_hi; bidx = bidx_hi; }
        int bi = bidx + 4 * h;
        float b = best;
        const float ob = __shfl_xor(b, 32);
        const int oi = __shfl_xor(bi, 32);
        const bool take = (ob < b) || (ob == b && oi < bi);
        bool near = false;
        if (TEMP) {
          const float sc = vq_t_merge_sec(b, sec, ob, __shfl_xor(sec, 32));
          const float m = take ? ob : b;
          const float qd = m / a.T;
          near = vq_t_near(m, sc, a.T, qd);
          b = qd;   // the winning distance as the reference reports it
        } else {
          b = take ? ob : b;
        }
        bi = take ? oi : bi;
        const int64_t n = (g + (int64_t)it * GW) * 32 + j;
        const bool valid = n < a.N;
        okf = valid && z2 <= VQ_NORM_LIMIT && !near;   // false for NaN
        win = bi;
        if (h == 0) {
          if (okf) {
            if (a.idx64) a.idx64[n] = (int64_t)bi + a.code_base;
            if (a.idx32) a.idx32[n] = (int32_t)(bi + a.code_base);
            if (a.dmin) a.dmin[n] = b;
          } else if (valid) {
            if (near && a.near_ok && z2 <= VQ_NORM_LIMIT) {
              a.nf_rows[a.N - 1 - atomicAdd(&a.control[VQ_CTL_NEAR_COUNT], 1)] = (int32_t)n;
              a.idx32[n] = 0;                   // the exact kernel's workgroups count themselves in here
              if (a.idx64) a.idx64[n] = -1;     // ... and keep the smallest (quotient, index) key here
            } else {
              a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
            }
          }
        }
      }
      }
      step_end();
    }
  }
  while (k < kend) { step_begin(false); step_end(); }   // group 0 keeps the barriers company to the end

  // ---- one loss partial per wave (the tiles of a wave in order, then a fixed tree: reproducible) ----
  if (a.q_out != nullptr && a.loss_partial != nullptr) {
    const float ws = wave_sum(lp);
    if (lane == 0) {
      a.loss_partial[g] = ws;
      for (int64_t i = g + GW; i < a.loss_slots; i += GW) a.loss_partial[i] = 0.0f;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t ov = (uint32_t)__shfl_xor((int)zmax_bits, o);
    zmax_bits = ov > zmax_bits ? ov : zmax_bits;
  }
  if (lane == 0 && zmax_bits) atomicMax(reinterpret_cast<unsigned int*>(&a.control[VQ_CTL_ZMAX]), zmax_bits);
}

// ---------------------------------------------------------------------------------------
// 1x1 convolution of the latents' rows on the same engine: out[n][o] = sum_c W[o][c] x[n][c] + b[o]
// (vqvae/custom_vqgan.py:50-51, applied at :68 and :72).  The weights take the codebook's place --
// their image streams through LDS, the slot of a stage's norms carries the bias -- the 32-row tile
// of x stays in registers, and where the search scans a finished accumulator tile this kernel
// stores it in x's layout.  An output is one fmaf chain over the input channels in increasing
// order, started from the bias.
// ---------------------------------------------------------------------------------------
struct ConvArgs {
  const float* x;
  const float* image;   // weights in image order, bias in the norms slot
  float* out;
  int64_t N;
  int64_t HW;
  int Cout;
  int nstages;
};

template <bool NCHW>
__device__ __forceinline__ void conv_store_tile(const f32x16& c, int code0, int h, int Cout, bool valid, float* obase,
                                                int64_t cstride) {
  if (!valid) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int d0 = code0 + 8 * q + 4 * h;
    if (!NCHW && d0 + 3 < Cout && (Cout & 3) == 0) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = c[4 * q + r];
      *reinterpret_cast<f32x4*>(obase + d0) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (d0 + r < Cout) obase[(int64_t)(d0 + r) * cstride] = c[4 * q + r];
    }
  }
}

// The same store when every output channel of the tile exists (C_out a multiple of 64): no
// predicate anywhere, so the number of store instructions per tile is fixed (16 in NCHW, 4 in rows).
template <bool NCHW>
__device__ __forceinline__ void conv_store_full(const f32x16& c, int code0, int h, float* obase, int64_t cstride) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int d0 = code0 + 8 * q + 4 * h;
    if (!NCHW) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = c[4 * q + r];
      *reinterpret_cast<f32x4*>(obase + d0) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) obase[(int64_t)(d0 + r) * cstride] = c[4 * q + r];
    }
  }
}

// FULL: every output channel of every tile exists (C_out a multiple of 64).  The stores are then
// unconditional, a wave issues a fixed number of them behind each stage's DMA, and the stage barrier
// waits with a counted s_waitcnt for the DMA alone instead of draining the stores too.
template <int D, bool NCHW, bool FULL>
__global__ __launch_bounds__(512, 2) void vq_conv1x1_kernel(ConvArgs a) {
  constexpr int NW = 8;
  constexpr int S = D / 2;
  constexpr int G = D / 8;
  constexpr int STAGE_F = VQ_STAGE_CODES * D + 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * VQ_WG_ROWS + wave * 32;

  stage_copy<D, NW>(a.image, 0, lds, wave, lane);

  // x tile into registers in B-operand order: xr[s] = x[row][2s + h]
  float xr[S];
  int64_t n = row0 + j;
  const bool valid = n < a.N;
  if (!valid) n = a.N - 1;
  float* obase;
  int64_t cstride;
  if (NCHW) {
    const int64_t b = n / a.HW, hw = n - b * a.HW;
    const float* src = a.x + (b * D + h) * a.HW + hw;
    const int64_t xstride = 2 * a.HW;
#pragma unroll
    for (int s = 0; s < S; ++s) xr[s] = src[(int64_t)s * xstride];
    obase = a.out + b * a.Cout * a.HW + hw;
    cstride = a.HW;
  } else {
    // (dword loads: a 16-byte load would hold four registers to leave two, and the tile fills the file)
    const float* src = a.x + n * D + h;
#pragma unroll
    for (int m = 0; m < D / 4; ++m) {
      xr[2 * m] = src[4 * m];
      xr[2 * m + 1] = src[4 * m + 2];
    }
    obase = a.out + n * a.Cout;
    cstride = 1;
  }

  f32x16 cA, cB;
#pragma unroll
  for (int i = 0; i < 16; ++i) cB[i] = 0.0f;

  __syncthreads();

  int cur = 0;
  for (int st = 0; st < a.nstages; ++st) {
    float* buf = lds + cur * STAGE_F;
    if (st + 1 < a.nstages) stage_copy<D, NW>(a.image, st + 1, lds + (cur ^ 1) * STAGE_F, wave, lane);
    const f32x4* A = reinterpret_cast<const f32x4*>(buf) + lane;
    const f32x4* bs = reinterpret_cast<const f32x4*>(buf + VQ_STAGE_CODES * D) + h;

    // first half: output channels [64 st, 64 st + 32) into set A, starting from their bias;
    // store set B of stage st-1
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = bs[2 * q];
#pragma unroll
      for (int r = 0; r < 4; ++r) cA[4 * q + r] = b4[r];
    }
    {
      f32x4 af = A[0];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 an = A[(g + 1 < G ? g + 1 : g) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) cA = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], xr[4 * g + q], cA, 0, 0, 0);
        af = an;
      }
    }
    if (st > 0) {
      if (FULL) conv_store_full<NCHW>(cB, (st - 1) * VQ_STAGE_CODES + 32, h, obase, cstride);
      else conv_store_tile<NCHW>(cB, (st - 1) * VQ_STAGE_CODES + 32, h, a.Cout, valid, obase, cstride);
    }

    // second half: output channels [64 st + 32, 64 st + 64) into set B; store set A
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = bs[8 + 2 * q];
#pragma unroll
      for (int r = 0; r < 4; ++r) cB[4 * q + r] = b4[r];
    }
    {
      f32x4 af = A[G * 64];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 an = A[(G + (g + 1 < G ? g + 1 : g)) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) cB = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], xr[4 * g + q], cB, 0, 0, 0);
        af = an;
      }
    }
    if (FULL) conv_store_full<NCHW>(cA, st * VQ_STAGE_CODES, h, obase, cstride);
    else conv_store_tile<NCHW>(cA, st * VQ_STAGE_CODES, h, a.Cout, valid, obase, cstride);

    if (FULL) {
      // The barrier must see this wave's DMA of stage st+1 landed, not its stores: vector memory
      // operations retire in issue order, and exactly SPT (first stage) or twice that many stores
      // were issued after the DMA, so waiting for all but those is enough.  (Rows beyond N are clamped
      // duplicates that rewrite the last row's values, so the count does not depend on `valid`.)
      constexpr int SPT = NCHW ? 16 : 4;
      if (st == 0) __builtin_amdgcn_s_waitcnt(((SPT & 15) | ((SPT >> 4) << 14)) | 0x0F70);
      else __builtin_amdgcn_s_waitcnt((((2 * SPT) & 15) | (((2 * SPT) >> 4) << 14)) | 0x0F70);
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    cur ^= 1;
  }
  if (FULL) conv_store_full<NCHW>(cB, (a.nstages - 1) * VQ_STAGE_CODES + 32, h, obase, cstride);
  else conv_store_tile<NCHW>(cB, (a.nstages - 1) * VQ_STAGE_CODES + 32, h, a.Cout, valid, obase, cstride);
}

// ---------------------------------------------------------------------------------------
// The same convolution, weights stationary: for C_out a multiple of 256.
//
// A persistent workgroup of 8 waves keeps a 256 x D block of W in registers for the whole launch --
// wave w the A fragments of output channels [32 w, 32 w + 32), D/2 registers per lane, read once
// from the image -- and the rows stream past it: tiles of 32 rows (32 x D floats) go through a ring
// of four LDS slots, copied by LDS-DMA three tiles ahead, every wave multiplies its 32 channels into
// each tile and stores the finished 32 x 32 block (the SIMD's other wave keeps the pipe busy meanwhile).  Loads and stores
// are then two even streams instead of a burst of row loads before each sweep and a burst of stores
// after it.  The chain of an output is unchanged (input channels in increasing order, started from
// the bias), so the result is the same bits as the kernels above.
//   NCHW: a slot is [channel][32 pixels]; lane (j, h) reads pixel j of channel 2s + h with
//         ds_read_b32, the 32 lanes of a half on 32 different banks.
//   rows: a slot is [row][D], channel c of row r at position c ^ r (r < 32) so that the 32 rows a
//         half reads for one k-step lie on 32 different banks; the DMA writes LDS linearly, so the
//         permutation is applied on the source side, one dword per lane.
// A wave's stage wait is counted: behind the DMA of the tile it is about to read lie exactly three
// tiles' stores and two tiles' DMA instructions.
// ---------------------------------------------------------------------------------------
template <int D, bool NCHW>
struct WsPlan {
  static constexpr int RING = 4;
  static constexpr int AHEAD = 3;
  static constexpr int TILE_F = 32 * D;
  static constexpr int DMA_PER_WAVE = NCHW ? D / 64 : D / 16;   // 1 KiB (16 B per lane) / 256 B (4 B per lane) pieces
  static constexpr int SPT = NCHW ? 16 : 4;                     // store instructions per finished tile
  static constexpr int BEHIND = 3 * SPT + 2 * DMA_PER_WAVE;
};

template <int D, bool NCHW>
__device__ __forceinline__ void ws_tile_copy(const ConvArgs& a, int64_t tile, float* slot, int wave, int lane) {
  using P = WsPlan<D, NCHW>;
  char* dst = reinterpret_cast<char*>(slot);
  if (NCHW) {
    const int64_t row0 = tile * 32;
    const int64_t img = row0 / a.HW, hw0 = row0 - img * a.HW;
    const float* base = a.x + img * D * a.HW + hw0 + 4 * (lane & 7);
#pragma unroll
    for (int k = 0; k < P::DMA_PER_WAVE; ++k) {
      const int p = wave + 8 * k;                 // channels [8p, 8p + 8)
      const float* g = base + (int64_t)(8 * p + (lane >> 3)) * a.HW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  } else {
    constexpr int SEG = D / 64;                   // 64-position pieces per row
#pragma unroll
    for (int k = 0; k < P::DMA_PER_WAVE; ++k) {
      const int p = wave + 8 * k;
      const int r = p / SEG, q = p % SEG;
      int64_t n = tile * 32 + r;
      n = n < a.N ? n : a.N - 1;
      const float* g = a.x + n * D + ((64 * q + lane) ^ r);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + (r * D + 64 * q) * 4), 4, 0, 0);
    }
  }
}

// One tile: acc = bias (from LDS), then the D/2 k-steps against the slot in chunks of eight, the B
// values of the next chunk in flight under the MFMAs of this one.  Nothing moves across a chunk's
// end, which bounds the registers the LDS reads may occupy.
template <int D, bool NCHW>
__device__ __forceinline__ void ws_tile_mfma(f32x16& acc, const f32x4* bias4, const float (&wr)[D / 2],
                                             const float* slot, int j, int h) {
  constexpr int CH = D / 16;                 // chunks of 8 k-steps
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = bias4[2 * q];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[4 * q + r] = v[r];
  }
  const char* sb = reinterpret_cast<const char*>(slot);
  // NCHW: pixel j of channel 2s + h.  rows: channel 2s + h of row j sits at position (2s + h) ^ j;
  // the low five bits carry the permutation, the rest of s is an immediate.
  const char* nb = sb + (NCHW ? (h * 32 + j) * 4 : j * D * 4);
  float bc[8], bn[8];
  auto load = [&](float (&b)[8], int c) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int s = 8 * c + q;
      if (NCHW) b[q] = *reinterpret_cast<const float*>(nb + s * 256);
      else b[q] = *reinterpret_cast<const float*>(nb + (((2 * (s & 15) + h) ^ j) * 4) + (s >> 4) * 128);
    }
  };
  load(bc, 0);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (c + 1 < CH) load(bn, c + 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[8 * c + q], bc[q], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) bc[q] = bn[q];
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int D, bool NCHW>
__global__ __launch_bounds__(512, 1) void vq_conv1x1_ws_kernel(ConvArgs a, int64_t ntiles) {
  using P = WsPlan<D, NCHW>;
  constexpr int G = D / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_bias = lds + P::RING * P::TILE_F;      // 256 floats

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int otile = 8 * blockIdx.y + wave;          // 32 output channels
  const int64_t first = blockIdx.x, step = gridDim.x;
  const int64_t cnt = (ntiles - first + step - 1) / step;    // tiles of this workgroup (>= 1: grid.x <= ntiles)

  // the first tiles are on their way while the weights load
#pragma unroll
  for (int i = 0; i < P::AHEAD; ++i)
    ws_tile_copy<D, NCHW>(a, first + (i < cnt ? i : cnt - 1) * step, lds + i * P::TILE_F, wave, lane);

  // weights: piece g of the tile holds, for this lane, the A values of k-steps 4g .. 4g + 3
  float wr[D / 2];
  {
    const float* st = a.image + (size_t)(otile >> 1) * vq_stage_floats(D);
    const f32x4* wp = reinterpret_cast<const f32x4*>(st) + (size_t)(otile & 1) * G * 64 + lane;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 v = wp[g * 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) wr[4 * g + q] = v[q];
    }
    if (lane < 32) lds_bias[32 * wave + lane] = st[VQ_STAGE_CODES * D + 32 * (otile & 1) + lane];
  }
  const f32x4* bias4 = reinterpret_cast<const f32x4*>(lds_bias + 32 * wave) + h;   // [2q]: rows 8q + 4h ..

  f32x16 acc;
  const int64_t cstride = NCHW ? a.HW : 1;

  for (int64_t ii = 0; ii < cnt; ++ii) {
    // this wave's share of tile ii has landed once all but the operations issued behind it are done:
    // the stores of three tiles and the DMA instructions of two
    if (ii >= 3) __builtin_amdgcn_s_waitcnt(((P::BEHIND & 15) | ((P::BEHIND >> 4) << 14)) | 0x0F70);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    {
      const int64_t nx = ii + P::AHEAD;
      ws_tile_copy<D, NCHW>(a, first + (nx < cnt ? nx : cnt - 1) * step, lds + (int)(nx & 3) * P::TILE_F, wave, lane);
    }
    ws_tile_mfma<D, NCHW>(acc, bias4, wr, lds + (int)(ii & 3) * P::TILE_F, j, h);
    const int64_t tile = first + ii * step;
    float* ob;
    if (NCHW) {
      const int64_t row0 = tile * 32;
      const int64_t img = row0 / a.HW, hw0 = row0 - img * a.HW;
      ob = a.out + img * a.Cout * a.HW + hw0 + j;
    } else {
      int64_t n = tile * 32 + j;
      n = n < a.N ? n : a.N - 1;
      ob = a.out + n * a.Cout;
    }
    conv_store_full<NCHW>(acc, 32 * otile, h, ob, cstride);
  }
}

template <int D>
int launch_conv_ws(const ConvArgs& a, bool nchw, hipStream_t s) {
  constexpr int lds_bytes = 4 * 32 * D * 4 + 1024;
  static int cus[64] = {};
  int dev = 0;
  const int rc = nchw ? vq_lds_ready<vq_conv1x1_ws_kernel<D, true>, lds_bytes>(&dev)
                      : vq_lds_ready<vq_conv1x1_ws_kernel<D, false>, lds_bytes>(&dev);
  if (rc != VQ_OK) return rc;
  if (!cus[dev & 63]) VQ_CHECK_HIP(hipDeviceGetAttribute(&cus[dev & 63], hipDeviceAttributeMultiprocessorCount, dev));
  const int64_t ntiles = (a.N + 31) / 32;
  const int gy = a.Cout / 256;
  int64_t gx = cus[dev & 63] / gy;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  const dim3 grid((unsigned)gx, (unsigned)gy), block(512);
  if (nchw) hipLaunchKernelGGL((vq_conv1x1_ws_kernel<D, true>), grid, block, lds_bytes, s, a, ntiles);
  else hipLaunchKernelGGL((vq_conv1x1_ws_kernel<D, false>), grid, block, lds_bytes, s, a, ntiles);
  return vq_hip_status(hipGetLastError());
}

template <bool NCHW, bool FULL> struct ConvTag { static constexpr bool nchw = NCHW, full = FULL; };

template <int D>
int launch_conv(const ConvArgs& a, bool nchw, hipStream_t s) {
  // whole 256-channel blocks of outputs, whole 32-row tiles per image, enough tiles for the ring to pay
  if constexpr (D >= 64) {
    if (a.Cout % 256 == 0 && (!nchw || a.HW % 32 == 0) && a.N >= 32 * 1024) return launch_conv_ws<D>(a, nchw, s);
  }
  constexpr int lds_bytes = (int)(2 * vq_stage_floats(D) * 4 + 256);
  const bool full = a.Cout % VQ_STAGE_CODES == 0;
  const int64_t blocks = (a.N + VQ_WG_ROWS - 1) / VQ_WG_ROWS;
  const dim3 grid((unsigned)blocks), block(512);
  auto go = [&](auto kern_tag) {
    constexpr bool N_ = decltype(kern_tag)::nchw, F_ = decltype(kern_tag)::full;
    const int arc = vq_lds_ready<vq_conv1x1_kernel<D, N_, F_>, lds_bytes>();
    if (arc != VQ_OK) return arc;
    hipLaunchKernelGGL((vq_conv1x1_kernel<D, N_, F_>), grid, block, lds_bytes, s, a);
    return vq_hip_status(hipGetLastError());
  };
  if (full && nchw) return go(ConvTag<true, true>{});
  if (full) return go(ConvTag<false, true>{});
  if (nchw) return go(ConvTag<true, false>{});
  return go(ConvTag<false, false>{});
}

// ---------------------------------------------------------------------------------------
// Exact scalar search: any D, torch.argmin's NaN rule, literal temperature division.
// One workgroup handles 8 rows at a time; thread t scans codes t, t+256, ...; the dot
// product is the same fmaf chain the MFMA computes, so both kernels agree bit for bit on
// rows both can handle.  Used for rows the MFMA kernel flags (non-finite input), for a
// flagged codebook, for degenerate temperatures and for shapes outside the MFMA kernel.
// ---------------------------------------------------------------------------------------
constexpr int XR = 8;  // rows per pass

__device__ __forceinline__ float z_load(const float* z, bool nchw, int64_t n, int c, int D, int64_t HW) {
  if (!nchw) return z[n * D + c];
  const int64_t b = n / HW, hw = n - b * HW;
  return z[(b * D + c) * HW + hw];
}

__device__ __forceinline__ float search_one_row(const SearchArgs& a, int D, int nchw, int64_t n, float* zs, float* rb,
                                                int* ri);
__device__ __forceinline__ float row_load(const SearchArgs& a, int D, int nchw, int64_t n, float* zs, float* rb);
__device__ __forceinline__ void row_scan(const SearchArgs& a, int D, float z2, int klo, int khi, const float* zs, float* rb,
                                         int* ri);
__device__ __forceinline__ float row_finish(const SearchArgs& a, int D, int nchw, int64_t n, int w, float q,
                                            const float* zs, float* rb);

// (quotient, index) as one 64-bit key whose unsigned order is "smaller quotient first, then lower index"
__device__ __forceinline__ unsigned long long vq_row_key(float q, int idx) {
  uint32_t u = __float_as_uint(q);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((unsigned long long)u << 32) | (uint32_t)idx;
}
__device__ __forceinline__ float vq_row_key_quotient(unsigned long long key) {
  uint32_t u = (uint32_t)(key >> 32);
  u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
  return __uint_as_float(u);
}

__device__ __forceinline__ float search_listed_rows(const SearchArgs& a, int D, int nchw, const int32_t* list, int list_step,
                                                    int count, int ticket_base, float* row_loss_slots, float* zs, float* rb,
                                                    int* ri);

__global__ __launch_bounds__(256) void vq_search_exact_kernel(SearchArgs a, int D, int nchw, int use_list,
                                                              float* loss_partial_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* zs = lds;                      // [XR][D]
  float* z2s = zs + (size_t)XR * D;     // [XR]
  float* rbest = z2s + XR;              // [XR][256]
  int* ridx = reinterpret_cast<int*>(rbest + XR * 256);  // [XR][256]
  const int t = threadIdx.x;
  const int64_t total = use_list ? (int64_t)a.control[VQ_CTL_NF_COUNT] : a.N;
  float block_loss = 0.0f;  // meaningful in thread 0

  for (int64_t base = (int64_t)blockIdx.x * XR; base < total; base += (int64_t)gridDim.x * XR) {
    const int nr = (int)((total - base) < XR ? (total - base) : XR);
    __syncthreads();
    for (int i = t; i < XR * D; i += 256) {
      const int r = i / D, c = i - r * D;
      float v = 0.0f;
      if (r < nr) {
        const int64_t n = use_list ? (int64_t)a.nf_rows[base + r] : base + r;
        v = z_load(a.z, nchw != 0, n, c, D, a.HW);
      }
      zs[i] = v;
    }
    __syncthreads();
    if (t < 2 * XR) {  // squared norms: even-channel chain and odd-channel chain
      const int r = t >> 1, hh = t & 1;
      float p = 0.0f;
      for (int c = hh; c < D; c += 2) p = __builtin_fmaf(zs[r * D + c], zs[r * D + c], p);
      const float o = __shfl_xor(p, 1);
      if (hh == 0) z2s[r] = p + o;
    }
    __syncthreads();

    float best[XR];
    int bi[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) { best[r] = 0.0f; bi[r] = -1; }
    for (int k = t; k < a.K; k += 256) {
      const float* e = a.E + (size_t)k * D;
      float dot[XR];
#pragma unroll
      for (int r = 0; r < XR; ++r) dot[r] = 0.0f;
      for (int c = 0; c < D; ++c) {
        const float ev = e[c];
#pragma unroll
        for (int r = 0; r < XR; ++r) dot[r] = __builtin_fmaf(zs[r * D + c], ev, dot[r]);
      }
      const float ek = a.e2[k];
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const float tt = z2s[r] + ek;
        float d = __builtin_fmaf(-2.0f, dot[r], tt);
        if (a.T != 1.0f) d = d / a.T;
        if (bi[r] < 0 || vq_argmin_takes(d, best[r])) { best[r] = d; bi[r] = k; }
      }
    }
#pragma unroll
    for (int r = 0; r < XR; ++r) { rbest[r * 256 + t] = best[r]; ridx[r * 256 + t] = bi[r]; }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if (t < off) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
          const float d2 = rbest[r * 256 + t + off];
          const int i2 = ridx[r * 256 + t + off];
          const float d1 = rbest[r * 256 + t];
          const int i1 = ridx[r * 256 + t];
          if (i2 >= 0 && (i1 < 0 || vq_merge_takes(d2, i2, d1, i1))) { rbest[r * 256 + t] = d2; ridx[r * 256 + t] = i2; }
        }
      }
      __syncthreads();
    }
    // winners: indices, distances, gather + straight-through + loss
    float part = 0.0f;
    for (int r = 0; r < nr; ++r) {
      const int64_t n = use_list ? (int64_t)a.nf_rows[base + r] : base + r;
      int w = ridx[r * 256];
      if (w < 0) w = 0;
      if (t == 0) {
        if (a.idx64) a.idx64[n] = (int64_t)w + a.code_base;
        if (a.idx32) a.idx32[n] = (int32_t)(w + a.code_base);
        if (a.dmin) a.dmin[n] = rbest[r * 256];
      }
      if (a.q_out) {
        const float* e = a.E + (size_t)w * D;
        for (int c = t; c < D; c += 256) {
          const float zv = zs[r * D + c];
          const float diff = e[c] - zv;
          part = __builtin_fmaf(diff, diff, part);
          int64_t off;
          if (nchw) { const int64_t b = n / a.HW, hw = n - b * a.HW; off = (b * D + c) * a.HW + hw; }
          else off = n * D + c;
          a.q_out[off] = zv + diff;
        }
      }
    }
    if (a.q_out) {
      __syncthreads();
      rbest[t] = part;
      __syncthreads();
      for (int off = 128; off >= 1; off >>= 1) {
        if (t < off) rbest[t] += rbest[t + off];
        __syncthreads();
      }
      if (t == 0) block_loss += rbest[0];
    }
  }
  if (use_list) {
    // the second list (from the end of nf_rows backwards): finite rows of a deferred-temperature sweep, one per workgroup
    const int near = a.control[VQ_CTL_NEAR_COUNT];
    block_loss += search_listed_rows(a, D, nchw, a.nf_rows + (a.N - 1), -1, near, 0, a.near_loss, zs, rbest, ridx);
  }
  if (t == 0 && loss_partial_out) loss_partial_out[blockIdx.x] = block_loss;
}

// ---------------------------------------------------------------------------------------
// One workgroup, one row: exact fp32 distances to every code (finite rows only, D % 4 == 0,
// D <= 1024), literal temperature division.  Serves short lists of rows -- those the filtered search
// cannot settle with its candidate lists, those the deferred-temperature sweep found a code close
// above the minimum for; the cost is ~K*D fma per row on the vector ALUs.
// Thread t takes codes t, t + 256, ... four at a time; same fma chain as everywhere else.
// zs: D floats (16-byte aligned), rb / ri: 256 entries each, all in LDS.  Returns (in thread 0) the
// row's squared-difference sum when a.q_out is set.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float row_load(const SearchArgs& a, int D, int nchw, int64_t n, float* zs, float* rb) {
  const int t = threadIdx.x;
  __syncthreads();
  for (int c = t; c < D; c += 256) zs[c] = z_load(a.z, nchw != 0, n, c, D, a.HW);
  __syncthreads();
  if (t < 2) {
    float p = 0.0f;
    for (int c = t; c < D; c += 2) p = __builtin_fmaf(zs[c], zs[c], p);
    const float o = __shfl_xor(p, 1);
    if (t == 0) rb[0] = p + o;
  }
  __syncthreads();
  const float z2 = rb[0];
  __syncthreads();
  return z2;
}

// codes klo .. khi - 1 (klo a multiple of 4); afterwards rb[0] / ri[0] hold the smallest quotient and its lowest
// index (0x7fffffff: the range was empty)
__device__ __forceinline__ void row_scan(const SearchArgs& a, int D, float z2, int klo, int khi, const float* zs, float* rb,
                                         int* ri) {
  const int t = threadIdx.x;
  float best = __builtin_inff();
  int bi = 0x7fffffff;
  for (int k0 = klo + t; k0 < khi; k0 += 1024) {
    int kk[4];
    const f32x4* er[4];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      kk[q] = k0 + 256 * q;
      er[q] = reinterpret_cast<const f32x4*>(a.E + (size_t)(kk[q] < khi ? kk[q] : k0) * D);
    }
#pragma unroll 4
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const f32x4 zv = reinterpret_cast<const f32x4*>(zs)[c4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 e = er[q][c4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[q] = __builtin_fmaf(zv[m], e[m], acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (kk[q] < khi) {
        float d = __builtin_fmaf(-2.0f, acc[q], z2 + a.e2[kk[q]]);
        if (a.T != 1.0f) d = d / a.T;
        if (d < best) { best = d; bi = kk[q]; }  // codes visited in increasing order: '<' keeps the lowest index
      }
    }
  }
  rb[t] = best;
  ri[t] = bi;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) {
      const float d2 = rb[t + off];
      const int i2 = ri[t + off];
      if (d2 < rb[t] || (d2 == rb[t] && i2 < ri[t])) { rb[t] = d2; ri[t] = i2; }
    }
    __syncthreads();
  }
}

// the row's outputs for winner w at quotient q; returns its squared-difference sum when a.q_out is set
__device__ __forceinline__ float row_finish(const SearchArgs& a, int D, int nchw, int64_t n, int w, float q,
                                            const float* zs, float* rb) {
  const int t = threadIdx.x;
  if (t == 0) {
    if (a.idx64) a.idx64[n] = (int64_t)w + a.code_base;
    if (a.idx32) a.idx32[n] = (int32_t)(w + a.code_base);
    if (a.dmin) a.dmin[n] = q;
  }
  float row_loss = 0.0f;
  if (a.q_out) {
    const float* e = a.E + (size_t)w * D;
    float part = 0.0f;
    for (int c = t; c < D; c += 256) {
      const float zv = zs[c];
      const float diff = e[c] - zv;
      part = __builtin_fmaf(diff, diff, part);
      int64_t off;
      if (nchw) { const int64_t b = n / a.HW, hw = n - b * a.HW; off = (b * D + c) * a.HW + hw; }
      else off = n * D + c;
      a.q_out[off] = zv + diff;
    }
    __syncthreads();
    rb[t] = part;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if (t < off) rb[t] += rb[t + off];
      __syncthreads();
    }
    row_loss = rb[0];
  }
  return row_loss;
}

__device__ __forceinline__ float search_one_row(const SearchArgs& a, int D, int nchw, int64_t n, float* zs, float* rb,
                                                int* ri) {
  const float z2 = row_load(a, D, nchw, n, zs, rb);
  row_scan(a, D, z2, 0, a.K, zs, rb, ri);
  const int w = ri[0] == 0x7fffffff ? 0 : ri[0];
  const float q = rb[0];
  __syncthreads();
  return row_finish(a, D, nchw, n, w, q, zs, rb);
}

// A list of rows (entry i at list[i * list_step]) through search_one_row.  A short list is shared out S <= 8
// workgroups to a row, each taking a range of the codes -- a row streams the whole codebook through one workgroup
// otherwise, a few hundred microseconds at 8 MiB however few the rows: the ranges fold their (quotient, index)
// winners with a 64-bit atomicMin in the row's idx64 slot (which must hold -1), count themselves in the row's idx32
// slot (which must hold ticket_base), and the last to arrive writes the row's outputs.  Its loss goes to
// row_loss_slots[i] when given (so that the partials do not depend on which workgroup that was), else into the
// return value, as do the losses of rows that one workgroup handled alone (meaningful in thread 0).
__device__ __forceinline__ float search_listed_rows(const SearchArgs& a, int D, int nchw, const int32_t* list, int list_step,
                                                    int count, int ticket_base, float* row_loss_slots, float* zs, float* rb,
                                                    int* ri) {
  float block_loss = 0.0f;
  int S = count > 0 ? (int)gridDim.x / count : 1;
  if (S > 8) S = 8;
  if (S > a.K / 256) S = a.K / 256;
  if (S < 2 || a.idx64 == nullptr) {
    for (int i = blockIdx.x; i < count; i += gridDim.x)
      block_loss += search_one_row(a, D, nchw, list[(int64_t)i * list_step], zs, rb, ri);
    return block_loss;
  }
  // count * S <= gridDim.x: at most one (row, range) per workgroup
  __shared__ int sh_last, sh_w;
  __shared__ float sh_q;
  const int w = blockIdx.x;
  if (w >= count * S) return 0.0f;
  const int i = w / S, part = w - i * S;
  const int64_t n = list[(int64_t)i * list_step];
  const int Kc = ((a.K + S - 1) / S + 3) & ~3;
  const int klo = part * Kc, khi = klo + Kc < a.K ? klo + Kc : a.K;
  const float z2 = row_load(a, D, nchw, n, zs, rb);
  row_scan(a, D, z2, klo, khi, zs, rb, ri);
  if (threadIdx.x == 0) {
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(&a.idx64[n]);
    if (ri[0] != 0x7fffffff) atomicMin(slot, vq_row_key(rb[0], ri[0]));
    __threadfence();
    const int ticket = atomicAdd(&a.idx32[n], 1);
    sh_last = ticket == ticket_base + S - 1;
    if (sh_last) {
      __threadfence();
      const unsigned long long key = atomicMin(slot, ~0ull);
      sh_w = (int)(uint32_t)key;
      sh_q = vq_row_key_quotient(key);
    }
  }
  __syncthreads();
  if (sh_last) {
    const float row_loss = row_finish(a, D, nchw, n, sh_w, sh_q, zs, rb);
    if (row_loss_slots) {
      if (threadIdx.x == 0) row_loss_slots[i] = row_loss;
    } else {
      block_loss = row_loss;
    }
  }
  return block_loss;
}

// The filtered search's last resort: the rows that both of its levels left open (list_b, counted in
// control[word_b]) -- or, when the first level left only a handful open (at most VQ_FILTER_FEW_ROWS, list_a /
// word_a), those rows directly: the second level's three launches then return at once, and their fixed cost
// (0.1 ms for a dozen rows) is spared.
__global__ __launch_bounds__(256) void vq_search_rows_kernel(SearchArgs a, int D, int nchw,
                                                             const int32_t* __restrict__ list_a, int word_a,
                                                             const int32_t* __restrict__ list_b, int word_b,
                                                             float* __restrict__ loss_out) {
  __shared__ __attribute__((aligned(16))) float zs[1024];
  __shared__ float rb[256];
  __shared__ int ri[256];
  float block_loss = 0.0f;  // meaningful in thread 0
  if (!vq_filter_gave_up(a.control, a.N)) {
    const int ca = a.control[word_a];
    const bool few = ca <= VQ_FILTER_FEW_ROWS;
    // (rows of either list hold -1 in idx32 and idx64: the marks the verification left)
    block_loss = search_listed_rows(a, D, nchw, few ? list_a : list_b, 1, few ? ca : a.control[word_b], -1, nullptr, zs, rb, ri);
  }
  if (threadIdx.x == 0 && loss_out) loss_out[blockIdx.x] = block_loss;
}

// ---------------------------------------------------------------------------------------
// Codebook preparation: squared norms + MFMA-fragment image.
// One wave per 32-code tile; lane (i = lane&31, h = lane>>5) walks row i's channels 2s+h,
// which is both the even/odd norm chain and the A-operand element order.
// ---------------------------------------------------------------------------------------
// Dreal: channels of a row of E in memory; Dreal < D lays the image out at D channels with zeros in
// channels Dreal .. D-1 (the norms, chains over the even and over the odd channels, are unchanged by them).
template <int D>
__global__ __launch_bounds__(64) void vq_prep_image_kernel(const float* __restrict__ E, int K, float* __restrict__ image,
                                                           float* __restrict__ e2, int32_t* control,
                                                           const float* __restrict__ bias = nullptr,
                                                           int conv_image = 0, int Dreal = D) {
  constexpr int G = D / 8;
  const int lane = threadIdx.x;
  const int i = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x;  // 32 codes
  const int code = tile * 32 + i;
  const bool real = code < K;
  const f32x4* row = reinterpret_cast<const f32x4*>(E + (size_t)(real ? code : 0) * D);
  const float* srow = E + (size_t)(real ? code : 0) * Dreal;
  const int stage = tile >> 1, rt = tile & 1;
  float* st_base = image + (size_t)stage * vq_stage_floats(D);
  f32x4* dst = reinterpret_cast<f32x4*>(st_base + (size_t)rt * 32 * D) + lane;
  float p = 0.0f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    f32x4 o;
    if (Dreal != D) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 8 * g + 2 * q + h;
        o[q] = (real && c < Dreal) ? srow[c] : 0.0f;
      }
    } else {
      f32x4 v0 = row[2 * g], v1 = row[2 * g + 1];
      if (!real) { v0 = f32x4{0, 0, 0, 0}; v1 = f32x4{0, 0, 0, 0}; }
      o[0] = h ? v0[1] : v0[0];
      o[1] = h ? v0[3] : v0[2];
      o[2] = h ? v1[1] : v1[0];
      o[3] = h ? v1[3] : v1[2];
    }
    dst[g * 64] = o;
#pragma unroll
    for (int q = 0; q < 4; ++q) p = __builtin_fmaf(o[q], o[q], p);
  }
  const float n2 = p + __shfl_xor(p, 32);
  if (h == 0 && conv_image) {
    // image of a 1x1 convolution's weights: the slot of the norms carries the bias
    st_base[VQ_STAGE_CODES * D + rt * 32 + i] = (real && bias) ? bias[code] : 0.0f;
  } else if (h == 0) {
    const float v = real ? n2 : __builtin_inff();
    e2[code] = v;
    st_base[VQ_STAGE_CODES * D + rt * 32 + i] = v;
    if (real && !(n2 <= VQ_NORM_LIMIT)) atomicOr(&control[VQ_CTL_CB_FLAG], 1);
  } else {
    // zero the unused tail of the norms piece once per stage (keeps the image defined)
    if (rt == 0) {
      for (int q = 0; q < 6; ++q) st_base[VQ_STAGE_CODES * D + 64 + q * 32 + i] = 0.0f;
    }
  }
}

// Norms only, any D (shapes that take the exact kernel).
__global__ __launch_bounds__(256) void vq_prep_norms_kernel(const float* __restrict__ E, int K, int Kpad, int D,
                                                            float* __restrict__ e2, int32_t* control) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int code = t >> 1, hh = t & 1;
  if (code >= Kpad) return;  // Kpad is even, both lanes of a pair take the same branch
  float p = 0.0f;
  if (code < K) {
    const float* row = E + (size_t)code * D;
    for (int c = hh; c < D; c += 2) p = __builtin_fmaf(row[c], row[c], p);
  }
  const float o = __shfl_xor(p, 1);
  if (hh == 0) {
    const float n2 = p + o;
    e2[code] = code < K ? n2 : __builtin_inff();
    if (code < K && !(n2 <= VQ_NORM_LIMIT)) atomicOr(&control[VQ_CTL_CB_FLAG], 1);
  }
}

// Merge of the codebook split: splits are in increasing code order, so "later split wins only
// if strictly smaller" keeps the lowest index.  Flagged rows (index -1) are left to the exact kernel.
__global__ __launch_bounds__(256) void vq_split_merge_kernel(const float* __restrict__ part_d,
                                                            const int32_t* __restrict__ part_i, int ksplit, int64_t N,
                                                            int64_t* __restrict__ idx64, int32_t* __restrict__ idx32,
                                                            float* __restrict__ dmin) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float best = part_d[n];
  int32_t bi = part_i[n];
  if (bi >= 0) {
    for (int y = 1; y < ksplit; ++y) {
      const float d = part_d[(size_t)y * N + n];
      if (d < best) { best = d; bi = part_i[(size_t)y * N + n]; }
    }
  }
  if (idx64) idx64[n] = bi;
  if (idx32) idx32[n] = bi;
  if (dmin && bi >= 0) dmin[n] = best;
}

// Standalone gather + straight-through + loss partials for given int64 indices.  Rows whose
// index is outside [0, K) are skipped (no output, no loss contribution).
__global__ __launch_bounds__(256) void vq_gather_st_kernel(const float* __restrict__ z, int nchw, int64_t N, int D,
                                                           int64_t HW, const float* __restrict__ E, int K,
                                                           const int64_t* __restrict__ idx, float* __restrict__ q_out,
                                                           float* __restrict__ loss_partial,
                                                           const int32_t* __restrict__ gate_control,
                                                           int64_t loss_slots) {
  // A workgroup owns 256 consecutive rows.  NCHW: thread = row, loop over channels, so the
  // z loads and q stores of a wave are contiguous along hw; the codebook rows come through
  // the caches.  ND: a wave walks one row at a time, lanes along channels.
  __shared__ float red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * VQ_WG_ROWS;
  float part = 0.0f;
  if (gate_control != nullptr && vq_filter_gave_up(gate_control, N)) {
    // filtered search that fell back to the fp32 kernel: that kernel writes every row itself
    if (t == 0 && loss_partial)
      for (int64_t i = blockIdx.x; i < loss_slots || i == blockIdx.x; i += gridDim.x) loss_partial[i] = 0.0f;
    return;
  }
  if (nchw) {
    const int64_t n = row0 + t;
    const int64_t k = n < N ? idx[n] : -1;
    if (k >= 0 && k < K) {
      const f32x4* e = reinterpret_cast<const f32x4*>(E + (size_t)k * D);
      const int64_t b = n / HW, hw = n - b * HW;
      const int64_t off0 = b * D * HW + hw;
      if ((D & 15) == 0) {
        // 16 channels per step: the four 16-byte reads of the code row hit one 64-byte sector
        for (int c16 = 0; c16 < D / 16; ++c16) {
          f32x4 ev[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) ev[q] = e[4 * c16 + q];
          // (the latents and the outputs pass through once: non-temporal, so that they do not push the code rows
          // out of L2 -- every row of the batch fetches one of K rows of E at random)
          float zv[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) zv[i] = __builtin_nontemporal_load(&z[off0 + (int64_t)(16 * c16 + i) * HW]);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float diff = ev[i >> 2][i & 3] - zv[i];
            part = __builtin_fmaf(diff, diff, part);
            __builtin_nontemporal_store(zv[i] + diff, &q_out[off0 + (int64_t)(16 * c16 + i) * HW]);
          }
        }
      } else {
        const float* es = E + (size_t)k * D;
        for (int c = 0; c < D; ++c) {
          const float zv = z[off0 + (int64_t)c * HW];
          const float diff = es[c] - zv;
          part = __builtin_fmaf(diff, diff, part);
          q_out[off0 + (int64_t)c * HW] = zv + diff;
        }
      }
    }
  } else if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(E)) & 15) == 0) {
    // rows: a wave takes rows wave, wave + 4, ... of the block, four at a time; a lane moves 16 bytes of each
    // (one wave instruction = 1 KiB = a whole row at D = 256); latents and outputs non-temporal as above
    const int D4 = D >> 2;
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
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float diff = ev[u][i] - zv[u][i];
              part = __builtin_fmaf(diff, diff, part);
              o[i] = zv[u][i] + diff;
            }
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(q_out + n[u] * D) + c4);
          }
      }
    }
  } else {
    for (int r = wave; r < VQ_WG_ROWS; r += 4) {
      const int64_t n = row0 + r;
      if (n >= N) break;
      const int64_t k = idx[n];
      if (k < 0 || k >= K) continue;
      const float* e = E + (size_t)k * D;
      for (int c = lane; c < D; c += 64) {
        const float zv = z[n * D + c];
        const float diff = e[c] - zv;
        part = __builtin_fmaf(diff, diff, part);
        q_out[n * D + c] = zv + diff;
      }
    }
  }
  const float ws = wave_sum(part);
  if (lane == 0) red[wave] = ws;
  __syncthreads();
  if (t == 0 && loss_partial) {
    loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    for (int64_t i = (int64_t)blockIdx.x + gridDim.x; i < loss_slots; i += gridDim.x) loss_partial[i] = 0.0f;
  }
}

template <int D, int CT, bool NCHW, int TEMP, bool PAIRWISE, bool PAD = false>
int launch_mfma(const SearchArgs& a, hipStream_t s) {
  constexpr int lds_bytes = (int)(2 * vq_stage_floats(D) * 4 + 256);
  auto kern = vq_search_mfma_kernel<D, CT, NCHW, TEMP, PAIRWISE, PAD>;
  const int arc = vq_lds_ready<vq_search_mfma_kernel<D, CT, NCHW, TEMP, PAIRWISE, PAD>, lds_bytes>();
  if (arc != VQ_OK) return arc;
  const int64_t blocks = (a.N + VQ_WG_ROWS - 1) / VQ_WG_ROWS;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)(a.ksplit > 1 ? a.ksplit : 1)), dim3(512 / CT), lds_bytes, s, a);
  return vq_hip_status(hipGetLastError());
}

// Column tiles per wave for each channel count (see the kernel's comment).
#ifndef VQ_CT_256
#define VQ_CT_256 1
#endif
template <int D> struct TileChoice { static constexpr int CT = 1; };
template <> struct TileChoice<256> { static constexpr int CT = VQ_CT_256; };

template <int D, int CT, bool NCHW, bool PAD>
int dispatch_mfma_temp(const SearchArgs& a, bool temp, hipStream_t s) {
  if (!temp) return launch_mfma<D, CT, NCHW, VQ_T_NONE, false, PAD>(a, s);
  // partial sweeps (their winners are merged by quotient) and the filtered search's stand-in divide every distance
  if (a.ksplit > 1 || a.fallback_role) return launch_mfma<D, CT, NCHW, VQ_T_LITERAL, false, PAD>(a, s);
  return launch_mfma<D, CT, NCHW, VQ_T_DEFERRED, false, PAD>(a, s);
}

template <int D>
int dispatch_mfma(const SearchArgs& a, bool nchw, bool temp, bool pairwise, hipStream_t s) {
  constexpr int CT = TileChoice<D>::CT;
  if (a.Dreal != D) {
    // fewer channels in memory than the kernel runs at: the zero-padded form (not for the pairwise sweep)
    if (pairwise || a.Dreal > D || a.Dreal <= 0) return VQ_ERR_UNSUPPORTED;
    return nchw ? dispatch_mfma_temp<D, 1, true, true>(a, temp, s) : dispatch_mfma_temp<D, 1, false, true>(a, temp, s);
  }
  if (pairwise) return launch_mfma<D, CT, false, VQ_T_NONE, true>(a, s);
  return nchw ? dispatch_mfma_temp<D, CT, true, false>(a, temp, s) : dispatch_mfma_temp<D, CT, false, false>(a, temp, s);
}

int run_mfma(const SearchArgs& a, int D, bool nchw, bool temp, bool pairwise, hipStream_t s) {
  switch (D) {
    case 16: return dispatch_mfma<16>(a, nchw, temp, pairwise, s);
    case 32: return dispatch_mfma<32>(a, nchw, temp, pairwise, s);
    case 64: return dispatch_mfma<64>(a, nchw, temp, pairwise, s);
    case 128: return dispatch_mfma<128>(a, nchw, temp, pairwise, s);
    case 256: return dispatch_mfma<256>(a, nchw, temp, pairwise, s);
    default: return VQ_ERR_UNSUPPORTED;
  }
}

// Loss partials the search itself owns: one per 256-row workgroup, or, for batches the persistent
// kernel may take (one partial per wave of its grid), at least as many as that grid has waves.
int64_t loss_slots_for(int64_t N) {
  const int64_t blocks = N > 0 ? (N + VQ_WG_ROWS - 1) / VQ_WG_ROWS : 0;
  return blocks >= 512 && blocks < VQ_CYC_WAVES ? VQ_CYC_WAVES : blocks;
}

// ---- the persistent kernel for short sweeps: when it runs, and how ---------------------------
struct CyclicTune {
  int enable = 1;
  int max_stages = 16;          // fewer than 128 channels: K <= 1024, beyond that the one-tile-per-workgroup kernel is as fast
  int64_t min_tiles_per_wave = 2;
  int force_streaming = 0;      // never keep the image resident (experiments)
};
CyclicTune g_cyc;


constexpr int CYC_LDS_MAX = 160 * 1024;

bool cyc_resident(int nstages, int D) {
  return !g_cyc.force_streaming && (size_t)nstages * vq_stage_floats(D) * 4 + 256 <= (size_t)CYC_LDS_MAX;
}

int device_cus(int dev) {
  static int cus[64] = {};
  if (dev < 0 || dev >= 64) return 256;
  if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
  return cus[dev];
}

bool cyc_eligible(int64_t N, int K, int D, bool nchw, int64_t HW) {
  if (!g_cyc.enable) return false;
  if (nchw && HW % 32 != 0) return false;      // a 32-row tile must lie inside one image
  if (loss_slots_for(N) < VQ_CYC_WAVES) return false;   // one loss partial per wave of the grid
  const int nstages = vq_kpad(K) / VQ_STAGE_CODES;
  // long sweeps: with at least 128 channels the persistent kernel is ahead at every codebook size (K = 8192, D = 256,
  // 4 Mi rows: 122.3 against 123.6 ms; D = 128: 64.0 against 64.9 ms), with 64 channels only up to 16 stages
  // (K = 8192, D = 64: 35.1 against 33.1 ms)
  if (D < 128 && nstages > g_cyc.max_stages) return false;
  const int64_t ntiles = (N + 31) / 32;
  if (ntiles < g_cyc.min_tiles_per_wave * VQ_CYC_WAVES) return false;
  // cycling stages: the two groups' memory steps must not overlap, or nobody is left to issue the DMA
  if (!cyc_resident(nstages, D) && nstages < cyc_mem_steps(D, false) + 1) return false;
  return true;
}

template <int D, bool NCHW, bool TEMP, bool RESIDENT>
int launch_cyclic(const SearchArgs& a, hipStream_t s) {
  constexpr int attr_bytes = RESIDENT ? CYC_LDS_MAX : (int)(2 * vq_stage_floats(D) * 4 + 256);
  int dev = 0;
  const int arc = vq_lds_ready<vq_search_cyclic_kernel<D, NCHW, TEMP, RESIDENT>, attr_bytes>(&dev);
  if (arc != VQ_OK) return arc;
  const size_t lds_bytes = RESIDENT ? (size_t)a.nstages * vq_stage_floats(D) * 4 + 256 : (size_t)attr_bytes;
  const int64_t ntiles = (a.N + 31) / 32;
  int64_t grid = device_cus(dev);
  if (grid > VQ_CYC_WAVES / 8) grid = VQ_CYC_WAVES / 8;
  if (grid > (ntiles + 7) / 8) grid = (ntiles + 7) / 8;
  hipLaunchKernelGGL((vq_search_cyclic_kernel<D, NCHW, TEMP, RESIDENT>), dim3((unsigned)grid), dim3(512), lds_bytes,
                     s, a, ntiles);
  return vq_hip_status(hipGetLastError());
}

template <int D, bool NCHW>
int dispatch_cyclic2(const SearchArgs& a, bool temp, hipStream_t s) {
  const bool res = cyc_resident(a.nstages, D);
  if (temp) return res ? launch_cyclic<D, NCHW, true, true>(a, s) : launch_cyclic<D, NCHW, true, false>(a, s);
  return res ? launch_cyclic<D, NCHW, false, true>(a, s) : launch_cyclic<D, NCHW, false, false>(a, s);
}

template <int D>
int dispatch_cyclic(const SearchArgs& a, bool nchw, bool temp, hipStream_t s) {
  return nchw ? dispatch_cyclic2<D, true>(a, temp, s) : dispatch_cyclic2<D, false>(a, temp, s);
}

int run_cyclic(const SearchArgs& a, int D, bool nchw, bool temp, hipStream_t s) {
  switch (D) {
    case 16: return dispatch_cyclic<16>(a, nchw, temp, s);
    case 32: return dispatch_cyclic<32>(a, nchw, temp, s);
    case 64: return dispatch_cyclic<64>(a, nchw, temp, s);
    case 128: return dispatch_cyclic<128>(a, nchw, temp, s);
    case 256: return dispatch_cyclic<256>(a, nchw, temp, s);
    default: return VQ_ERR_UNSUPPORTED;
  }
}

int run_exact(const SearchArgs& a, int D, bool nchw, bool use_list, float* loss_out, int blocks, hipStream_t s) {
  const size_t lds_bytes = ((size_t)XR * D + XR + 2 * XR * 256) * 4;
  const int arc = vq_lds_ready<vq_search_exact_kernel, (XR * 4096 + XR + 2 * XR * 256) * 4>();  // sized for D = 4096
  if (arc != VQ_OK) return arc;
  hipLaunchKernelGGL(vq_search_exact_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, s, a, D, nchw ? 1 : 0,
                     use_list ? 1 : 0, loss_out);
  return vq_hip_status(hipGetLastError());
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Codebook split.  Batches too small to fill the chip -- at most 128 row tiles (32 Ki rows) -- divide
// the sweep so that row tiles x parts stays within one round of 256 workgroups.  Larger batches
// (up to 2048 row tiles) divide it in 2, 4 or 8 when their last round of workgroups would run
// nearly empty: 257 row tiles are two rounds of whole sweeps undivided, but five rounds of quarter
// sweeps; a division is taken when it saves at least a tenth of the rounds x stages.
struct SplitPlan { int ksplit; int sps; };
SplitPlan split_plan(int64_t N, int K) {
  constexpr int64_t CUS = 256;
  const int64_t blocks = (N + VQ_WG_ROWS - 1) / VQ_WG_ROWS;
  const int nstages = vq_kpad(K) / VQ_STAGE_CODES;
  if (nstages < 8) return {1, nstages};
  if (blocks > 128) {
    if (blocks > 2048) return {1, nstages};
    const int64_t whole = (blocks + CUS - 1) / CUS * nstages;
    int64_t best = whole;
    SplitPlan plan{1, nstages};
    for (int want = 2; want <= 8 && want <= nstages / 4; want *= 2) {
      const int sps = (nstages + want - 1) / want;
      const int ks = (nstages + sps - 1) / sps;
      const int64_t cost = (blocks * ks + CUS - 1) / CUS * sps;
      if (cost * 10 <= whole * 9 && cost < best) {
        best = cost;
        plan = {ks, sps};
      }
    }
    return plan;
  }
  int ks = (int)(256 / blocks);
  if (ks > nstages / 4) ks = nstages / 4;
  if (ks > 64) ks = 64;
  if (ks < 2) return {1, nstages};
  const int sps = (nstages + ks - 1) / ks;
  return {(nstages + sps - 1) / sps, sps};
}

extern "C" {

size_t vq_search_split_bytes(int64_t N, int K) {
  if (N <= 0 || K <= 0) return 0;
  const SplitPlan p = split_plan(N, K);
  return p.ksplit > 1 ? (size_t)p.ksplit * (size_t)N * 8 : 0;
}

size_t vq_image_bytes(int K, int D) {
  if (K <= 0 || D <= 0 || vq_pad_dim(D) == 0) return 256;
  return (size_t)(vq_kpad(K) / VQ_STAGE_CODES) * vq_stage_floats(vq_pad_dim(D)) * 4;
}
size_t vq_norms_bytes(int K) { return K > 0 ? (size_t)vq_kpad(K) * 4 : 256; }
int64_t vq_search_blocks(int64_t N) { return N > 0 ? (N + VQ_WG_ROWS - 1) / VQ_WG_ROWS : 0; }
// (the sweep's own partials, then the VQ_NEAR_SLOTS row partials that the exact kernel's shared rows use)
static int64_t loss_slots(int64_t N) { return loss_slots_for(N) + VQ_NEAR_SLOTS; }
size_t vq_loss_partial_bytes(int64_t N) { return (size_t)(loss_slots(N) + VQ_FIX_WGS) * 4; }

// Internal (not in the public header): knobs of the persistent short-sweep kernel, for experiments
// and A/B timing.  what: 0 enable, 1 max stages, 2 min tiles per wave, 4 force cycling stages.  Returns the previous value.
int vq_internal_tune(int what, int value) {
  int old = -1;
  switch (what) {
    case 0: old = g_cyc.enable; g_cyc.enable = value; break;
    case 1: old = g_cyc.max_stages; g_cyc.max_stages = value; break;
    case 2: old = (int)g_cyc.min_tiles_per_wave; g_cyc.min_tiles_per_wave = value; break;
    case 4: old = g_cyc.force_streaming; g_cyc.force_streaming = value; break;
    default: break;
  }
  return old;
}
size_t vq_control_bytes(void) { return VQ_CTL_WORDS * 4; }

int vq_search_uses_persistent_kernel(int64_t N, int K, int D, int layout, int64_t HW) {
  if (N <= 0 || K <= 0 || D <= 0 || vq_pad_dim(D) != D) return 0;
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  const SplitPlan sp = split_plan(N, K);
  if (sp.ksplit > 1) return 0;   // (a caller that passes no split workspace gets the undivided sweep instead)
  return cyc_eligible(N, K, D, nchw, nchw ? HW : 1) ? 1 : 0;
}

int vq_control_reset(int32_t* control, void* stream) {
  if (!control) return VQ_ERR_INVALID_ARG;
  return vq_hip_status(hipMemsetAsync(control, 0, VQ_CTL_WORDS * 4, (hipStream_t)stream));
}

int vq_codebook_prep(const float* E, int K, int D, float* image, float* e2, int32_t* control, void* stream) {
  if (!E || !e2 || !control || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (D > 4096) return VQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int Kpad = vq_kpad(K);
  if (vq_pad_dim(D) != 0 && image != nullptr) {
    // (a row of E is read 16 bytes at a time only when its length is one of the kernels' own channel counts)
    if ((vq_fast_dim(D) && !aligned16(E)) || !aligned16(image)) return VQ_ERR_INVALID_ARG;
    const int tiles = Kpad / 32;
    const float* nobias = nullptr;
    switch (vq_pad_dim(D)) {
      case 16: hipLaunchKernelGGL(vq_prep_image_kernel<16>, dim3(tiles), dim3(64), 0, s, E, K, image, e2, control, nobias, 0, D); break;
      case 32: hipLaunchKernelGGL(vq_prep_image_kernel<32>, dim3(tiles), dim3(64), 0, s, E, K, image, e2, control, nobias, 0, D); break;
      case 64: hipLaunchKernelGGL(vq_prep_image_kernel<64>, dim3(tiles), dim3(64), 0, s, E, K, image, e2, control, nobias, 0, D); break;
      case 128: hipLaunchKernelGGL(vq_prep_image_kernel<128>, dim3(tiles), dim3(64), 0, s, E, K, image, e2, control, nobias, 0, D); break;
      default: hipLaunchKernelGGL(vq_prep_image_kernel<256>, dim3(tiles), dim3(64), 0, s, E, K, image, e2, control, nobias, 0, D); break;
    }
  } else {
    hipLaunchKernelGGL(vq_prep_norms_kernel, dim3((2 * Kpad + 255) / 256), dim3(256), 0, s, E, K, Kpad, D, e2, control);
  }
  return vq_hip_status(hipGetLastError());
}

static int search_common(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E,
                         const float* image, const float* e2, int K, float temperature, int64_t code_base,
                         int64_t* idx64, int32_t* idx32, float* dmin, float* q_out, float* loss_partial,
                         int32_t* nf_rows, int32_t* control, void* split_ws, bool force_exact, hipStream_t s) {
  if (!z || !E || !e2 || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  if (D > 4096 || N >= (int64_t)1 << 31 || (int64_t)K + code_base >= (int64_t)1 << 31) return VQ_ERR_UNSUPPORTED;
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  SearchArgs a{};
  a.z = z; a.E = E; a.image = image; a.e2 = e2; a.N = N; a.HW = nchw ? HW : 1; a.K = K; a.Dreal = D;
  a.nstages = vq_kpad(K) / VQ_STAGE_CODES; a.T = temperature; a.code_base = code_base;
  a.idx64 = idx64; a.idx32 = idx32; a.dmin = dmin; a.q_out = q_out; a.loss_partial = loss_partial;
  a.nf_rows = nf_rows; a.control = control;
  a.near_ok = (D % 4 == 0 && D <= 1024 && aligned16(E)) ? 1 : 0;   // what search_one_row takes
  const int64_t blocks = vq_search_blocks(N);
  const int64_t slots = loss_slots(N);
  a.loss_slots = slots;
  a.near_loss = loss_partial ? loss_partial + slots - VQ_NEAR_SLOTS : nullptr;
  // T > 0 and finite keeps the MFMA path (division by it is monotone); anything else is
  // evaluated literally by the exact kernel.
  const bool t_regular = temperature > 0.0f && temperature < __builtin_inff();
  // the matrix-core kernels: any D up to 256 (channel counts other than 16 / 32 / 64 / 128 / 256 run zero-padded
  // to the next of those and read z and E element by element, so only the kernels' own counts need alignment)
  const int Dp = vq_pad_dim(D);
  const bool padded = Dp != D;
  const bool fast = !force_exact && Dp != 0 && image != nullptr && t_regular && (padded || (aligned16(z) && aligned16(E))) &&
                    aligned16(image) && idx32 != nullptr && nf_rows != nullptr && control != nullptr;
  if (fast) {
    const SplitPlan sp = split_plan(N, K);
    a.ksplit = 1;
    a.stages_per_split = a.nstages;
    int rc;
    if (split_ws != nullptr && sp.ksplit > 1 && idx64 != nullptr && (q_out == nullptr || code_base == 0)) {
      // small batch: split the codebook sweep over blockIdx.y, merge, then gather separately
      a.ksplit = sp.ksplit;
      a.stages_per_split = sp.sps;
      a.part_d = static_cast<float*>(split_ws);
      a.part_i = reinterpret_cast<int32_t*>(a.part_d + (size_t)sp.ksplit * N);
      a.q_out = nullptr;
      a.loss_partial = nullptr;
      rc = run_mfma(a, Dp, nchw, temperature != 1.0f, false, s);
      if (rc != VQ_OK) return rc;
      hipLaunchKernelGGL(vq_split_merge_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a.part_d, a.part_i,
                         sp.ksplit, N, idx64, idx32, dmin);
      if (q_out) {
        hipLaunchKernelGGL(vq_gather_st_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, nchw ? 1 : 0, N, D, a.HW,
                           E, K, idx64, q_out, loss_partial, (const int32_t*)nullptr, slots);
      }
      VQ_CHECK_HIP(hipGetLastError());
      a.q_out = q_out;
      a.loss_partial = loss_partial;
      a.ksplit = 1;
    } else if (!padded && cyc_eligible(N, K, D, nchw, a.HW)) {
      // short sweep, large batch: the persistent kernel (same winners, distances and outputs)
      rc = run_cyclic(a, D, nchw, temperature != 1.0f, s);
      if (rc != VQ_OK) return rc;
    } else {
      rc = run_mfma(a, Dp, nchw, temperature != 1.0f, false, s);
      if (rc != VQ_OK) return rc;
    }
    // rows flagged by the MFMA kernel (normally none: the kernel then exits at once)
    float* fix_loss = loss_partial ? loss_partial + slots : nullptr;
    return run_exact(a, D, nchw, true, fix_loss, VQ_FIX_WGS, s);
  }
  // exact path for everything: the search's own partials are zeroed, the kernel fills the tail
  if (loss_partial) VQ_CHECK_HIP(hipMemsetAsync(loss_partial, 0, (size_t)slots * 4, s));
  if (control) {
    int rc = vq_launch_absmax(z, N * D, control, s);
    if (rc != VQ_OK) return rc;
  }
  float* fix_loss = loss_partial ? loss_partial + slots : nullptr;
  return run_exact(a, D, nchw, false, fix_loss, VQ_FIX_WGS, s);
}

int vq_search(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* image,
              const float* e2, int K, float temperature, int64_t code_base, int64_t* idx64, int32_t* idx32,
              float* dmin, float* q_out, float* loss_partial, int32_t* nf_rows, int32_t* control, void* split_ws,
              void* stream) {
  if (!idx32 || !nf_rows || !control) return VQ_ERR_INVALID_ARG;
  return search_common(z, layout, N, D, HW, E, image, e2, K, temperature, code_base, idx64, idx32, dmin, q_out,
                       loss_partial, nf_rows, control, split_ws, false, (hipStream_t)stream);
}

// loss partials of the filtered search: [0, blocks) gather of the settled rows, [blocks, +FIX) exact kernel,
// then [.., +blocks) fp32 kernel in the fall-back role, then [.., +FIX) per-row kernel
size_t vq_filter_loss_partial_bytes(int64_t N) { return (size_t)(2 * vq_search_blocks(N) + 2 * VQ_FIX_WGS) * 4; }

int vq_search_filtered(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* image,
                       const float* e2, const void* fimage, int K, float temperature, int64_t code_base,
                       int64_t* idx64, int32_t* idx32, float* dmin, float* q_out, float* loss_partial,
                       int32_t* nf_rows, int32_t* control, void* work, void* stream) {
  if (!z || !E || !image || !e2 || !fimage || !idx64 || !idx32 || !nf_rows || !control || !work) return VQ_ERR_INVALID_ARG;
  if (N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  const bool t_regular = temperature > 0.0f && temperature < __builtin_inff();
  if (!vq_filter_supported(K, D) || !t_regular || N >= (int64_t)1 << 31 || (int64_t)K + code_base >= (int64_t)1 << 31)
    return VQ_ERR_UNSUPPORTED;
  if (!aligned16(z) || !aligned16(E) || !aligned16(image) || !aligned16(fimage) || !aligned16(work)) return VQ_ERR_INVALID_ARG;
  if (q_out != nullptr && code_base != 0) return VQ_ERR_UNSUPPORTED;  // the gather below indexes E by the global code
  hipStream_t s = (hipStream_t)stream;
  const bool nchw = layout == VQ_LAYOUT_NCHW;
  const int64_t blocks = vq_search_blocks(N);
  // 1. bf16 filter + exact verification of the candidates, two levels: settles almost every row
  int32_t* hard_rows = nullptr;
  int32_t* open_rows = nullptr;   // what the first level left open (the second level's input)
  int rc = vq_filter_verify_launch(z, nchw ? 1 : 0, N, D, HW, E, e2, fimage, K, temperature, code_base, idx64, idx32,
                                   dmin, nf_rows, &open_rows, &hard_rows, control, work, stream);
  if (rc != VQ_OK) return rc;
  if (q_out) {
    // 2. settled rows: gather + straight-through + loss partials (rows marked -1 are skipped)
    hipLaunchKernelGGL(vq_gather_st_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, nchw ? 1 : 0, N, D,
                       nchw ? HW : 1, E, K, idx64, q_out, loss_partial, (const int32_t*)control, (int64_t)0);
    VQ_CHECK_HIP(hipGetLastError());
  }
  SearchArgs a{};
  a.z = z; a.E = E; a.image = image; a.e2 = e2; a.N = N; a.HW = nchw ? HW : 1; a.K = K; a.Dreal = D;
  a.nstages = vq_kpad(K) / VQ_STAGE_CODES; a.T = temperature; a.code_base = code_base;
  a.idx64 = idx64; a.idx32 = idx32; a.dmin = dmin; a.q_out = q_out;
  a.nf_rows = nf_rows; a.control = control; a.ksplit = 1; a.stages_per_split = a.nstages;
  // 3. the few rows with more close codes than both candidate lists hold (or the handful the first level left
  //    open, when the second level stood aside): every code for each of them
  float* row_loss = loss_partial ? loss_partial + 2 * blocks + VQ_FIX_WGS : nullptr;
  hipLaunchKernelGGL(vq_search_rows_kernel, dim3(VQ_FIX_WGS), dim3(256), 0, s, a, D, nchw ? 1 : 0,
                     (const int32_t*)open_rows, (int)VQ_CTL_OVF_COUNT, (const int32_t*)hard_rows, (int)VQ_CTL_OVF2_COUNT,
                     row_loss);
  VQ_CHECK_HIP(hipGetLastError());
  // 4. fall-back: a batch the filter could not separate (over N/8 rows unsettled after the first
  //    pass) is redone by the fp32 kernel; otherwise its workgroups exit at once
  a.loss_partial = loss_partial ? loss_partial + blocks + VQ_FIX_WGS : nullptr;
  if (a.loss_partial) VQ_CHECK_HIP(hipMemsetAsync(a.loss_partial, 0, (size_t)blocks * 4, s));
  a.fallback_role = 1;
  rc = run_mfma(a, D, nchw, temperature != 1.0f, false, s);
  if (rc != VQ_OK) return rc;
  // 5. rows that are not finite or degenerate (and every row of a flagged codebook): the exact kernel
  a.fallback_role = 0;
  a.loss_partial = loss_partial;
  float* fix_loss = loss_partial ? loss_partial + blocks : nullptr;
  return run_exact(a, D, nchw, true, fix_loss, VQ_FIX_WGS, s);
}

int vq_search_exact(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* e2,
                    int K, float temperature, int64_t code_base, int64_t* idx64, int32_t* idx32, float* dmin,
                    float* q_out, float* loss_partial, void* stream) {
  return search_common(z, layout, N, D, HW, E, nullptr, e2, K, temperature, code_base, idx64, idx32, dmin, q_out,
                       loss_partial, nullptr, nullptr, nullptr, true, (hipStream_t)stream);
}

int vq_gather_st(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, int K,
                 const int64_t* idx64, float* q_out, float* loss_partial, void* stream) {
  if (!z || !E || !idx64 || !q_out || N <= 0 || K <= 0 || D <= 0) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  const int64_t blocks = vq_search_blocks(N);
  hipLaunchKernelGGL(vq_gather_st_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                     layout == VQ_LAYOUT_NCHW ? 1 : 0, N, D, layout == VQ_LAYOUT_NCHW ? HW : 1, E, K, idx64, q_out,
                     loss_partial, (const int32_t*)nullptr, (int64_t)0);
  return vq_hip_status(hipGetLastError());
}

size_t vq_conv1x1_image_bytes(int C_out, int C_in) { return vq_image_bytes(C_out, C_in); }

int vq_conv1x1_prep(const float* W, const float* bias, int C_out, int C_in, float* image, void* stream) {
  if (!W || !image || C_out <= 0 || C_in <= 0) return VQ_ERR_INVALID_ARG;
  if (!vq_fast_dim(C_in)) return VQ_ERR_UNSUPPORTED;
  if (!aligned16(W) || !aligned16(image)) return VQ_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int tiles = vq_kpad(C_out) / 32;
  float* none = nullptr;
  int32_t* nocontrol = nullptr;
  switch (C_in) {
    case 16: hipLaunchKernelGGL(vq_prep_image_kernel<16>, dim3(tiles), dim3(64), 0, s, W, C_out, image, none, nocontrol, bias, 1); break;
    case 32: hipLaunchKernelGGL(vq_prep_image_kernel<32>, dim3(tiles), dim3(64), 0, s, W, C_out, image, none, nocontrol, bias, 1); break;
    case 64: hipLaunchKernelGGL(vq_prep_image_kernel<64>, dim3(tiles), dim3(64), 0, s, W, C_out, image, none, nocontrol, bias, 1); break;
    case 128: hipLaunchKernelGGL(vq_prep_image_kernel<128>, dim3(tiles), dim3(64), 0, s, W, C_out, image, none, nocontrol, bias, 1); break;
    default: hipLaunchKernelGGL(vq_prep_image_kernel<256>, dim3(tiles), dim3(64), 0, s, W, C_out, image, none, nocontrol, bias, 1); break;
  }
  return vq_hip_status(hipGetLastError());
}

int vq_conv1x1(const float* x, int layout, int64_t N, int C_in, int64_t HW, const float* image, int C_out, float* out,
               void* stream) {
  if (!x || !image || !out || N <= 0 || C_in <= 0 || C_out <= 0) return VQ_ERR_INVALID_ARG;
  if (layout != VQ_LAYOUT_ND && layout != VQ_LAYOUT_NCHW) return VQ_ERR_INVALID_ARG;
  if (layout == VQ_LAYOUT_NCHW && (HW <= 0 || N % HW != 0)) return VQ_ERR_INVALID_ARG;
  if (!vq_fast_dim(C_in) || N >= (int64_t)1 << 31) return VQ_ERR_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(image) || !aligned16(out)) return VQ_ERR_INVALID_ARG;
  const bool nchw = layout == VQ_LAYOUT_NCHW && HW > 1;
  ConvArgs a{};
  a.x = x; a.image = image; a.out = out; a.N = N; a.HW = nchw ? HW : 1; a.Cout = C_out;
  a.nstages = vq_kpad(C_out) / VQ_STAGE_CODES;
  hipStream_t s = (hipStream_t)stream;
  switch (C_in) {
    case 16: return launch_conv<16>(a, nchw, s);
    case 32: return launch_conv<32>(a, nchw, s);
    case 64: return launch_conv<64>(a, nchw, s);
    case 128: return launch_conv<128>(a, nchw, s);
    default: return launch_conv<256>(a, nchw, s);
  }
}

int vq_pairwise_search_launch(const float* E, const float* image, const float* e2, int K, int D, int64_t row_base,
                              int64_t n_rows, float* nn_dist, int64_t* nn_idx, int32_t* idx32, double* sum_partial,
                              float* mm_partial, int32_t* nf_rows, int32_t* control, void* stream) {
  // internal helper of vq_pairwise_stats / vq_pairwise_rows (vq_stats.hip); not part of the public header.
  // Rows [row_base, row_base + n_rows) of the codebook play the latents, all K rows the codes.
  SearchArgs a{};
  a.z = E + (size_t)row_base * D; a.E = E; a.image = image; a.e2 = e2; a.N = n_rows; a.HW = 1; a.K = K; a.Dreal = D;
  a.nstages = vq_kpad(K) / VQ_STAGE_CODES; a.T = 1.0f; a.code_base = 0;
  a.idx64 = nn_idx; a.idx32 = idx32; a.dmin = nn_dist; a.q_out = nullptr; a.loss_partial = nullptr;
  a.nf_rows = nf_rows; a.control = control; a.pw_sum = sum_partial; a.pw_minmax = mm_partial;
  a.pw_row_base = row_base;
  a.ksplit = 1; a.stages_per_split = a.nstages;
  return run_mfma(a, D, false, false, true, (hipStream_t)stream);
}

}  // extern "C"
