// Filtered nearest-codebook search for gfx950 (MI355X): the same result as the fp32 search of
// vq_search.hip -- the argmin of the fp32 fma-chain distances of
// vqvae/ema_vector_quantizer.py:42-53, lowest index on ties -- at bf16 matrix-core speed.
//
//   1. Filter (vqf_filter_kernel).  Approximate distances from bf16-rounded operands on
//      v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), in the z-stationary layout of the fp32
//      kernel: latents in registers as the B operand, the bf16 codebook image streamed through LDS
//      by DMA, 16 values of one latent per lane and 32-code tile.  What the matrix core accumulates
//      is a lower bound L[n,k] of the (shifted) fp32 distance -- see "Error bound" -- and each lane
//      keeps the R smallest keys it has seen; a key is L with its low 12 bits replaced by the code's
//      position in the sweep, so one and-or, one unsigned min and R-1 medians-of-three per value
//      maintain values and indices together.
//   2. Verify (vqf_verify_kernel).  Only codes whose key lies below a threshold derived from the
//      smallest key can be the fp32 argmin.  These few candidates (1.6 per row on Gaussian data) are
//      evaluated with the exact fp32 chain, in the order the fp32 MFMA and the CPU oracle use, and
//      the smallest (lowest index on ties) wins.
//   3. A row whose list may be incomplete (all R keys of a lane under the threshold) is filtered
//      again with R = 8 (list mode); what is still open after that goes to a kernel that evaluates
//      all codes for one row (vq_search.hip).  Rows that are not finite or degenerate, and every row
//      of a non-finite codebook, are handed to the exact scalar kernel through the same list the
//      fp32 path uses.  A batch whose first pass leaves more than N/8 rows open is finished by the
//      fp32 MFMA kernel.  All of this is decided on the device from counters.
//
// Error bound.  The filter works on centred data, z' = fl(z - mu), e' = fl(e - mu) with mu the
// codebook's mean row (distances do not change under a common shift; a common offset of the data
// would otherwise eat the bf16 mantissa).  With z' = z~ + dz and e' = e~ + de (bf16 round-to-nearest),
//   z'.e' - z~.e~ = z~.de + dz.e', so |2 z'.e' - 2 z~.e~| <= 2 (||z~|| ||de_k|| + ||dz|| ||e'_k||)
// by Cauchy-Schwarz; ||dz||, ||z~|| are computed per row and ||de_k||, ||e'_k|| per code from the
// actual rounding errors.  Products of two bf16 are exact in fp32.  The fp32 accumulation of the
// matrix core, the roundings of the centring and of ||e'_k||^2 are covered by g_f = (D + 24) 2^-21
// times the magnitudes on the filter's side, (||z'|| + ||e'_k||)^2 (this allows the matrix core 7 ulp
// per accumulated product); the roundings of the fp32 chain itself by g_o = (D + 4) 2^-24 times
// (||z|| + ||e_k||)^2, uncentred.  All of it is linear in five per-code numbers with per-row
// coefficients,
//   delta[n,k] = const_n + c1 ||de_k|| + c2 ||e'_k|| + c3 ||e'_k||^2 + c4 ||e_k|| + c5 ||e_k||^2,
// so the matrix core subtracts the code-dependent part itself: five more slots of the extra k-step
// hold the per-code numbers against (-c1 .. -c5), all rounded up in magnitude to bf16, and a key
// is a LOWER bound L[n,k] of d + c_n - const_n.  With k^ the code of the smallest key, every code
// that can be the fp32 argmin has
//   L[n,k] <= L[n,k^] + 2 (delta[n,k^] - const_n) + 2 const_n        (+ the key truncation),
// so a code pays for its own norms only and the threshold for the best code's.
// ||e'_k||^2 itself enters the contraction as three bf16 terms times 1.0 (exactly the fp32 number
// the preparation computed), and c_n >= ||z'||^2 + max_k delta[n,k] (rounded up to bf16, times
// 1.0) keeps every key positive, so that keys order as unsigned integers and as floats alike.
#include "vq_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int FQ_STAGE_TILES = 4;            // 32-code tiles per LDS stage (128 codes)
constexpr int FQ_WG_ROWS = 512;              // 8 waves x 2 column tiles x 32 latents
constexpr uint32_t FQ_KEY_MASK = 0xFFFFF000u;  // 12 low bits carry (tile << 4) | register
constexpr int FQ_MAX_TILES = 256;            // => K <= 8192 per launch
constexpr float FQ_PAD_NORM = 1.0e38f;       // squared norm of padding codes (finite, and finite after adding c_n)
constexpr float FQ_TINY_NORM = 1.0e-30f;     // rows with ||z||^2 below this take the exact kernel

__device__ __forceinline__ uint32_t bf16_pair(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
// smallest bf16 value >= x, as a float (x >= 0, finite)
__device__ __forceinline__ float bf16_up(float x) { return __uint_as_float((__float_as_uint(x) + 0xffffu) & 0xffff0000u); }
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

__host__ __device__ inline int fq_steps(int D) { return D / 16 + 1; }
__host__ __device__ inline int fq_tiles(int K) { return (K + 32 * FQ_STAGE_TILES - 1) / (32 * FQ_STAGE_TILES) * FQ_STAGE_TILES; }

// Codebook split of a filter launch that has too few row workgroups to fill the chip: the sweep is
// divided over blockIdx.y into S parts of sps stages each (at least two stages per part); every part
// leaves its own key lists and vqf_merge_kernel folds them into the lists an undivided sweep would
// have produced.  Evaluated on the host (first pass: the row count is known) and on the device
// (second pass: the row count is a counter in device memory) with the same result.
constexpr int FQ_MAX_SPLIT = 8;          // first pass: the work buffer holds up to this many key lists per row
#ifndef FQ_LIST_SPLIT
#define FQ_LIST_SPLIT 16   // measured at 64 Ki rows: 8 parts 0.549 / 0.664 ms per training / eval step, 16: 0.528 / 0.642, 32: 0.546 / 0.728
#endif
constexpr int FQ_MAX_SPLIT_LIST = FQ_LIST_SPLIT;    // second pass (a list of rows): rows x parts must stay within N lists
// parts wanted for a launch of `wgs` row workgroups, before the codebook's size limits it
__host__ __device__ inline int fq_split_wanted(int64_t wgs, int cap = FQ_MAX_SPLIT) {
  const int s = wgs >= 192 ? 1 : (int)(256 / (wgs > 0 ? wgs : 1));
  return s > cap ? cap : s;
}
// the second pass's limit: its lists live in a region sized for N of them, and at most N/8 rows reach it
__host__ __device__ inline int fq_list_cap(int64_t nrows, int64_t N) {
  const int64_t c = nrows > 0 ? N / nrows : FQ_MAX_SPLIT_LIST;
  return c > FQ_MAX_SPLIT_LIST ? FQ_MAX_SPLIT_LIST : (c < 1 ? 1 : (int)c);
}
__host__ __device__ inline void fq_split(int64_t wgs, int nstages, int& S, int& sps, int cap = FQ_MAX_SPLIT) {
  int s = fq_split_wanted(wgs, cap);
  if (s > nstages / 2) s = nstages / 2;
  if (s < 1) s = 1;
  sps = (nstages + s - 1) / s;
  S = (nstages + sps - 1) / sps;   // <= s
}

// ---------------------------------------------------------------------------------------
// mu[c] = mean over the codes of E[k][c], in two steps with a fixed summation order: 32 row
// ranges x D/64 channel groups of partial sums (coalesced 256-byte row segments), then one thread
// per channel adds the 32 partials.  Any mu is valid for the filter; this one is reproducible.
// ---------------------------------------------------------------------------------------
constexpr int FQ_MEAN_PARTS = 32;

__global__ __launch_bounds__(256) void vqf_mean_partial_kernel(const float* __restrict__ E, int K, int D,
                                                               float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (K + FQ_MEAN_PARTS - 1) / FQ_MEAN_PARTS;
  const int k0 = blockIdx.y * per, k1 = k0 + per < K ? k0 + per : K;
  float acc = 0.0f;
  if (c < D)
    for (int k = k0 + r; k < k1; k += 4) acc += E[(size_t)k * D + c];
  part[r][cl] = acc;
  __syncthreads();
  if (r == 0 && c < D) partial[blockIdx.y * D + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

__global__ __launch_bounds__(256) void vqf_mean_final_kernel(const float* __restrict__ partial, int K, int D,
                                                             float* __restrict__ mu) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float acc = 0.0f;
  for (int p = 0; p < FQ_MEAN_PARTS; ++p) acc += partial[p * D + c];
  mu[c] = acc / (float)K;
}

// ---------------------------------------------------------------------------------------
// bf16 codebook image: [tile][step][lane] fragments of 8 bf16.  With e' = fl(e - mu): lane (i, g)
// of step s holds -2 e~'[32 tile + i][16 s + 8 g + 0..7]; the last step holds (n2_hi, n2_mid, n2_lo,
// 1, ||de||, ||e'||, ||e'||^2, ||e||) in the g = 0 lanes and (||e||^2, 0...) in the g = 1 lanes, where
// n2 = ||e'||^2 as summed here and the five norms are rounded up to bf16; they are also stored as
// floats in aux[2 code], aux[2 code + 1] for the verification.  hdr[0..4] = their maxima (bit
// patterns, atomicMax).
// ---------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void vqf_prep_kernel(const float* __restrict__ E, const float* __restrict__ e2, int K,
                                                      const float* __restrict__ mu, u32x4* __restrict__ image,
                                                      uint32_t* __restrict__ hdr, f32x4* __restrict__ aux) {
  constexpr int FS = D / 16 + 1;
  const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
  const int tile = blockIdx.x, code = tile * 32 + i;
  const bool real = code < K;
  const f32x4* row = reinterpret_cast<const f32x4*>(E + (size_t)(real ? code : 0) * D);
  const f32x4* mrow = reinterpret_cast<const f32x4*>(mu);
  u32x4* dst = image + (size_t)tile * FS * 64 + lane;
  float en = 0.0f, de = 0.0f;
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    f32x4 v0 = row[4 * s + 2 * g] - mrow[4 * s + 2 * g], v1 = row[4 * s + 2 * g + 1] - mrow[4 * s + 2 * g + 1];
    if (!real) { v0 = f32x4{0, 0, 0, 0}; v1 = f32x4{0, 0, 0, 0}; }
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t w = bf16_pair(-2.0f * v[2 * p], -2.0f * v[2 * p + 1]);
      o[p] = w;
      const float r0 = v[2 * p] - (-0.5f * bf16_lo(w)), r1 = v[2 * p + 1] - (-0.5f * bf16_hi(w));
      en = __builtin_fmaf(v[2 * p], v[2 * p], en);
      en = __builtin_fmaf(v[2 * p + 1], v[2 * p + 1], en);
      de = __builtin_fmaf(r0, r0, de);
      de = __builtin_fmaf(r1, r1, de);
    }
    dst[s * 64] = o;
  }
  en += __shfl_xor(en, 32);
  de += __shfl_xor(de, 32);
  // per-code terms of the error bound, rounded up to bf16 (padding codes: 0, they never win)
  float raw = real ? e2[code] : 0.0f;                   // ||e||^2, uncentred, as the fp32 chain uses it
  if (!(raw <= VQ_NORM_LIMIT)) raw = VQ_NORM_LIMIT;     // non-finite codebooks take the exact kernel anyway
  const float pde = real ? bf16_up(__builtin_sqrtf(de) * 1.0001f) : 0.0f;
  const float pen = real ? bf16_up(__builtin_sqrtf(en) * 1.0001f) : 0.0f;
  const float pe2 = real ? bf16_up(en * 1.0001f) : 0.0f;
  const float pun = bf16_up(__builtin_sqrtf(raw) * 1.0001f);
  const float pu2 = bf16_up(raw * 1.0001f);
  u32x4 o = u32x4{0, 0, 0, 0};
  if (g == 0) {
    float n2 = real ? en : FQ_PAD_NORM;
    if (!(n2 <= FQ_PAD_NORM)) n2 = FQ_PAD_NORM;
    const float hi = __uint_as_float(__float_as_uint(n2) & 0xffff0000u);
    const float r1 = n2 - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float lo = r1 - mid;
    o[0] = bf16_pair(hi, mid);
    o[1] = bf16_pair(lo, 1.0f);
    o[2] = bf16_pair(pde, pen);
    o[3] = bf16_pair(pe2, pun);
    aux[2 * code] = f32x4{pde, pen, pe2, pun};
    aux[2 * code + 1] = f32x4{pu2, 0.0f, 0.0f, 0.0f};
    if (real) {
      atomicMax(&hdr[0], __float_as_uint(pde));
      atomicMax(&hdr[1], __float_as_uint(pen));
      atomicMax(&hdr[2], __float_as_uint(pe2));
      atomicMax(&hdr[3], __float_as_uint(pun));
      atomicMax(&hdr[4], __float_as_uint(pu2));
    }
  } else {
    o[0] = bf16_pair(pu2, 0.0f);
  }
  dst[(D / 16) * 64] = o;
}

struct FilterArgs {
  const float* z;
  const u32x4* image;
  const uint32_t* hdr;
  const float* mu;   // [D] centre of the codebook
  int64_t N;
  int64_t HW;
  int nstages;
  u32x4* cand;     // [rows][2][R/4] : the R smallest keys of each lane half (unused slots +inf)
  f32x4* rowaux;   // [rows][2] : const_n, c1 .. c5 of the error bound (c's as rounded to bf16)
  int32_t* control;
  // list mode: the launch covers rows row_list[0 .. control[count_word]); cand / rowaux are indexed
  // by the position in the list
  const int32_t* row_list;
  int count_word;
};

template <int D>
__device__ __forceinline__ void fq_stage_copy(__amdgpu_buffer_rsrc_t image, int stage, char* lds_buf, int wave, int lane) {
  constexpr int PIECES = FQ_STAGE_TILES * (D / 16 + 1);
  // waves 0-3 issue the DMA (one per SIMD); their partners 4-7 start the stage's MFMAs at once.  A wave takes a
  // contiguous run of the stage's 1 KiB pieces (dma_run: no vector instruction per piece)
  constexpr int NL = 4;
  constexpr int PER = (PIECES + NL - 1) / NL;
  if (wave >= NL) return;
  const int first = wave * PER;
  const int count = PIECES - first < PER ? PIECES - first : PER;
  dma_run<0, PER>(image, reinterpret_cast<float*>(lds_buf) + first * 256, (uint32_t)lane * 16u,
                  (uint32_t)stage * (uint32_t)(PIECES * 1024) + (uint32_t)first * 1024u, count);
}

// Insert the 16 keys of one finished accumulator tile into the lane's sorted R smallest.
// Keys are positive floats, so the unsigned minimum and the float median-of-three agree on their
// order; neither inserts a canonicalising move.  maskv holds FQ_KEY_MASK in a vector register so
// that mask, merge and the scalar position fit one v_and_or_b32.
template <int R>
__device__ __forceinline__ void fq_scan(const f32x16& c, uint32_t tilebits, uint32_t maskv, float (&m)[R]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    // no inline asm here: the compiler must see the read of the MFMA result to place the wait states
    const uint32_t kb = (__float_as_uint(c[r]) & maskv) | (tilebits + r);
    const float x = __uint_as_float(kb);
    const uint32_t m0b = __float_as_uint(m[0]);
    float nm[R];
    nm[0] = __uint_as_float(kb < m0b ? kb : m0b);
#pragma unroll
    for (int i = 1; i < R; ++i) nm[i] = __builtin_amdgcn_fmed3f(m[i - 1], m[i], x);
#pragma unroll
    for (int i = 0; i < R; ++i) m[i] = nm[i];
  }
}

// R (4 or 8) = keys kept per lane.  The first pass runs R = 4; rows whose list overflowed are
// filtered again with R = 8 (list mode).
#ifndef FQ_CT
#define FQ_CT 2   // 32-latent column tiles per wave (4 = four waves per workgroup, one per SIMD: measured 17 % slower, the scan loses the SIMD partner whose MFMAs it hides under)
#endif
template <int D, bool NCHW, int R>
__global__ __launch_bounds__(1024 / FQ_CT, FQ_CT == 2 ? 2 : 1) void vqf_filter_kernel(FilterArgs a) {
  constexpr int CT = FQ_CT;
  constexpr int ZS = D / 16, FS = ZS + 1;
  constexpr int STAGE_B = FQ_STAGE_TILES * FS * 1024;
  extern __shared__ __attribute__((aligned(16))) char flds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * FQ_WG_ROWS + wave * (32 * CT);
  if (a.control[VQ_CTL_CB_FLAG] != 0) return;  // the verify kernel flags every row
  const bool listed = a.row_list != nullptr;
  int64_t nrows = a.N;
  if (listed) {
    nrows = a.control[a.count_word];
    if (nrows * 8 > a.N) return;  // too many unsettled rows: the fp32 kernel redoes the batch
    if (nrows <= VQ_FILTER_FEW_ROWS) return;  // a handful: the per-row kernel takes them directly
  }
  if ((int64_t)blockIdx.x * FQ_WG_ROWS >= nrows) return;  // list mode: the grid covers the worst case
  int nsplit, sps;
  fq_split((nrows + FQ_WG_ROWS - 1) / FQ_WG_ROWS, a.nstages, nsplit, sps, listed ? fq_list_cap(nrows, a.N) : FQ_MAX_SPLIT);
  const int part = blockIdx.y;
  if (part >= nsplit) return;
  const int st_begin = part * sps;
  const int st_end = st_begin + sps < a.nstages ? st_begin + sps : a.nstages;

  // the codebook mean, through LDS (behind the two stage buffers); the barrier comes before the
  // first DMA is issued, so that it does not wait for that copy
  float* mus = reinterpret_cast<float*>(flds + 2 * STAGE_B);
  for (int c = threadIdx.x; c < D; c += 1024 / CT) mus[c] = a.mu[c];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.image), 0, (int)((size_t)a.nstages * FQ_STAGE_TILES * FS * 1024), 0x00020000);
  fq_stage_copy<D>(img, st_begin, flds, wave, lane);   // lands while z is being loaded

  float hmax[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) hmax[i] = __uint_as_float(a.hdr[i]);

  u32x4 zb[CT][FS];
  uint32_t zmax_bits = 0;  // max |z| (bit pattern), for the EMA scatter's fixed-point scale
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int64_t n = row0 + ct * 32 + j;
    if (n >= nrows) n = nrows - 1;
    if (listed) n = a.row_list[n];
    float zt2 = 0.0f, dz2 = 0.0f, zu2 = 0.0f;
    uint32_t am = 0;
#pragma unroll
    for (int s = 0; s < ZS; ++s) {
      // keep at most four steps' loads in flight: more would not fit the register file
      if ((s & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      const f32x4* mp = reinterpret_cast<const f32x4*>(mus + 16 * s + 8 * g);
      const f32x4 m0 = mp[0], m1 = mp[1];
      const float mc[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
      float v[8];
      if (NCHW) {
        const int64_t b = n / a.HW, hw = n - b * a.HW;
        const float* src = a.z + (b * D + 16 * s + 8 * g) * a.HW + hw;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[(int64_t)i * a.HW];
      } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.z + n * D + 16 * s + 8 * g);
        const f32x4 x0 = src[0], x1 = src[1];
        v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
        v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t ab = __float_as_uint(v[i]) & 0x7fffffffu;
        am = ab > am ? ab : am;
        zu2 = __builtin_fmaf(v[i], v[i], zu2);
        v[i] = v[i] - mc[i];                               // centred
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const uint32_t w = bf16_pair(v[2 * p], v[2 * p + 1]);
        zb[ct][s][p] = w;
        const float t0 = bf16_lo(w), t1 = bf16_hi(w);
        const float r0 = v[2 * p] - t0, r1 = v[2 * p + 1] - t1;
        zt2 = __builtin_fmaf(t0, t0, zt2);
        zt2 = __builtin_fmaf(t1, t1, zt2);
        dz2 = __builtin_fmaf(r0, r0, dz2);
        dz2 = __builtin_fmaf(r1, r1, dz2);
      }
      // the running maximum is pinned here: left to itself the compiler defers the whole max
      // reduction to the end of the loop and keeps (spills) every raw value until then
      asm volatile("" : "+v"(am));
    }
    zmax_bits = am > zmax_bits ? am : zmax_bits;
    zt2 += __shfl_xor(zt2, 32);
    dz2 += __shfl_xor(dz2, 32);
    zu2 += __shfl_xor(zu2, 32);
    // coefficients of the error bound for this row (see the file comment)
    const float nz = __builtin_sqrtf(zt2) * 1.0001f, ndz = __builtin_sqrtf(dz2) * 1.0001f;
    const float zc = nz + ndz;                                              // >= ||z'||
    const float zn = __builtin_sqrtf(zu2) * 1.0001f;                        // >= ||z||
    const float gf = (float)(D + 24) * 4.76837158203125e-07f * 1.01f;       // (D + 24) 2^-21
    const float go = (float)(D + 4) * 5.9604644775390625e-08f * 1.01f;      // (D + 4) 2^-24
    const float c1 = bf16_up(2.0f * nz * 1.0001f);                          // times ||de_k||
    const float c2 = bf16_up((2.0f * ndz + 2.02f * gf * zc) * 1.0001f);     // times ||e'_k||
    const float c3 = bf16_up(1.01f * gf);                                   // times ||e'_k||^2
    const float c4 = bf16_up(2.0f * go * zn * 1.0001f);                     // times ||e_k||
    const float c5 = bf16_up(go * 1.0001f);                                 // times ||e_k||^2
    // largest code-dependent part over the codebook
    const float wmax = (c1 * hmax[0] + c2 * hmax[1] + c3 * hmax[2] + c4 * hmax[3] + c5 * hmax[4]) * 1.0001f;
    float cst = (gf * (1.01f * zc * zc + 3.1f * wmax) + go * zn * zn * 1.0001f + 1.0e-30f + nz * 1.0e-35f) * 1.001f;
    // rows that are not finite, or so small that bf16 would flush them: a NaN constant sends them
    // to the exact kernel (the verification tests its threshold for NaN)
    if (!(zu2 <= VQ_NORM_LIMIT) || zu2 < FQ_TINY_NORM) cst = __builtin_nanf("");
    {
      // written now rather than after the sweep: 16 registers less to carry through it
      const int64_t pos = row0 + ct * 32 + j;
      if (g == 0 && pos < nrows && part == 0) {
        a.rowaux[2 * pos] = f32x4{cst, c1, c2, c3};
        a.rowaux[2 * pos + 1] = f32x4{c4, c5, 0.0f, 0.0f};
      }
    }
    const float cup = bf16_up(zc * zc * 1.0001f + (2.0f * wmax + cst) * 1.0001f);  // keeps every key positive
    u32x4 extra = u32x4{0, 0, 0, 0};
    if (g == 0) {
      extra[0] = bf16_pair(1.0f, 1.0f);
      extra[1] = bf16_pair(1.0f, cup);
      extra[2] = bf16_pair(-c1, -c2);
      extra[3] = bf16_pair(-c3, -c4);
    } else {
      extra[0] = bf16_pair(-c5, 0.0f);
    }
    zb[ct][ZS] = extra;
  }
  if (!listed && part == 0) {  // the first level sees every row of the batch
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)zmax_bits, off);
      zmax_bits = o > zmax_bits ? o : zmax_bits;
    }
    if (lane == 0 && zmax_bits) atomicMax(reinterpret_cast<unsigned int*>(&a.control[VQ_CTL_ZMAX]), zmax_bits);
  }

  uint32_t maskv = FQ_KEY_MASK;
  asm volatile("" : "+v"(maskv));
  float m[CT][R];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < R; ++i) m[ct][i] = __builtin_inff();

  __syncthreads();  // stage 0 has landed

  int cur = 0;
  for (int st = st_begin; st < st_end; ++st) {
    char* buf = flds + cur * STAGE_B;
    if (st + 1 < st_end) fq_stage_copy<D>(img, st + 1, flds + (cur ^ 1) * STAGE_B, wave, lane);
    const u32x4* A = reinterpret_cast<const u32x4*>(buf) + lane;
#pragma unroll
    for (int t = 0; t < FQ_STAGE_TILES; ++t) {
      f32x16 c[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[ct][i] = 0.0f;
#pragma unroll
      for (int s = 0; s < FS; ++s) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, A[(t * FS + s) * 64]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          c[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, zb[ct][s]), c[ct], 0, 0, 0);
      }
      const uint32_t tilebits = (uint32_t)(st * FQ_STAGE_TILES + t) << 4;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) fq_scan<R>(c[ct], tilebits, maskv, m[ct]);
    }
    __syncthreads();  // every wave is done with buf; the DMA of stage st+1 has landed
    cur ^= 1;
  }

#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int64_t pos = row0 + ct * 32 + j;   // cand / rowaux are indexed by position (= row unless listed)
    if (pos < nrows) {
#pragma unroll
      for (int q = 0; q < R / 4; ++q)
        a.cand[((pos * nsplit + part) * 2 + g) * (R / 4) + q] = u32x4{__float_as_uint(m[ct][4 * q]), __float_as_uint(m[ct][4 * q + 1]),
                                                   __float_as_uint(m[ct][4 * q + 2]), __float_as_uint(m[ct][4 * q + 3])};
    }
  }
}

// Folds the key lists of a divided sweep: for every (row, lane half) the R smallest of the parts'
// S x R keys, sorted -- exactly the list one sweep over the whole codebook leaves (the R smallest
// of a union are among the R smallest of each part).
template <int R>
__global__ __launch_bounds__(256) void vqf_merge_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t N,
                                                        int nstages, const int32_t* __restrict__ control, int count_word) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t pos = t >> 1;
  const int g = (int)(t & 1);
  int64_t nrows = N;
  if (count_word >= 0) {
    nrows = control[count_word];
    if (nrows * 8 > N || nrows <= VQ_FILTER_FEW_ROWS) return;
  }
  if (control[VQ_CTL_CB_FLAG] != 0 || pos >= nrows) return;
  int nsplit, sps;
  fq_split((nrows + FQ_WG_ROWS - 1) / FQ_WG_ROWS, nstages, nsplit, sps, count_word >= 0 ? fq_list_cap(nrows, N) : FQ_MAX_SPLIT);
  uint32_t m[R];
#pragma unroll
  for (int i = 0; i < R; ++i) m[i] = 0x7f800000u;
  for (int y = 0; y < nsplit; ++y) {
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      const u32x4 v = in[((pos * nsplit + y) * 2 + g) * (R / 4) + q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t x = v[j];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const uint32_t lo = m[i] < x ? m[i] : x;
          x = m[i] < x ? x : m[i];
          m[i] = lo;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < R / 4; ++q)
    out[(pos * 2 + g) * (R / 4) + q] = u32x4{m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]};
}

struct VerifyArgs {
  const float* z;
  const float* E;
  const float* e2;
  const u32x4* cand;
  const f32x4* rowaux;
  const f32x4* aux;      // per-code norms as the filter used them: [2 k] = (||de||, ||e'||, ||e'||^2, ||e||), [2 k + 1] = (||e||^2, ..)
  int64_t N;
  int64_t HW;
  int K;
  int D;
  float T;
  int64_t code_base;
  int64_t* idx64;
  int32_t* idx32;
  float* dmin;
  int32_t* nf_rows;    // rows for the exact scalar kernel (not finite, degenerate, flagged codebook)
  int32_t* ovf_rows;   // finite rows whose candidate list may be incomplete: next level
  int ovf_word;        // control word counting ovf_rows
  int32_t* control;
  const int32_t* row_list;  // list mode (see FilterArgs)
  int count_word;
};

__device__ __forceinline__ int fq_decode(uint32_t key, int g) {
  const int r = key & 15, tile = (key >> 4) & 255;
  return tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
}

// One thread per row: exact fp32 chain distances of the surviving candidates.  Slot 0 is the
// code of the smallest key (always a candidate); the other slots are evaluated under the lane's
// own predicate, so that codes outside the band cost no memory traffic.
template <bool NCHW, int R>
__global__ __launch_bounds__(256) void vqf_verify_kernel(VerifyArgs a) {
  constexpr int SL = 2 * (R - 1);  // candidate slots: R-1 per lane half (the R-th in band = overflow)
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool listed = a.row_list != nullptr;
  if (listed && (int64_t)a.control[a.count_word] * 8 > a.N) return;  // the fp32 kernel redoes the batch
  if (listed && a.control[a.count_word] <= VQ_FILTER_FEW_ROWS) return;  // a handful: the per-row kernel takes them directly
  if (pos >= (listed ? (int64_t)a.control[a.count_word] : a.N)) return;
  const int64_t n = listed ? (int64_t)a.row_list[pos] : pos;
  bool exact_row = a.control[VQ_CTL_CB_FLAG] != 0;
  bool overflow = false;
  uint32_t kh[2][R];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      const u32x4 v = a.cand[(pos * 2 + g) * (R / 4) + q];
#pragma unroll
      for (int i = 0; i < 4; ++i) kh[g][4 * q + i] = v[i];
    }
  const f32x4 ra = a.rowaux[2 * pos], rb = a.rowaux[2 * pos + 1];
  const float v0 = __uint_as_float(kh[0][0] & FQ_KEY_MASK), v1 = __uint_as_float(kh[1][0] & FQ_KEY_MASK);
  const int g0 = v1 < v0 ? 1 : 0;                 // half that holds the smallest key
  const float mn = g0 ? v1 : v0;
  // threshold: twice the best code's own error terms plus twice the row constant (file comment);
  // key truncation: value in (L (1 - 2^-11), L]
  int kbest = fq_decode(g0 ? kh[1][0] : kh[0][0], g0);
  if (kbest >= a.K) kbest = 0;                    // garbage keys (flagged codebook): the row goes to the exact kernel
  const f32x4 ca = a.aux[2 * kbest], cb = a.aux[2 * kbest + 1];
  const float wbest = (ra[1] * ca[0] + ra[2] * ca[1] + ra[3] * ca[2] + rb[0] * ca[3] + rb[1] * cb[0]) * 1.0001f;
  const float thr = mn * 1.0009765625f + 2.0f * (wbest + ra[0]) * 1.0001f;
  if (!(thr < __builtin_inff()) || !(mn >= 0.0f)) exact_row = true;  // also catches NaN
  int code[SL];
  bool ok[SL];
  // slot order: (best half, 0), (other half, 0), (best half, 1), (other half, 1), ...
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int g = q ? 1 - g0 : g0;
      const uint32_t key = g ? kh[1][i] : kh[0][i];
      const int c = fq_decode(key, g);
      code[2 * i + q] = c;
      ok[2 * i + q] = __uint_as_float(key & FQ_KEY_MASK) <= thr && c < a.K;
    }
#pragma unroll
  for (int g = 0; g < 2; ++g)
    if (__uint_as_float(kh[g][R - 1] & FQ_KEY_MASK) <= thr) overflow = true;  // the list may be incomplete
  if (!ok[0]) exact_row = true;
  if (exact_row || overflow) {
    // -2: the exact kernel's row, -1: next level's row (the gather skips both)
    if (a.idx64) a.idx64[n] = exact_row ? -2 : -1;
    a.idx32[n] = exact_row ? -2 : -1;
    if (exact_row) a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
    else a.ovf_rows[atomicAdd(&a.control[a.ovf_word], 1)] = (int32_t)n;
    return;
  }
  bool several = false;
#pragma unroll
  for (int s = 1; s < SL; ++s) several = several || ok[s];
  if (!several && a.dmin == nullptr) {
    // a single candidate is the argmin; without a distance to report the row needs no arithmetic
    if (a.idx64) a.idx64[n] = (int64_t)code[0] + a.code_base;
    a.idx32[n] = (int32_t)(code[0] + a.code_base);
    return;
  }

  float acc[SL];
#pragma unroll
  for (int s = 0; s < SL; ++s) acc[s] = 0.0f;
  float pe = 0.0f, po = 0.0f;
  const float* zrow;
  int64_t zst;
  if (NCHW) { const int64_t b = n / a.HW, hw = n - b * a.HW; zrow = a.z + b * a.D * a.HW + hw; zst = a.HW; }
  else { zrow = a.z + n * a.D; zst = 1; }
  const f32x4* erow[SL];
#pragma unroll
  for (int s = 0; s < SL; ++s) erow[s] = reinterpret_cast<const f32x4*>(a.E + (size_t)(ok[s] ? code[s] : code[0]) * a.D);
  // 16 channels per step: a lane's four 16-byte reads of a code row are issued back to back, so
  // the 64-byte sector fetched by the first serves the other three (rows of different lanes share
  // nothing, so a sector used once would cost four times the traffic).
  const int n16 = a.D / 16;
  float zn[16];  // the next step's latent values are in flight while this step computes
#pragma unroll
  for (int i = 0; i < 16; ++i) zn[i] = zrow[(int64_t)i * zst];
  for (int c16 = 0; c16 < n16; ++c16) {
    float zv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) zv[i] = zn[i];
    const int cn = c16 + 1 < n16 ? c16 + 1 : c16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zn[i] = zrow[(int64_t)(16 * cn + i) * zst];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      pe = __builtin_fmaf(zv[i], zv[i], pe);
      po = __builtin_fmaf(zv[i + 1], zv[i + 1], po);
    }
    {
      f32x4 e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = erow[0][4 * c16 + q];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0] = __builtin_fmaf(zv[i], e[i >> 2][i & 3], acc[0]);
    }
#pragma unroll
    for (int s = 1; s < SL; ++s) {
      if (ok[s]) {
        f32x4 e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = erow[s][4 * c16 + q];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s] = __builtin_fmaf(zv[i], e[i >> 2][i & 3], acc[s]);
      }
    }
  }
  const float z2 = pe + po;
  if (!(z2 <= VQ_NORM_LIMIT) || z2 < FQ_TINY_NORM) exact_row = true;   // (the filter flags these rows already)
  float best = __builtin_inff();
  int bi = 0x7fffffff;
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    float d = __builtin_fmaf(-2.0f, acc[s], z2 + a.e2[ok[s] ? code[s] : code[0]]);   // unused slots may hold any code
    if (a.T != 1.0f) d = d / a.T;
    const bool take = ok[s] && (d < best || (d == best && code[s] < bi));
    best = take ? d : best;
    bi = take ? code[s] : bi;
  }
  if (exact_row) {
    if (a.idx64) a.idx64[n] = -2;
    a.idx32[n] = -2;
    a.nf_rows[atomicAdd(&a.control[VQ_CTL_NF_COUNT], 1)] = (int32_t)n;
  } else {
    if (a.idx64) a.idx64[n] = (int64_t)bi + a.code_base;
    a.idx32[n] = (int32_t)(bi + a.code_base);
    if (a.dmin) a.dmin[n] = best;
  }
}

template <int D, bool NCHW, int R>
int launch_filter(const FilterArgs& a, hipStream_t s) {
  constexpr int STAGE_B = FQ_STAGE_TILES * (D / 16 + 1) * 1024;
  constexpr int lds_bytes = 2 * STAGE_B + D * 4;
  auto kern = vqf_filter_kernel<D, NCHW, R>;
  const int arc = vq_lds_ready<vqf_filter_kernel<D, NCHW, R>, lds_bytes>();
  if (arc != VQ_OK) return arc;
  // first pass: every row, the split known here; second pass (list mode): at most N/8 rows reach it,
  // and the kernel derives its split from the counter, so the grid covers the largest one
  const int64_t rows = a.row_list ? (a.N + 7) / 8 : a.N;
  const int64_t blocks = (rows + FQ_WG_ROWS - 1) / FQ_WG_ROWS;
  int nsplit = 1, sps = a.nstages;
  if (a.row_list) nsplit = a.nstages / 2 < FQ_MAX_SPLIT_LIST ? (a.nstages / 2 > 0 ? a.nstages / 2 : 1) : FQ_MAX_SPLIT_LIST;
  else fq_split(blocks, a.nstages, nsplit, sps);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)nsplit), dim3(1024 / FQ_CT), lds_bytes, s, a);
  return vq_hip_status(hipGetLastError());
}

template <int R>
int dispatch_filter(const FilterArgs& f, int D, bool nchw, hipStream_t s) {
  switch (D) {
    case 64: return nchw ? launch_filter<64, true, R>(f, s) : launch_filter<64, false, R>(f, s);
    case 128: return nchw ? launch_filter<128, true, R>(f, s) : launch_filter<128, false, R>(f, s);
    default: return nchw ? launch_filter<256, true, R>(f, s) : launch_filter<256, false, R>(f, s);
  }
}

template <int R>
int launch_verify(const VerifyArgs& v, bool nchw, hipStream_t s) {
  const unsigned vb = (unsigned)((v.N + 255) / 256);
  if (nchw) hipLaunchKernelGGL((vqf_verify_kernel<true, R>), dim3(vb), dim3(256), 0, s, v);
  else hipLaunchKernelGGL((vqf_verify_kernel<false, R>), dim3(vb), dim3(256), 0, s, v);
  return vq_hip_status(hipGetLastError());
}

template <int R>
int launch_merge(const u32x4* in, u32x4* out, int64_t N, int64_t rows_max, int nstages, const int32_t* control,
                 int count_word, hipStream_t s) {
  const unsigned mb = (unsigned)((2 * rows_max + 255) / 256);
  hipLaunchKernelGGL((vqf_merge_kernel<R>), dim3(mb), dim3(256), 0, s, in, out, N, nstages, control, count_word);
  return vq_hip_status(hipGetLastError());
}

// rows of the first pass's divided sweep need S lists each: they go behind the fixed regions
inline size_t fq_split_extra_bytes(int64_t N) {
  const int S = fq_split_wanted((N + FQ_WG_ROWS - 1) / FQ_WG_ROWS);   // the largest split any codebook allows for this N
  return S > 1 ? (size_t)((N + 3) & ~(int64_t)3) * 32 * S : 0;
}

}  // namespace

bool vq_filter_supported(int K, int D) { return (D == 64 || D == 128 || D == 256) && K > 0 && fq_tiles(K) <= FQ_MAX_TILES; }

extern "C" {

size_t vq_filter_image_bytes(int K, int D) {
  if (!vq_filter_supported(K, D)) return 256;
  // image, header, per-code terms (8 floats), codebook mean and its partial sums
  return (size_t)fq_tiles(K) * fq_steps(D) * 1024 + 256 + (size_t)fq_tiles(K) * 32 * 32 + (size_t)D * 4 * (1 + FQ_MEAN_PARTS);
}


int vq_filter_prep(const float* E, const float* e2, int K, int D, void* fimage, void* stream) {
  if (!E || !e2 || !fimage) return VQ_ERR_INVALID_ARG;
  if (!vq_filter_supported(K, D)) return VQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int tiles = fq_tiles(K);
  u32x4* image = static_cast<u32x4*>(fimage);
  uint32_t* hdr = reinterpret_cast<uint32_t*>(static_cast<char*>(fimage) + (size_t)tiles * fq_steps(D) * 1024);
  f32x4* aux = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(hdr) + 256);
  float* mu = reinterpret_cast<float*>(reinterpret_cast<char*>(aux) + (size_t)tiles * 32 * 32);
  VQ_CHECK_HIP(hipMemsetAsync(hdr, 0, 256, s));
  float* mu_partial = mu + D;
  hipLaunchKernelGGL(vqf_mean_partial_kernel, dim3((D + 63) / 64, FQ_MEAN_PARTS), dim3(256), 0, s, E, K, D, mu_partial);
  hipLaunchKernelGGL(vqf_mean_final_kernel, dim3((D + 255) / 256), dim3(256), 0, s, (const float*)mu_partial, K, D, mu);
  switch (D) {
    case 64: hipLaunchKernelGGL(vqf_prep_kernel<64>, dim3(tiles), dim3(64), 0, s, E, e2, K, mu, image, hdr, aux); break;
    case 128: hipLaunchKernelGGL(vqf_prep_kernel<128>, dim3(tiles), dim3(64), 0, s, E, e2, K, mu, image, hdr, aux); break;
    default: hipLaunchKernelGGL(vqf_prep_kernel<256>, dim3(tiles), dim3(64), 0, s, E, e2, K, mu, image, hdr, aux); break;
  }
  return vq_hip_status(hipGetLastError());
}

// Work buffer: [keys of pass 1: N x 32][row terms: N x 32][list A: N x 4][keys of pass 2: N x 64][row terms: N x 32][list B: N x 4]
// (regions laid out for N rounded up to a multiple of 4, which keeps each 16-byte aligned), then, for
// batches small enough that the first pass divides its sweep, the S key lists per row of that pass.
// How a filter pass over `rows` rows of a codebook of K codes divides its sweep: parts[0] = number of
// parts (blockIdx.y), parts[1] = stages of 128 codes per part.  Host-side view of fq_split, exported
// so that the layout rules (every part non-empty, rows x parts within the work buffer) can be
// checked without a GPU.
int vq_filter_split(int64_t rows, int K, int* parts) {
  if (rows < 0 || K <= 0 || !parts) return VQ_ERR_INVALID_ARG;
  int S, sps;
  fq_split((rows + FQ_WG_ROWS - 1) / FQ_WG_ROWS, fq_tiles(K) / FQ_STAGE_TILES, S, sps);
  parts[0] = S;
  parts[1] = sps;
  return VQ_OK;
}

size_t vq_filter_work_bytes(int64_t N) {
  return N > 0 ? (size_t)((N + 3) & ~(int64_t)3) * 168 + fq_split_extra_bytes(N) : 256;
}

// Internal: filter + verify, two levels.  Settled rows get idx / dmin; rows for the exact kernel
// are appended to nf_rows (control[VQ_CTL_NF_COUNT]); rows that overflowed both levels are left
// in *hard_rows_out (control[VQ_CTL_OVF2_COUNT]) for the per-row kernel.  Unsettled rows hold -1.
int vq_filter_verify_launch(const float* z, int nchw, int64_t N, int D, int64_t HW, const float* E, const float* e2,
                            const void* fimage, int K, float T, int64_t code_base, int64_t* idx64, int32_t* idx32,
                            float* dmin, int32_t* nf_rows, int32_t** open_rows_out, int32_t** hard_rows_out,
                            int32_t* control, void* work, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int tiles = fq_tiles(K);
  char* w = static_cast<char*>(work);
  const size_t NP = (size_t)((N + 3) & ~(int64_t)3);   // region stride: keeps every region 16-byte aligned
  int32_t* listA = reinterpret_cast<int32_t*>(w + NP * 64);
  int32_t* listB = reinterpret_cast<int32_t*>(w + NP * 164);
  *open_rows_out = listA;
  *hard_rows_out = listB;
  FilterArgs f{};
  f.z = z; f.image = static_cast<const u32x4*>(fimage);
  f.hdr = reinterpret_cast<const uint32_t*>(static_cast<const char*>(fimage) + (size_t)tiles * fq_steps(D) * 1024);
  f.mu = reinterpret_cast<const float*>(reinterpret_cast<const char*>(f.hdr) + 256 + (size_t)tiles * 32 * 32);
  f.N = N; f.HW = nchw ? HW : 1; f.nstages = tiles / FQ_STAGE_TILES;
  u32x4* keys1 = reinterpret_cast<u32x4*>(w);
  int split1, sps1;
  fq_split((N + FQ_WG_ROWS - 1) / FQ_WG_ROWS, f.nstages, split1, sps1);
  f.cand = split1 > 1 ? reinterpret_cast<u32x4*>(w + NP * 168) : keys1;
  f.rowaux = reinterpret_cast<f32x4*>(w + NP * 32);
  f.control = control;
  int rc = dispatch_filter<4>(f, D, nchw != 0, s);
  if (rc != VQ_OK) return rc;
  if (split1 > 1) {
    rc = launch_merge<4>(f.cand, keys1, N, N, f.nstages, control, -1, s);
    if (rc != VQ_OK) return rc;
  }
  VerifyArgs v{};
  v.z = z; v.E = E; v.e2 = e2; v.cand = keys1; v.rowaux = f.rowaux; v.N = N; v.HW = f.HW; v.K = K; v.D = D; v.T = T;
  v.code_base = code_base; v.idx64 = idx64; v.idx32 = idx32; v.dmin = dmin; v.nf_rows = nf_rows; v.control = control;
  v.aux = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(f.hdr) + 256);
  v.ovf_rows = listA; v.ovf_word = VQ_CTL_OVF_COUNT;
  rc = launch_verify<4>(v, nchw != 0, s);
  if (rc != VQ_OK) return rc;
  // second level over list A: 8 keys per lane
  f.cand = reinterpret_cast<u32x4*>(w + NP * 68);
  f.rowaux = reinterpret_cast<f32x4*>(w + NP * 132);
  f.row_list = listA; f.count_word = VQ_CTL_OVF_COUNT;
  rc = dispatch_filter<8>(f, D, nchw != 0, s);
  if (rc != VQ_OK) return rc;
  // its lists (one per part of the sweep; rows x parts <= N) are folded into the unused tail of the
  // second pass's row-term region (at most N/8 rows reach this pass: row terms N x 4 bytes, folded
  // lists N x 8 bytes of the region's N x 32)
  u32x4* keys2 = reinterpret_cast<u32x4*>(w + NP * 140);
  rc = launch_merge<8>(f.cand, keys2, N, (N + 7) / 8, f.nstages, control, VQ_CTL_OVF_COUNT, s);
  if (rc != VQ_OK) return rc;
  v.cand = keys2; v.rowaux = f.rowaux; v.row_list = listA; v.count_word = VQ_CTL_OVF_COUNT;
  v.ovf_rows = listB; v.ovf_word = VQ_CTL_OVF2_COUNT;
  return launch_verify<8>(v, nchw != 0, s);
}

}  // extern "C"
