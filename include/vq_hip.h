/*
 * C ABI of libvqhip.so: the MI355X (gfx950) vector-quantizer hot path.
 *
 * Every entry point takes plain device pointers and sizes, enqueues its kernels on the
 * HIP stream it is given (a hipStream_t passed as void*; NULL = the default stream),
 * never allocates, never synchronises and never throws.  The return value is a status:
 * 0 = VQ_OK, VQ_ERR_* for argument errors, VQ_ERR_HIP_BASE + hipError_t for runtime
 * errors.  All tensors are fp32, indices int64 (plus an int32 working copy), one
 * caller thread per device.
 *
 * "ref:" comments name the lines of the reference (heimaoqqq/VQ-VAE) each entry point
 * replaces.  A training forward of the reference's EMAVectorQuantizer
 * (vqvae/ema_vector_quantizer.py:33-108) is, in order:
 *   vq_codebook_prep -> vq_search -> vq_code_histogram -> vq_ema_scatter ->
 *   vq_step_scalars -> vq_ema_cluster_update -> vq_ema_codebook_update
 */
#ifndef VQ_HIP_H
#define VQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  VQ_OK = 0,
  VQ_ERR_INVALID_ARG = 1,   /* null pointer, non-positive size, misaligned buffer          */
  VQ_ERR_UNSUPPORTED = 2,   /* shape outside what the kernels handle (D > 4096, ...)       */
  VQ_ERR_TOO_LARGE = 3,     /* size beyond the fixed maximum of an entry point without scratch */
  VQ_ERR_HIP_BASE = 1000    /* + hipError_t                                                */
};

/* Layout of the latent tensor z (and of the quantized output). */
enum {
  VQ_LAYOUT_ND = 0,   /* [N][D] row major (ref: vqvae/trainers/vqtrainer.py:60 reshapes to this) */
  VQ_LAYOUT_NCHW = 1  /* [B][D][HW]; row n = b*HW + hw. Fuses the permute+contiguous of
                         ref: vqvae/ema_vector_quantizer.py:35-36 and :98 into the loads/stores */
};

int vq_version(void);
const char* vq_status_string(int status);

/*
 * Optional one-time set-up of a device, for callers that must keep the first call free of set-up
 * work.  No counterpart in the reference.  For GPU
 * `device` it raises the dynamic-LDS limit of every kernel of the library that needs more than the
 * 64 KiB default (hipFuncSetAttribute) and looks up the device address of the library's static
 * scratch array (vq_pairwise_stats_static).  Without this call each entry point does the same for
 * its own kernels the first time it runs on a device.  The call makes `device` current while it runs
 * and restores the caller's current device; it allocates nothing and may be repeated.
 * (Recording the step's entry points into a HIP graph has been tried with this set-up done first and
 * is NOT supported: replays of a captured vq_quantize_step returned wrong indices and faulted.  Only
 * vq_pairwise_stats_static is written for use under capture.)
 */
int vq_init(int device);
/* Number of kernels vq_init prepares (entered in the library's table when it is loaded); diagnostic. */
int vq_large_lds_kernels(void);

/* ---- sizes of the caller-owned work buffers (bytes; every buffer 256-byte aligned) ---- */
size_t vq_image_bytes(int K, int D);            /* codebook image, the search's MFMA A operand     */
size_t vq_norms_bytes(int K);                   /* squared norms of the (padded) codebook          */
int64_t vq_search_blocks(int64_t N);            /* workgroups of vq_search = loss partials it writes*/
size_t vq_loss_partial_bytes(int64_t N);        /* floats: the sweep's partials + the exact kernel's */
/* Which of its two matrix-core kernels vq_search runs for this problem (both give the same bits): 1 = the
 * persistent kernel (large batches; every codebook size at D >= 128, up to 1024 codes below), 0 = one workgroup
 * per 256 rows (small batches, divided sweeps, zero-padded channel counts).  Host-side rule only; for profiles and
 * benchmarks that have to name the kernel they measure.  No counterpart in the reference. */
int vq_search_uses_persistent_kernel(int64_t N, int K, int D, int layout, int64_t HW);
size_t vq_control_bytes(void);                  /* small block of control words                    */
size_t vq_search_split_bytes(int64_t N, int K); /* scratch of the codebook split (0: this N, K is not split) */
int vq_scatter_slabs(int64_t N, int K, int D);  /* number of partial-sum slabs vq_ema_scatter writes (NCHW)*/
int vq_scatter_slabs_for(int64_t N, int K, int D, int layout, int64_t HW); /* the same for a given layout */

/*
 * Codebook preparation.  ref: vqvae/ema_vector_quantizer.py:44 (sum(E**2, dim=1)).
 * Writes e2[k] = ||E[k]||^2 for k < K (+inf for the padding up to a multiple of 64) and,
 * when D is one of 16/32/64/128/256, the codebook re-laid out in MFMA-fragment order
 * ("image").  Sets control word 1 when any norm is NaN or above 1e30 (the search then
 * takes the exact path for every row).  control must have been zeroed by the caller
 * (vq_control_reset) before the first kernel of a step.
 */
int vq_control_reset(int32_t* control, void* stream);
int vq_codebook_prep(const float* E, int K, int D, float* image, float* e2, int32_t* control, void* stream);

/*
 * Nearest-codebook search, fused with gather, straight-through output and the commitment
 * loss partial sums.  ref: vqvae/ema_vector_quantizer.py:35-58 (distances, temperature,
 * argmin, one-hot @ E as a gather), :92 (sum((q - z)^2)), :95 (z + (q - z)), :98 (layout).
 *   z            [N rows], layout as above, D channels; HW only used for NCHW
 *   E            [K][D] row major codebook (rows code_base .. code_base+K-1 of the full one)
 *   image, e2    from vq_codebook_prep for the same E
 *   temperature  distances are divided by it when != 1 (ref :49-50).  Whole-codebook sweeps compare undivided
 *                distances and divide the winner only: division by T > 0 is monotone, so the result -- index
 *                and quotient -- is that of dividing every distance, and the few rows where a code before the
 *                winner lies within rounding of it are redone with every distance divided (same bits)
 *   code_base    added to every stored index (codebook-row sharding)
 *   idx64        [N] int64 out (ref :53), may be NULL
 *   idx32        [N] int32 out, required (feeds histogram / scatter)
 *   dmin         [N] winning distance out, may be NULL
 *   q_out        quantized output in z's layout, may be NULL (then loss_partial unused)
 *   loss_partial [vq_loss_partial_bytes/4] per-workgroup sums of (q - z)^2, may be NULL
 *   nf_rows      [N] int32 scratch: rows routed to the exact path -- rows that are not finite from the front
 *                (count in control word 0), rows of a temperature sweep with a close earlier code from the
 *                end backwards (count in control word 6)
 *   control      control words (the two counters, codebook flag, max |z| for the EMA scatter)
 *   split_ws     vq_search_split_bytes(N, K) bytes of scratch, may be NULL.  With fewer than
 *                32 Ki rows the row tiles alone cannot fill 256 CUs, and up to 512 Ki rows the
 *                last round of row tiles can leave most of them idle (257 tiles: two rounds);
 *                given this scratch (and idx64) the codebook sweep is split over extra
 *                workgroups where that saves time, and merged, with identical results.
 * Ties go to the lowest index; NaN handling follows torch.argmin.
 */
int vq_search(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* image,
              const float* e2, int K, float temperature, int64_t code_base, int64_t* idx64, int32_t* idx32,
              float* dmin, float* q_out, float* loss_partial, int32_t* nf_rows, int32_t* control, void* split_ws,
              void* stream);

/*
 * Filtered search: the same outputs as vq_search -- the argmin of the fp32 distances of
 * vqvae/ema_vector_quantizer.py:42-53 with the lowest index on ties, the gather (:58), the
 * straight-through output (:95) and the loss partials (:92) -- computed in two steps: approximate
 * distances from bf16-rounded operands on the bf16 matrix cores select, under a proven error
 * bound, the few codes that can be the fp32 argmin; those are then evaluated with the exact fp32
 * chain.  Rows the bound cannot settle go through the fp32 MFMA kernel (`image` of
 * vq_codebook_prep), rows that are not finite through the exact kernel.  Supported for D in
 * {64, 128, 256}, K <= 8192 per call and finite T > 0 (VQ_ERR_UNSUPPORTED otherwise; use
 * vq_search).  q_out requires code_base == 0.
 *   vq_filter_image_bytes(K, D)   bytes of the bf16 codebook image
 *   vq_filter_work_bytes(N)       bytes of the candidate buffer `work`
 *   vq_filter_loss_partial_bytes(N)  bytes of `loss_partial` (more partials than vq_search writes;
 *                                 all of them are defined on return and feed vq_step_scalars)
 *   vq_filter_prep                builds the bf16 image from E and the norms e2 of vq_codebook_prep
 *                                 (call after it, on the same stream, whenever E changed)
 */
size_t vq_filter_image_bytes(int K, int D);
size_t vq_filter_work_bytes(int64_t N);
/* How a filter pass over `rows` rows divides its codebook sweep when it has too few row workgroups
 * to fill the chip: parts[0] = parts of the sweep, parts[1] = stages (128 codes) per part.  Host
 * code only; the work buffer of vq_filter_work_bytes(N) holds the key lists of every part of the
 * first pass (rows = N) and of the second pass (rows <= N/8). */
int vq_filter_split(int64_t rows, int K, int* parts);
int vq_filter_prep(const float* E, const float* e2, int K, int D, void* fimage, void* stream);
size_t vq_filter_loss_partial_bytes(int64_t N);
int vq_search_filtered(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* image,
                       const float* e2, const void* fimage, int K, float temperature, int64_t code_base,
                       int64_t* idx64, int32_t* idx32, float* dmin, float* q_out, float* loss_partial,
                       int32_t* nf_rows, int32_t* control, void* work, void* stream);

/* Same search through the exact scalar kernel only (any D <= 4096); used as the in-library
 * cross-check of the MFMA kernel and for shapes the MFMA kernel does not cover. */
int vq_search_exact(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, const float* e2,
                    int K, float temperature, int64_t code_base, int64_t* idx64, int32_t* idx32, float* dmin,
                    float* q_out, float* loss_partial, void* stream);

/*
 * Gather + straight-through + loss partials for given indices (used after a sharded
 * search has merged its winners).  ref: vqvae/ema_vector_quantizer.py:58, :92, :95, :98.
 * loss_partial gets vq_search_blocks(N) floats.  Rows whose index is outside [0, K) are
 * skipped: no output is written for them and they add nothing to the loss.
 */
int vq_gather_st(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, int K,
                 const int64_t* idx64, float* q_out, float* loss_partial, void* stream);

/*
 * Code histogram: counts[k] = #(idx == k).  ref: vqvae/ema_vector_quantizer.py:65
 * (column sum of the one-hot matrix), vqvae/codebook_utilization_check.py:93-95,
 * vqvae/vqgan_trainer.py:601-603.  idx is int32 when idx_is_64 == 0, else int64.
 * counts is zeroed first unless accumulate != 0.  Out-of-range indices are counted in
 * control word 2 and otherwise ignored.
 */
int vq_code_histogram(const void* idx, int idx_is_64, int64_t M, int K, uint32_t* counts, int accumulate,
                      int32_t* control, void* stream);

/*
 * EMA embedding sums: slab[s][k][c] = sum of z[n][c] over the rows n of slab s with
 * idx32[n] == code_base + k.  ref: vqvae/ema_vector_quantizer.py:84 (encodings.t() @ flat_input).
 * Writes `nslab` slabs of K*D floats, nslab = vq_scatter_slabs_for(N, K, D, layout, HW) (0 stands for
 * vq_scatter_slabs(N, K, D)); their sum over s is dw.  NCHW: a workgroup owns two channels and all
 * codes; row-major (also what a channels_last tensor is in memory): a workgroup owns a range of
 * codes with all channels and reads each matching row once, whole.  The sums are
 * accumulated in LDS in 64-bit fixed point scaled from max|z| (control word 3, written by
 * vq_search for the same z; pass zmax_valid = 0 to have it computed here), so they do not
 * depend on arrival order: bit-reproducible, each element rounded once at max|z| * 2^-40 or
 * finer.  A batch holding NaN/inf takes a float-atomic path that propagates them like the
 * reference.  Rows whose index falls outside [code_base, code_base + K) are skipped
 * (codebook-row sharding: a rank accumulates only the rows its shard won).
 * Limits: N < 2^31 rows per call and code_base + K < 2^31 (row numbers and codes are 32-bit inside
 * the kernels; N is an int64_t only for uniformity of the ABI): VQ_ERR_UNSUPPORTED beyond them, as
 * for vq_search.  A batch of 2^31 rows or more is passed as several calls whose slabs the caller adds.
 */
int vq_ema_scatter(const float* z, int layout, int64_t N, int D, int64_t HW, const int32_t* idx32, int K,
                   int64_t code_base, int zmax_valid, int32_t* control, float* slabs, int nslab, void* stream);

/*
 * The same sums for row-major z (layout VQ_LAYOUT_ND, which is also a channels_last tensor's memory)
 * by counting sort: the rows are ranked by code (per-chunk LDS histograms, a scan, LDS cursors) and
 * every row is then read whole, once, in code order; lane groups accumulate runs of one code in
 * registers and add them to the code's accumulator with 64-bit integer atomics (fixed point as
 * above: the result does not depend on any ordering).  ref: vqvae/ema_vector_quantizer.py:84.
 * dw gets K*D floats (one slab: pass it on with nslab = 1).  Supported for K <= 16384 and D a
 * multiple of 4 with D/4 a power of two up to 64 (vq_scatter_sorted_supported); z 16-byte aligned,
 * work = vq_scatter_sorted_work_bytes(N, K, D) bytes, 256-byte aligned.
 */
int vq_scatter_sorted_supported(int K, int D);
size_t vq_scatter_sorted_work_bytes(int64_t N, int K, int D);
int vq_ema_scatter_sorted(const float* z, int64_t N, int D, const int32_t* idx32, int K, int64_t code_base,
                          int zmax_valid, int32_t* control, float* dw, void* work, void* stream);

/* dw[k][c] = sum_s slabs[s][k][c] in slab order (packs the data-parallel all-reduce buffer). */
int vq_reduce_slabs(const float* slabs, int nslab, int K, int D, float* dw, void* stream);

/*
 * Scalars of one forward.  ref: vqvae/ema_vector_quantizer.py:92 (loss = beta * mean) and
 * :105-106 (perplexity = exp(-sum p ln(p + 1e-10)), p = counts / N_total).
 * sq_sum_out (may be NULL) receives the plain sum of squares, for data-parallel reduction.
 */
int vq_step_scalars(const float* loss_partial, int64_t n_partial, int64_t N_total, int D, float beta,
                    const uint32_t* counts, int K, float* loss_out, float* perplexity_out, double* sq_sum_out,
                    void* stream);

/*
 * EMA cluster-size update with the persisted Laplace smoothing.
 * ref: vqvae/ema_vector_quantizer.py:67 (usage_count += counts), :71-72, :75-80.
 * decay, one_minus_decay, eps, k_eps are the reference's Python-double expressions
 * (decay, 1 - decay, epsilon, K * epsilon) rounded to fp32 by the caller.
 * counts64 (may be NULL) overrides counts with int64 values (all-reduced counts).
 */
int vq_ema_cluster_update(const uint32_t* counts, const int64_t* counts64, int K, float decay,
                          float one_minus_decay, float eps, float k_eps, float* usage, float* cluster_size,
                          void* stream);

/*
 * EMA embedding update and renormalisation.  ref: vqvae/ema_vector_quantizer.py:85, :89.
 * ema_w = ema_w*decay + (1-decay)*sum_s slabs[s]; E = ema_w / cluster_size[:, None].
 */
int vq_ema_codebook_update(const float* slabs, int nslab, int K, int D, float decay, float one_minus_decay,
                           const float* cluster_size, float* ema_w, float* E, void* stream);

/*
 * The same three K-sized steps with fewer launches, for steps so short that launches are what they
 * cost (a 4096-vector batch is ten launches of 5-12 us of which the arithmetic is a fraction):
 *   vq_step_finish             = vq_step_scalars + vq_ema_cluster_update in one single-workgroup kernel
 *                                (usage = cluster_size = NULL: the scalars only, as in eval mode);
 *   vq_ema_codebook_update_end = vq_ema_codebook_update.
 * Same arithmetic in the same order as the separate entry points (ref: vqvae/ema_vector_quantizer.py:67-89,
 * :92, :105-106).  Whichever of the two runs last in a step may also do the step's housekeeping, so that
 * the next step needs no vq_control_reset and no zeroing of its histogram: cleanup_control (may be NULL):
 * the 16 control words -- words 0..7 are copied to words 8..15, where the step's counters stay readable,
 * then zeroed; cleanup_counts (may be NULL): K histogram bins to zero, i.e. the buffer the NEXT step will
 * count into with accumulate = 1 (alternate two buffers: this step's counts stay readable).
 */
int vq_step_finish(const float* loss_partial, int64_t n_partial, int64_t N_total, int D, float beta,
                   const uint32_t* counts, int K, float decay, float one_minus_decay, float eps, float k_eps,
                   float* usage, float* cluster_size, float* loss_out, float* perplexity_out,
                   int32_t* cleanup_control, uint32_t* cleanup_counts, void* stream);
int vq_ema_codebook_update_end(const float* slabs, int nslab, int K, int D, float decay, float one_minus_decay,
                               const float* cluster_size, float* ema_w, float* E, int32_t* cleanup_control,
                               uint32_t* cleanup_counts, void* stream);

/*
 * Backward of the forward's differentiable part.  ref: autograd of
 * vqvae/ema_vector_quantizer.py:92 and :95:
 *   grad_z = grad_q + grad_loss * coef * (z - q_st),  coef = 2*beta/(N*D)
 * grad_q may be NULL (treated as zero); grad_loss is a device scalar (NULL = 0).
 */
int vq_backward(const float* grad_q, const float* grad_loss, const float* z, const float* q_st, int64_t numel,
                float coef, float* grad_z, void* stream);

/*
 * The quantizer followed by a 1x1 convolution.  ref: vqvae/custom_vqgan.py:51, :72 (post_quant_conv)
 * applied at :95 to the output of the quantizer (:81); loss as vqvae/ema_vector_quantizer.py:92.
 * Every quantized row is a codebook row, so the convolution of the N rows equals a gather from the
 * convolved codebook  P = E W^T + b  (K x DP, computed by the caller with any GEMM):
 *   out[n, :] = P[idx64[n], :]   in the layout of z (NCHW with DP channels, or (N, DP) rows)
 *   loss_partial[w] = sum over the 256 rows of workgroup w of |E[idx64[n]] - z[n]|^2  (may be NULL)
 * loss_partial gets vq_search_blocks(N) floats.  E and P 16-byte aligned.  Rows whose index is
 * outside [0, K) are skipped, as in vq_gather_st.
 */
int vq_gather_project(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E, int K,
                      const int64_t* idx64, const float* P, int DP, float* out, float* loss_partial, void* stream);

/*
 * vq_backward with the quantized row read from a codebook instead of a stored N x D tensor:
 *   grad_z[n, :] = grad_q[n, :] + grad_loss * coef * (z[n, :] - E[idx64[n], :])
 * E is the codebook the forward searched (the caller keeps a copy across the EMA update, K x D
 * floats instead of N x D).  ref: autograd of vqvae/ema_vector_quantizer.py:92 and :95.
 * grad_q may be NULL; grad_loss is a device scalar (NULL = 0); a row whose index is outside [0, K)
 * passes grad_q through.
 */
int vq_backward_gather(const float* grad_q, const float* grad_loss, const float* z, int layout, int64_t N, int D,
                       int64_t HW, const float* E, int K, const int64_t* idx64, float coef, float* grad_z,
                       void* stream);

/*
 * 1x1 convolution of latent rows on the search's MFMA engine: out[n][o] = sum_c W[o][c] x[n][c] + b[o],
 * x and out in the same layout (NCHW with C_in / C_out channels, or rows).  ref: the two
 * nn.Conv2d(.., 1) layers either side of the quantizer, vqvae/custom_vqgan.py:50-51, applied at :68
 * (quant_conv) and :72 (post_quant_conv); with W transposed it is also their input gradient.
 * Each output is one fp32 fmaf chain over the input channels in increasing order, started from b[o].
 * vq_conv1x1_prep lays W ([C_out][C_in] row major, bias may be NULL) out as the `image` the kernel
 * streams (vq_conv1x1_image_bytes bytes).  C_in in {16, 32, 64, 128, 256}; x, out, W, image 16-byte
 * aligned.  (NCHW with HW == 1 is the row layout.)
 */
size_t vq_conv1x1_image_bytes(int C_out, int C_in);
int vq_conv1x1_prep(const float* W, const float* bias, int C_out, int C_in, float* image, void* stream);
int vq_conv1x1(const float* x, int layout, int64_t N, int C_in, int64_t HW, const float* image, int C_out, float* out,
               void* stream);

/*
 * Parameter gradients of that convolution:  dW[o][i] = sum_n gy[n][o] x[n][i],  db[o] = sum_n gy[n][o]
 * (db may be NULL).  ref: autograd of the nn.Conv2d(.., 1) layers of vqvae/custom_vqgan.py:50-51, whose
 * parameters the reference's optimiser updates (vqvae/vqgan_trainer.py:98-104).  gy and x share one
 * layout: NCHW (HW a multiple of 32) or rows (channel counts multiples of 4); up to 1024 channels
 * each; gy, x, work 16-byte aligned.  fp32 MFMA products, partial sums of disjoint row ranges added
 * in a fixed order: the result is the same bits on every run, and equals the sum over n up to fp32
 * rounding of the summation order.  `work`: vq_conv1x1_wgrad_work_bytes bytes.
 * vq_conv1x1_wgrad_supported returns 1 for the shapes the kernel takes.
 */
int vq_conv1x1_wgrad_supported(int layout, int64_t N, int C_out, int C_in, int64_t HW);
int64_t vq_conv1x1_wgrad_work_bytes(int64_t N, int C_out, int C_in);
int vq_conv1x1_wgrad(const float* gy, const float* x, int layout, int64_t N, int C_out, int C_in, int64_t HW,
                     float* dW, float* db, void* work, void* stream);

/*
 * Re-seed codes in place (maintenance).  ref: the per-code loops of
 * vqvae/ema_vector_quantizer.py:185-196 (reset_dead_codes: noise 0.2, scale 0.1), :262-274
 * (reset_low_usage_codes: noise 0.4, scale 0.2) and the new rows of :131-148 (expand_codebook:
 * noise 0.2, scale 0).  For every i < n_targets, with k = targets[i] and a donor d drawn
 * uniformly from donors[0 .. n_donors):
 *   E[k] = E[d] + noise * N(0, 1);  ema_w[k] = scale * ema_w[d];
 *   cluster_size[k] = scale * cluster_size[d];  usage[k] = 0.
 * All donors are read before any target is written (`stage`: vq_reseed_stage_bytes(n_targets, D)
 * bytes).  Random numbers come from Philox4x32-10 keyed by `seed`: the draws follow the
 * reference's distributions, not torch's generator sequence.
 */
size_t vq_reseed_stage_bytes(int n_targets, int D);
int vq_reseed_codes(float* E, float* ema_w, float* cluster_size, float* usage, int K, int D, const int64_t* targets,
                    int n_targets, const int64_t* donors, int n_donors, float noise, float scale, uint64_t seed,
                    float* stage, void* stream);

/* One-hot matrix enc[N][K] (fp32).  ref: vqvae/ema_vector_quantizer.py:54-55. */
int vq_onehot(const int64_t* idx64, int64_t N, int K, float* enc, void* stream);

/*
 * get_codebook_stats.  ref: vqvae/ema_vector_quantizer.py:205-232.
 * out[5] (double, device): active count, utilisation, entropy (log2), normalised entropy,
 * total usage.
 */
int vq_cluster_stats(const float* cluster_size, const float* usage, int K, double* out, void* stream);

/*
 * Index statistics from a histogram.  out64[5]: ref vqvae/codebook_utilization_check.py:86-108
 * (#unique, utilisation, entropy, normalised entropy, the "gini" of :108), float64 like
 * NumPy.  out32[4]: ref vqvae/vqgan_trainer.py:594-612 (#unique, utilisation, entropy,
 * normalised entropy), fp32 like the torch code.  Either may be NULL.
 */
int vq_index_stats(const uint32_t* counts, int K, int64_t M, double* out64, float* out32, void* stream);

/*
 * Pairwise codebook distances (the analysis ref README.md:26-29 describes; the reference
 * holds no code for it).  nn_dist[i] / nn_idx[i]: nearest other code of i; out[3] (double):
 * min, mean, max over ordered pairs i != j (NaN when the codebook holds non-finite rows).
 * sum_partial: vq_search_blocks(K) doubles of scratch; mm_partial: 2*vq_search_blocks(K) + K
 * 4-byte words of scratch.
 */
int vq_pairwise_stats(const float* E, const float* image, const float* e2, int K, int D, float* nn_dist,
                      int64_t* nn_idx, double* sum_partial, float* mm_partial, double* out, int32_t* control,
                      void* stream);

/*
 * The same sweep for a range of rows: rows [row_base, row_base + n_rows) of the codebook against all
 * K codes -- one rank's share when the K x K sweep is sharded by rows over several GPUs (the
 * reference has no multi-device code; the analysis is the one README.md:26-29 describes).
 * nn_dist / nn_idx get n_rows entries (nn_idx holds global code numbers); out[3] (double): min, SUM
 * (not mean) and max over the pairs (i, j), i in the range, j != i, so that ranks combine them with
 * MIN / SUM / MAX.  sum_partial: vq_search_blocks(n_rows) doubles; mm_partial:
 * 2*vq_search_blocks(n_rows) + n_rows words.
 */
int vq_pairwise_rows(const float* E, const float* image, const float* e2, int K, int D, int64_t row_base,
                     int64_t n_rows, float* nn_dist, int64_t* nn_idx, double* sum_partial, float* mm_partial,
                     double* out, int32_t* control, void* stream);

/*
 * vq_pairwise_stats for callers that cannot allocate device memory (code under HIP graph capture,
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels
 * and same results as vq_pairwise_stats, but sum_partial / mm_partial live in a device array inside
 * the library (about 262 KiB per GPU, held until the process ends) instead of caller memory; the
 * call allocates and frees nothing.  K is limited to VQ_PAIRWISE_STATIC_MAX_K (VQ_ERR_TOO_LARGE
 * beyond it).  All calls on one GPU share that array: calls that overlap on one GPU (two streams,
 * two threads) race; calls in stream order do not, and nothing has to be cleared between them.
 * The array used is that of the GPU `stream` belongs to (for the NULL stream: the current device),
 * which must also be where E, image, ... live.  The first call on a GPU looks the array's address up
 * (hipGetSymbolAddress, host-side work that may load the code object): under stream capture call
 * vq_init(device) beforehand, so that no call inside the capture is a first call.
 */
#define VQ_PAIRWISE_STATIC_MAX_K 65536
int vq_pairwise_stats_static(const float* E, const float* image, const float* e2, int K, int D, float* nn_dist,
                             int64_t* nn_idx, double* out, int32_t* control, void* stream);

/*
 * Data-parallel training (one process per GPU; the reference has no multi-device code on this path, the
 * exchange sits between the statistics of vqvae/ema_vector_quantizer.py:65/:84 and the update of :71-89):
 * every rank packs its EMA statistics into one fp32 buffer, the ranks SUM-all-reduce it (RCCL), and every
 * rank finishes the step from the reduced buffer, so all replicas apply the same update.
 *   buf = [dw (K*D, only if slabs != NULL) | counts mod 4096 (K) | counts div 4096 (K) | rows mod 4096 | rows div 4096]
 * vq_packed_stats_floats(K, D, with_sums) floats.  The split halves of the counts and of the row count add
 * exactly in fp32 for up to 4096 ranks of fewer than 2^24 rows each (VQ_ERR_UNSUPPORTED beyond that).
 * vq_pack_stats: slabs as written by vq_ema_scatter (summed in slab order, as vq_reduce_slabs) or NULL in
 * eval mode.  vq_step_finish_reduced: vq_step_finish with the counts and the global row count taken from
 * reduced_tail = buf + (K*D or 0); the loss stays the mean over the N_local rows of this rank (as under
 * DistributedDataParallel), perplexity and cluster sizes come from the global counts; counts64_out (K) and
 * n_total_out (1) receive the reduced counts and row count if not NULL.  Then
 * vq_ema_codebook_update_end(buf, 1, ...) applies the reduced sums.
 */
size_t vq_packed_stats_floats(int K, int D, int with_sums);
int vq_pack_stats(const float* slabs, int nslab, int K, int D, const uint32_t* counts, int64_t n_local, float* buf,
                  void* stream);
int vq_step_finish_reduced(const float* loss_partial, int64_t n_partial, int64_t N_local, int D, float beta,
                           const float* reduced_tail, int K, float decay, float one_minus_decay, float eps, float k_eps,
                           float* usage, float* cluster_size, float* loss_out, float* perplexity_out,
                           int64_t* counts64_out, int64_t* n_total_out, int32_t* cleanup_control,
                           uint32_t* cleanup_counts, void* stream);

/*
 * One whole forward of the quantizer in ONE call: vq_codebook_prep -> vq_search -> vq_code_histogram ->
 * (training) vq_ema_scatter[_sorted] -> vq_step_finish -> (training) vq_ema_codebook_update_end, enqueued
 * back to back on `stream`.  ref: vqvae/ema_vector_quantizer.py:33-108, the forward as a whole.  For a host
 * whose per-call overhead matters (a Python binding pays microseconds per call; a 4096-vector step is
 * a dozen launches of a few microseconds each).  Nothing here that the separate entry points do not do;
 * buffers and their sizes are those of the separate entry points:
 *   image, e2, control         vq_image_bytes, vq_norms_bytes, vq_control_bytes
 *   idx32, nf_rows             N int32 each
 *   loss_partial               vq_loss_partial_bytes(N)
 *   split_ws                   vq_search_split_bytes(N, K) bytes, or NULL
 *   counts, counts_next        K uint32 each: the two alternating histogram buffers (see vq_step_finish)
 *   slabs, nslab               training: vq_scatter_slabs_for(...) * K * D floats; with sort_work != NULL
 *                              (row-major z, vq_scatter_sorted_supported) one slab and the counting-sort path
 *   sort_work                  vq_scatter_sorted_work_bytes(N, K, D) bytes, or NULL
 * control_clean != 0: the previous call on these buffers finished and nothing else has touched `control` or
 * `counts` since, so they are zero already; otherwise they are zeroed first (two memsets).
 * Outputs: idx64 [N], q_out (layout of z), loss_out, perplexity_out (device scalars), dmin [N] or NULL;
 * training != 0 also updates usage, cluster_size, ema_w and E in place.
 */
typedef struct vq_step_args {
  const float* z; int layout; int64_t N; int D; int64_t HW;
  float* E; float* ema_w; float* cluster_size; float* usage; int K;
  float beta, decay, one_minus_decay, eps, k_eps, temperature;
  int training;
  int64_t* idx64; float* q_out; float* loss_out; float* perplexity_out; float* dmin;
  float* image; float* e2; int32_t* control; int32_t* idx32; int32_t* nf_rows; float* loss_partial; void* split_ws;
  uint32_t* counts; uint32_t* counts_next; float* slabs; int nslab; void* sort_work;
  int control_clean;
} vq_step_args;
int vq_quantize_step(const vq_step_args* args, void* stream);

/*
 * Codebook-row-sharded search support: pack (distance, index) into one int64 whose
 * signed order is the order torch.argmin would see on the unsharded distance matrix
 * (NaN first, then ascending distance, then ascending index), and unpack the winner after
 * an all-reduce(MIN) over int64.  ref: vqvae/ema_vector_quantizer.py:53 applied across shards.
 */
int vq_pack_dist_index(const float* dmin, const int64_t* idx64, int64_t N, int64_t* packed, void* stream);
int vq_unpack_dist_index(const int64_t* packed, int64_t N, float* dmin, int64_t* idx64, int32_t* idx32,
                         void* stream);
/* acc[n] = min(acc[n], other[n]) over packed keys: merges the winners of two codebook-row ranges
 * on one device (what the MIN all-reduce does across devices).  ref: the argmin of
 * vqvae/ema_vector_quantizer.py:53 taken over row ranges of the codebook. */
int vq_packed_min(int64_t* acc, const int64_t* other, int64_t N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQ_HIP_H */
