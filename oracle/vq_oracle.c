/*
 * CPU oracle for the vector-quantization hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the algorithm of the reference's
 * EMAVectorQuantizer.forward (reference: vqvae/ema_vector_quantizer.py:33-108),
 * its get_codebook_stats (:205-232), the index statistics of
 * vqvae/codebook_utilization_check.py:83-108 and vqvae/vqgan_trainer.py:594-612.
 * Only tests, the smoke check and the benchmark's cpu_baseline leg may load it.
 * The product path never calls into this file.
 *
 * Numerics.  The reference leaves the summation order of its three reductions to
 * the BLAS / ATen build it runs on (torch.matmul, torch.sum).  The oracle fixes
 * one order and documents it, so that an implementation which follows the same
 * order can be compared bit for bit:
 *   - dot(z, e)    : one fp32 fused-multiply-add chain over the channels in
 *                    increasing order, starting from +0.0f;
 *   - ||x||^2      : two fp32 fma chains, one over the even channels and one
 *                    over the odd channels, each in increasing order from +0.0f,
 *                    then even + odd;
 *   - distance     : d = fma(-2, dot, (||z||^2 + ||e||^2))   (the reference's
 *                    association, ema_vector_quantizer.py:42-46; 2*dot is exact,
 *                    so the fma rounds exactly like fl(t - fl(2*dot)));
 *   - temperature  : d = d / T, IEEE division, only when T != 1
 *                    (ema_vector_quantizer.py:49-50);
 *   - argmin       : torch.argmin's rule (ema_vector_quantizer.py:53): NaN is
 *                    smaller than everything, first NaN wins; among equal
 *                    minima the lowest index wins.
 * Everything that is not an index (loss, EMA sums, entropies) accumulates in
 * double and is compared with a tolerance.
 *
 * Parity pinning: checked against golden vectors produced by importing the
 * reference quantizer (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VQO_LAYOUT_ND 0   /* z is [N][D] row major                          */
#define VQO_LAYOUT_NCHW 1 /* z is [B][D][HW], row n = b*HW + hw (fused :35) */

static inline float z_at(const float* z, int layout, int64_t n, int c, int D, int64_t HW) {
  if (layout == VQO_LAYOUT_ND) return z[n * D + c];
  int64_t b = n / HW, hw = n % HW;
  return z[(b * D + c) * HW + hw];
}

static inline int64_t out_off(int layout, int64_t n, int c, int D, int64_t HW) {
  if (layout == VQO_LAYOUT_ND) return n * D + c;
  int64_t b = n / HW, hw = n % HW;
  return (b * D + c) * HW + hw;
}

/* ||x||^2 with the even/odd chain order described above. */
static inline float sqnorm_row(const float* x, int D) {
  float p0 = 0.0f, p1 = 0.0f;
  for (int c = 0; c + 1 < D; c += 2) {
    p0 = fmaf(x[c], x[c], p0);
    p1 = fmaf(x[c + 1], x[c + 1], p1);
  }
  if (D & 1) p0 = fmaf(x[D - 1], x[D - 1], p0);
  return p0 + p1;
}

/* e2[k] = ||E[k]||^2  (reference :44). */
void vqo_codebook_norms(const float* E, int K, int D, float* e2) {
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) e2[k] = sqnorm_row(E + (size_t)k * D, D);
}

/* torch.argmin update rule: returns 1 when candidate d replaces best. */
static inline int argmin_takes(float d, float best) {
  if (d < best) return 1;
  if (isnan(d) && !isnan(best)) return 1;
  return 0;
}

/*
 * Nearest-codebook search (reference :35-53).  idx[n] (int64) and, when dmin is
 * not NULL, the winning distance.  code_base is added to every index (used by
 * the codebook-row-sharded search, where E holds rows [code_base, code_base+K)).
 * skip_self != 0 masks the pair (n, n) (pairwise codebook analysis, z := E).
 */
void vqo_search(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E,
                const float* e2, int K, float T, int64_t code_base, int skip_self, int64_t* idx,
                float* dmin) {
#pragma omp parallel
  {
    float* zr = (float*)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      for (int c = 0; c < D; ++c) zr[c] = z_at(z, layout, n, c, D, HW);
      const float z2 = sqnorm_row(zr, D);
      float best = 0.0f;
      int64_t bi = -1;
      for (int k = 0; k < K; ++k) {
        if (skip_self && (int64_t)k + code_base == n) continue;
        const float* e = E + (size_t)k * D;
        float dot = 0.0f;
        for (int c = 0; c < D; ++c) dot = fmaf(zr[c], e[c], dot);
        const float t = z2 + e2[k];
        float d = fmaf(-2.0f, dot, t);
        if (T != 1.0f) d = d / T;
        if (bi < 0 || argmin_takes(d, best)) {
          best = d;
          bi = k;
        }
      }
      if (bi < 0) bi = 0;
      idx[n] = bi + code_base;
      if (dmin) dmin[n] = best;
    }
    free(zr);
  }
}

/*
 * Gather + straight-through + commitment-loss sum (reference :58, :92, :95, :98).
 *   q_st = z + (E[idx] - z)        (two fp32 roundings, as the reference's :95)
 *   *sq_sum = sum over all N*D elements of (E[idx] - z)^2 (double accumulation)
 * q_out has the layout of z.
 */
void vqo_gather_st(const float* z, int layout, int64_t N, int D, int64_t HW, const float* E,
                   const int64_t* idx, float* q_out, double* sq_sum) {
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int64_t n = 0; n < N; ++n) {
    const float* e = E + (size_t)idx[n] * D;
    double s = 0.0;
    for (int c = 0; c < D; ++c) {
      const float zv = z_at(z, layout, n, c, D, HW);
      const float diff = e[c] - zv;
      s += (double)diff * (double)diff;
      if (q_out) q_out[out_off(layout, n, c, D, HW)] = zv + diff;
    }
    total += s;
  }
  if (sq_sum) *sq_sum = total;
}

/* counts[k] = #(idx == k) (reference :65 column sum of the one-hot matrix). */
void vqo_histogram(const int64_t* idx, int64_t M, int K, int64_t* counts) {
  memset(counts, 0, sizeof(int64_t) * (size_t)K);
  for (int64_t i = 0; i < M; ++i) {
    int64_t k = idx[i];
    if (k >= 0 && k < K) counts[k] += 1;
  }
}

/* dw[k][c] = sum over rows with idx == k of z[n][c]  (reference :84), double accumulation. */
void vqo_embed_sum(const float* z, int layout, int64_t N, int D, int64_t HW, const int64_t* idx,
                   int K, float* dw) {
  double* acc = (double*)calloc((size_t)K * D, sizeof(double));
  for (int64_t n = 0; n < N; ++n) {
    int64_t k = idx[n];
    if (k < 0 || k >= K) continue;
    double* a = acc + (size_t)k * D;
    for (int c = 0; c < D; ++c) a[c] += (double)z_at(z, layout, n, c, D, HW);
  }
  for (size_t i = 0; i < (size_t)K * D; ++i) dw[i] = (float)acc[i];
  free(acc);
}

/* perplexity = exp(-sum p ln(p + 1e-10)), p = counts / N   (reference :105-106). */
float vqo_perplexity(const int64_t* counts, int K, int64_t N) {
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    const float p = (float)counts[k] / (float)N;
    s += (double)(p * logf(p + 1e-10f));
  }
  return expf(-(float)s);
}

/*
 * EMA codebook update (reference :65-89).  The scalar constants arrive exactly
 * as the reference's Python expressions evaluate them before they meet an fp32
 * tensor: decay, (1 - decay) computed in double, epsilon, K * epsilon computed
 * in double -- each then rounded to fp32.  The Laplace-smoothed cluster sizes are
 * stored back (:76-80).
 */
void vqo_ema_update(const int64_t* counts, const float* dw, int K, int D, float decay,
                    float one_minus_decay, float eps, float k_eps, float* usage, float* cs,
                    float* ema_w, float* E) {
  double n_acc = 0.0;
  for (int k = 0; k < K; ++k) {
    const float c = (float)counts[k];
    usage[k] = usage[k] + c;
    cs[k] = cs[k] * decay + one_minus_decay * c;
    n_acc += (double)cs[k];
  }
  const float n = (float)n_acc;
  const float denom = n + k_eps;
  for (int k = 0; k < K; ++k) cs[k] = (cs[k] + eps) / denom * n;
  for (int k = 0; k < K; ++k) {
    for (int c = 0; c < D; ++c) {
      const size_t i = (size_t)k * D + c;
      ema_w[i] = ema_w[i] * decay + one_minus_decay * dw[i];
      E[i] = ema_w[i] / cs[k];
    }
  }
}

/*
 * get_codebook_stats (reference :205-232) on the EMA cluster sizes:
 *   out[0] active = #(cs > 0), out[1] utilisation, out[2] entropy (log2),
 *   out[3] normalised entropy, out[4] total usage.
 */
void vqo_cluster_stats(const float* cs, const float* usage, int K, double* out) {
  int64_t active = 0;
  double total = 0.0, use = 0.0;
  for (int k = 0; k < K; ++k) {
    if (cs[k] > 0.0f) active++;
    total += (double)cs[k];
    use += (double)usage[k];
  }
  double ent = 0.0, nent = 0.0;
  if (active > 0) {
    const float tot = (float)total;
    for (int k = 0; k < K; ++k) {
      const float p = cs[k] / tot;
      ent -= (double)(p * log2f(p + 1e-10f));
    }
    nent = ent / (double)log2f((float)K);
  }
  out[0] = (double)active;
  out[1] = (double)active / (double)K;
  out[2] = ent;
  out[3] = nent;
  out[4] = use;
}

/*
 * Index statistics of analyze_codebook_utilization
 * (reference: vqvae/codebook_utilization_check.py:83-108), float64 like NumPy:
 *   out[0] #unique, out[1] utilisation, out[2] entropy, out[3] normalised entropy,
 *   out[4] "gini" = 1 - sum((p+/mean(p+))^2)/|p+|  (literal, not the textbook Gini).
 * counts are over M indices.
 */
void vqo_index_stats_f64(const int64_t* counts, int K, int64_t M, double* out) {
  int64_t uniq = 0;
  double ent = 0.0, psum = 0.0;
  for (int k = 0; k < K; ++k)
    if (counts[k] > 0) {
      uniq++;
      const double p = (double)counts[k] / (double)M;
      ent -= p * log2(p + 1e-10);
      psum += p;
    }
  double gini = 0.0;
  if (uniq > 0) {
    const double mean = psum / (double)uniq;
    double s = 0.0;
    for (int k = 0; k < K; ++k)
      if (counts[k] > 0) {
        const double r = ((double)counts[k] / (double)M) / mean;
        s += r * r;
      }
    gini = 1.0 - s / (double)uniq;
  }
  out[0] = (double)uniq;
  out[1] = (double)uniq / (double)K;
  out[2] = ent;
  out[3] = ent / log2((double)K);
  out[4] = gini;
}

/*
 * Batch statistics of monitor_codebook (reference: vqvae/vqgan_trainer.py:594-612),
 * fp32 like the torch code: out[0] #unique, out[1] utilisation, out[2] entropy,
 * out[3] normalised entropy.  probs = counts / sum(counts).
 */
void vqo_index_stats_f32(const int64_t* counts, int K, float* out) {
  int64_t uniq = 0, total = 0;
  for (int k = 0; k < K; ++k) total += counts[k];
  double ent = 0.0;
  const float tot = (float)total;
  for (int k = 0; k < K; ++k)
    if (counts[k] > 0) {
      uniq++;
      const float p = (float)counts[k] / tot;
      if (p > 0.0f) ent -= (double)(p * log2f(p + 1e-10f));
    }
  out[0] = (float)uniq;
  out[1] = (float)uniq / (float)K;
  out[2] = (float)ent;
  out[3] = (float)ent / log2f((float)K);
}

/*
 * Pairwise codebook distances (the analysis the reference's README.md:26-29
 * describes; the reference holds no code for it, so this definition is the
 * build's own): P[i][j] = fma(-2, dot(e_i, e_j), e2[i] + e2[j]) for i != j, with
 * the same dot and norm orders as the search.  nn_dist[i] / nn_idx[i] = nearest
 * other code of i (lowest j on ties); out[0] min, out[1] mean, out[2] max over
 * all ordered pairs i != j (double accumulation).
 */
void vqo_pairwise(const float* E, int K, int D, float* nn_dist, int64_t* nn_idx, double* out) {
  float* e2 = (float*)malloc(sizeof(float) * (size_t)K);
  vqo_codebook_norms(E, K, D, e2);
  double gsum = 0.0;
  float gmin = INFINITY, gmax = -INFINITY;
#pragma omp parallel for schedule(static) reduction(+ : gsum) reduction(min : gmin) reduction(max : gmax)
  for (int i = 0; i < K; ++i) {
    const float* a = E + (size_t)i * D;
    float best = INFINITY;
    int64_t bi = -1;
    for (int j = 0; j < K; ++j) {
      if (j == i) continue;
      const float* b = E + (size_t)j * D;
      float dot = 0.0f;
      for (int c = 0; c < D; ++c) dot = fmaf(a[c], b[c], dot);
      const float d = fmaf(-2.0f, dot, e2[i] + e2[j]);
      if (bi < 0 || argmin_takes(d, best)) {
        best = d;
        bi = j;
      }
      gsum += (double)d;
      if (d < gmin) gmin = d;
      if (d > gmax) gmax = d;
    }
    nn_dist[i] = best;
    nn_idx[i] = bi < 0 ? 0 : bi;
  }
  out[0] = gmin;
  out[1] = K > 1 ? gsum / ((double)K * (double)(K - 1)) : 0.0;
  out[2] = gmax;
  free(e2);
}

int vqo_version(void) { return 1; }
