/*
 * Sanitizer run of the CPU oracle (test infrastructure): compiles vq_oracle.c into this driver with
 * -fsanitize=address,undefined and walks every entry point over small problems with awkward sizes
 * (one row, one code, odd D, NCHW with HW not a power of two, a shard offset, masked self pairs,
 * NaN / inf inputs, out-of-range indices).  Any out-of-bounds access, misaligned access or signed
 * overflow aborts; a clean run prints "asan_check ok".
 */
#include "vq_oracle.c"

#include <stdio.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static void run_case(int64_t B, int D, int64_t HW, int K, int layout, float T, int64_t code_base, int poison) {
  const int64_t N = B * HW;
  float* z = (float*)malloc(sizeof(float) * (size_t)(N * D));
  float* E = (float*)malloc(sizeof(float) * (size_t)K * D);
  float* e2 = (float*)malloc(sizeof(float) * (size_t)K);
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
  float* dmin = (float*)malloc(sizeof(float) * (size_t)N);
  float* q = (float*)malloc(sizeof(float) * (size_t)(N * D));
  int64_t* counts = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
  float* dw = (float*)malloc(sizeof(float) * (size_t)K * D);
  float* usage = (float*)calloc((size_t)K, sizeof(float));
  float* cs = (float*)calloc((size_t)K, sizeof(float));
  float* ema_w = (float*)calloc((size_t)K * D, sizeof(float));
  for (int64_t i = 0; i < N * D; ++i) z[i] = rnd();
  for (int64_t i = 0; i < (int64_t)K * D; ++i) E[i] = 0.5f * rnd();
  if (poison) {
    z[0] = NAN;
    if (N * D > 3) z[3] = INFINITY;
    if (N * D > 5) z[5] = -INFINITY;
  }
  vqo_codebook_norms(E, K, D, e2);
  vqo_search(z, layout, N, D, HW, E, e2, K, T, code_base, 0, idx, dmin);
  vqo_search(z, layout, N, D, HW, E, e2, K, T, code_base, 1, idx, NULL);
  vqo_search(z, layout, N, D, HW, E, e2, K, T, 0, 0, idx, dmin);
  double sq = 0.0;
  vqo_gather_st(z, layout, N, D, HW, E, idx, q, &sq);
  vqo_gather_st(z, layout, N, D, HW, E, idx, NULL, NULL);
  if (N > 1) idx[N - 1] = K + 7;   /* out-of-range entries are skipped by the statistics */
  if (N > 2) idx[N - 2] = -3;
  vqo_histogram(idx, N, K, counts);
  vqo_embed_sum(z, layout, N, D, HW, idx, K, dw);
  (void)vqo_perplexity(counts, K, N);
  vqo_ema_update(counts, dw, K, D, 0.95f, 0.05f, 1e-5f, (float)K * 1e-5f, usage, cs, ema_w, E);
  double out5[5];
  float out4[4];
  vqo_cluster_stats(cs, usage, K, out5);
  vqo_index_stats_f64(counts, K, N, out5);
  vqo_index_stats_f32(counts, K, out4);
  float* nn = (float*)malloc(sizeof(float) * (size_t)K);
  int64_t* ni = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
  double out3[3];
  vqo_pairwise(E, K, D, nn, ni, out3);
  free(z); free(E); free(e2); free(idx); free(dmin); free(q); free(counts); free(dw);
  free(usage); free(cs); free(ema_w); free(nn); free(ni);
}

int main(void) {
  const int Ds[] = {1, 3, 8, 17, 64};
  const int Ks[] = {1, 2, 7, 64, 100};
  for (unsigned a = 0; a < sizeof(Ds) / sizeof(Ds[0]); ++a)
    for (unsigned b = 0; b < sizeof(Ks) / sizeof(Ks[0]); ++b) {
      run_case(1, Ds[a], 1, Ks[b], 0, 1.0f, 0, 0);
      run_case(3, Ds[a], 5, Ks[b], 1, 2.5f, 1000, 0);
      run_case(2, Ds[a], 7, Ks[b], 1, 1.0f, 0, 1);
      run_case(9, Ds[a], 1, Ks[b], 0, -1.0f, 5, 1);
      run_case(4, Ds[a], 4, Ks[b], 1, 0.0f, 0, 0);
    }
  printf("asan_check ok (version %d)\n", vqo_version());
  return 0;
}
