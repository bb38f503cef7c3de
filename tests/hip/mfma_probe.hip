// Test-only probe of the gfx950 bf16 matrix core (not part of the product library).
//
// The filtered search budgets an error for the matrix core's own fp32 accumulation inside
// v_mfma_f32_32x32x16_bf16.  The instruction's internal summation order and rounding are not
// documented, so the budget is an assumption; this probe runs the same instruction chain the
// filter kernel runs (KK/16 dependent MFMAs into one accumulator tile) on caller-provided bf16
// operands and returns the raw 32 x 32 fp32 result, so a test can compare it with the exact
// value of the same sum of products.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A: [32][KK] bf16 bit patterns (rows of the result), B: [32][KK] (columns), C: [32][32] fp32,
// C[r][c] = sum_k A[r][k] * B[c][k].  One wave.
__global__ __launch_bounds__(64) void mfma_probe_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                        int KK, float* __restrict__ C) {
  const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int s = 0; s < KK / 16; ++s) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(A + (size_t)i * KK + 16 * s + 8 * g);
    const u32x4 b = *reinterpret_cast<const u32x4*>(B + (size_t)i * KK + 16 * s + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
    C[row * 32 + i] = acc[r];
  }
}

extern "C" int mfma_probe_bf16(const uint16_t* A, const uint16_t* B, int KK, float* C, void* stream) {
  if (!A || !B || !C || KK <= 0 || KK % 16 != 0) return 1;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, KK, C);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
