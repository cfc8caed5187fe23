// FP8 (e4m3fn x e4m3fn -> fp32 -> bf16/fp16) GEMM with row/column/tensor scale epilogue.
//   out[m][n] = (sum_k A[m][k] * B[n][k]) * sa[m|0] * sb[n|0] (+ bias[n])
// A [M,K] row-major, B stored [N,K] row-major (both K-contiguous, "TN").
//
// v1 "weight-streaming" kernel: every wave owns a 16-row strip of B (the weights), streams
// it straight from HBM into MFMA fragments (each lane 32 contiguous bytes of a row, four
// lanes = one full 128-B line) and multiplies it against MT 16-row tiles of A that the four
// waves of the workgroup read through L1/L2.  MFMA 16x16x32 fp8: the weights are the A
// operand (rows = n), the activations the B operand (cols = m) so each lane ends up with four
// consecutive n of one output row (one 8-byte store).
// Bound at decode (M <= 128): HBM (weights read once); at prefill: MFMA.
#include "common.h"

typedef long fp8x8_t;  // 8 fp8 values = one MFMA 16x16x32 fp8 operand

struct GemmParams {
  const uint8_t* a;
  const uint8_t* b;
  const float* sa;
  const float* sb;
  const void* bias;
  void* out;
  int64_t M, N, K, lda, ldb, ldo;
  int sa_row, sb_row;
};

template <typename OutT, int MT>
__global__ __launch_bounds__(256) void fp8_gemm_kernel(const GemmParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  const int64_t m0 = (int64_t)blockIdx.y * (MT * 16);
  if (n0 >= p.N) return;

  const int64_t nrow = min(n0 + r16, p.N - 1);
  const uint8_t* wp = p.b + nrow * p.ldb + 32 * q;
  const uint8_t* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = p.a + min(m0 + t * 16 + r16, p.M - 1) * p.lda + 32 * q;

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t K = p.K;
  const int64_t kfull = K & ~(int64_t)127;
  for (int64_t kb = 0; kb < kfull; kb += 128) {
    const uint4 w0 = *(const uint4*)(wp + kb);
    const uint4 w1 = *(const uint4*)(wp + kb + 16);
    const fp8x8_t wf[4] = {(long)(((uint64_t)w0.y << 32) | w0.x), (long)(((uint64_t)w0.w << 32) | w0.z),
                           (long)(((uint64_t)w1.y << 32) | w1.x), (long)(((uint64_t)w1.w << 32) | w1.z)};
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const uint4 x0 = *(const uint4*)(xp[t] + kb);
      const uint4 x1 = *(const uint4*)(xp[t] + kb + 16);
      const fp8x8_t xf[4] = {(long)(((uint64_t)x0.y << 32) | x0.x), (long)(((uint64_t)x0.w << 32) | x0.z),
                             (long)(((uint64_t)x1.y << 32) | x1.x), (long)(((uint64_t)x1.w << 32) | x1.z)};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[s], xf[s], acc[t], 0, 0, 0);
    }
  }
  if (kfull < K) {  // K tail (multiple of 16): zero-fill what lies beyond K
    const int64_t kb = kfull;
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0;
    if (kb + 32 * q < K) w0 = *(const uint4*)(wp + kb);
    if (kb + 32 * q + 16 < K) w1 = *(const uint4*)(wp + kb + 16);
    const fp8x8_t wf[4] = {(long)(((uint64_t)w0.y << 32) | w0.x), (long)(((uint64_t)w0.w << 32) | w0.z),
                           (long)(((uint64_t)w1.y << 32) | w1.x), (long)(((uint64_t)w1.w << 32) | w1.z)};
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      uint4 x0 = make_uint4(0, 0, 0, 0), x1 = x0;
      if (kb + 32 * q < K) x0 = *(const uint4*)(xp[t] + kb);
      if (kb + 32 * q + 16 < K) x1 = *(const uint4*)(xp[t] + kb + 16);
      const fp8x8_t xf[4] = {(long)(((uint64_t)x0.y << 32) | x0.x), (long)(((uint64_t)x0.w << 32) | x0.z),
                             (long)(((uint64_t)x1.y << 32) | x1.x), (long)(((uint64_t)x1.w << 32) | x1.z)};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[s], xf[s], acc[t], 0, 0, 0);
    }
  }

  // epilogue: lane holds out[m = m0+16t+r16][n = n0 + 4q + r], r = 0..3
  const int64_t nb = n0 + 4 * q;
  float sbv[4], bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t n = min(nb + r, p.N - 1);
    sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
    bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int64_t m = m0 + t * 16 + r16;
    if (m >= p.M) continue;
    const float sav = p.sa_row ? p.sa[m] : p.sa[0];
    OutT* o = (OutT*)p.out + m * p.ldo + nb;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[t][r] * sav * sbv[r] + bv[r];
    if (nb + 3 < p.N && (p.ldo & 3) == 0) {
      *(uint2*)o = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.N) o[r] = (OutT)v[r];
    }
  }
}

template <typename OutT>
static void launch_fp8_gemm(const GemmParams& p, hipStream_t st) {
  const unsigned gx = (unsigned)cdiv64(p.N, 64);
  if (p.M <= 16) fp8_gemm_kernel<OutT, 1><<<dim3(gx, (unsigned)cdiv64(p.M, 16)), 256, 0, st>>>(p);
  else if (p.M <= 32) fp8_gemm_kernel<OutT, 2><<<dim3(gx, (unsigned)cdiv64(p.M, 32)), 256, 0, st>>>(p);
  else if (p.M <= 64) fp8_gemm_kernel<OutT, 4><<<dim3(gx, (unsigned)cdiv64(p.M, 64)), 256, 0, st>>>(p);
  else fp8_gemm_kernel<OutT, 8><<<dim3(gx, (unsigned)cdiv64(p.M, 128)), 256, 0, st>>>(p);
}

extern "C" int mi_fp8_gemm(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                           const void* bias, void* out, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldo, int scale_a_mode, int scale_b_mode, int out_dtype,
                           void* stream) {
  MI_CHECK_ARG(M >= 0 && N >= 0 && K > 0);
  if (M == 0 || N == 0) return MI_OK;
  MI_CHECK_ARG(a && b_nk && scale_a && scale_b && out);
  MI_CHECK_ARG(out_dtype == MI_BF16 || out_dtype == MI_FP16);
  MI_CHECK_ARG(scale_a_mode == MI_SCALE_TENSOR || scale_a_mode == MI_SCALE_ROW);
  MI_CHECK_ARG(scale_b_mode == MI_SCALE_TENSOR || scale_b_mode == MI_SCALE_ROW);
  if (K % 16 != 0 || lda % 16 != 0 || ldb % 16 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_gemm: K, lda, ldb must be multiples of 16 (got %lld %lld %lld)",
            (long long)K, (long long)lda, (long long)ldb);
  MI_CHECK_ARG((((uintptr_t)a | (uintptr_t)b_nk) & 15) == 0);
  MI_CHECK_ARG(cdiv64(M, 16) <= 65535 * 8);
  GemmParams p;
  p.a = (const uint8_t*)a; p.b = (const uint8_t*)b_nk; p.sa = scale_a; p.sb = scale_b;
  p.bias = bias; p.out = out; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldo = ldo;
  p.sa_row = scale_a_mode == MI_SCALE_ROW; p.sb_row = scale_b_mode == MI_SCALE_ROW;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MI_BF16) launch_fp8_gemm<bf16_t>(p, st);
  else launch_fp8_gemm<f16_t>(p, st);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
