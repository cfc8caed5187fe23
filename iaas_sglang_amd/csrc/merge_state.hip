// lse-weighted merge of two partial attention results (cascade / split prefix+suffix).
// One thread per 16-byte pack of the output; fp32 arithmetic; HBM-bound, trivial.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void merge_state_kernel(const T* __restrict__ a,
                                                          const float* __restrict__ la,
                                                          const T* __restrict__ b,
                                                          const float* __restrict__ lb, T* __restrict__ out,
                                                          float* __restrict__ out_lse, int64_t n_heads_total,
                                                          int d) {
  const int packs = d / 8;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n_heads_total * packs) return;
  const int64_t th = gid / packs;
  const int pk = (int)(gid % packs);
  float x = la[th], y = lb[th];
  x = isinf(x) ? -INFINITY : x;
  y = isinf(y) ? -INFINITY : y;
  const float mx = fmaxf(x, y);
  const float ea = expf(x - mx), eb = expf(y - mx);
  const float se = ea + eb;
  const float sa = ea / se, sb = eb / se;
  const uint4 va = *(const uint4*)(a + th * d + pk * 8);
  const uint4 vb = *(const uint4*)(b + th * d + pk * 8);
  const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
  uint32_t wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = Elem<T>::lo(wa[j]) * sa + Elem<T>::lo(wb[j]) * sb;
    const float hi = Elem<T>::hi(wa[j]) * sa + Elem<T>::hi(wb[j]) * sb;
    wo[j] = pack2<T>(lo, hi);
  }
  *(uint4*)(out + th * d + pk * 8) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
  if (out_lse && pk == 0) out_lse[th] = logf(se) + mx;
}

extern "C" int mi_merge_state(const void* o_a, const float* lse_a, const void* o_b, const float* lse_b,
                              void* out, float* out_lse, int64_t n, int64_t h, int64_t d, int dtype,
                              void* stream) {
  MI_CHECK_ARG(n >= 0 && h > 0 && d > 0);
  if (n == 0) return MI_OK;
  MI_CHECK_ARG(o_a && lse_a && o_b && lse_b && out);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (d % 8 != 0) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_merge_state: head size %lld not a multiple of 8", (long long)d);
  MI_CHECK_ARG((((uintptr_t)o_a | (uintptr_t)o_b | (uintptr_t)out) & 15) == 0);
  const int64_t total = n * h * (d / 8);
  const unsigned blocks = (unsigned)cdiv64(total, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16)
    merge_state_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)o_a, lse_a, (const bf16_t*)o_b, lse_b,
                                                        (bf16_t*)out, out_lse, n * h, (int)d);
  else
    merge_state_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)o_a, lse_a, (const f16_t*)o_b, lse_b,
                                                       (f16_t*)out, out_lse, n * h, (int)d);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
