// FP8 (OCP e4m3fn, gfx950) activation quantisation: per-tensor (dynamic / static) and
// per-token.  Bit-exact with the reference arithmetic: scale = absmax / 448 (IEEE divide),
// q = rne_fp8(clamp(x * (1/scale), +-448)).  HBM-bound: 16-B loads, 8-B stores.
// One deliberate difference: absmax == 0 gives q = 0 instead of 0*inf = NaN.
#include "common.h"

#define FP8_MAX 448.0f

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

template <typename T> __device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = Elem<T>::lo(w[j]);
    f[2 * j + 1] = Elem<T>::hi(w[j]);
  }
}

__device__ __forceinline__ float clamp448(float v) { return fmaxf(fminf(v, FP8_MAX), -FP8_MAX); }

// 8 floats -> 8 fp8 bytes (RNE), scaled by inv
__device__ __forceinline__ uint2 quant8(const float (&f)[8], float inv) {
  uint32_t lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[0] * inv), clamp448(f[1] * inv), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[2] * inv), clamp448(f[3] * inv), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[4] * inv), clamp448(f[5] * inv), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[6] * inv), clamp448(f[7] * inv), hi, true);
  return make_uint2(lo, hi);
}

// ------------------------------------------------------------------ per tensor
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                     int64_t M, int64_t K, int64_t ldx) {
  __shared__ float red[4];
  const int64_t vec_per_row = K / 8;
  const int64_t total = M * vec_per_row;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vec_per_row, c = i % vec_per_row;
    float f[8];
    unpack8<T>(*(const uint4*)(x + r * ldx + c * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(f[j]));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void quant_tensor_kernel(const T* __restrict__ x, uint8_t* __restrict__ q,
                                                           float* __restrict__ scale, int64_t M,
                                                           int64_t K, int64_t ldx, int mode,
                                                           const float* __restrict__ partial, int npartial) {
  // mode 1 (static): *scale is given.  modes 0/2 (dynamic): every block reduces the per-block maxima
  // of absmax_kernel (<= 2048 floats, L2-hot) to the tensor amax; block 0 publishes the scale.
  //   0: scale = amax/448, multiplier 1/scale                    (per_tensor_quant_fp8.cu:42,54)
  //   2: multiplier 448/max(amax,1e-12), scale = 1/multiplier    (input_to_float8, fp8_utils.py:314-325)
  __shared__ float red[4];
  float s, inv;
  if (mode == 1) {
    s = *scale;
    inv = s > 0.f ? 1.0f / s : 0.f;
  } else {
    float mx = 0.f;
    for (int i = threadIdx.x; i < npartial; i += 256) mx = fmaxf(mx, partial[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (mode == 0) {
      s = mx / FP8_MAX;
      inv = s > 0.f ? 1.0f / s : 0.f;
    } else {
      inv = FP8_MAX / fmaxf(mx, 1e-12f);
      s = 1.0f / inv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale = s;
  }
  const int64_t vec_per_row = K / 8;
  const int64_t total = M * vec_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vec_per_row, c = i % vec_per_row;
    float f[8];
    unpack8<T>(*(const uint4*)(x + r * ldx + c * 8), f);
    *(uint2*)(q + r * K + c * 8) = quant8(f, inv);
  }
}

// scratch for the per-block maxima: a __device__ array of the code object (no allocation, graph
// safe).  Calls are stream-ordered; two calls on DIFFERENT streams must not overlap, which holds for
// the single-stream model runner this library plugs into.
__device__ float g_absmax_partial[2048];
static float* partial_buffer() {
  static float* ptr = nullptr;
  if (!ptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_absmax_partial)) != hipSuccess) return nullptr;
    ptr = (float*)p;
  }
  return ptr;
}

extern "C" int mi_fp8_quant_per_tensor(const void* x, void* q, float* scale, int64_t M, int64_t K,
                                       int64_t ldx, int is_static, int dtype, void* stream) {
  MI_CHECK_ARG(M >= 0 && K > 0);
  MI_CHECK_ARG(scale != nullptr);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  hipStream_t st = (hipStream_t)stream;
  MI_CHECK_ARG(is_static >= 0 && is_static <= 2);
  if (M == 0) {
    if (is_static != 1 && hipMemsetAsync(scale, 0, sizeof(float), st) != hipSuccess)
      MI_FAIL(MI_ERR_LAUNCH, "mi_fp8_quant_per_tensor: memset failed");
    return MI_OK;
  }
  MI_CHECK_ARG(x && q);
  if (K % 8 != 0 || ldx % 8 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_quant_per_tensor: K and ldx must be multiples of 8");
  MI_CHECK_ARG((((uintptr_t)x & 15) | ((uintptr_t)q & 7)) == 0);
  const int64_t total = M * (K / 8);
  unsigned blocks = (unsigned)(cdiv64(total, 256) < 2048 ? cdiv64(total, 256) : 2048);
  float* partial = nullptr;
  if (is_static != 1) {
    partial = partial_buffer();
    if (!partial) MI_FAIL(MI_ERR_LAUNCH, "mi_fp8_quant_per_tensor: reduction scratch symbol not found");
    // fewer, fatter blocks for the max pass: 8 vectors per thread
    unsigned ab = (unsigned)(cdiv64(total, 256 * 8) < 2048 ? cdiv64(total, 256 * 8) : 2048);
    if (ab < 1) ab = 1;
    if (dtype == MI_BF16) absmax_kernel<bf16_t><<<ab, 256, 0, st>>>((const bf16_t*)x, partial, M, K, ldx);
    else absmax_kernel<f16_t><<<ab, 256, 0, st>>>((const f16_t*)x, partial, M, K, ldx);
    MI_CHECK_LAUNCH();
    if (dtype == MI_BF16)
      quant_tensor_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, (uint8_t*)q, scale, M, K, ldx, is_static, partial, (int)ab);
    else
      quant_tensor_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, (uint8_t*)q, scale, M, K, ldx, is_static, partial, (int)ab);
    MI_CHECK_LAUNCH();
    return MI_OK;
  }
  if (dtype == MI_BF16)
    quant_tensor_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, (uint8_t*)q, scale, M, K, ldx, 1, nullptr, 0);
  else
    quant_tensor_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, (uint8_t*)q, scale, M, K, ldx, 1, nullptr, 0);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------------------------- per token
// One 256-thread workgroup per row; the row stays in registers between the two passes
// when K <= 256*8*RV elements (RV vectors per thread), else it is re-read (L2-hot).
template <typename T, int RV>
__global__ __launch_bounds__(256) void quant_token_kernel(const T* __restrict__ x, uint8_t* __restrict__ q,
                                                          float* __restrict__ scales, int64_t K, int64_t ldx) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * ldx;
  uint8_t* qr = q + row * K;
  const int64_t nvec = K / 8;
  uint4 keep[RV];
  float mx = 0.f;
#pragma unroll
  for (int v = 0; v < RV; ++v) {
    const int64_t i = threadIdx.x + v * 256;
    keep[v] = make_uint4(0, 0, 0, 0);
    if (i < nvec) keep[v] = *(const uint4*)(xr + i * 8);
    float f[8];
    unpack8<T>(keep[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(f[j]));
  }
  for (int64_t i = threadIdx.x + RV * 256; i < nvec; i += 256) {
    float f[8];
    unpack8<T>(*(const uint4*)(xr + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(f[j]));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = mx / FP8_MAX;
  if (threadIdx.x == 0) scales[row] = s;
  const float inv = s > 0.f ? 1.0f / s : 0.f;
#pragma unroll
  for (int v = 0; v < RV; ++v) {
    const int64_t i = threadIdx.x + v * 256;
    if (i < nvec) {
      float f[8];
      unpack8<T>(keep[v], f);
      *(uint2*)(qr + i * 8) = quant8(f, inv);
    }
  }
  for (int64_t i = threadIdx.x + RV * 256; i < nvec; i += 256) {
    float f[8];
    unpack8<T>(*(const uint4*)(xr + i * 8), f);
    *(uint2*)(qr + i * 8) = quant8(f, inv);
  }
}

extern "C" int mi_fp8_quant_per_token(const void* x, void* q, float* scales, int64_t M, int64_t K,
                                      int64_t ldx, int dtype, void* stream) {
  MI_CHECK_ARG(M >= 0 && K > 0);
  if (M == 0) return MI_OK;
  MI_CHECK_ARG(x && q && scales);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (K % 8 != 0 || ldx % 8 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_quant_per_token: K and ldx must be multiples of 8");
  MI_CHECK_ARG((((uintptr_t)x & 15) | ((uintptr_t)q & 7)) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16)
    quant_token_kernel<bf16_t, 2><<<(unsigned)M, 256, 0, st>>>((const bf16_t*)x, (uint8_t*)q, scales, K, ldx);
  else
    quant_token_kernel<f16_t, 2><<<(unsigned)M, 256, 0, st>>>((const f16_t*)x, (uint8_t*)q, scales, K, ldx);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
