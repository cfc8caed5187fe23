// Memory-bound elementwise ops that sit between the hot-path kernels in a Llama layer
// (SURVEY section 8f "next" rows): fused add + RMSNorm, NeoX RoPE, SiLU-and-mul.  Arithmetic
// follows the reference's *native* (torch) forms op by op, including their roundings.
// All 16-byte vector loads/stores; HBM/L2-bound, no MFMA.
#include "common.h"

template <typename T> __device__ __forceinline__ float rnd(float v) { return round_to<T>(v); }

template <typename T> __device__ __forceinline__ void unpack8f(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = Elem<T>::lo(w[j]);
    f[2 * j + 1] = Elem<T>::hi(w[j]);
  }
}
template <typename T> __device__ __forceinline__ uint4 pack8f(const float (&f)[8]) {
  return make_uint4(pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7]));
}

// ---------------------------------------------------------------- (fused add +) RMSNorm
// layers/layernorm.py:128-146 (forward_native): x32 = x (+ residual); residual = x32 -> T;
// out = (x32 * rsqrt(mean(x32^2) + eps) * w) -> T.   One 256-thread workgroup per row.
// 8 floats (already rounded to T) -> 8 fp8 bytes with the static per-tensor scale: identical bits to
// running mi_fp8_quant_per_tensor (mode 1) on the T-typed output.
__device__ __forceinline__ uint2 quant8_static(const float (&f)[8], float inv) {
  uint32_t lo = 0, hi = 0;
  auto c = [](float v) { return fmaxf(fminf(v, 448.0f), -448.0f); };
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[0] * inv), c(f[1] * inv), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[2] * inv), c(f[3] * inv), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[4] * inv), c(f[5] * inv), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[6] * inv), c(f[7] * inv), hi, true);
  return make_uint2(lo, hi);
}

template <typename T, int VPT>  // VPT 16-byte vectors per thread (H <= 256*8*VPT)
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, T* __restrict__ residual,
                                                      const T* __restrict__ w, T* __restrict__ out, int64_t H,
                                                      int64_t ldx, int64_t ldr, int64_t ldo, float eps,
                                                      uint8_t* __restrict__ q_out, const float* __restrict__ q_scale) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t nvec = H / 8;
  float v[VPT][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t c = threadIdx.x + i * 256;
    if (c < nvec) {
      unpack8f<T>(*(const uint4*)(x + row * ldx + c * 8), v[i]);
      if (residual) {
        float r[8];
        unpack8f<T>(*(const uint4*)(residual + row * ldr + c * 8), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += r[j];
        *(uint4*)(residual + row * ldr + c * 8) = pack8f<T>(v[i]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  ss = red[0] + red[1] + red[2] + red[3];
  const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t c = threadIdx.x + i * 256;
    if (c < nvec) {
      float wf[8], o[8];
      unpack8f<T>(*(const uint4*)(w + c * 8), wf);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[i][j] * inv * wf[j];
      if (out) *(uint4*)(out + row * ldo + c * 8) = pack8f<T>(o);
      if (q_out) {
        const float qs = *q_scale;
        const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rnd<T>(o[j]);
        *(uint2*)(q_out + row * H + c * 8) = quant8_static(o, qinv);
      }
    }
  }
}

static int rmsnorm_impl(const void* x, void* residual, const void* weight, void* out, int64_t M, int64_t H,
                        int64_t ldx, int64_t ldr, int64_t ldo, float eps, int dtype, void* stream, void* q_out,
                        const float* q_scale) {
  MI_CHECK_ARG(M >= 0 && H > 0);
  if (M == 0) return MI_OK;
  MI_CHECK_ARG(x && weight && (out || q_out));
  MI_CHECK_ARG(!q_out || q_scale);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (H % 8 != 0 || H > 256 * 8 * 8 || ldx % 8 || ldo % 8 || (residual && ldr % 8))
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_rmsnorm: H must be a multiple of 8 and <= 16384 (H=%lld)", (long long)H);
  hipStream_t st = (hipStream_t)stream;
  const int vpt = (int)cdiv64(H / 8, 256);
#define LAUNCH_RMS(TT, V) \
  rmsnorm_kernel<TT, V><<<(unsigned)M, 256, 0, st>>>((const TT*)x, (TT*)residual, (const TT*)weight, (TT*)out, H, ldx, ldr, ldo, eps, (uint8_t*)q_out, q_scale)
  if (dtype == MI_BF16) {
    if (vpt <= 1) LAUNCH_RMS(bf16_t, 1); else if (vpt <= 2) LAUNCH_RMS(bf16_t, 2); else if (vpt <= 4) LAUNCH_RMS(bf16_t, 4); else LAUNCH_RMS(bf16_t, 8);
  } else {
    if (vpt <= 1) LAUNCH_RMS(f16_t, 1); else if (vpt <= 2) LAUNCH_RMS(f16_t, 2); else if (vpt <= 4) LAUNCH_RMS(f16_t, 4); else LAUNCH_RMS(f16_t, 8);
  }
#undef LAUNCH_RMS
  MI_CHECK_LAUNCH();
  return MI_OK;
}

extern "C" int mi_rmsnorm(const void* x, void* residual, const void* weight, void* out, int64_t M, int64_t H,
                          int64_t ldx, int64_t ldr, int64_t ldo, float eps, int dtype, void* stream) {
  MI_CHECK_ARG(out != nullptr);
  return rmsnorm_impl(x, residual, weight, out, M, H, ldx, ldr, ldo, eps, dtype, stream, nullptr, nullptr);
}

extern "C" int mi_rmsnorm_fp8(const void* x, void* residual, const void* weight, void* out, void* q_out,
                              const float* q_scale, int64_t M, int64_t H, int64_t ldx, int64_t ldr, int64_t ldo,
                              float eps, int dtype, void* stream) {
  MI_CHECK_ARG(q_out != nullptr && q_scale != nullptr);
  return rmsnorm_impl(x, residual, weight, out, M, H, ldx, ldr, ldo, eps, dtype, stream, q_out, q_scale);
}

// ------------------------------------------------------------------------- NeoX RoPE
// layers/rotary_embedding.py:49-74,138-166 (forward_native, neox style, rotary_dim == head_dim):
// cos/sin are cast to T; o1 = x1*cos - x2*sin, o2 = x2*cos + x1*sin with every product and
// sum rounded to T like torch's elementwise ops.  cos_sin_cache [max_pos, D] = [cos | sin], fp32.
// One thread per 8 rotation pairs; grid = tokens x (Hq + Hkv) heads.
template <typename T>
__global__ __launch_bounds__(256) void rope_neox_kernel(T* __restrict__ q, T* __restrict__ k,
                                                        const int64_t* __restrict__ positions,
                                                        const float* __restrict__ cos_sin, int64_t tokens, int Hq,
                                                        int Hkv, int D, int64_t ldq, int64_t ldk) {
  const int half = D / 2, vph = half / 8;  // vectors per half head
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_tok = (int64_t)(Hq + Hkv) * vph;
  if (gid >= tokens * per_tok) return;
  const int64_t t = gid / per_tok;
  const int rem = (int)(gid % per_tok);
  const int h = rem / vph, c = rem % vph;
  T* base = (h < Hq) ? q + t * ldq + (int64_t)h * D : k + t * ldk + (int64_t)(h - Hq) * D;
  const float* cs = cos_sin + positions[t] * D;
  float x1[8], x2[8], o1[8], o2[8];
  unpack8f<T>(*(const uint4*)(base + c * 8), x1);
  unpack8f<T>(*(const uint4*)(base + half + c * 8), x2);
  // cos / sin of the 8 pairs as four 16-byte loads (rows of the cache are D floats: 16-byte aligned for D % 16 == 0)
  const f32x4 c0 = *(const f32x4*)(cs + c * 8), c1 = *(const f32x4*)(cs + c * 8 + 4);
  const f32x4 s0 = *(const f32x4*)(cs + half + c * 8), s1 = *(const f32x4*)(cs + half + c * 8 + 4);
  const float cv[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float co = rnd<T>(cv[j]), si = rnd<T>(sv[j]);
    o1[j] = rnd<T>(rnd<T>(x1[j] * co) - rnd<T>(x2[j] * si));
    o2[j] = rnd<T>(rnd<T>(x2[j] * co) + rnd<T>(x1[j] * si));
  }
  *(uint4*)(base + c * 8) = pack8f<T>(o1);
  *(uint4*)(base + half + c * 8) = pack8f<T>(o2);
}

extern "C" int mi_rope_neox(void* q, void* k, const int64_t* positions, const float* cos_sin_cache,
                            int64_t tokens, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                            int64_t ldq, int64_t ldk, int dtype, void* stream) {
  MI_CHECK_ARG(tokens >= 0);
  if (tokens == 0) return MI_OK;
  MI_CHECK_ARG(q && k && positions && cos_sin_cache && ((uintptr_t)cos_sin_cache & 15) == 0);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (head_dim % 16 != 0 || ldq % 8 || ldk % 8)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_rope_neox: head_dim must be a multiple of 16");
  const int64_t total = tokens * (num_q_heads + num_kv_heads) * (head_dim / 16);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16)
    rope_neox_kernel<bf16_t><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>((bf16_t*)q, (bf16_t*)k, positions, cos_sin_cache, tokens, (int)num_q_heads, (int)num_kv_heads, (int)head_dim, ldq, ldk);
  else
    rope_neox_kernel<f16_t><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>((f16_t*)q, (f16_t*)k, positions, cos_sin_cache, tokens, (int)num_q_heads, (int)num_kv_heads, (int)head_dim, ldq, ldk);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// --------------------------------------------------------------------- SiLU and mul
// layers/activation.py:56-58: F.silu(x[..., :d]) * x[..., d:]  (silu rounded to T, then the product)
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t M,
                                                       int64_t I, int64_t ldx, int64_t ldo,
                                                       uint8_t* __restrict__ q_out, const float* __restrict__ q_scale) {
  float qinv = 0.f;
  if (q_out) {
    const float qs = *q_scale;
    qinv = qs > 0.f ? 1.0f / qs : 0.f;
  }
  const int64_t vpr = I / 8;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < M * vpr; gid += (int64_t)gridDim.x * 256) {
    const int64_t r = gid / vpr, c = gid % vpr;
    float a[8], b[8], o[8];
    unpack8f<T>(*(const uint4*)(x + r * ldx + c * 8), a);
    unpack8f<T>(*(const uint4*)(x + r * ldx + I + c * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rnd<T>(silu_f32(a[j])) * b[j];
    if (out) *(uint4*)(out + r * ldo + c * 8) = pack8f<T>(o);
    if (q_out) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rnd<T>(o[j]);
      *(uint2*)(q_out + r * I + c * 8) = quant8_static(o, qinv);
    }
  }
}

static int silu_impl(const void* x, void* out, int64_t M, int64_t I, int64_t ldx, int64_t ldo, int dtype, void* stream,
                     void* q_out, const float* q_scale) {
  MI_CHECK_ARG(M >= 0 && I > 0);
  if (M == 0) return MI_OK;
  MI_CHECK_ARG(x && (out || q_out));
  MI_CHECK_ARG(!q_out || q_scale);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (I % 8 || ldx % 8 || ldo % 8) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_silu_and_mul: sizes must be multiples of 8");
  const int64_t total = M * (I / 8);
  const unsigned blocks = (unsigned)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16) silu_mul_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)out, M, I, ldx, ldo, (uint8_t*)q_out, q_scale);
  else silu_mul_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, (f16_t*)out, M, I, ldx, ldo, (uint8_t*)q_out, q_scale);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

extern "C" int mi_silu_and_mul(const void* x, void* out, int64_t M, int64_t I, int64_t ldx, int64_t ldo, int dtype,
                               void* stream) {
  MI_CHECK_ARG(out != nullptr);
  return silu_impl(x, out, M, I, ldx, ldo, dtype, stream, nullptr, nullptr);
}

extern "C" int mi_silu_and_mul_fp8(const void* x, void* out, void* q_out, const float* q_scale, int64_t M, int64_t I,
                                   int64_t ldx, int64_t ldo, int dtype, void* stream) {
  MI_CHECK_ARG(q_out != nullptr && q_scale != nullptr);
  return silu_impl(x, out, M, I, ldx, ldo, dtype, stream, q_out, q_scale);
}
