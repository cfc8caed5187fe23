// Fused consumers of split-K slabs (widened): the FP8 decode GEMM leaves
// raw fp32 accumulators in S slabs [S][M][N]; these kernels sum them, apply the GEMM's own epilogue
// (x = round_T(sum * sa * sb)) and then the next op(s) of the layer in the same pass, removing the
// standalone reduce / rope / kv-write / norm / quant launches (each ~5 us at M = 128).
// Every rounding of the unfused sequence is reproduced, so results are bit-identical to it.
// Public entry points (include/mi_hotpath.h): mi_fp8_gemm_add_rmsnorm_fp8, mi_fp8_gemm_rope_kvwrite,
// mi_fp8_gemm_silu_mul_fp8 = partial GEMM (fp8_gemm.hip) + one consumer launch.
#include "common.h"

MI_INTERNAL int mi_fp8_gemm_plan_splits(int64_t M, int64_t N, int64_t K);
MI_INTERNAL int mi_fp8_gemm_partial(const void* a, const void* b_nk, float* slabs, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldb, void* stream);
MI_INTERNAL int mi_fp8_gemm_silu_epilogue(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                                         void* q_out, const float* q_scale, int64_t M, int64_t I, int64_t K, int64_t lda,
                                         int64_t ldb, int dtype, void* stream);   // rc 1 = shape not eligible

template <typename T> __device__ __forceinline__ float rndT(float v) { return round_to<T>(v); }

template <typename T> __device__ __forceinline__ void unpack8g(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = Elem<T>::lo(w[j]); f[2 * j + 1] = Elem<T>::hi(w[j]); }
}
template <typename T> __device__ __forceinline__ uint4 pack8g(const float (&f)[8]) {
  return make_uint4(pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7]));
}
__device__ __forceinline__ uint2 quant8g(const float (&f)[8], float inv) {
  uint32_t lo = 0, hi = 0;
  auto c = [](float v) { return fmaxf(fminf(v, 448.0f), -448.0f); };
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[0] * inv), c(f[1] * inv), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[2] * inv), c(f[3] * inv), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[4] * inv), c(f[5] * inv), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[6] * inv), c(f[7] * inv), hi, true);
  return make_uint2(lo, hi);
}

// ------------------------------------------------ slabs -> (+residual) -> RMSNorm -> fp8
// == fp8_gemm reduce epilogue, then RMSNorm.forward_native with residual (layernorm.py:128-146),
// then static per-tensor quant.  One workgroup per row.
template <typename T, int VPT>
__global__ __launch_bounds__(256) void slab_add_rmsnorm_fp8_kernel(const float* __restrict__ slab, int S,
                                                                   const float* __restrict__ sa, const float* __restrict__ sb,
                                                                   T* __restrict__ residual, const T* __restrict__ w,
                                                                   uint8_t* __restrict__ q_out, const float* __restrict__ q_scale,
                                                                   T* __restrict__ out, int64_t M, int64_t H, float eps) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t nvec = H / 8;
  float v[VPT][8];
  float ss = 0.f;
  // slab sum in split order, then the GEMM epilogue exactly as fp8_gemm_reduce_kernel does it
  // (acc * sa * sb, one rounding to T), then the residual add in fp32.  The kernel is latency-bound (one
  // row per workgroup): all slab / residual loads of a thread are requested before the first add.
  constexpr int SB = 8;   // slabs per batch of loads
  const float sav = sa[0], sbv = sb[0];
  // Latency-bound (one row per workgroup, a few dependent round trips): EVERY load of the thread -- the first SB slabs
  // of all its vectors, the residual and the norm weight -- is requested before the first add, so the kernel pays one
  // memory latency for them instead of one per vector and one more for the weight.
  // (hidden sizes up to 4096: two vectors per thread = ~210 VGPRs; wider rows keep one vector's loads at a time)
  constexpr bool PRE = VPT <= 2;
  constexpr int NPRE = PRE ? VPT : 1;
  f32x4 la[NPRE][SB], lb[NPRE][SB];
  uint4 rres[VPT], wreg[VPT];
  auto load_first = [&](int i, int slot) __attribute__((always_inline)) {
    const int64_t c = threadIdx.x + i * 256;
#pragma unroll
    for (int s2 = 0; s2 < SB; ++s2)
      if (s2 < S) {
        la[slot][s2] = *(const f32x4*)(slab + (s2 * M + row) * H + c * 8);
        lb[slot][s2] = *(const f32x4*)(slab + (s2 * M + row) * H + c * 8 + 4);
      }
  };
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t c = threadIdx.x + i * 256;
    rres[i] = make_uint4(0, 0, 0, 0);
    wreg[i] = make_uint4(0, 0, 0, 0);
    if (c < nvec) {
      if constexpr (PRE) load_first(i, i);
      if (residual) rres[i] = *(const uint4*)(residual + row * H + c * 8);
      wreg[i] = *(const uint4*)(w + c * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t c = threadIdx.x + i * 256;
    if (c < nvec) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      if constexpr (!PRE) load_first(i, 0);
#pragma unroll
      for (int s2 = 0; s2 < SB; ++s2)
        if (s2 < S) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += la[PRE ? i : 0][s2][j]; acc[4 + j] += lb[PRE ? i : 0][s2][j]; }
        }
      for (int s0 = SB; s0 < S; s0 += SB) {      // more than SB slabs: further batches, same split order
        f32x4 a[SB], b[SB];
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2)
          if (s0 + s2 < S) {
            a[s2] = *(const f32x4*)(slab + ((s0 + s2) * M + row) * H + c * 8);
            b[s2] = *(const f32x4*)(slab + ((s0 + s2) * M + row) * H + c * 8 + 4);
          }
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2)
          if (s0 + s2 < S) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += a[s2][j]; acc[4 + j] += b[s2][j]; }
          }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = rndT<T>(acc[j] * sav * sbv);
      if (residual) {
        float r[8];
        unpack8g<T>(rres[i], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += r[j];
        *(uint4*)(residual + row * H + c * 8) = pack8g<T>(v[i]);
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
  float qinv = 0.f;
  if (q_out) {
    const float qs = *q_scale;
    qinv = qs > 0.f ? 1.0f / qs : 0.f;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t c = threadIdx.x + i * 256;
    if (c < nvec) {
      float wf[8], o[8];
      unpack8g<T>(wreg[i], wf);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rndT<T>(v[i][j] * inv * wf[j]);
      if (out) *(uint4*)(out + row * H + c * 8) = pack8g<T>(o);
      if (q_out) *(uint2*)(q_out + row * H + c * 8) = quant8g(o, qinv);
    }
  }
}

static int slab_add_rmsnorm_fp8(const float* slab, int S, const float* scale_a, const float* scale_b,
                                void* residual, const void* weight, void* q_out, const float* q_scale,
                                void* out, int64_t M, int64_t H, float eps, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vpt = (int)cdiv64(H / 8, 256);
#define LAUNCH(TT, V) slab_add_rmsnorm_fp8_kernel<TT, V><<<(unsigned)M, 256, 0, st>>>(slab, S, scale_a, scale_b, (TT*)residual, (const TT*)weight, (uint8_t*)q_out, q_scale, (TT*)out, M, H, eps)
  if (dtype == MI_BF16) { if (vpt <= 1) LAUNCH(bf16_t, 1); else if (vpt <= 2) LAUNCH(bf16_t, 2); else if (vpt <= 4) LAUNCH(bf16_t, 4); else LAUNCH(bf16_t, 8); }
  else { if (vpt <= 1) LAUNCH(f16_t, 1); else if (vpt <= 2) LAUNCH(f16_t, 2); else if (vpt <= 4) LAUNCH(f16_t, 4); else LAUNCH(f16_t, 8); }
#undef LAUNCH
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------ slabs (qkv) -> RoPE -> q out + KV-pool write
// == fp8_gemm reduce epilogue, RotaryEmbedding.forward_native (neox, rotary_embedding.py:49-166) on q,k
// and set_kv_buffer (memory_pool.py:454-455).  One thread per 8 rotation pairs / 16 v elements.
template <typename T>
__global__ __launch_bounds__(256) void slab_rope_kvwrite_kernel(const float* __restrict__ slab, int S,
                                                                const float* __restrict__ sa, const float* __restrict__ sb,
                                                                const int64_t* __restrict__ positions,
                                                                const float* __restrict__ cos_sin, T* __restrict__ q_out,
                                                                T* __restrict__ k_cache, T* __restrict__ v_cache,
                                                                const int64_t* __restrict__ loc, int64_t tokens, int Hq,
                                                                int Hkv, int D, int64_t ldq, int64_t cache_stride_k,
                                                                int64_t cache_stride_v) {
  const int half = D / 2, vph = half / 8;
  const int heads = Hq + 2 * Hkv;
  const int64_t N = (int64_t)heads * D;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_tok = (int64_t)heads * vph;
  if (gid >= tokens * per_tok) return;
  const int64_t t = gid / per_tok;
  const int rem = (int)(gid % per_tok);
  const int h = rem / vph, c = rem % vph;
  const int64_t col = (int64_t)h * D + c * 8;
  // requested first, so the dependent cos/sin loads can go out while the slab loads are still in flight
  const int64_t pos = (h < Hq + Hkv) ? positions[t] : 0;
  const int64_t slot = (h >= Hq) ? loc[t] : 0;
  float acc1[8], acc2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc1[j] = 0.f; acc2[j] = 0.f; }
  constexpr int SB = 8;   // slabs per batch: all loads of a batch are requested before the first add
  for (int s0 = 0; s0 < S; s0 += SB) {
    f32x4 a[SB], b[SB], e[SB], f[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s)
      if (s0 + s < S) {
        const float* base = slab + ((s0 + s) * tokens + t) * N + col;
        a[s] = *(const f32x4*)base; b[s] = *(const f32x4*)(base + 4);
        e[s] = *(const f32x4*)(base + half); f[s] = *(const f32x4*)(base + half + 4);
      }
#pragma unroll
    for (int s = 0; s < SB; ++s)
      if (s0 + s < S) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[j] += a[s][j]; acc1[4 + j] += b[s][j]; acc2[j] += e[s][j]; acc2[4 + j] += f[s][j]; }
      }
  }
  float x1[8], x2[8], o1[8], o2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x1[j] = rndT<T>(acc1[j] * sa[0] * sb[0]); x2[j] = rndT<T>(acc2[j] * sa[0] * sb[0]); }
  if (h < Hq + Hkv) {  // q or k head: rotate
    const float* cs = cos_sin + pos * D;
    const f32x4 c0 = *(const f32x4*)(cs + c * 8), c1 = *(const f32x4*)(cs + c * 8 + 4);
    const f32x4 s0 = *(const f32x4*)(cs + half + c * 8), s1 = *(const f32x4*)(cs + half + c * 8 + 4);
    const float cv[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float co = rndT<T>(cv[j]), si = rndT<T>(sv[j]);
      o1[j] = rndT<T>(rndT<T>(x1[j] * co) - rndT<T>(x2[j] * si));
      o2[j] = rndT<T>(rndT<T>(x2[j] * co) + rndT<T>(x1[j] * si));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { o1[j] = x1[j]; o2[j] = x2[j]; }
  }
  T* dst;
  if (h < Hq) dst = q_out + t * ldq + (int64_t)h * D;
  else if (h < Hq + Hkv) dst = k_cache + slot * cache_stride_k + (int64_t)(h - Hq) * D;
  else dst = v_cache + slot * cache_stride_v + (int64_t)(h - Hq - Hkv) * D;
  *(uint4*)(dst + c * 8) = pack8g<T>(o1);
  *(uint4*)(dst + half + c * 8) = pack8g<T>(o2);
}

static int slab_rope_kvwrite(const float* slab, int S, const float* scale_a, const float* scale_b,
                             const int64_t* positions, const float* cos_sin_cache, void* q_out, void* k_cache,
                             void* v_cache, const int64_t* loc, int64_t tokens, int64_t num_q_heads,
                             int64_t num_kv_heads, int64_t head_dim, int64_t ldq, int64_t cache_stride_k,
                             int64_t cache_stride_v, int dtype, void* stream) {
  const int64_t total = tokens * (num_q_heads + 2 * num_kv_heads) * (head_dim / 16);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16)
    slab_rope_kvwrite_kernel<bf16_t><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(slab, S, scale_a, scale_b, positions, cos_sin_cache, (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache, loc, tokens, (int)num_q_heads, (int)num_kv_heads, (int)head_dim, ldq, cache_stride_k, cache_stride_v);
  else
    slab_rope_kvwrite_kernel<f16_t><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(slab, S, scale_a, scale_b, positions, cos_sin_cache, (f16_t*)q_out, (f16_t*)k_cache, (f16_t*)v_cache, loc, tokens, (int)num_q_heads, (int)num_kv_heads, (int)head_dim, ldq, cache_stride_k, cache_stride_v);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------ slabs (gate_up) -> SiLU(gate) * up -> fp8
// == fp8_gemm reduce epilogue, SiluAndMul.forward_native (activation.py:56-58), static per-tensor quant.
// q_scale == nullptr: T-typed output (q_out points at T [M, I]): the 16-bit-activation form for the int4 linears
template <typename T>
__global__ __launch_bounds__(256) void slab_silu_mul_fp8_kernel(const float* __restrict__ slab, int S,
                                                                const float* __restrict__ sa, const float* __restrict__ sb,
                                                                uint8_t* __restrict__ q_out, const float* __restrict__ q_scale,
                                                                int64_t M, int64_t I) {
  const float qs = q_scale ? *q_scale : 1.f;
  const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
  const int64_t vpr = I / 8, N = 2 * I;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < M * vpr; gid += (int64_t)gridDim.x * 256) {
    const int64_t r = gid / vpr, c = gid % vpr;
    float g[8], u[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { g[j] = 0.f; u[j] = 0.f; }
    for (int s = 0; s < S; ++s) {
      const float* base = slab + (s * M + r) * N + c * 8;
      const f32x4 a = *(const f32x4*)base, b = *(const f32x4*)(base + 4);
      const f32x4 e = *(const f32x4*)(base + I), f = *(const f32x4*)(base + I + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { g[j] += a[j]; g[4 + j] += b[j]; u[j] += e[j]; u[4 + j] += f[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gv = rndT<T>(g[j] * sa[0] * sb[0]), uv = rndT<T>(u[j] * sa[0] * sb[0]);
      o[j] = rndT<T>(rndT<T>(silu_f32(gv)) * uv);
    }
    if (q_scale) *(uint2*)(q_out + r * I + c * 8) = quant8g(o, qinv);
    else *(uint4*)((T*)q_out + r * I + c * 8) = pack8g<T>(o);
  }
}

// ============================================================================ public entry points
// Batches beyond one pass of the GEMM (256 rows with the 8-wave kernel, else 128; graph batch sizes up to 512) run
// the same fused pair once per chunk of rows, through the same slabs: every output row is produced exactly as in a <= 128-row call on its chunk.
#define FUSED_MAX_M 512
MI_INTERNAL int64_t mi_fp8_gemm_partial_max_rows(int64_t N);   // fp8_gemm.hip: 256 (8-wave deep-ring kernel) or 128
extern "C" int64_t mi_fp8_gemm_fused_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (N > 0 && M > mi_fp8_gemm_partial_max_rows(N) && M <= FUSED_MAX_M) M = mi_fp8_gemm_partial_max_rows(N);
  const int S = mi_fp8_gemm_plan_splits(M, N, K);
  return S > 0 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
}

#define FUSED_PROLOGUE(NAME)                                                                              \
  MI_CHECK_ARG(M >= 0 && N > 0 && K > 0);                                                                 \
  if (M == 0) return MI_OK;                                                                               \
  MI_CHECK_ARG(a && b_nk && scale_a && scale_b);                                                          \
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);                                                     \
  const int S = mi_fp8_gemm_plan_splits(M, N, K);                                                         \
  if (S <= 0) MI_FAIL(MI_ERR_UNSUPPORTED, NAME ": decode shapes only (M <= 512, K %% 128 == 0)");         \
  const int64_t need = (int64_t)S * M * N * (int64_t)sizeof(float);                                       \
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))                                \
    MI_FAIL(MI_ERR_INVALID, NAME ": workspace of mi_fp8_gemm_fused_workspace_bytes() = %lld bytes needed", (long long)need)

extern "C" int mi_fp8_gemm_add_rmsnorm_fp8(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                                           void* residual, const void* norm_weight, void* out, void* q_out,
                                           const float* q_scale, int64_t M, int64_t N, int64_t K, int64_t lda,
                                           int64_t ldb, float eps, int dtype, void* workspace,
                                           int64_t workspace_bytes, void* stream) {
  const int64_t FUSED_CHUNK = N > 0 ? mi_fp8_gemm_partial_max_rows(N) : 128;
  if (M > FUSED_CHUNK && M <= FUSED_MAX_M) {
    MI_CHECK_ARG(a && b_nk && (out || q_out));
    const int64_t es = 2;   // bf16 / fp16
    for (int64_t m0 = 0; m0 < M; m0 += FUSED_CHUNK) {
      const int rc = mi_fp8_gemm_add_rmsnorm_fp8(
          (const char*)a + m0 * lda, b_nk, scale_a, scale_b, residual ? (char*)residual + m0 * N * es : nullptr,
          norm_weight, out ? (char*)out + m0 * N * es : nullptr, q_out ? (char*)q_out + m0 * N : nullptr, q_scale,
          M - m0 < FUSED_CHUNK ? M - m0 : FUSED_CHUNK, N, K, lda, ldb, eps, dtype, workspace, workspace_bytes, stream);
      if (rc != MI_OK) return rc;
    }
    return MI_OK;
  }
  FUSED_PROLOGUE("mi_fp8_gemm_add_rmsnorm_fp8");
  MI_CHECK_ARG(norm_weight && (out || q_out) && (!q_out || q_scale));
  if (N % 8 != 0 || N > 256 * 8 * 8) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_gemm_add_rmsnorm_fp8: N must be a multiple of 8, <= 16384");
  const int rc = mi_fp8_gemm_partial(a, b_nk, (float*)workspace, M, N, K, lda, ldb, stream);
  if (rc != MI_OK) return rc;
  return slab_add_rmsnorm_fp8((const float*)workspace, S, scale_a, scale_b, residual, norm_weight, q_out, q_scale, out,
                              M, N, eps, dtype, stream);
}

extern "C" int mi_fp8_gemm_rope_kvwrite(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                                        const int64_t* positions, const float* cos_sin_cache, void* q_out,
                                        void* k_cache, void* v_cache, const int64_t* loc, int64_t M,
                                        int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t K,
                                        int64_t lda, int64_t ldb, int64_t ldq, int64_t cache_stride_k,
                                        int64_t cache_stride_v, int dtype, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  const int64_t FUSED_CHUNK = mi_fp8_gemm_partial_max_rows((num_q_heads + 2 * num_kv_heads) * head_dim);
  if (M > FUSED_CHUNK && M <= FUSED_MAX_M) {
    MI_CHECK_ARG(a && b_nk && positions && q_out && loc);
    for (int64_t m0 = 0; m0 < M; m0 += FUSED_CHUNK) {
      const int rc = mi_fp8_gemm_rope_kvwrite(
          (const char*)a + m0 * lda, b_nk, scale_a, scale_b, positions ? positions + m0 : nullptr, cos_sin_cache,
          q_out ? (char*)q_out + m0 * ldq * 2 : nullptr, k_cache, v_cache, loc ? loc + m0 : nullptr,
          M - m0 < FUSED_CHUNK ? M - m0 : FUSED_CHUNK, num_q_heads, num_kv_heads, head_dim, K, lda, ldb, ldq,
          cache_stride_k, cache_stride_v, dtype, workspace, workspace_bytes, stream);
      if (rc != MI_OK) return rc;
    }
    return MI_OK;
  }
  const int64_t N = (num_q_heads + 2 * num_kv_heads) * head_dim;
  FUSED_PROLOGUE("mi_fp8_gemm_rope_kvwrite");
  MI_CHECK_ARG(positions && cos_sin_cache && q_out && k_cache && v_cache && loc && num_q_heads > 0 && num_kv_heads > 0);
  MI_CHECK_ARG(((uintptr_t)cos_sin_cache & 15) == 0);
  if (head_dim % 16 != 0 || ldq % 8 || cache_stride_k % 8 || cache_stride_v % 8)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_gemm_rope_kvwrite: head_dim %% 16 and 16-byte aligned rows required");
  const int rc = mi_fp8_gemm_partial(a, b_nk, (float*)workspace, M, N, K, lda, ldb, stream);
  if (rc != MI_OK) return rc;
  return slab_rope_kvwrite((const float*)workspace, S, scale_a, scale_b, positions, cos_sin_cache, q_out, k_cache,
                           v_cache, loc, M, num_q_heads, num_kv_heads, head_dim, ldq, cache_stride_k, cache_stride_v,
                           dtype, stream);
}

extern "C" int mi_fp8_gemm_silu_mul_fp8(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                                        void* q_out, const float* q_scale, int64_t M, int64_t I, int64_t K,
                                        int64_t lda, int64_t ldb, int dtype, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  MI_CHECK_ARG(I > 0 && q_out && q_scale);
  const int64_t FUSED_CHUNK = mi_fp8_gemm_partial_max_rows(2 * I);
  if (M > FUSED_CHUNK && M <= FUSED_MAX_M) {
    MI_CHECK_ARG(a && b_nk);
    for (int64_t m0 = 0; m0 < M; m0 += FUSED_CHUNK) {
      const int rc = mi_fp8_gemm_silu_mul_fp8((const char*)a + m0 * lda, b_nk, scale_a, scale_b, (char*)q_out + m0 * I,
                                              q_scale, M - m0 < FUSED_CHUNK ? M - m0 : FUSED_CHUNK, I, K, lda, ldb,
                                              dtype, workspace, workspace_bytes, stream);
      if (rc != MI_OK) return rc;
    }
    return MI_OK;
  }
  if (M > 0 && a && b_nk && scale_a && scale_b) {   // in-kernel epilogue when the GEMM needs no split-K
    const int rc = mi_fp8_gemm_silu_epilogue(a, b_nk, scale_a, scale_b, q_out, q_scale, M, I, K, lda, ldb, dtype, stream);
    if (rc <= 0) return rc;
  }
  const int64_t N = 2 * I;
  FUSED_PROLOGUE("mi_fp8_gemm_silu_mul_fp8");
  if (I % 8 != 0) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_gemm_silu_mul_fp8: I must be a multiple of 8");
  const int rc = mi_fp8_gemm_partial(a, b_nk, (float*)workspace, M, N, K, lda, ldb, stream);
  if (rc != MI_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = M * (I / 8);
  const unsigned blocks = (unsigned)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  if (dtype == MI_BF16)
    slab_silu_mul_fp8_kernel<bf16_t><<<blocks, 256, 0, st>>>((const float*)workspace, S, scale_a, scale_b, (uint8_t*)q_out, q_scale, M, I);
  else
    slab_silu_mul_fp8_kernel<f16_t><<<blocks, 256, 0, st>>>((const float*)workspace, S, scale_a, scale_b, (uint8_t*)q_out, q_scale, M, I);
  MI_CHECK_LAUNCH();
  return MI_OK;
}


// ============================================================================ int4 (AWQ / GPTQ) linears fused with
// their consumer: the same three consumer kernels on the slabs of w4a16_xw_kernel, 16-bit activations in and out (unit
// scales: x = round_T(sum of the slabs), exactly what w4_reduce_kernel gives without a bias).  The C4 decode layer
// (Llama-2-7B AWQ) drops from 19 launches to 9: norm (first layer) | qkv + rope + kv-write | attention, merge |
// o + add + norm | gate_up + silu*mul | down + add + norm.  Bit-identical to the unfused sequence
// (tests/test_fused_gpu.py::test_w4_fused_*).
MI_INTERNAL int mi_w4a16_plan_splits(int64_t M, int64_t N, int64_t K, int64_t group);
MI_INTERNAL int mi_w4a16_gemm_partial(const void* x, const void* qw_native, const void* zs_native, float* slabs, int64_t M,
                                     int64_t N, int64_t K, int64_t group, int64_t ldx, int dtype, void* stream);
__device__ float mi_unit_scale = 1.0f;
static const float* unit_scale_ptr() {
  static const float* p = nullptr;
  if (!p) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(mi_unit_scale)) == hipSuccess) p = (const float*)d;
  }
  return p;
}
extern "C" int64_t mi_w4a16_fused_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t group_size) {
  if (M <= 0 || M > 128) return 0;
  const int S = mi_w4a16_plan_splits(M, N, K, group_size > 0 ? group_size : K);
  return S > 0 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
}
#define W4_FUSED_PROLOGUE(NAME)                                                                          \
  MI_CHECK_ARG(M > 0 && N > 0 && K > 0 && x && qw_native && zs_native);                                  \
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);                                                    \
  if (group_size <= 0) group_size = K;                                                                   \
  const int S = M <= 128 ? mi_w4a16_plan_splits(M, N, K, group_size) : 0;                                \
  if (S <= 0) MI_FAIL(MI_ERR_UNSUPPORTED, NAME ": decode shapes only (M <= 128, K %% 128 == 0, N %% 64 == 0, group %% 128 == 0)"); \
  const int64_t need = (int64_t)S * M * N * (int64_t)sizeof(float);                                      \
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))                               \
    MI_FAIL(MI_ERR_INVALID, NAME ": workspace of mi_w4a16_fused_workspace_bytes() = %lld bytes needed", (long long)need); \
  const float* one = unit_scale_ptr();                                                                   \
  if (!one) MI_FAIL(MI_ERR_LAUNCH, NAME ": device constant not available")

extern "C" int mi_w4a16_gemm_add_rmsnorm(const void* x, const void* qw_native, const void* zs_native, void* residual,
                                         const void* norm_weight, void* out, int64_t M, int64_t N, int64_t K,
                                         int64_t group_size, int64_t ldx, float eps, int dtype, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  W4_FUSED_PROLOGUE("mi_w4a16_gemm_add_rmsnorm");
  MI_CHECK_ARG(norm_weight && out);
  if (N % 8 != 0 || N > 256 * 8 * 8) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4a16_gemm_add_rmsnorm: N must be a multiple of 8, <= 16384");
  const int rc = mi_w4a16_gemm_partial(x, qw_native, zs_native, (float*)workspace, M, N, K, group_size, ldx, dtype, stream);
  if (rc != MI_OK) return rc;
  return slab_add_rmsnorm_fp8((const float*)workspace, S, one, one, residual, norm_weight, nullptr, nullptr, out, M, N, eps,
                              dtype, stream);
}

extern "C" int mi_w4a16_gemm_rope_kvwrite(const void* x, const void* qw_native, const void* zs_native,
                                          const int64_t* positions, const float* cos_sin_cache, void* q_out, void* k_cache,
                                          void* v_cache, const int64_t* loc, int64_t M, int64_t num_q_heads,
                                          int64_t num_kv_heads, int64_t head_dim, int64_t K, int64_t group_size, int64_t ldx,
                                          int64_t ldq, int64_t cache_stride_k, int64_t cache_stride_v, int dtype,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
  const int64_t N = (num_q_heads + 2 * num_kv_heads) * head_dim;
  W4_FUSED_PROLOGUE("mi_w4a16_gemm_rope_kvwrite");
  MI_CHECK_ARG(positions && cos_sin_cache && q_out && k_cache && v_cache && loc && num_q_heads > 0 && num_kv_heads > 0);
  MI_CHECK_ARG(((uintptr_t)cos_sin_cache & 15) == 0);
  if (head_dim % 16 != 0 || ldq % 8 || cache_stride_k % 8 || cache_stride_v % 8)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4a16_gemm_rope_kvwrite: head_dim %% 16 and 16-byte aligned rows required");
  const int rc = mi_w4a16_gemm_partial(x, qw_native, zs_native, (float*)workspace, M, N, K, group_size, ldx, dtype, stream);
  if (rc != MI_OK) return rc;
  return slab_rope_kvwrite((const float*)workspace, S, one, one, positions, cos_sin_cache, q_out, k_cache, v_cache, loc, M,
                           num_q_heads, num_kv_heads, head_dim, ldq, cache_stride_k, cache_stride_v, dtype, stream);
}

extern "C" int mi_w4a16_gemm_silu_mul(const void* x, const void* qw_native, const void* zs_native, void* out, int64_t M,
                                      int64_t I, int64_t K, int64_t group_size, int64_t ldx, int dtype, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  MI_CHECK_ARG(I > 0 && out);
  const int64_t N = 2 * I;
  W4_FUSED_PROLOGUE("mi_w4a16_gemm_silu_mul");
  if (I % 8 != 0 || ((uintptr_t)out & 15)) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4a16_gemm_silu_mul: I %% 8 == 0 and a 16-byte aligned output required");
  const int rc = mi_w4a16_gemm_partial(x, qw_native, zs_native, (float*)workspace, M, N, K, group_size, ldx, dtype, stream);
  if (rc != MI_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = M * (I / 8);
  const unsigned blocks = (unsigned)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  if (dtype == MI_BF16)
    slab_silu_mul_fp8_kernel<bf16_t><<<blocks, 256, 0, st>>>((const float*)workspace, S, one, one, (uint8_t*)out, nullptr, M, I);
  else
    slab_silu_mul_fp8_kernel<f16_t><<<blocks, 256, 0, st>>>((const float*)workspace, S, one, one, (uint8_t*)out, nullptr, M, I);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
