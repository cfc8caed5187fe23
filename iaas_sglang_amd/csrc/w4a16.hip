// int4 weight-only (AWQ / GPTQ) linear: dequant fused into an MFMA f16/bf16 GEMM.
//
// Checkpoint layouts pack nibbles along N (AWQ) or along K in act-order (GPTQ); neither gives
// a lane the 8 consecutive-k values an MFMA fragment wants, so weights are repacked ONCE at
// load time (mi_w4_repack, called from process_weights_after_loading) into the native layout
//   qw  : u32 [N/16][K/128][64 lanes][4]   lane l, dword s = the 8 weights
//         W[k = 128*kb + 32*s + 8*(l>>4) + j][n = 16*nt + (l&15)], j = 0..7, stored with
//         element 2i in nibble i and element 2i+1 in nibble i+4 so that
//         (w >> 4i) & 0x000F000F is the 16-bit pair (e_2i, e_2i+1) ready for the magic-number
//         int->float trick; one wave-wide 16-B load = 1 KiB contiguous = 16 n x 128 k.
//   zs  : u32 [K/g][N]  lo = scale (f16/bf16 bits), hi = zero point as float16(1024+z) /
//         bfloat16(128+z)  (GPTQ's stored-minus-one already added back)
//   perm: i32 [K] (GPTQ act-order only): native row k' holds checkpoint row perm[k'].
// The dequantised fragment is bit-identical to the reference's (w - z) * s in the activation
// dtype, so the fused GEMM equals dequant + matmul up to fp32 accumulation order.
// Bound at decode: HBM (int4 weights read once).
#include "common.h"

__device__ __constant__ int kAwqNibbleOfCol[8] = {0, 4, 1, 5, 2, 6, 3, 7};  // col 8c+j <- nibble

template <typename T> __device__ __forceinline__ uint16_t to_bits(float f) {
  T x = (T)f;
  return __builtin_bit_cast(uint16_t, x);
}

// ------------------------------------------------------------------- source accessors
__device__ __forceinline__ int src_weight(const int32_t* qweight, int layout, int64_t N, int64_t k, int64_t n) {
  if (layout == MI_W4_AWQ) {
    const uint32_t w = (uint32_t)qweight[k * (N / 8) + n / 8];
    return (w >> (4 * kAwqNibbleOfCol[n & 7])) & 0xF;
  }
  const uint32_t w = (uint32_t)qweight[(k / 8) * N + n];
  return (w >> (4 * (k & 7))) & 0xF;
}
__device__ __forceinline__ int src_zero(const int32_t* qzeros, int layout, int64_t N, int64_t g, int64_t n) {
  const uint32_t w = (uint32_t)qzeros[g * (N / 8) + n / 8];
  if (layout == MI_W4_AWQ) return (w >> (4 * kAwqNibbleOfCol[n & 7])) & 0xF;
  return ((w >> (4 * (n & 7))) & 0xF) + 1;  // AutoGPTQ v1: stored minus one
}

// int -> float magic offsets: 0x6400|q = float16(1024+q), bfloat16 128+q is exact for q < 128
template <typename T> struct Magic;
template <> struct Magic<f16_t> { static constexpr float value = 1024.f; };
template <> struct Magic<bf16_t> { static constexpr float value = 128.f; };

// --------------------------------------------------------------------------- repack
__global__ __launch_bounds__(256) void w4_repack_kernel(const int32_t* __restrict__ qweight,
                                                        const int32_t* __restrict__ perm, uint32_t* __restrict__ qw,
                                                        int64_t N, int64_t K, int layout) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per output dword
  const int64_t total = (N / 16) * (K / 128) * 256;
  if (gid >= total) return;
  const int s = gid & 3, lane = (gid >> 2) & 63;
  const int64_t blk = gid >> 8;
  const int64_t kb = blk % (K / 128), nt = blk / (K / 128);
  const int64_t n = nt * 16 + (lane & 15);
  const int64_t k0 = kb * 128 + 32 * s + 8 * (lane >> 4);
  uint32_t w = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t ks = perm ? perm[k0 + j] : k0 + j;
    const uint32_t v = src_weight(qweight, layout, N, ks, n);
    const int nib = (j >> 1) + 4 * (j & 1);
    w |= v << (4 * nib);
  }
  qw[gid] = w;
}

template <typename T>
__global__ __launch_bounds__(256) void w4_zs_kernel(const int32_t* __restrict__ qzeros, const T* __restrict__ scales,
                                                    uint32_t* __restrict__ zs, int64_t N, int64_t G, int layout) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= G * N) return;
  const int64_t g = gid / N, n = gid % N;
  const int z = src_zero(qzeros, layout, N, g, n);
  const uint16_t sb = __builtin_bit_cast(uint16_t, scales[gid]);
  zs[gid] = (uint32_t)sb | ((uint32_t)to_bits<T>(Magic<T>::value + (float)z) << 16);
}

extern "C" int mi_w4_repack(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                            const int32_t* perm, void* qw_native, void* zs_native, int64_t N, int64_t K,
                            int64_t group_size, int layout, int dtype, void* stream) {
  MI_CHECK_ARG(qweight && qzeros && scales && qw_native && zs_native);
  MI_CHECK_ARG(layout == MI_W4_AWQ || layout == MI_W4_GPTQ);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (group_size <= 0) group_size = K;
  if (N % 16 != 0 || K % 128 != 0 || group_size % 32 != 0 || K % group_size != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4_repack: need N%%16==0, K%%128==0, group%%32==0 (N=%lld K=%lld g=%lld)",
            (long long)N, (long long)K, (long long)group_size);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (N / 16) * (K / 128) * 256;
  w4_repack_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(qweight, perm, (uint32_t*)qw_native, N, K, layout);
  MI_CHECK_LAUNCH();
  const int64_t G = K / group_size;
  if (dtype == MI_FP16)
    w4_zs_kernel<f16_t><<<(unsigned)cdiv64(G * N, 256), 256, 0, st>>>(qzeros, (const f16_t*)scales, (uint32_t*)zs_native, N, G, layout);
  else
    w4_zs_kernel<bf16_t><<<(unsigned)cdiv64(G * N, 256), 256, 0, st>>>(qzeros, (const bf16_t*)scales, (uint32_t*)zs_native, N, G, layout);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// -------------------------------------------------------------------- dequant helpers
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;

// one packed dword (8 weights) -> MFMA fragment of 8 T values, exact (w - z) * s
template <typename T> struct Deq;
template <> struct Deq<f16_t> {
  static __device__ __forceinline__ f16x8 run(uint32_t w, uint32_t zs) {
    const uint32_t s2 = (zs & 0xffffu) * 0x00010001u;
    const uint32_t z2 = (zs >> 16) * 0x00010001u;
    const half2_t sv = __builtin_bit_cast(half2_t, s2), zv = __builtin_bit_cast(half2_t, z2);
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t pair = ((w >> (4 * i)) & 0x000F000Fu) | 0x64006400u;  // (1024+e, 1024+e')
      half2_t h = __builtin_bit_cast(half2_t, pair);
      h = (h - zv) * sv;  // exact difference, one rounding in the product
      o[i] = __builtin_bit_cast(uint32_t, h);
    }
    return __builtin_bit_cast(f16x8, u32x4{o[0], o[1], o[2], o[3]});
  }
};
template <> struct Deq<bf16_t> {
  static __device__ __forceinline__ bf16x8 run(uint32_t w, uint32_t zs) {
    const float s = __uint_as_float(zs << 16);
    const float z = __uint_as_float(zs & 0xffff0000u) - 128.f;  // exact
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = (float)((w >> (4 * i)) & 0xFu);
      const float b = (float)((w >> (4 * i + 16)) & 0xFu);
      o[i] = pack2<bf16_t>((a - z) * s, (b - z) * s);
    }
    return __builtin_bit_cast(bf16x8, u32x4{o[0], o[1], o[2], o[3]});
  }
};

// ------------------------------------------------------------------------------ GEMM
struct W4Params {
  const void* x;
  const uint32_t* qw;
  const uint32_t* zs;
  const int32_t* perm;
  const void* bias;
  void* out;
  int64_t M, N, K, group, ldx, ldo;
};

template <typename T, int MT, bool PERM>
__global__ __launch_bounds__(256) void w4a16_gemm_kernel(const W4Params p) {
  typedef typename Elem<T>::vec8 vec8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t nt = (int64_t)blockIdx.x * 4 + wave;
  const int64_t n0 = nt * 16;
  const int64_t m0 = (int64_t)blockIdx.y * (MT * 16);
  if (n0 >= p.N) return;
  const int64_t KB = p.K / 128;
  const uint4* wq = (const uint4*)p.qw + nt * KB * 64 + lane;
  const uint32_t* zsp = p.zs + n0 + r16;
  const T* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = (const T*)p.x + min(m0 + t * 16 + r16, p.M - 1) * p.ldx + 8 * q;

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t kb = 0; kb < KB; ++kb) {
    const uint4 wv = wq[kb * 64];
    const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t k0 = kb * 128 + 32 * s;
      const uint32_t zsv = zsp[(k0 / p.group) * p.N];
      const vec8 wf = Deq<T>::run(ww[s], zsv);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        vec8 xf;
        if constexpr (PERM) {
          T tmp[8];
          const T* xr = xp[t] - 8 * q;
#pragma unroll
          for (int j = 0; j < 8; ++j) tmp[j] = xr[p.perm[k0 + 8 * q + j]];
          xf = *(vec8*)tmp;
        } else {
          xf = __builtin_bit_cast(vec8, *(const uint4*)(xp[t] + k0));
        }
        acc[t] = Elem<T>::mfma16(wf, xf, acc[t]);
      }
    }
  }

  const int64_t nb = n0 + 4 * q;
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = p.bias ? (float)((const T*)p.bias)[nb + r] : 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int64_t m = m0 + t * 16 + r16;
    if (m >= p.M) continue;
    T* o = (T*)p.out + m * p.ldo + nb;
    *(uint2*)o = make_uint2(pack2<T>(acc[t][0] + bv[0], acc[t][1] + bv[1]),
                            pack2<T>(acc[t][2] + bv[2], acc[t][3] + bv[3]));
  }
}

template <typename T, bool PERM> static void launch_w4(const W4Params& p, hipStream_t st) {
  const unsigned gx = (unsigned)cdiv64(p.N, 64);
  if (p.M <= 16) w4a16_gemm_kernel<T, 1, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 16)), 256, 0, st>>>(p);
  else if (p.M <= 32) w4a16_gemm_kernel<T, 2, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 32)), 256, 0, st>>>(p);
  else w4a16_gemm_kernel<T, 4, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 64)), 256, 0, st>>>(p);
}

extern "C" int mi_w4a16_gemm(const void* x, const void* qw_native, const void* zs_native, const int32_t* perm,
                             const void* bias, void* out, int64_t M, int64_t N, int64_t K, int64_t group_size,
                             int64_t ldx, int64_t ldo, int dtype, void* stream) {
  MI_CHECK_ARG(M >= 0 && N > 0 && K > 0);
  if (M == 0) return MI_OK;
  MI_CHECK_ARG(x && qw_native && zs_native && out);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (group_size <= 0) group_size = K;
  if (N % 16 != 0 || K % 128 != 0 || group_size % 32 != 0 || ldx % 8 != 0 || ldo % 4 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4a16_gemm: need N%%16==0, K%%128==0, group%%32==0, ldx%%8==0, ldo%%4==0");
  MI_CHECK_ARG((((uintptr_t)x | (uintptr_t)qw_native) & 15) == 0 && ((uintptr_t)out & 7) == 0);
  W4Params p;
  p.x = x; p.qw = (const uint32_t*)qw_native; p.zs = (const uint32_t*)zs_native; p.perm = perm;
  p.bias = bias; p.out = out; p.M = M; p.N = N; p.K = K; p.group = group_size; p.ldx = ldx; p.ldo = ldo;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_FP16) {
    if (perm) launch_w4<f16_t, true>(p, st); else launch_w4<f16_t, false>(p, st);
  } else {
    if (perm) launch_w4<bf16_t, true>(p, st); else launch_w4<bf16_t, false>(p, st);
  }
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------------- unfused dequant (checkpoint layout)
template <typename T>
__global__ __launch_bounds__(256) void w4_dequant_kernel(const int32_t* __restrict__ qweight,
                                                         const int32_t* __restrict__ qzeros,
                                                         const T* __restrict__ scales, const int32_t* __restrict__ g_idx,
                                                         T* __restrict__ w_out, int64_t N, int64_t K, int64_t group,
                                                         int layout) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= N * K) return;
  const int64_t k = gid / N, n = gid % N;
  const int64_t g = g_idx ? g_idx[k] : k / group;
  const int wv = src_weight(qweight, layout, N, k, n);
  const int z = src_zero(qzeros, layout, N, g, n);
  w_out[gid] = (T)((float)(wv - z) * (float)scales[g * N + n]);
}

extern "C" int mi_w4_dequantize(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                                const int32_t* g_idx, void* w_out, int64_t N, int64_t K, int64_t group_size,
                                int layout, int dtype, void* stream) {
  MI_CHECK_ARG(qweight && qzeros && scales && w_out && N > 0 && K > 0);
  MI_CHECK_ARG(layout == MI_W4_AWQ || layout == MI_W4_GPTQ);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (group_size <= 0) group_size = K;
  if (N % 8 != 0 || (layout == MI_W4_GPTQ && K % 8 != 0))
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4_dequantize: N (and K for GPTQ) must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)cdiv64(N * K, 256);
  if (dtype == MI_FP16)
    w4_dequant_kernel<f16_t><<<blocks, 256, 0, st>>>(qweight, qzeros, (const f16_t*)scales, g_idx, (f16_t*)w_out, N, K, group_size, layout);
  else
    w4_dequant_kernel<bf16_t><<<blocks, 256, 0, st>>>(qweight, qzeros, (const bf16_t*)scales, g_idx, (bf16_t*)w_out, N, K, group_size, layout);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
