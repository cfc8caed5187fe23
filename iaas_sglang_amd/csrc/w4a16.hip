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
template <int I> __device__ __forceinline__ float ub(uint32_t v) { return (float)((v >> (8 * I)) & 0xffu); }   // v_cvt_f32_ubyteI
template <> struct Deq<bf16_t> {
  // (e - z) * s with e, z small integers and s a bf16: e*s and z*s are exact in fp32 (4 + 8 significant bits) and
  // so is their difference, so ONE fma per element gives the exact product and v_cvt_pk_bf16_f32 rounds it once --
  // the bits of round_bf16((float)(e - z) * s).  Bytes instead of nibbles feed v_cvt_f32_ubyteN directly:
  // w & 0x0f0f0f0f holds elements (0, 4, 1, 5), (w >> 4) & 0x0f0f0f0f elements (2, 6, 3, 7) in bytes 0..3.
  static __device__ __forceinline__ bf16x8 run(uint32_t w, uint32_t zs) {
    const float s = __uint_as_float(zs << 16);
    const float nzs = -((__uint_as_float(zs & 0xffff0000u) - 128.f) * s);   // -(z * s), exact
    const uint32_t a = w & 0x0f0f0f0fu, b = (w >> 4) & 0x0f0f0f0fu;
    uint32_t o[4];
    o[0] = pack2<bf16_t>(fmaf(ub<0>(a), s, nzs), fmaf(ub<2>(a), s, nzs));
    o[1] = pack2<bf16_t>(fmaf(ub<0>(b), s, nzs), fmaf(ub<2>(b), s, nzs));
    o[2] = pack2<bf16_t>(fmaf(ub<1>(a), s, nzs), fmaf(ub<3>(a), s, nzs));
    o[3] = pack2<bf16_t>(fmaf(ub<1>(b), s, nzs), fmaf(ub<3>(b), s, nzs));
    return __builtin_bit_cast(bf16x8, u32x4{o[0], o[1], o[2], o[3]});
  }
};

// The same dequantisation in four parts (one packed pair of the fragment each), so that a kernel can place them between
// its MFMAs: DeqQ<T>::Prep holds what a (tile, phase) or a packed dword shares, part<I>() returns dword I of the
// fragment Deq<T>::run would give -- the same instructions on the same values, bit for bit.
template <typename T> struct DeqQ;
template <> struct DeqQ<f16_t> {
  struct Zs { half2_t sv, zv; };
  static __device__ __forceinline__ Zs zs(uint32_t v) {
    return Zs{__builtin_bit_cast(half2_t, (v & 0xffffu) * 0x00010001u), __builtin_bit_cast(half2_t, (v >> 16) * 0x00010001u)};
  }
  template <int I> static __device__ __forceinline__ uint32_t part(uint32_t w, const Zs& z) {
    // (w >> 4 I) & 0x000F000F | 0x64006400 in ONE v_bfi_b32 (hipcc emits v_and + v_or for the C form)
    uint32_t pair;
    const uint32_t sh = I ? (w >> (4 * I)) : w;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(pair) : "s"(0x000F000Fu), "v"(sh), "v"(0x64006400u));   // one SGPR per VOP3 (constant bus)
    half2_t h = __builtin_bit_cast(half2_t, pair);
    h = (h - z.zv) * z.sv;
    return __builtin_bit_cast(uint32_t, h);
  }
};
template <> struct DeqQ<bf16_t> {
  struct Zs { float s, nzs; };
  static __device__ __forceinline__ Zs zs(uint32_t v) {
    const float s = __uint_as_float(v << 16);
    return Zs{s, -((__uint_as_float(v & 0xffff0000u) - 128.f) * s)};
  }
  template <int I> static __device__ __forceinline__ uint32_t part(uint32_t w, const Zs& z) {
    const uint32_t h = ((I & 1) ? (w >> 4) : w) & 0x0f0f0f0fu;       // I = 0, 2: elements (0,4,1,5); 1, 3: (2,6,3,7)
    if constexpr (I < 2) return pack2<bf16_t>(fmaf(ub<0>(h), z.s, z.nzs), fmaf(ub<2>(h), z.s, z.nzs));
    else return pack2<bf16_t>(fmaf(ub<1>(h), z.s, z.nzs), fmaf(ub<3>(h), z.s, z.nzs));
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


// ---------------------------------------------------------------------------------------------
// Decode-shaped fused dequant GEMM (M <= 128, no act-order): same structure as fp8_gemm_xs_kernel.
//   workgroup (nb, sp) = NWV waves = NWV 16-column weight tiles x a range of 128-element k-phases;
//   per phase the activation block [M x 256 B] (fp16/bf16) and each wave's packed int4 block (one
//   1-KiB native tile = 16 n x 128 k) go global -> LDS by inline-asm LDS-DMA into a ring of R stages, R - 1 phases
//   ahead, with counted vmcnt waits (every wave issues the same number of pieces per stage); with two stages and a
//   full drain per phase every phase exposed most of a DMA latency (16 phases x ~1.2 us = the 22 us of the qkv
//   shape at M = 64); the (scale, zero) words of the workgroup's whole k-range are staged
//   once; x rows are unpadded with the 16-byte slot XOR-swizzled by (row & 15) on the source side.
//   Per phase and wave: 1 ds_read_b128 of packed weights -> 4 dequantised MFMA fragments (exact
//   (w - z) * s), 4 x MT MFMA 16x16x32.  Split-K partials go to fp32 slabs (w4_reduce_kernel).
// Bound: HBM (N*K/2 bytes of weights read once) + per-CU ingest of the activation block.
// ring depth of w4a16_xs_kernel: 4 stages (3 phases ahead) while R * stage + 16 KiB of zs fits the 160 KiB of LDS
static constexpr int w4_xs_ring(int mt) { return mt >= 8 ? 3 : 4; }

template <typename T, int MT, int NWV>
__global__ __launch_bounds__(NWV * 64) void w4a16_xs_kernel(const W4Params p, float* __restrict__ slab, int S,
                                                            int phases_per_wg) {
  typedef typename Elem<T>::vec8 vec8;
  constexpr int PW = 256;                            // bytes of x per row per phase (128 elements)
  constexpr int ROWS = MT * 16;
  constexpr int XBYTES = ROWS * PW;
  constexpr int STAGE = XBYTES + NWV * 1024;         // + one packed weight tile per wave
  constexpr int MAXPH = 32;                          // zs staging capacity (phases per workgroup)
  constexpr int R = w4_xs_ring(MT);                  // ring stages (LDS: R * STAGE + 16 KiB of zs)
  constexpr int LA = R - 1;                          // phases requested ahead
  constexpr int XP = (ROWS / 4 + NWV - 1) / NWV;     // x pieces per wave and stage (clamped duplicates when ROWS / 4 < NWV)
  constexpr int OPS = 1 + XP;                        // vmcnt entries per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* zs_lds = (uint32_t*)(smem + R * STAGE);  // [NWV][MAXPH][16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t nt = (int64_t)blockIdx.x * NWV + wave;
  const int64_t n0 = nt * 16;
  const int sp = blockIdx.y;
  const int64_t KB = p.K / 128;
  const int64_t ph0 = (int64_t)sp * phases_per_wg;
  const int64_t ph1 = min(KB, ph0 + phases_per_wg);
  const bool tile_ok = n0 < p.N;
  const uint32_t lds_base = lds_addr_of(smem);
  const int drow = lane >> 4, dslot = lane & 15;
  const uint4* wq = (const uint4*)p.qw + min(nt, p.N / 16 - 1) * KB * 64 + lane;

  // (scale, zero) words of this wave's 16 columns for every phase of the workgroup: staged once
  if (tile_ok) {
    for (int i = lane; i < (int)(ph1 - ph0) * 16; i += 64) {
      const int64_t ph = ph0 + i / 16;
      zs_lds[(wave * MAXPH + i / 16) * 16 + (i & 15)] = p.zs[((ph * 128) / p.group) * p.N + n0 + (i & 15)];
    }
  }

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

#define W4_STAGE(ph_, st_)                                                                             \
  {                                                                                                    \
    glds16(wq + (int64_t)(ph_) * 64, lds_base + (st_) * STAGE + XBYTES + wave * 1024);                 \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                   \
      /* a wave past the last piece re-requests the last one (same bytes, same place): every wave has OPS entries */ \
      const int rr_ = min((i * NWV + wave) * 4, ROWS - 4);                                             \
      const int row_ = rr_ + drow;                                                                     \
      const int ss_ = dslot ^ (row_ & 15);                                                             \
      glds16((const T*)p.x + min((int64_t)row_, p.M - 1) * p.ldx + (int64_t)(ph_) * 128 + ss_ * 8,     \
             lds_base + (st_) * STAGE + rr_ * PW);                                                     \
    }                                                                                                  \
  }
#define W4_MMA(ph_, st_)                                                                               \
  if (tile_ok) {                                                                                       \
    const uint4 wv_ = *(const uint4*)(smem + (st_) * STAGE + XBYTES + wave * 1024 + lane * 16);        \
    const uint32_t ww_[4] = {wv_.x, wv_.y, wv_.z, wv_.w};                                              \
    const uint32_t zsv_ = zs_lds[(wave * MAXPH + (int)((ph_) - ph0)) * 16 + r16];                      \
    const char* xb_ = smem + (st_) * STAGE + r16 * PW;                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
      const vec8 wf_ = Deq<T>::run(ww_[s], zsv_);                                                      \
      const int o_ = ((s * 4 + q) ^ r16) * 16;                                                         \
      _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                                 \
        const vec8 xf_ = __builtin_bit_cast(vec8, *(const uint4*)(xb_ + t * 16 * PW + o_));            \
        acc[t] = Elem<T>::mfma16(wf_, xf_, acc[t]);                                                    \
      }                                                                                                \
    }                                                                                                  \
  }
  // issue order: stages 0 .. LA-1 | iteration i: [wait stage i] barrier, request stage i + LA (into the slot stage
  // i - 1 used: every wave is past it), compute stage i.  Younger than stage i at its wait: min(LA - 1, n - 1 - i) stages.
  if (ph0 < ph1) {
    const int n = (int)(ph1 - ph0);
#pragma unroll
    for (int a = 0; a < LA; ++a) {
      const int64_t ph_a = ph0 + a;
      if (a < n) W4_STAGE(ph_a, a);
    }
    for (int it = 0; it < n; ++it) {       // (not `i`: the macros have loops of their own over that name)
      const int y = min(LA - 1, n - 1 - it);
      if (y >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
      else if (y == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int64_t ph_req = ph0 + it + LA, ph_now = ph0 + it;
      const int st_req = (it + LA) % R, st_now = it % R;
      if (it + LA < n) W4_STAGE(ph_req, st_req);
      W4_MMA(ph_now, st_now);
    }
  }
#undef W4_STAGE
#undef W4_MMA
  if (!tile_ok) return;

  const int64_t nb = n0 + 4 * q;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int64_t m = (int64_t)t * 16 + r16;
    if (m >= p.M) continue;
    if (S > 1) {
      *(f32x4*)(slab + ((int64_t)sp * p.M + m) * p.N + nb) = acc[t];
    } else {
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = p.bias ? (float)((const T*)p.bias)[nb + r] : 0.f;
      *(uint2*)((T*)p.out + m * p.ldo + nb) = make_uint2(pack2<T>(acc[t][0] + bv[0], acc[t][1] + bv[1]),
                                                        pack2<T>(acc[t][2] + bv[2], acc[t][3] + bv[3]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// w4a16_xw_kernel: the decode-shaped int4 GEMM with wave ROLES (round 3; the structure of fp8_gemm_xw_kernel).
// What the stamps of the FP8 kernel showed applies here more strongly: in w4a16_xs_kernel every one of 8 waves issues
// DMA, reads the WHOLE activation stage (8 x 16 KiB at M = 64 for 8 KiB of packed weights), dequantises and runs its
// MFMAs in lock-step (measured: 0.14 of the HBM rate at the C4 shapes, SQ_WAIT_ANY 0.44-0.49, MFMA 0.08-0.12 busy).
//   * waves 8-11 are LOADERS (LDS-DMA only: activation pieces of 4 rows x 256 B, the 1-KiB native weight tiles, one
//     1-KiB row of (scale, zero) words per 128-k phase), counted vmcnt, one barrier per phase, phases walked cyclically
//     from (7 b) % count in workgroup b;
//   * waves 0-7 are CONSUMERS, two per SIMD (with one per SIMD the dequant VALU of a wave and its own MFMAs do not
//     overlap: 30.5 us for gate_up against 26.1 with two), each owning NT/2 16-column tiles x all M rows of the
//     workgroup's 64 NT columns: an activation fragment feeds NT/2 MFMAs, and the block shares one pass over the
//     activations -- at fp16/bf16 activations against int4 weights the ACTIVATION block is most of what a CU ingests
//     ((M x 256 B) per phase against NT x 4 KiB of weights), so wide column blocks + split-K beat 128-column blocks
//     (w4_xw_plan);
//   * the dequantisation of step s + 1 is placed by hand between the MFMAs of step s (one quarter of a fragment per
//     MFMA, DeqQ<T>): left to the scheduler each step's dequant sits in one block in front of its MFMAs;
//   * the epilogue goes through LDS: whole-row stores of the fp32 slab tile or of the T-typed output by all 12 waves.
// Dequant, MFMA operands and the per-element accumulation order over a workgroup's phases are those of
// w4a16_xs_kernel up to the rotation of the phase walk (fp32 sums of the same terms in another order).
#ifdef MI_TUNING
// diagnostic build only (MI_W4_STAMPS=1): per (workgroup, wave) cycle sums of w4a16_xw_kernel's phase segments
__device__ unsigned long long mi_w4_stamps[512 * 12 * 8];
extern "C" int mi_debug_w4_stamps(unsigned long long* host_out, int clear) {
  if (clear) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(mi_w4_stamps)) != hipSuccess) return -1;
    return hipMemset(d, 0, sizeof(mi_w4_stamps)) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi_w4_stamps), sizeof(mi_w4_stamps)) == hipSuccess ? 0 : -1;
}
static int w4_stamps_on() {
  static const int v = mi_tune("MI_W4_STAMPS", 0);
  return v;
}
#endif
template <int MT, int NT> struct W4XW {
  static constexpr int ROWS = MT * 16, WR = 64 * NT;
  static constexpr int XBYTES = ROWS * 256, WBYTES = 4 * NT * 1024, ZBYTES = 1024;
  static constexpr int STAGE = XBYTES + WBYTES + ZBYTES;
  static constexpr int TILE = ROWS * (WR * 4 + 16);                      // staged fp32 output tile
  static constexpr int R = (160 * 1024 / STAGE) >= 4 ? 4 : (160 * 1024 / STAGE);   // ring stages
  static constexpr int LDS = R * STAGE > TILE ? R * STAGE : TILE;
  static_assert(R >= 3 && LDS <= 160 * 1024, "ring of at least 3 stages within the CU's LDS");
};

template <typename T, int MT, int NT>
__global__ __launch_bounds__(768) void w4a16_xw_kernel(const W4Params p, float* __restrict__ slab, int S,
                                                       int phases_per_wg, int dbg /* bit 0: stamps, bit 1: slab even when S == 1 */) {
  typedef typename Elem<T>::vec8 vec8;
  typedef W4XW<MT, NT> C;
  constexpr int ROWS = C::ROWS, WR = C::WR, XBYTES = C::XBYTES, WBYTES = C::WBYTES, STAGE = C::STAGE, R = C::R;
  constexpr int D = R - 1, NL = 4, NCW = 8;               // 8 consumer waves (two per SIMD) + 4 loader waves
  constexpr int TPC = 4 * NT >= NCW ? 4 * NT / NCW : 1;   // 16-column tiles per consumer (NT = 1: consumers 4-7 idle)
  constexpr int NWAVES = NCW + NL;
  constexpr int XPIECES = ROWS / 4;                       // activation DMA pieces (4 rows x 256 B) per phase
  constexpr int XL = XPIECES >= NL ? XPIECES / NL : 1;    // per loader (small M: loaders re-fetch a piece)
  constexpr int WL = NT;                                  // weight tiles per loader (4 NT per workgroup)
  constexpr int E = XL + WL + 1;                          // vmcnt entries per loader per phase (+ the zs row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int sp = blockIdx.y;
  const int64_t KB = p.K / 128;
  const int64_t ph0 = (int64_t)sp * phases_per_wg;
  const int64_t ph1 = min(KB, ph0 + phases_per_wg);
  const int64_t col0 = (int64_t)blockIdx.x * WR;          // first output column of the workgroup
  const uint32_t lds_base = lds_addr_of(smem);

  f32x4 acc[TPC][MT];
#pragma unroll
  for (int tn = 0; tn < TPC; ++tn)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[tn][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool has_tiles = wave * TPC < 4 * NT;             // (consumer waves only)

  if (wave >= NCW) {
    // ---------------------------------------------------------------- loaders
    const int ld = wave - NCW;
    const int drow = lane >> 4, dslot = lane & 15;
    const int64_t ntiles = p.N / 16;
    const int64_t nphw = ph1 - ph0;
    const int64_t rot = nphw > 0 ? ((int64_t)blockIdx.x * 7) % nphw : 0;
    const uint4* wsrc[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int64_t nt = min((int64_t)blockIdx.x * (4 * NT) + ld * WL + i, ntiles - 1);
      wsrc[i] = (const uint4*)p.qw + nt * KB * 64 + lane;
    }
    const T* xsrc[XL];
    uint32_t xdst[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int rr = ((ld * XL + i) % XPIECES) * 4;
      const int row = rr + drow;
      xsrc[i] = (const T*)p.x + min((int64_t)row, p.M - 1) * p.ldx + (dslot ^ (row & 15)) * 8;
      xdst[i] = lds_base + rr * 256;
    }
    const int64_t zcol = min(col0 + lane * 4, p.N - 4);   // (scale, zero) words of 4 columns per lane
    auto issue = [&](int64_t ph) __attribute__((always_inline)) {
      const uint32_t st = (uint32_t)((ph - ph0) % R) * STAGE;
      int64_t kph = ph + rot;
      kph = kph >= ph1 ? kph - nphw : kph;
#pragma unroll
      for (int i = 0; i < XL; ++i) glds16(xsrc[i] + kph * 128, xdst[i] + st);
#pragma unroll
      for (int i = 0; i < WL; ++i) glds16(wsrc[i] + kph * 64, lds_base + st + XBYTES + (ld * WL + i) * 1024);
      glds16(p.zs + ((kph * 128) / p.group) * p.N + zcol, lds_base + st + XBYTES + WBYTES);
    };
    static_assert(D >= 2 && D <= 3, "the wait enumerates up to 2 phases in flight behind the awaited one");
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (ph0 + d < ph1) issue(ph0 + d);
#ifdef MI_TUNING
    unsigned long long t_iss = 0, t_wait = 0, t_bar = 0;
#endif
    for (int64_t ph = ph0; ph < ph1; ++ph) {
#ifdef MI_TUNING
      const unsigned long long ta = (dbg & 1) ? __builtin_amdgcn_s_memtime() : 0;
#endif
      const int64_t after = min((int64_t)(D - 1), ph1 - 1 - ph);   // phases issued after ph may stay in flight
      if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * E) : "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MI_TUNING
      const unsigned long long tb = (dbg & 1) ? __builtin_amdgcn_s_memtime() : 0;
#endif
      __builtin_amdgcn_s_barrier();                      // phase ph published; the consumers are done with ph-1
#ifdef MI_TUNING
      const unsigned long long tc = (dbg & 1) ? __builtin_amdgcn_s_memtime() : 0;
#endif
      if (ph + D < ph1) issue(ph + D);                   // into the stage phase ph-1 used
#ifdef MI_TUNING
      if (dbg & 1) { const unsigned long long td = __builtin_amdgcn_s_memtime(); t_wait += tb - ta; t_bar += tc - tb; t_iss += td - tc; }
#endif
    }
#ifdef MI_TUNING
    if ((dbg & 1) && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
      unsigned long long* o = mi_w4_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      o[0] += t_iss; o[1] += t_wait; o[2] += t_bar; o[3] += (unsigned long long)(ph1 - ph0);
    }
#endif
    __builtin_amdgcn_s_barrier();                        // (the consumers' "LDS may be reused" barrier)
  } else {
    // ---------------------------------------------------------------- consumers
#ifdef MI_TUNING
    unsigned long long t_cmp = 0, t_bar = 0, t_prev = 0;
    const unsigned long long t_k0 = (dbg & 1) ? __builtin_amdgcn_s_memtime() : 0, r_k0 = (dbg & 1) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
    for (int64_t ph = ph0; ph < ph1; ++ph) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef MI_TUNING
      const unsigned long long ta = (dbg & 1) ? __builtin_amdgcn_s_memtime() : 0;
#endif
      __builtin_amdgcn_s_barrier();                      // phase ph landed (the loaders waited for it)
#ifdef MI_TUNING
      if (dbg & 1) { const unsigned long long tb = __builtin_amdgcn_s_memtime(); if (t_prev) t_cmp += ta - t_prev; t_bar += tb - ta; t_prev = tb; }
#endif
      const char* xb = smem + ((ph - ph0) % R) * STAGE;
      const char* wb = xb + XBYTES;
      if (!has_tiles) continue;
      uint32_t ww[TPC][4], zsv[TPC];
#pragma unroll
      for (int tn = 0; tn < TPC; ++tn) {
        const int tile = wave * TPC + tn;
        const uint4 wv = *(const uint4*)(wb + tile * 1024 + lane * 16);
        ww[tn][0] = wv.x; ww[tn][1] = wv.y; ww[tn][2] = wv.z; ww[tn][3] = wv.w;
        zsv[tn] = *(const uint32_t*)(wb + WBYTES + (tile * 16 + r16) * 4);
      }
      // Software pipeline over the four 32-k steps of the phase, in SOURCE order behind scheduling fences: after each
      // MFMA of step s comes one quarter of a weight fragment of step s + 1 (4-5 VALU), the activation fragments of
      // step s + 1 are requested in front of the MFMAs of step s.  A wave issues in order: left to the scheduler, each
      // step's dequant (16-24 VALU per tile) sits in one block in front of its MFMAs and the matrix pipe idles
      // meanwhile (stamps: 1600-2300 cycles per phase for 512 cycles of MFMA per wave).
      typename DeqQ<T>::Zs zq[TPC];
#pragma unroll
      for (int tn = 0; tn < TPC; ++tn) zq[tn] = DeqQ<T>::zs(zsv[tn]);
      uint4 xr[2][MT];
      uint32_t wfr[2][TPC][4];
#pragma unroll
      for (int t = 0; t < MT; ++t) xr[0][t] = *(const uint4*)(xb + (t * 16 + r16) * 256 + ((q ^ r16) * 16));
#pragma unroll
      for (int tn = 0; tn < TPC; ++tn) {
        wfr[0][tn][0] = DeqQ<T>::template part<0>(ww[tn][0], zq[tn]);
        wfr[0][tn][1] = DeqQ<T>::template part<1>(ww[tn][0], zq[tn]);
        wfr[0][tn][2] = DeqQ<T>::template part<2>(ww[tn][0], zq[tn]);
        wfr[0][tn][3] = DeqQ<T>::template part<3>(ww[tn][0], zq[tn]);
      }
      constexpr int NM = TPC * MT, NP = TPC * 4;                          // MFMAs / fragment quarters per step
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        __builtin_amdgcn_sched_barrier(0);
        if (s < 3) {
          const int o = (((s + 1) * 4 + q) ^ r16) * 16;
#pragma unroll
          for (int t = 0; t < MT; ++t) xr[nxt][t] = *(const uint4*)(xb + (t * 16 + r16) * 256 + o);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NM; ++j) {
          const int t = j / TPC, tn = j % TPC;
          const vec8 wfv = __builtin_bit_cast(vec8, u32x4{wfr[cur][tn][0], wfr[cur][tn][1], wfr[cur][tn][2], wfr[cur][tn][3]});
          acc[tn][t] = Elem<T>::mfma16(wfv, __builtin_bit_cast(vec8, xr[cur][t]), acc[tn][t]);
          __builtin_amdgcn_sched_barrier(0);
          if (s < 3) {
#pragma unroll
            for (int pi = j * NP / NM; pi < (j + 1) * NP / NM; ++pi) {   // this MFMA's share of the next step's quarters
              const int ptn = pi / 4;
              const uint32_t wn = ww[ptn][s + 1];
              switch (pi & 3) {
                case 0: wfr[nxt][ptn][0] = DeqQ<T>::template part<0>(wn, zq[ptn]); break;
                case 1: wfr[nxt][ptn][1] = DeqQ<T>::template part<1>(wn, zq[ptn]); break;
                case 2: wfr[nxt][ptn][2] = DeqQ<T>::template part<2>(wn, zq[ptn]); break;
                default: wfr[nxt][ptn][3] = DeqQ<T>::template part<3>(wn, zq[ptn]); break;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
#ifdef MI_TUNING
    if ((dbg & 1) && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
      unsigned long long* o = mi_w4_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      o[0] += t_cmp; o[2] += t_bar; o[3] += (unsigned long long)(ph1 - ph0);
      o[4] += __builtin_amdgcn_s_memtime() - t_k0; o[5] += __builtin_amdgcn_s_memrealtime() - r_k0;
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every stage consumed: the epilogue may reuse LDS
  }

  // ------------------------------------------------------------------ epilogue through LDS: [ROWS][WR] fp32, rows
  // padded to 4 WR + 16 bytes (conflict-free ds_write_b128), then whole-row stores by all 8 waves
  constexpr int TP = WR * 4 + 16;
  const bool to_slab = S > 1 || (dbg & 2);
  if (wave < NCW && has_tiles) {
#pragma unroll
    for (int tn = 0; tn < TPC; ++tn) {
      const int cb = (wave * TPC + tn) * 16 + 4 * q;       // block column of the lane's 4 values
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (!to_slab && p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (float)((const T*)p.bias)[min(col0 + cb + r, p.N - 1)];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        f32x4 v = acc[tn][t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
        *(f32x4*)(smem + (t * 16 + r16) * TP + cb * 4) = v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (to_slab) {
    float* sbp = slab + (int64_t)sp * p.M * p.N + col0;
    constexpr int LPR = WR / 4, RPW = 64 / LPR;           // lanes per row (16 B each), rows per wave instruction
#pragma unroll
    for (int i = 0; i < (ROWS + NWAVES * RPW - 1) / (NWAVES * RPW); ++i) {
      const int row = (i * NWAVES + wave) * RPW + lane / LPR, c = lane % LPR;
      if (row < ROWS && row < p.M && col0 + c * 4 < p.N)
        *(f32x4*)(sbp + (int64_t)row * p.N + c * 4) = *(const f32x4*)(smem + row * TP + c * 16);
    }
  } else {
    T* op = (T*)p.out + col0;
    constexpr int LPR = WR / 8, RPW = 64 / LPR;           // 8 columns per lane
#pragma unroll
    for (int i = 0; i < (ROWS + NWAVES * RPW - 1) / (NWAVES * RPW); ++i) {
      const int row = (i * NWAVES + wave) * RPW + lane / LPR, c = lane % LPR;
      if (row < ROWS && row < p.M && col0 + c * 8 < p.N) {
        const f32x4 a = *(const f32x4*)(smem + row * TP + c * 32), b = *(const f32x4*)(smem + row * TP + c * 32 + 16);
        *(uint4*)(op + (int64_t)row * p.ldo + c * 8) =
            make_uint4(pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3]), pack2<T>(b[0], b[1]), pack2<T>(b[2], b[3]));
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void w4_reduce_kernel(const W4Params p, const float* __restrict__ slab, int S) {
  const int64_t nq = p.N / 4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= p.M * nq) return;
  const int64_t m = gid / nq, nb = (gid % nq) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) v += *(const f32x4*)(slab + ((int64_t)s * p.M + m) * p.N + nb);
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = p.bias ? (float)((const T*)p.bias)[nb + r] : 0.f;
  *(uint2*)((T*)p.out + m * p.ldo + nb) = make_uint2(pack2<T>(v[0] + bv[0], v[1] + bv[1]), pack2<T>(v[2] + bv[2], v[3] + bv[3]));
}

// ---------------------------------------------------------------------------------------------
// Prefill-shaped fused dequant GEMM (M > 512): 256 (m) x 256 (n) output tile per 8-wave workgroup, the dequantisation
// fused into the MFMA loop -- replaces "materialise W, then a dense GEMM" (awq.py:199-203), which writes and re-reads
// 2 N K bytes of dense weights per call and hands the product to a library.
//   * k is walked in phases of 64 elements.  The activation tile [256 rows x 128 B] goes global -> LDS by LDS-DMA into
//     a ring of 3 stages, two phases ahead (the LDS image of fp8_gemm_xd_kernel: 8-row x 128-B pieces, lane-linear,
//     two rows per 256-B line, 16-byte position XOR-swizzled with (line & 15) on the source side: conflict-free
//     ds_read_b128 fragments);
//   * wave w owns output columns 32 w .. 32 w + 31 (two 16-column tiles) for ALL 256 rows, so every packed weight is
//     dequantised exactly once per workgroup (a 2 x 4 wave grid would dequantise each twice).  Its packed weights --
//     two 1-KiB native tiles per 128-k block -- and the 32 (scale, zero) words of the block arrive by LDS-DMA in a
//     wave-private ring of 3 blocks, two blocks ahead;
//   * per phase and wave: 2 ds_read_b128 of packed weights -> 4 dequantised fragments (exact (w - z) * s, the bits of
//     the dense dequant), 32 ds_read_b128 of activations, 64 MFMA 16x16x32 (1024 cycles): the ~64-160 VALU
//     instructions of the dequant issue in the MFMA shadow;
//   * counted vmcnt waits (7 entries stay in flight: the next phase's 4 activation pieces + the next block's 3 weight
//     pieces), ONE barrier per phase; no compiler-tracked vector load inside the loop.
// LDS: 3 x 32 KiB (activations) + 8 waves x 3 x 2304 B (weights + zs) = 150 KiB.
// Bound: MFMA (bf16/fp16 16x16x32); per-CU ingest 41 KiB per phase (0.85 us of MFMA) = 48 GB/s.
template <typename T>
__global__ __launch_bounds__(512) void w4a16_tile_kernel(const W4Params p, int mblocks, int nblocks,
                                                         float* __restrict__ slab, int S) {
  typedef typename Elem<T>::vec8 vec8;
  constexpr int BM = 256, BN = 256, PB = 128;            // 64 elements of k = 128 bytes of an activation row
  constexpr int XSTAGE = BM * PB, XR = 3;
  constexpr int WBLK = 2 * 1024 + 256, WR = 3;           // per wave and 128-k block: 2 native tiles + 64 zs words
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;

  // XCD-aware renumbering: the ~32 tiles resident on one XCD form an 8 (m) x 4 (n) patch that shares both operands
  // in that XCD's L2 (same scheme as fp8_gemm_tile_kernel)
  const int nwg = mblocks * nblocks;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
  const int tid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
  constexpr int GM = 8;
  const int per_group = GM * nblocks;
  const int grp = tid / per_group, in_grp = tid % per_group;
  const int first_m = grp * GM;
  const int gsz = min(mblocks - first_m, GM);
  const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
  const int64_t m0 = (int64_t)mb * BM, n0 = (int64_t)nb * BN + wave * 32;

  // k range of this workgroup in 128-k blocks (split-K: grid.y = S)
  const int64_t KB = p.K / 128;
  const int64_t bper = (KB + S - 1) / S;
  const int64_t b0 = (int64_t)blockIdx.y * bper;
  const int64_t b1 = min(KB, b0 + bper);
  const int nph = (int)(b1 - b0) * 2;

  const uint32_t lds_base = lds_addr_of(smem);
  const uint32_t wlds = __builtin_amdgcn_readfirstlane(lds_base + XR * XSTAGE + wave * (WR * WBLK));
  // activation DMA geometry (4 of the 32 pieces of a phase per wave)
  const int dline = lane >> 4, dpos = lane & 15;
  uint32_t xoff[4], xlds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int line = piece * 4 + dline;
    const int logical = dpos ^ (line & 15);
    const int row = line * 2 + (logical >> 3);
    xoff[i] = (uint32_t)((min(m0 + row, p.M - 1) - m0) * p.ldx * 2 + (logical & 7) * 16);
    xlds[i] = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
  }
  // packed weights: native tile (nt, kb) = 1 KiB at ((nt * KB + kb) * 1024); zs word of column n in group row g
  const int64_t ntiles = p.N / 16;
  uint32_t woff[2];
  bool tile_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t nt = n0 / 16 + j;
    tile_ok[j] = nt < ntiles;
    woff[j] = (uint32_t)(min(nt, ntiles - 1) * KB * 1024 + lane * 16);
  }
  const uint32_t zoff = (uint32_t)(min(n0 + (lane & 31), p.N - 1) * 4);
  const uint8_t* xbase = (const uint8_t*)p.x + m0 * p.ldx * 2;

  auto issue_x = [&](int ph) __attribute__((always_inline)) {          // phase ph (0-based inside the k range)
    const uint8_t* xa = uniform_ptr(xbase + (b0 * 128 + (int64_t)ph * 64) * 2);
    const uint32_t st = (uint32_t)(ph % XR) * XSTAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_s(xoff[i], xa, xlds[i] + st);
  };
  auto issue_w = [&](int b) __attribute__((always_inline)) {           // block b (0-based inside the k range)
    const int64_t kb = b0 + b;
    const uint8_t* wa = uniform_ptr((const uint8_t*)p.qw + kb * 1024);
    const uint8_t* za = uniform_ptr(p.zs + (int64_t)((int)kb * 128 / (int)p.group) * p.N);
    const uint32_t dst = wlds + (uint32_t)(b % WR) * WBLK;
    glds16_s(woff[0], wa, dst);
    glds16_s(woff[1], wa, dst + 1024);
    glds4_s(zoff, za, dst + 2048);
  };
  auto frag = [&](const char* base, int row, int slot) __attribute__((always_inline)) -> uint4 {
    return *(const uint4*)(base + (row >> 1) * 256 + (((((row & 1) << 3) | slot) ^ ((row >> 1) & 15)) * 16));
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nph > 0) {
    // issue order: W(0) W(1) X(0) X(1) | iteration ph: [wait X(ph)] barrier, X(ph+2), (ph even) W(ph/2 + 2), compute.
    // vmcnt entries younger than X(ph) at its wait: X(ph+1) (4) and the one weight block issued after X(ph) (3):
    // W(ph/2 + 1) for even ph >= 2 (issued in iteration ph-2), W((ph+3)/2) for odd ph (iteration ph-1).
    const int nblk = nph / 2;
    issue_w(0);
    if (nblk > 1) issue_w(1);
    issue_x(0);
    if (nph > 1) issue_x(1);
    for (int ph = 0; ph < nph; ++ph) {
      const bool x_after = ph + 1 < nph;
      const bool w_after = ph == 0 ? false : (ph & 1) ? (ph + 3) / 2 < nblk : ph / 2 + 1 < nblk;
      if (x_after && w_after) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if (x_after) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (w_after) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                    // X(ph) landed everywhere; everyone is done with phase ph-1
      if (ph + 2 < nph) issue_x(ph + 2);                   // into the stage phase ph-1 used
      if (!(ph & 1) && ph / 2 + 2 < nblk) issue_w(ph / 2 + 2);   // into the slot block ph/2 - 1 used (wave-private)
      const char* xb = smem + (ph % XR) * XSTAGE;
      const char* wb = smem + XR * XSTAGE + wave * (WR * WBLK) + ((ph >> 1) % WR) * WBLK;
      vec8 wf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint4 wv = *(const uint4*)(wb + j * 1024 + lane * 16);
        const uint32_t zsv = *(const uint32_t*)(wb + 2048 + (j * 16 + r16) * 4);
        const uint32_t lo = (ph & 1) ? wv.z : wv.x, hi = (ph & 1) ? wv.w : wv.y;
        wf[j][0] = Deq<T>::run(lo, zsv);
        wf[j][1] = Deq<T>::run(hi, zsv);
      }
      // activation fragments in groups of 4 (two m tiles x two k sub-steps), software-pipelined one group ahead: the
      // ds_read_b128 of group g+1 are issued before the 8 MFMAs of group g, so an MFMA never waits on a read issued
      // just in front of it (compiler-ordered reads had 1-2 instructions of lookahead: half the MFMA rate)
      uint4 xg[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xg[0][i] = frag(xb, (i >> 1) * 16 + r16, (i & 1) * 4 + q);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xg[(g + 1) & 1][i] = frag(xb, ((g + 1) * 2 + (i >> 1)) * 16 + r16, (i & 1) * 4 + q);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = g * 2 + (i >> 1), ks = i & 1;
          const vec8 xf = __builtin_bit_cast(vec8, xg[g & 1][i]);
          acc[t][0] = Elem<T>::mfma16(wf[0][ks], xf, acc[t][0]);
          acc[t][1] = Elem<T>::mfma16(wf[1][ks], xf, acc[t][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: lane holds out[m = m0 + 16 t + r16][n = n0 + 16 j + 4 q + r]
  // Full tiles without split-K: two neighbouring waves (64 + 64 bytes of every row) stage their outputs in 32 KiB of the
  // now free rings -- [256 rows][128 B], 16-byte chunk c of row R at position c ^ (R & 7) -- and the pair writes whole
  // 128-byte lines, 8 rows per non-temporal store (the MFMA layout gives 8-byte stores, 16 rows x 32 B per instruction:
  // partial lines, which cost fp8_gemm_tile_kernel a third of its time before the same change).
  const bool staged = S == 1 && tile_ok[0] && tile_ok[1] && ((int64_t)nb * BN + BN <= p.N) && (p.ldo & 7) == 0 &&
                      ((uintptr_t)p.out & 15) == 0;     // uniform over the workgroup (nb, not the wave's n0)
  if (staged) {
    __syncthreads();                                    // every wave is done with the rings
    char* stg = smem + (wave >> 1) * 32768;
    const int half = wave & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t nbase = n0 + j * 16 + 4 * q;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = p.bias ? (float)((const T*)p.bias)[nbase + r] : 0.f;
      const int c = half * 4 + j * 2 + (q >> 1);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = t * 16 + r16;
        *(uint2*)(stg + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) =
            make_uint2(pack2<T>(acc[t][j][0] + bv[0], acc[t][j][1] + bv[1]), pack2<T>(acc[t][j][2] + bv[2], acc[t][j][3] + bv[3]));
      }
    }
    __syncthreads();
    const int pr = lane >> 3, pc = lane & 7;
    typedef __attribute__((ext_vector_type(4))) uint32_t st_u32x4;
    const int64_t ncol = (int64_t)nb * BN + (wave >> 1) * 64 + pc * 8;
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) {
      const int row = half * 128 + ps * 8 + pr;
      const uint4 v = *(const uint4*)(stg + row * 128 + ((pc ^ (row & 7)) << 4));
      const int64_t m = m0 + row;
      if (m < p.M) __builtin_nontemporal_store(st_u32x4{v.x, v.y, v.z, v.w}, (st_u32x4*)((T*)p.out + m * p.ldo + ncol));
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (!tile_ok[j]) continue;
    const int64_t nbase = n0 + j * 16 + 4 * q;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (p.bias && S == 1) ? (float)((const T*)p.bias)[nbase + r] : 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int64_t m = m0 + t * 16 + r16;
      if (m >= p.M) continue;
      if (S > 1) {
        *(f32x4*)(slab + ((int64_t)blockIdx.y * p.M + m) * p.N + nbase) = acc[t][j];
      } else {
        *(uint2*)((T*)p.out + m * p.ldo + nbase) = make_uint2(pack2<T>(acc[t][j][0] + bv[0], acc[t][j][1] + bv[1]),
                                                             pack2<T>(acc[t][j][2] + bv[2], acc[t][j][3] + bv[3]));
      }
    }
  }
}

// split-K factor of the tile kernel: only grids that leave most of the chip idle are split (a phase costs ~0.85 us of
// MFMA per workgroup; the slabs cost a write and a read of S * M * N fp32)
static int w4_tile_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = cdiv64(M, 256) * cdiv64(N, 256), blocks = K / 128;
  int best = 1;
  double best_t = (double)cdiv64(tiles, 256) * (double)blocks * 1.7 + 5.0;
  for (int s = 2; s <= 4 && tiles * s <= 256 && blocks / s >= 8; ++s) {
    const double t = (double)cdiv64(blocks, s) * 1.7 + (double)s * (double)M * (double)N * 8.0 / 4e6 + 10.0;
    if (t < best_t) { best_t = t; best = s; }
  }
  return best;
}

template <typename T>
static void launch_w4_tile(const W4Params& p, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  const int mblocks = (int)cdiv64(p.M, 256), nblocks = (int)cdiv64(p.N, 256);
  int S = w4_tile_splits(p.M, p.N, p.K);
  if (S > 1 && (!workspace || workspace_bytes < (int64_t)S * p.M * p.N * (int64_t)sizeof(float))) S = 1;
  const size_t lds = (size_t)3 * 256 * 128 + (size_t)8 * 3 * 2304;
  w4a16_tile_kernel<T><<<dim3((unsigned)(mblocks * nblocks), (unsigned)S), 512, lds, st>>>(p, mblocks, nblocks,
                                                                                             (float*)workspace, S);
  if (S > 1) w4_reduce_kernel<T><<<(unsigned)cdiv64(p.M * (p.N / 4), 256), 256, 0, st>>>(p, (const float*)workspace, S);
}

static void w4_plan(int64_t N, int64_t K, int* S, int* ppw) {
  const int64_t nblk = cdiv64(N, 128), nph = K / 128;
  int64_t want = nblk >= 200 ? 1 : 256 / nblk;
  if (want < 1) want = 1;
  if (want > nph) want = nph;
  int64_t per = cdiv64(nph, want);
  if (per > 32) per = 32;                 // zs staging capacity
  *ppw = (int)per;
  *S = (int)cdiv64(nph, per);
}

// Plan of the role kernel: column block of 64 NT columns (NT = 4, 2, 1) and split-K factor S.  Per 128-k phase a
// workgroup ingests the activation block (rows x 256 B) + NT x 4 KiB of weights + 1 KiB of (scale, zero) words at
// ~45 GB/s per CU (tools/xd_stamps.py on the FP8 kernel: the LDS-DMA issue rate of a CU), and every slab byte is written
// and read again (~4.5 TB/s each way): the cheapest (NT, S) by that model, S chosen to fill one round of 256 CUs.
// C4 (M = 64): gate_up 22016 x 4096 -> NT 4, S 3; qkv 12288 x 4096 -> NT 4 / S 5 or NT 2 / S 2; o, down -> NT 2 or 1.
static bool w4_xw_plan(int64_t M, int64_t N, int64_t K, int64_t group, int* S, int* ppw, int* nt_out) {
  static const int enable = mi_tune("MI_W4_XW", 1);
  if (!enable || M > 128 || K % 128 != 0 || group % 128 != 0 || N % 64 != 0) return false;
  const int64_t nph = K / 128, rows = M <= 16 ? 16 : M <= 32 ? 32 : M <= 64 ? 64 : 128;
  double best = 0;
  bool found = false;
  static const int force_nt = mi_tune("MI_W4_NT", 0), force_s = mi_tune("MI_W4_S", 0);   // tuning builds: sweep the plan
  for (int nt = 4; nt >= 1; nt >>= 1) {
    const int64_t wr = 64 * nt;
    if (N % wr != 0 || (force_nt && nt != force_nt)) continue;
    if (rows * 256 + nt * 4096 + 1024 > 160 * 1024 / 3) continue;         // three stages must fit
    const int64_t nblk = N / wr;
    int64_t want = nblk >= 200 ? 1 : 256 / nblk;
    if (force_s) want = force_s;
    if (want < 1) want = 1;
    if (want > nph) want = nph;
    const int64_t per = cdiv64(nph, want), s = cdiv64(nph, per);
    const double ingest_us = (double)(rows * 256 + nt * 4096 + 1024) * (double)per / 45e3;
    const double slab_us = s > 1 ? 2.0 * (double)s * (double)M * (double)N * 4.0 / 4.5e6 : 0.0;
    const double cost = (double)cdiv64(nblk * s, 256) * ingest_us + slab_us;
    if (!found || cost < best) { best = cost; found = true; *S = (int)s; *ppw = (int)per; *nt_out = nt; }
  }
  return found;
}

template <typename T, int MT, int NT>
static void launch_w4_xw_nt(const W4Params& p, float* slab, int S, int ppw, hipStream_t st, bool partial) {
  dim3 grid((unsigned)(p.N / (64 * NT)), (unsigned)S);
#ifdef MI_TUNING
  const int dbg = w4_stamps_on() ? 1 : 0;
#else
  const int dbg = 0;
#endif
  w4a16_xw_kernel<T, MT, NT><<<grid, 768, W4XW<MT, NT>::LDS, st>>>(p, slab, S, ppw, dbg | (partial ? 2 : 0));
}
// `reduce`: sum the slabs with w4_reduce_kernel; false = the partial form (slabs even when S == 1, a fused consumer follows)
template <typename T, int MT>
static void launch_w4_xw(const W4Params& p, float* slab, int S, int ppw, int nt, hipStream_t st, bool reduce) {
  if (nt == 4) launch_w4_xw_nt<T, MT, 4>(p, slab, S, ppw, st, !reduce);
  else if (nt == 2) launch_w4_xw_nt<T, MT, 2>(p, slab, S, ppw, st, !reduce);
  else launch_w4_xw_nt<T, MT, 1>(p, slab, S, ppw, st, !reduce);
  if (S > 1 && reduce) w4_reduce_kernel<T><<<(unsigned)cdiv64(p.M * (p.N / 4), 256), 256, 0, st>>>(p, slab, S);
}
template <typename T>
static void launch_w4_xw_m(const W4Params& p, float* slab, int S, int ppw, int nt, hipStream_t st, bool reduce) {
  if (p.M <= 16) launch_w4_xw<T, 1>(p, slab, S, ppw, nt, st, reduce);
  else if (p.M <= 32) launch_w4_xw<T, 2>(p, slab, S, ppw, nt, st, reduce);
  else if (p.M <= 64) launch_w4_xw<T, 4>(p, slab, S, ppw, nt, st, reduce);
  else launch_w4_xw<T, 8>(p, slab, S, ppw, nt, st, reduce);
}

#define W4_CHUNK_MAX_M 512   // batches of 129..512 rows: the decode kernel once per 128-row chunk (above: the tile kernel)
extern "C" int64_t mi_w4a16_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || K % 128 != 0 || N % 16 != 0) return 0;
  if (M > W4_CHUNK_MAX_M) {
    const int S = w4_tile_splits(M, N, K);
    return S > 1 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
  }
  if (M > 128) M = 128;
  int S, ppw, nt;
  w4_plan(N, K, &S, &ppw);
  int64_t need = S > 1 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
  // the role kernel's plan (group size unknown here: assume it applies; its slabs are never larger than 8 x M x N x 4)
  if (w4_xw_plan(M, N, K, 128, &S, &ppw, &nt) && S > 1 && (int64_t)S * M * N * (int64_t)sizeof(float) > need)
    need = (int64_t)S * M * N * (int64_t)sizeof(float);
  return need;
}

template <typename T, int MT>
static void launch_w4_xs(const W4Params& p, float* slab, int S, int ppw, hipStream_t st) {
  constexpr int NWV = 8;
  const size_t lds = (size_t)w4_xs_ring(MT) * (MT * 16 * 256 + NWV * 1024) + (size_t)NWV * 32 * 16 * 4;
  dim3 grid((unsigned)cdiv64(p.N, 16 * NWV), (unsigned)S);
  w4a16_xs_kernel<T, MT, NWV><<<grid, NWV * 64, lds, st>>>(p, slab, S, ppw);
  if (S > 1) w4_reduce_kernel<T><<<(unsigned)cdiv64(p.M * (p.N / 4), 256), 256, 0, st>>>(p, slab, S);
}

// returns false when the shape / workspace does not allow this path
template <typename T>
static bool try_w4_xs(const W4Params& p, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (p.M > W4_CHUNK_MAX_M || p.perm || p.group % 128 != 0) return false;
  if (p.M > 128) {
    // decode batches beyond 128 rows (graph batch sizes up to 512): the fallback kernel below re-reads and re-dequantises
    // the weights per 64-row block (measured M=256: 70-350 us per Llama-2-7B GEMM against 21-46 us at M=128);
    // the x-stationary kernel once per 128-row chunk streams them twice at full rate instead
    for (int64_t m0 = 0; m0 < p.M; m0 += 128) {
      W4Params q = p;
      q.M = p.M - m0 < 128 ? p.M - m0 : 128;
      q.x = (const char*)p.x + m0 * p.ldx * 2;
      q.out = (char*)p.out + m0 * p.ldo * 2;
      if (!try_w4_xs<T>(q, workspace, workspace_bytes, st)) return false;   // only possible for the first chunk
    }
    return true;
  }
  int S, ppw, nt;
  if ((((uintptr_t)p.zs | (uintptr_t)p.out) & 15) == 0 && p.ldo % 8 == 0 && w4_xw_plan(p.M, p.N, p.K, p.group, &S, &ppw, &nt)) {
    const int64_t need_xw = S > 1 ? (int64_t)S * p.M * p.N * (int64_t)sizeof(float) : 0;
    if (need_xw <= workspace_bytes && (need_xw == 0 || workspace)) {
      launch_w4_xw_m<T>(p, (float*)workspace, S, ppw, nt, st, true);
      return true;
    }
  }
  w4_plan(p.N, p.K, &S, &ppw);
  const int64_t need = S > 1 ? (int64_t)S * p.M * p.N * (int64_t)sizeof(float) : 0;
  if (need > workspace_bytes || (need > 0 && !workspace)) {
    if (p.K / 128 > 32) return false;      // cannot hold the whole k-range's zs words without split-K
    S = 1;
    ppw = (int)(p.K / 128);
  }
  float* slab = (float*)workspace;
  if (p.M <= 16) launch_w4_xs<T, 1>(p, slab, S, ppw, st);
  else if (p.M <= 32) launch_w4_xs<T, 2>(p, slab, S, ppw, st);
  else if (p.M <= 64) launch_w4_xs<T, 4>(p, slab, S, ppw, st);
  else launch_w4_xs<T, 8>(p, slab, S, ppw, st);
  return true;
}

template <typename T, bool PERM> static void launch_w4(const W4Params& p, hipStream_t st) {
  const unsigned gx = (unsigned)cdiv64(p.N, 64);
  if (p.M <= 16) w4a16_gemm_kernel<T, 1, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 16)), 256, 0, st>>>(p);
  else if (p.M <= 32) w4a16_gemm_kernel<T, 2, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 32)), 256, 0, st>>>(p);
  else w4a16_gemm_kernel<T, 4, PERM><<<dim3(gx, (unsigned)cdiv64(p.M, 64)), 256, 0, st>>>(p);
}

extern "C" int mi_w4a16_gemm(const void* x, const void* qw_native, const void* zs_native, const int32_t* perm,
                             const void* bias, void* out, int64_t M, int64_t N, int64_t K, int64_t group_size,
                             int64_t ldx, int64_t ldo, int dtype, void* workspace, int64_t workspace_bytes,
                             void* stream) {
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
  MI_CHECK_ARG(((uintptr_t)workspace & 15) == 0 && workspace_bytes >= 0);
  // prefill batches: the 256 x 256 tile kernel with the dequant in the MFMA loop (act-order callers hand in
  // x[:, perm] -- mi_gather_columns -- so the native k order is the contraction order)
  if (M > W4_CHUNK_MAX_M && !perm && group_size % 128 == 0) {
    if (dtype == MI_FP16) launch_w4_tile<f16_t>(p, workspace, workspace_bytes, st);
    else launch_w4_tile<bf16_t>(p, workspace, workspace_bytes, st);
    MI_CHECK_LAUNCH();
    return MI_OK;
  }
  if (dtype == MI_FP16) {
    if (try_w4_xs<f16_t>(p, workspace, workspace_bytes, st)) { MI_CHECK_LAUNCH(); return MI_OK; }
    if (perm) launch_w4<f16_t, true>(p, st); else launch_w4<f16_t, false>(p, st);
  } else {
    if (try_w4_xs<bf16_t>(p, workspace, workspace_bytes, st)) { MI_CHECK_LAUNCH(); return MI_OK; }
    if (perm) launch_w4<bf16_t, true>(p, st); else launch_w4<bf16_t, false>(p, st);
  }
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---- split-K partial form for the fused consumers (fused_glue.hip): raw fp32 accumulators in `slabs` [S][M][N], no
// epilogue.  Returns the split count S (>= 1), or 0 when the role kernel does not take the shape.
MI_INTERNAL int mi_w4a16_plan_splits(int64_t M, int64_t N, int64_t K, int64_t group) {
  int S, ppw, nt;
  return w4_xw_plan(M, N, K, group, &S, &ppw, &nt) ? S : 0;
}
MI_INTERNAL int mi_w4a16_gemm_partial(const void* x, const void* qw_native, const void* zs_native, float* slabs, int64_t M,
                                     int64_t N, int64_t K, int64_t group, int64_t ldx, int dtype, void* stream) {
  MI_CHECK_ARG(x && qw_native && zs_native && slabs && M > 0 && M <= 128);
  MI_CHECK_ARG((((uintptr_t)x | (uintptr_t)qw_native | (uintptr_t)zs_native | (uintptr_t)slabs) & 15) == 0 && ldx % 8 == 0);
  int S, ppw, nt;
  if (!w4_xw_plan(M, N, K, group, &S, &ppw, &nt)) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4a16_gemm_partial: shape not supported");
  W4Params p;
  p.x = x; p.qw = (const uint32_t*)qw_native; p.zs = (const uint32_t*)zs_native; p.perm = nullptr; p.bias = nullptr;
  p.out = nullptr; p.M = M; p.N = N; p.K = K; p.group = group; p.ldx = ldx; p.ldo = N;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_FP16) launch_w4_xw_m<f16_t>(p, slabs, S, ppw, nt, st, false);
  else launch_w4_xw_m<bf16_t>(p, slabs, S, ppw, nt, st, false);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------------- act-order: activations in native k order
// GPTQ act-order stores native row k' = checkpoint row perm[k'] (mi_w4_repack), so the contraction needs x[:, perm].
// out[m][k'] = x[m][perm[k']], 2-byte elements, 4 per thread (one 8-byte store); the source row stays in L1/L2.
__global__ __launch_bounds__(256) void gather_columns_kernel(const uint16_t* __restrict__ x, const int32_t* __restrict__ perm,
                                                             uint16_t* __restrict__ out, int64_t K, int64_t ldx,
                                                             int64_t ldo) {
  const int64_t m = blockIdx.y;
  const int64_t k0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (k0 >= K) return;
  const uint16_t* xr = x + m * ldx;
  const int4 pi = *(const int4*)(perm + k0);
  const uint32_t lo = (uint32_t)xr[pi.x] | ((uint32_t)xr[pi.y] << 16), hi = (uint32_t)xr[pi.z] | ((uint32_t)xr[pi.w] << 16);
  *(uint2*)(out + m * ldo + k0) = make_uint2(lo, hi);
}

extern "C" int mi_gather_columns(const void* x, const int32_t* perm, void* out, int64_t M, int64_t K, int64_t ldx,
                                 int64_t ldo, void* stream) {
  MI_CHECK_ARG(M >= 0 && K > 0);
  if (M == 0) return MI_OK;
  MI_CHECK_ARG(x && perm && out);
  if (K % 4 != 0 || ldo % 4 != 0 || (((uintptr_t)out | (uintptr_t)perm) & 7) != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_gather_columns: need K %% 4 == 0, ldo %% 4 == 0, 8-byte aligned out / perm");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t m0 = 0; m0 < M; m0 += 65535) {     // grid.y limit
    const int64_t rows = M - m0 < 65535 ? M - m0 : 65535;
    gather_columns_kernel<<<dim3((unsigned)cdiv64(K, 1024), (unsigned)rows), 256, 0, st>>>(
        (const uint16_t*)x + m0 * ldx, perm, (uint16_t*)out + m0 * ldo, K, ldx, ldo);
  }
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ------------------------------------------------------- native layout -> dense [N, K]
// The dense W^T [N, K] of a repacked weight (inspection / tests: the product path never materialises it any more --
// prefill batches run w4a16_tile_kernel).  One thread per native dword = 8 consecutive k of one output column, one
// 16-byte store.  Same Deq<T> as the fused kernels: bit-identical weights.
template <typename T>
__global__ __launch_bounds__(256) void w4_dequant_native_kernel(const uint32_t* __restrict__ qw,
                                                                const uint32_t* __restrict__ zs, T* __restrict__ w_nk,
                                                                int64_t N, int64_t K, int64_t group) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (N / 16) * (K / 128) * 256) return;
  const int s = gid & 3, lane = (gid >> 2) & 63;
  const int64_t blk = gid >> 8;
  const int64_t kb = blk % (K / 128), nt = blk / (K / 128);
  const int64_t n = nt * 16 + (lane & 15);
  const int64_t k0 = kb * 128 + 32 * s + 8 * (lane >> 4);
  const auto v = Deq<T>::run(qw[gid], zs[(k0 / group) * N + n]);
  *(uint4*)(w_nk + n * K + k0) = __builtin_bit_cast(uint4, v);
}

extern "C" int mi_w4_dequantize_native(const void* qw_native, const void* zs_native, void* w_nk, int64_t N, int64_t K,
                                       int64_t group_size, int dtype, void* stream) {
  MI_CHECK_ARG(qw_native && zs_native && w_nk && N > 0 && K > 0);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (group_size <= 0) group_size = K;
  if (N % 16 != 0 || K % 128 != 0 || group_size % 8 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_w4_dequantize_native: need N%%16==0, K%%128==0, group%%8==0");
  MI_CHECK_ARG((((uintptr_t)qw_native | (uintptr_t)w_nk) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((N / 16) * (K / 128));
  if (dtype == MI_FP16)
    w4_dequant_native_kernel<f16_t><<<blocks, 256, 0, st>>>((const uint32_t*)qw_native, (const uint32_t*)zs_native, (f16_t*)w_nk, N, K, group_size);
  else
    w4_dequant_native_kernel<bf16_t><<<blocks, 256, 0, st>>>((const uint32_t*)qw_native, (const uint32_t*)zs_native, (bf16_t*)w_nk, N, K, group_size);
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
