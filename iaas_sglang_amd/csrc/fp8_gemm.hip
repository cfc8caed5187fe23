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
  int sa_row, sb_row, rotate;
  int var;      // fp8_gemm_xd_kernel schedule bits (xd_var())
  int rot_step; // fp8_gemm_xw_kernel: phase rotation step across workgroups
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


// ---------------------------------------------------------------------------------------------
// Decode-shaped GEMM (M <= 128): weights are read exactly once, every CU gets a 16-column strip.
//   workgroup = NW waves, tile = (MT*16 rows) x 16 columns x all of K; wave w owns the 128-byte
//   k-chunks c = w, w+NW, ... (adjacent waves read adjacent chunks of the same 16 weight rows, so
//   the workgroup streams whole DRAM pages).  Every operand goes straight from HBM/L2 into MFMA
//   fragments: a lane holds bytes [16q,16q+16) and [64+16q,64+16q+16) of its row's chunk (each
//   wave load covers 64 contiguous bytes of 16 rows); the same k-slot assignment is used for both
//   operands, which is all the MFMA needs.  One v_mfma_scale_f32_16x16x128_f8f6f4 per (m-tile,
//   chunk) with unit block scales (E8M0 0x7F) -- twice the rate of the unscaled fp8 MFMA.
//   Next chunk's loads are issued before the current chunk's MFMAs (register double buffer).
//   The NW partial tiles are summed through LDS and wave t applies the epilogue of m-tile t.
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <typename OutT, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void fp8_gemm_skinny_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [NW][MT][64 lanes][4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  const int64_t K = p.K;
  const int64_t KC = (K + 127) / 128;

  const uint8_t* wp = p.b + min(n0 + r16, p.N - 1) * p.ldb + 16 * q;
  const uint8_t* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = p.a + min((int64_t)t * 16 + r16, p.M - 1) * p.lda + 16 * q;

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // chunk j of this wave is c = wave + NW*j over the KF full chunks; workgroups start at different
  // j so that at any instant the chip touches many different weight/activation columns
  const int64_t KF = K / 128;
  const int64_t J = (KF > wave) ? (KF - wave + NW - 1) / NW : 0;
  const int64_t rot = (p.rotate & 1) ? (int64_t)(blockIdx.x % (unsigned)(J > 0 ? J : 1)) : 0;

  // Two register buffers (a/b) as plain arrays + macros: no structs/lambdas, so nothing can end up
  // in scratch.  Loads never sit under a condition (a per-load select makes hipcc branch and drain
  // vmcnt around every load): past the end they re-read the last chunk, which is harmless.
  uint4 wa0, wa1, wb0, wb1, xa0[MT], xa1[MT], xb0[MT], xb1[MT];
#define SK_CHUNK(j_) (wave + NW * (((j_) + rot) >= J ? ((j_) + rot) - J : ((j_) + rot)))
#define SK_LOAD(W0, W1, X0, X1, c_)                         \
  {                                                         \
    const int64_t kb_ = (int64_t)(c_) * 128;                \
    W0 = *(const uint4*)(wp + kb_);                         \
    W1 = *(const uint4*)(wp + kb_ + 64);                    \
    const int64_t kx_ = kb_;                                \
    _Pragma("unroll") for (int t = 0; t < MT; ++t) {        \
      X0[t] = *(const uint4*)(xp[t] + kx_);                 \
      X1[t] = *(const uint4*)(xp[t] + kx_ + 64);            \
    }                                                       \
  }
#define SK_MMA(W0, W1, X0, X1)                                                                          \
  {                                                                                                     \
    const i32x8 wf_ = {(int)W0.x, (int)W0.y, (int)W0.z, (int)W0.w, (int)W1.x, (int)W1.y, (int)W1.z, (int)W1.w}; \
    _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                                    \
      const i32x8 xf_ = {(int)X0[t].x, (int)X0[t].y, (int)X0[t].z, (int)X0[t].w,                        \
                         (int)X1[t].x, (int)X1[t].y, (int)X1[t].z, (int)X1[t].w};                       \
      acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf_, xf_, acc[t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
    }                                                                                                   \
  }
  if (J > 0) {
    SK_LOAD(wa0, wa1, xa0, xa1, SK_CHUNK(0));
    for (int64_t j = 0; j + 1 < J; j += 2) {
      SK_LOAD(wb0, wb1, xb0, xb1, SK_CHUNK(j + 1));
      SK_MMA(wa0, wa1, xa0, xa1);
      const int64_t jn = j + 2 < J ? j + 2 : J - 1;
      SK_LOAD(wa0, wa1, xa0, xa1, SK_CHUNK(jn));
      SK_MMA(wb0, wb1, xb0, xb1);
    }
    if (J & 1) SK_MMA(wa0, wa1, xa0, xa1);
  }
  if (KF < KC && wave == (int)(KF % NW)) {  // partial last chunk (K % 128 != 0): zero-fill beyond K
    const int64_t kb = KF * 128;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const bool ok0 = kb + 16 * q < K, ok1 = kb + 64 + 16 * q < K;
    wa0 = z; wa1 = z;
    if (ok0) wa0 = *(const uint4*)(wp + kb);
    if (ok1) wa1 = *(const uint4*)(wp + kb + 64);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      xa0[t] = z; xa1[t] = z;
      if (ok0) xa0[t] = *(const uint4*)(xp[t] + kb);
      if (ok1) xa1[t] = *(const uint4*)(xp[t] + kb + 64);
    }
    SK_MMA(wa0, wa1, xa0, xa1);
  }
#undef SK_CHUNK
#undef SK_LOAD
#undef SK_MMA

  // ---- cross-wave (split-K) reduction through LDS, then the scale/bias epilogue
#pragma unroll
  for (int t = 0; t < MT; ++t) *(f32x4*)(red + ((wave * MT + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  for (int t = wave; t < MT; t += NW) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += *(const f32x4*)(red + ((w * MT + t) * 64 + lane) * 4);
    const int64_t m = (int64_t)t * 16 + r16;
    const int64_t nb = n0 + 4 * q;
    if (m < p.M) {
      const float sav = p.sa_row ? p.sa[m] : p.sa[0];
      OutT* o = (OutT*)p.out + m * p.ldo + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = min(nb + r, p.N - 1);
        const float sbv = p.sb_row ? p.sb[n] : p.sb[0];
        const float bv = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
        v[r] = sum[r] * sav * sbv + bv;
      }
      if (nb + 3 < p.N && (p.ldo & 3) == 0) {
        *(uint2*)o = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < p.N) o[r] = (OutT)v[r];
      }
    }
  }
}

template <typename OutT, int MT>
static void launch_skinny(const GemmParams& p, hipStream_t st) {
  constexpr int NW = 8;
  const size_t lds = (size_t)NW * MT * 64 * 4 * sizeof(float);
  fp8_gemm_skinny_kernel<OutT, MT, NW><<<(unsigned)cdiv64(p.N, 16), NW * 64, lds, st>>>(p);
}


// ---------------------------------------------------------------------------------------------
// Decode-shaped GEMM (M <= 128): every operand is staged global -> LDS by LDS-DMA, MFMA reads LDS.
//
// History of this kernel (measured on MI355X, M=128, tools/gemm_bench.py):
//   fragments straight from HBM/L2 (kernel above): 0.6-0.9 TB/s of weights -- 16 activation loads per
//     2 weight loads share one in-order vmcnt queue, only ~2 KiB of weights per wave in flight;
//   activations in LDS, weights in registers: 1.0-2.3 TB/s -- hipcc cannot count register loads that
//     are pending across the loop back-edge and drains vmcnt(0) in front of the first MFMA;
//   this form: NO compiler-tracked vector loads in the loop at all.
// Structure: workgroup (nb, sp) = NWV waves = NWV 16-row weight tiles x a range of 256-byte k-phases
// (split-K factor S chosen so that ~320 workgroups exist; S = 1 when N alone provides them).  Per
// phase the x block [M x 256 B] and each wave's weight block [16 x 256 B] are copied into one of TWO
// LDS stages by inline-asm LDS-DMA (1 KiB per wave instruction = four 256-B row segments) while the
// previous phase computes; rows are unpadded (the DMA image is lane-linear) and the 16-byte slot is
// XOR-swizzled with (row & 15) on the SOURCE side, so ds_read_b128 fragment reads are conflict-free.
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales; one `vmcnt(0)` + barrier per phase.
// S == 1: scale/bias epilogue directly; S > 1: fp32 partial tile to slab[sp][M][N], summed by
// fp8_gemm_reduce_kernel.
struct SiluEpi {
  uint8_t* q_out;
  const float* q_scale;
};

// Epilogue of the decode-shaped kernel (a function so that kernel variants can share it): every wave is past
// the last phase barrier when it is called, so LDS may be reused (EPI == 1 exchange buffer).
template <typename OutT, int MT, int NWV, int EPI>
__device__ __forceinline__ void xs_epilogue(const GemmParams& p, float* __restrict__ slab, int S, int sp, int force_slab,
                                            const SiluEpi& epi, f32x4 (&acc)[MT], char* smem, int64_t n0, int64_t c0,
                                            int64_t Ihalf, bool tile_ok, int lane, int wave) {
  const int r16 = lane & 15, q = lane >> 4;
  if constexpr (EPI == 1) {
    // every wave is past the last phase barrier: the stages are free.  up waves publish round_T(acc*sa*sb+bias)
    f32x4* xch = (f32x4*)smem;                  // [NWV/2][MT][64] f32x4
    float sbv[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = min(n0 + 4 * q + r, p.N - 1);
      sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
      bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t m = min((int64_t)t * 16 + r16, p.M - 1);
      const float sav = p.sa_row ? p.sa[m] : p.sa[0];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = round_to<OutT>(acc[t][r] * sav * sbv[r] + bv[r]);
    }
    if (wave >= NWV / 2) {
#pragma unroll
      for (int t = 0; t < MT; ++t) xch[((wave - NWV / 2) * MT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (wave >= NWV / 2 || !tile_ok) return;
    const float qs = *epi.q_scale;
    const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t m = (int64_t)t * 16 + r16;
      if (m >= p.M) continue;
      const f32x4 u = xch[(wave * MT + t) * 64 + lane];
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = acc[t][r];
        o[r] = round_to<OutT>(round_to<OutT>(silu_f32(g)) * u[r]);
        o[r] = fmaxf(fminf(o[r] * qinv, 448.0f), -448.0f);
      }
      uint32_t w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w, true);
      *(uint32_t*)(epi.q_out + m * Ihalf + c0 + 4 * q) = w;
    }
    return;
  }
  if (!tile_ok) return;

  const int64_t nb = n0 + 4 * q;
  if (S > 1 || force_slab) {  // fp32 partial tile -> slab[sp][m][n]
    float* sb = slab + (int64_t)sp * p.M * p.N;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t m = (int64_t)t * 16 + r16;
      if (m >= p.M) continue;
      float* o = sb + m * p.N + nb;
      if (nb + 3 < p.N && (p.N & 3) == 0) {
        *(f32x4*)o = acc[t];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < p.N) o[r] = acc[t][r];
      }
    }
    return;
  }
  float sbv[4], bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t n = min(nb + r, p.N - 1);
    sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
    bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int64_t m = (int64_t)t * 16 + r16;
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

// EPI = 1 (gate_up of a gated MLP, N = 2*I, S == 1, NWV == 8): waves 0-3 own four 16-row GATE tiles, waves
// 4-7 the UP tiles of the same output columns; the epilogue exchanges the up values through LDS and writes
// fp8(silu(gate) * up) [M, I] with the static scale *q_scale -- bit-identical to the bf16 GEMM output
// followed by mi_silu_and_mul_fp8 (activation.py:56-58 + static quant), without the [M, 2I] round trip.
template <typename OutT, int MT, int NWV, int NST, int EPI = 0>   // NST = LDS ring depth (2 or 3 stages)
__global__ __launch_bounds__(NWV * 64) void fp8_gemm_xs_kernel(const GemmParams p, float* __restrict__ slab,
                                                               int S, int phases_per_wg, int force_slab,
                                                               const SiluEpi epi = SiluEpi{nullptr, nullptr}) {
  constexpr int PW = 256;                       // phase width (bytes of K) = 2 MFMA k-chunks
  constexpr int ROWS = MT * 16;
  constexpr int XBYTES = ROWS * PW;             // x block of one stage
  constexpr int WBYTES = NWV * 16 * PW;         // one 16-row weight block per wave
  constexpr int STAGE = XBYTES + WBYTES;
  // NST == 5: SPLIT ring -- x in 2 stages [0, 2*XBYTES), weights in 3 wave-private stages behind them.  The
  // weight block of phase ph+2 is requested two phases ahead and is the LAST thing each wave issues, so the
  // per-phase wait is vmcnt(4) (x of ph+1 and weights of ph+1 landed, weights of ph+2 still flying): HBM
  // latency is spread over two phases instead of one.  (All loads of a wave retire in order, so x -- needed
  // one phase ahead -- must not be queued behind more than one weight block.)
  constexpr bool SPLIT = (NST == 5);
  auto x_off = [&](int st) { return SPLIT ? st * XBYTES : st * STAGE; };
  auto w_off = [&](int st) { return SPLIT ? 2 * XBYTES + st * WBYTES : st * STAGE + XBYTES; };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t Ihalf = p.N / 2;                // EPI only
  const int64_t c0 = ((int64_t)blockIdx.x * (NWV / 2) + (wave & (NWV / 2 - 1))) * 16;   // EPI: output column of the tile
  const int64_t n0 = EPI ? (wave < NWV / 2 ? c0 : Ihalf + c0) : ((int64_t)blockIdx.x * NWV + wave) * 16;
  const int sp = blockIdx.y;
  const int64_t KC = p.K / 128;                 // K % 128 == 0 on this path
  const int64_t NPH = (KC + 1) / 2;
  const int64_t ph0 = (int64_t)sp * phases_per_wg;
  const int64_t ph1 = min(NPH, ph0 + phases_per_wg);
  const bool tile_ok = EPI ? c0 < Ihalf : n0 < p.N;
  const uint32_t lds_base = lds_addr_of(smem);

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA geometry: one instruction = 4 rows x 256 B; lane L -> row (L >> 4), LDS slot (L & 15);
  // the global source slot is (L & 15) ^ (row & 15)
  const int drow = lane >> 4, dslot = lane & 15;
  const uint8_t* wrow[4];                       // this wave's weight rows, 4 per DMA instruction
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = p.b + min(n0 + i * 4 + drow, p.N - 1) * p.ldb;

#define XS_STAGE_W(ph_, st_)                                                                        \
  {                                                                                                 \
    const int64_t kb_ = (int64_t)(ph_) * PW;                                                        \
    const int nslot_ = (int)min((int64_t)16, (p.K - kb_) / 16);                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {             /* weights: rows 4i .. 4i+3 */      \
      const int row_ = i * 4 + drow;                                                                \
      int ss_ = dslot ^ (row_ & 15);                                                                \
      ss_ = ss_ < nslot_ ? ss_ : 0;                                                                 \
      glds16(wrow[i] + kb_ + ss_ * 16, lds_base + w_off(st_) + (wave * 16 + i * 4) * PW);           \
    }                                                                                               \
  }
#define XS_STAGE_X(ph_, st_)                                                                        \
  {                                                                                                 \
    const int64_t kb_ = (int64_t)(ph_) * PW;                                                        \
    const int nslot_ = (int)min((int64_t)16, (p.K - kb_) / 16);                                     \
    _Pragma("unroll") for (int i = 0; i < (ROWS / 4 + NWV - 1) / NWV; ++i) {   /* x rows */         \
      const int rr_ = (i * NWV + wave) * 4;                                                         \
      if (rr_ < ROWS) {                                                                             \
        const int row_ = rr_ + drow;                                                                \
        int ss_ = dslot ^ (row_ & 15);                                                              \
        ss_ = ss_ < nslot_ ? ss_ : 0;                                                               \
        glds16(p.a + min((int64_t)row_, p.M - 1) * p.lda + kb_ + ss_ * 16, lds_base + x_off(st_) + rr_ * PW); \
      }                                                                                             \
    }                                                                                               \
  }
#define XS_STAGE(ph_, st_)   \
  {                          \
    XS_STAGE_W(ph_, st_);    \
    XS_STAGE_X(ph_, st_);    \
  }
#define XS_MMA(ph_, st_) XS_MMA2(ph_, st_, st_)
#define XS_MMA2(ph_, xst_, wst_)                                                                           \
  if (tile_ok) {                                                                                           \
    const int nch_ = (int)min((int64_t)2, KC - (int64_t)(ph_) * 2);                                        \
    const char* xb_ = smem + x_off(xst_) + r16 * PW;                                                       \
    const char* wb_ = smem + w_off(wst_) + (wave * 16 + r16) * PW;                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                        \
      if (c < nch_) {                                                                                      \
        const int o0_ = ((c * 8 + q) ^ r16) * 16, o1_ = ((c * 8 + 4 + q) ^ r16) * 16;                      \
        const uint4 w0_ = *(const uint4*)(wb_ + o0_), w1_ = *(const uint4*)(wb_ + o1_);                    \
        const i32x8 wf_ = {(int)w0_.x, (int)w0_.y, (int)w0_.z, (int)w0_.w,                                 \
                           (int)w1_.x, (int)w1_.y, (int)w1_.z, (int)w1_.w};                                \
        _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                                   \
          const uint4 x0_ = *(const uint4*)(xb_ + t * 16 * PW + o0_);                                      \
          const uint4 x1_ = *(const uint4*)(xb_ + t * 16 * PW + o1_);                                      \
          const i32x8 xf_ = {(int)x0_.x, (int)x0_.y, (int)x0_.z, (int)x0_.w,                               \
                             (int)x1_.x, (int)x1_.y, (int)x1_.z, (int)x1_.w};                              \
          acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf_, xf_, acc[t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  }
  // NST stages in an LDS ring: NST-1 phases are in flight while one computes.  Each wave issues DPP DMA
  // instructions per phase, so "phase ph+1 has landed, ph+2 may still fly" is a COUNTED wait vmcnt(DPP)
  // (loads retire in order); the barrier then publishes every wave's pieces of phase ph+1.
  constexpr int DPP = 4 + (ROWS / 4 + NWV - 1) / NWV;   // DMA instructions per wave per phase
#define XS_BARRIER_AFTER(N_)                                      \
  {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");     \
    __syncthreads();                                              \
  }
  if (ph0 < ph1) {
    if constexpr (SPLIT) {
      XS_STAGE_X(ph0, 0);
      XS_STAGE_W(ph0, 0);
      if (ph0 + 1 < ph1) {
        XS_STAGE_W(ph0 + 1, 1);
        XS_BARRIER_AFTER(4);
      } else {
        XS_BARRIER_AFTER(0);
      }
      int ws = 0;
      for (int64_t ph = ph0; ph < ph1; ++ph) {
        const int xs = (int)((ph - ph0) & 1);
        const int ws2 = ws == 0 ? 2 : ws - 1;   // (ws + 2) % 3
        if (ph + 1 < ph1) XS_STAGE_X(ph + 1, xs ^ 1);
        if (ph + 2 < ph1) {
          XS_STAGE_W(ph + 2, ws2);
          XS_MMA2(ph, xs, ws);
          XS_BARRIER_AFTER(4);                  // x(ph+1), w(ph+1) landed; w(ph+2) in flight
        } else {
          XS_MMA2(ph, xs, ws);
          XS_BARRIER_AFTER(0);
        }
        ws = ws == 2 ? 0 : ws + 1;
      }
    } else if constexpr (NST == 2) {
      XS_STAGE(ph0, 0);
      XS_BARRIER_AFTER(0);
      for (int64_t ph = ph0; ph < ph1; ++ph) {
        const int st = (int)((ph - ph0) & 1);
        if (ph + 1 < ph1) XS_STAGE(ph + 1, st ^ 1);
        XS_MMA(ph, st);
        XS_BARRIER_AFTER(0);
      }
    } else {
      XS_STAGE(ph0, 0);
      if (ph0 + 1 < ph1) {
        XS_STAGE(ph0 + 1, 1);
        XS_BARRIER_AFTER(DPP);
      } else {
        XS_BARRIER_AFTER(0);
      }
      int st = 0;
      for (int64_t ph = ph0; ph < ph1; ++ph) {
        const int st2 = st == 0 ? 2 : st - 1;   // (st + 2) % 3
        if (ph + 2 < ph1) {
          XS_STAGE(ph + 2, st2);
          XS_MMA(ph, st);
          XS_BARRIER_AFTER(DPP);                // phase ph+1 landed, ph+2 in flight
        } else {
          XS_MMA(ph, st);
          XS_BARRIER_AFTER(0);
        }
        st = st == 2 ? 0 : st + 1;
      }
    }
  }
#undef XS_BARRIER_AFTER
#undef XS_STAGE
#undef XS_STAGE_W
#undef XS_STAGE_X
#undef XS_MMA
#undef XS_MMA2
  xs_epilogue<OutT, MT, NWV, EPI>(p, slab, S, sp, force_slab, epi, acc, smem, n0, c0, Ihalf, tile_ok, lane, wave);
}



// schedule bits of fp8_gemm_xd_kernel: 1 = non-temporal weight DMA, 2 = waves 4-7 compute before they issue (stagger),
// 4 = all fragment reads of a phase ahead of its MFMAs, 8 = s_setprio 1 on waves 4-7
#ifdef MI_TUNING
// diagnostic build only: per (workgroup, wave) cycle sums of the phase segments of fp8_gemm_xd_kernel (var bit 16)
__device__ unsigned long long mi_xd_stamps[512 * 12 * 8];
extern "C" int mi_debug_xd_stamps(unsigned long long* host_out, int clear) {
  if (clear) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(mi_xd_stamps)) != hipSuccess) return -1;
    return hipMemset(d, 0, sizeof(unsigned long long) * 512 * 12 * 8) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi_xd_stamps), sizeof(unsigned long long) * 512 * 12 * 8) == hipSuccess ? 0 : -1;
}
#endif
static int xw_rot() {
  static const int v = mi_tune("MI_XW_ROT", 7);
  return v;
}
static int xd_var() {
  static const int v = mi_tune("MI_XD_VAR", 0);
  return v;
}

// ---------------------------------------------------------------------------------------------
// fp8_gemm_xd_kernel: the decode-shaped GEMM with a DEEP ring of 128-byte k-phases.
//
// What bounds the decode GEMM (tools/src/lds_fill.hip, a pure LDS-DMA skeleton of the gate_up shape, no MFMA):
//   64-KiB stages (32 KiB shared x + 32 KiB private weights), issue -> wait -> barrier:   36 us  (= fp8_gemm_xs_kernel)
//   the same with one stage in flight across the wait:                                     25 us
//   32-KiB stages (one 128-byte k-phase), ring of 2 / 3 / 4 / 5:                     25.9 / 24.8 / 24.8 / 25.0 us
// i.e. a stage takes ~2.2 us to land under full load wherever it comes from (L2 or HBM), and 160 KiB of LDS
// cannot hold more than two 64-KiB stages.  So: 128-byte phases, x block [M x 128 B] + 8 wave-private weight
// blocks [16 x 128 B] = 32 KiB per stage at M = 128, a ring of R = 4 stages, every stage requested 3 phases
// ahead, counted vmcnt (2 phases in flight across the barrier), ONE barrier per phase.
// LDS image as in fp8_gemm_tile_kernel: a DMA piece = 8 rows x 128 B (1 KiB, lane-linear), two consecutive rows
// share one 256-byte LDS line, the 16-byte position inside the line is XOR-swizzled with (line & 15) on the
// SOURCE side -> conflict-free ds_read_b128 fragments.  One v_mfma_scale_f32_16x16x128_f8f6f4 per (m tile, phase);
// phases are walked in order, so results are bit-identical to fp8_gemm_xs_kernel (same fp32 accumulation order).
template <typename OutT, int MT, int EPI, int R>
__global__ __launch_bounds__(512) void fp8_gemm_xd_kernel(const GemmParams p, float* __restrict__ slab, int S,
                                                          int phases_per_wg /* 128-byte phases */, int force_slab,
                                                          const SiluEpi epi) {
  constexpr int NWV = 8, PB = 128, D = R - 1;
  constexpr int ROWS = MT * 16;
  constexpr int XBYTES = ROWS * PB, WBYTES = NWV * 16 * PB, STAGE = XBYTES + WBYTES;
  constexpr int XP = ROWS / 8;                          // x DMA pieces (8 rows x 128 B) per phase
  constexpr int XD = XP >= NWV ? XP / NWV : 1;          // per wave (small M: several waves fetch the same piece)
  constexpr int E = XD + 2;                             // vmcnt entries per wave per phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t Ihalf = p.N / 2;
  const int64_t c0 = ((int64_t)blockIdx.x * (NWV / 2) + (wave & (NWV / 2 - 1))) * 16;
  const int64_t n0 = EPI ? (wave < NWV / 2 ? c0 : Ihalf + c0) : ((int64_t)blockIdx.x * NWV + wave) * 16;
  const int sp = blockIdx.y;
  const int64_t NPH = p.K / PB;
  const int64_t ph0 = (int64_t)sp * phases_per_wg;
  const int64_t ph1 = min(NPH, ph0 + phases_per_wg);
  const bool tile_ok = EPI ? c0 < Ihalf : n0 < p.N;
  const uint32_t lds_base = lds_addr_of(smem);

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA geometry of a piece: lane L -> LDS byte L*16 of the piece; line pair l = 4*piece + (L >> 4), position
  // P = L & 15 holds logical (row bit, slot) = P ^ (l & 15); 32-bit byte offsets from the scalar phase base
  const int dline = lane >> 4, dpos = lane & 15;
  uint32_t woff[2], xoff[XD], xlds[XD];
#pragma unroll
  for (int i = 0; i < 2; ++i) {                         // this wave's own 16 weight rows = pieces 2*wave, 2*wave+1
    const int line = (wave * 2 + i) * 4 + dline;
    const int logical = dpos ^ (line & 15);
    const int wrow = i * 8 + dline * 2 + (logical >> 3);                    // row inside the wave's 16
    woff[i] = (uint32_t)(min(n0 + wrow, p.N - 1) * p.ldb + (logical & 7) * 16);
  }
#pragma unroll
  for (int i = 0; i < XD; ++i) {
    const int piece = (i * NWV + wave) % XP;
    const int line = piece * 4 + dline;
    const int logical = dpos ^ (line & 15);
    const int row = line * 2 + (logical >> 3);
    xoff[i] = (uint32_t)(min((int64_t)row, p.M - 1) * p.lda + (logical & 7) * 16);
    xlds[i] = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
  }
  const uint32_t wlds = __builtin_amdgcn_readfirstlane(lds_base + XBYTES + wave * 2 * 1024);
  const int var = p.var;      // diagnostic builds: bit 16 = phase stamps, bit 32 = finer stamps
  (void)var;

  auto issue = [&](int64_t ph) __attribute__((always_inline)) {
    const uint32_t st = (uint32_t)((ph - ph0) % R) * STAGE;
    const uint8_t* xa = p.a + ph * PB;                   // wave-uniform: scalar arithmetic
    const uint8_t* wa = p.b + ph * PB;
#pragma unroll
    for (int i = 0; i < XD; ++i) glds16_s(xoff[i], xa, xlds[i] + st);
    glds16_s(woff[0], wa, wlds + st);
    glds16_s(woff[1], wa, wlds + st + 1024);
  };
  // fragment (row, 16-byte slot) of a tile whose rows are 128 B: line pair = row >> 1, physical position
  // ((row & 1) * 8 + slot) ^ (line & 15)
  auto frag = [&](const char* base, int row, int slot) __attribute__((always_inline)) -> uint4 {
    return *(const uint4*)(base + (row >> 1) * 256 + (((((row & 1) << 3) | slot) ^ ((row >> 1) & 15)) * 16));
  };
  // phases issued AFTER `ph` so far may stay in flight while we wait for `ph`: 0 .. D-1 of them
  auto wait_phase = [&](int64_t ph) __attribute__((always_inline)) {
    const int64_t after = min((int64_t)(D - 1), ph1 - 1 - ph);
    if (D >= 4 && after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * E) : "memory");
    else if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * E) : "memory");
    else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  static_assert(D >= 2 && D <= 4, "wait_phase enumerates up to 3 phases in flight");

  if (ph0 < ph1) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (ph0 + d < ph1) issue(ph0 + d);
#ifdef MI_TUNING
    unsigned long long t_cmp = 0, t_wait = 0, t_bar = 0, t_prev = 0, t_iss = 0, t_rd = 0, t_mma = 0;
    const bool stamp = (var & 16) != 0;
#endif
    for (int64_t ph = ph0; ph < ph1; ++ph) {
#ifdef MI_TUNING
      if (stamp) {   // wait_phase split into its two halves
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        const int64_t after = min((int64_t)(D - 1), ph1 - 1 - ph);
        if (D >= 4 && after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * E) : "memory");
        else if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * E) : "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
        if (t_prev) t_cmp += ta - t_prev;
        t_wait += tb - ta;
        t_bar += tc - tb;
        t_prev = tc;
        if (var & 32) {   // finer: DMA issue | fragment reads | MFMAs, each closed by its own stamp
          if (ph + D < ph1) issue(ph + D);
          const unsigned long long t1 = __builtin_amdgcn_s_memtime();
          const char* xb = smem + ((ph - ph0) % R) * STAGE;
          const char* wb = xb + XBYTES;
          const uint4 w0 = frag(wb, wave * 16 + r16, q), w1 = frag(wb, wave * 16 + r16, 4 + q);
          const i32x8 wf = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
          uint4 x0[MT], x1[MT];
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            x0[t] = frag(xb, t * 16 + r16, q);
            x1[t] = frag(xb, t * 16 + r16, 4 + q);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const i32x8 xf = {(int)x0[t].x, (int)x0[t].y, (int)x0[t].z, (int)x0[t].w,
                              (int)x1[t].x, (int)x1[t].y, (int)x1[t].z, (int)x1[t].w};
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xf, acc[t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
          }
          const unsigned long long t3 = __builtin_amdgcn_s_memtime();
          t_iss += t1 - tc; t_rd += t2 - t1; t_mma += t3 - t2;
          continue;
        }
      } else
#endif
      wait_phase(ph);                                    // phase ph landed everywhere; everyone is done with ph-1
      if (ph + D < ph1) issue(ph + D);                   // into the stage phase ph-1 used
      if (tile_ok) {
        const char* xb = smem + ((ph - ph0) % R) * STAGE;
        const char* wb = xb + XBYTES;
        const uint4 w0 = frag(wb, wave * 16 + r16, q), w1 = frag(wb, wave * 16 + r16, 4 + q);
        const i32x8 wf = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const uint4 x0 = frag(xb, t * 16 + r16, q), x1 = frag(xb, t * 16 + r16, 4 + q);
          const i32x8 xf = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
          acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xf, acc[t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
      }
    }
#ifdef MI_TUNING
    if (stamp && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
      unsigned long long* o = mi_xd_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      o[0] += t_cmp; o[1] += t_wait; o[2] += t_bar; o[3] += (unsigned long long)(ph1 - ph0);
      o[4] += t_iss; o[5] += t_rd; o[6] += t_mma;
    }
#endif
    __syncthreads();                                     // the epilogue may reuse LDS
  }
  xs_epilogue<OutT, MT, NWV, EPI>(p, slab, S, sp, force_slab, epi, acc, smem, n0, c0, Ihalf, tile_ok, lane, wave);
}
// ---------------------------------------------------------------------------------------------
// fp8_gemm_xw_kernel: the decode-shaped GEMM with wave ROLES (round 3).
//
// In-kernel stamps of fp8_gemm_xd_kernel (tools/xd_stamps.py, gate_up shape 128 x 28672 x 4096, cycles per 128-byte
// phase and wave): 376 blocked in the issue of its 4 DMA pieces (all 8 waves queue 32 pieces on the CU's one address
// path at once), 630 for its 18 fragment reads (every wave reads the WHOLE x stage: 8 x 18 KiB = 144 KiB per phase
// through a 256 B/clk LDS = 576 cycles), ~260 of MFMA, 280 in the barrier (the older half waits for the younger) and
// only ~100 waiting for data: the phase is a sum of three lock-stepped segments, not memory.  Re-ordering inside the
// wave (stagger, fragment prefetch, priorities, non-temporal weights) changed nothing measurable.  So:
//   * waves 4-7 are LOADERS: they only issue LDS-DMA (4 weight + XL activation pieces each per phase), wait for
//     their pieces with a counted vmcnt and meet the barrier; blocked issue slots cost nobody anything;
//   * waves 0-3 are CONSUMERS, one per SIMD, each owning 32 weight rows x all M rows: an x fragment now feeds two
//     MFMAs, the LDS read volume per phase drops from 144 KiB to 4 x 20 KiB, and nothing but ds_read + MFMA is in
//     their instruction stream.
// Same LDS image, ring, phase order and MFMA operands as fp8_gemm_xd_kernel: results are bit-identical to it.
// Weight-block row j of the workgroup (0..127) is output column nb*128 + j, or with EPI = 1 (gate_up + SiLU*mul):
// j < 64 -> gate column 64*nb + j, else the up column I + 64*nb + j - 64; consumers 0,1 end with gate values and
// consumers 2,3 with the up values of the same columns in the same registers.
template <typename OutT, int MT, int EPI, int R, int NT = 2>   // NT = 16-row weight tiles per consumer: block of 64 NT rows
__global__ __launch_bounds__(512) void fp8_gemm_xw_kernel(const GemmParams p, float* __restrict__ slab, int S,
                                                          int phases_per_wg /* 128-byte phases */, int force_slab,
                                                          const SiluEpi epi, int staged /* epilogue through LDS */) {
  constexpr int PB = 128, D = R - 1, NL = 4;
  constexpr int ROWS = MT * 16;
  constexpr int WR = 64 * NT;                           // weight rows (= output columns) of the workgroup
  constexpr int XBYTES = ROWS * PB, WBYTES = WR * PB, STAGE = XBYTES + WBYTES;
  static_assert(NT == 2 || (NT == 1 && EPI == 0), "the SiLU form pairs 64 gate with 64 up rows");
  constexpr int XP = ROWS / 8;                          // x DMA pieces (8 rows x 128 B) per phase
  constexpr int XL = XP >= NL ? XP / NL : 1;            // per loader (small M: loaders re-fetch a piece)
  constexpr int WL = WR / 8 / NL;                       // weight pieces per loader
  constexpr int E = XL + WL;                            // vmcnt entries per loader per phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t Ihalf = p.N / 2;
  const int sp = blockIdx.y;
  const int64_t NPH = p.K / PB;
  const int64_t ph0 = (int64_t)sp * phases_per_wg;
  const int64_t ph1 = min(NPH, ph0 + phases_per_wg);
  const uint32_t lds_base = lds_addr_of(smem);
#ifdef MI_TUNING
  const unsigned long long t_k0 = (p.var & 16) ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long r_k0 = (p.var & 16) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
  // output column of weight-block row j
  auto col_of = [&](int j) __attribute__((always_inline)) -> int64_t {
    if (EPI) return (j < 64 ? 0 : Ihalf - 64) + (int64_t)blockIdx.x * 64 + j;
    return (int64_t)blockIdx.x * WR + j;
  };

  if (wave >= 4) {
    // ---------------------------------------------------------------- loaders
    const int ld = wave - 4;
    const int dline = lane >> 4, dpos = lane & 15;
    uint32_t woff[WL], xoff[XL], wlds[WL], xlds[XL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int piece = ld * WL + i;
      const int line = piece * 4 + dline;
      const int logical = dpos ^ (line & 15);
      const int j = line * 2 + (logical >> 3);
      woff[i] = (uint32_t)(min(col_of(j), p.N - 1) * p.ldb + (logical & 7) * 16);
      wlds[i] = __builtin_amdgcn_readfirstlane(lds_base + XBYTES + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int piece = (ld * XL + i) % XP;
      const int line = piece * 4 + dline;
      const int logical = dpos ^ (line & 15);
      const int row = line * 2 + (logical >> 3);
      xoff[i] = (uint32_t)(min((int64_t)row, p.M - 1) * p.lda + (logical & 7) * 16);
      xlds[i] = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
    }
    // Workgroup b walks its phases cyclically from phase (7 b) % count: at any moment the workgroups of the chip then
    // ask for different k-slices of x and of their weight rows, instead of the same 128-byte column of every row
    // (same L2 / memory channels for everybody).  Measured on the gate_up shape once the kernel was DMA-bound:
    // 35.4 -> 30.8 us, any odd step 3..23 and an XCD-wise offset alike; split-K shapes (4-14 phases) are unchanged.
    // Only the loaders know: the consumers accumulate stages in arrival order (a fixed order per workgroup).
    const int64_t nphw = ph1 - ph0;
    const int64_t rot = nphw > 0 ? ((int64_t)blockIdx.x * p.rot_step) % nphw : 0;
    auto issue = [&](int64_t ph) __attribute__((always_inline)) {
      const uint32_t st = (uint32_t)((ph - ph0) % R) * STAGE;
      int64_t kph = ph + rot;
      kph = kph >= ph1 ? kph - nphw : kph;
      const uint8_t* xa = p.a + kph * PB;                // wave-uniform: scalar arithmetic
      const uint8_t* wa = p.b + kph * PB;
#pragma unroll
      for (int i = 0; i < XL; ++i) glds16_s(xoff[i], xa, xlds[i] + st);
#pragma unroll
      for (int i = 0; i < WL; ++i) glds16_s(woff[i], wa, wlds[i] + st);
    };
    static_assert(D >= 2 && D <= 4, "the wait enumerates up to 3 phases in flight");
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (ph0 + d < ph1) issue(ph0 + d);
#ifdef MI_TUNING
    unsigned long long t_iss = 0, t_wait = 0, t_bar = 0;
    const bool stamp = (p.var & 16) != 0;
#endif
    for (int64_t ph = ph0; ph < ph1; ++ph) {
#ifdef MI_TUNING
      const unsigned long long ta = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
      const int64_t after = min((int64_t)(D - 1), ph1 - 1 - ph);   // phases issued after ph may stay in flight
      if (D >= 4 && after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * E) : "memory");
      else if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * E) : "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MI_TUNING
      const unsigned long long tb = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
      __builtin_amdgcn_s_barrier();                      // phase ph published; the consumers are done with ph-1
#ifdef MI_TUNING
      const unsigned long long tc = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
      if (ph + D < ph1) issue(ph + D);                   // into the stage phase ph-1 used
#ifdef MI_TUNING
      if (stamp) {
        const unsigned long long td = __builtin_amdgcn_s_memtime();
        t_wait += tb - ta; t_bar += tc - tb; t_iss += td - tc;
      }
#endif
    }
#ifdef MI_TUNING
    if (stamp && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
      unsigned long long* o = mi_xd_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      o[0] += t_iss; o[1] += t_wait; o[2] += t_bar; o[3] += (unsigned long long)(ph1 - ph0);
    }
#endif
    __builtin_amdgcn_s_barrier();                        // (the consumers' "LDS may be reused" barrier)
    if (!staged) {
      if constexpr (EPI == 1) __builtin_amdgcn_s_barrier();   // (the exchange barrier of the per-lane epilogue)
      return;
    }
  }

  // ------------------------------------------------------------------ consumers
  f32x4 acc[NT][MT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[tn][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment (row, 16-byte slot) of a tile whose rows are 128 B: line pair = row >> 1, physical position
  // ((row & 1) * 8 + slot) ^ (line & 15)
  auto frag = [&](const char* base, int row, int slot) __attribute__((always_inline)) -> uint4 {
    return *(const uint4*)(base + (row >> 1) * 256 + (((((row & 1) << 3) | slot) ^ ((row >> 1) & 15)) * 16));
  };
  if (wave < 4) {
#ifdef MI_TUNING
  unsigned long long t_cmp = 0, t_bar = 0, t_prev = 0;
  const bool stamp = (p.var & 16) != 0;
#endif
  for (int64_t ph = ph0; ph < ph1; ++ph) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef MI_TUNING
    const unsigned long long ta = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
    __builtin_amdgcn_s_barrier();                        // phase ph landed (the loaders waited for it)
#ifdef MI_TUNING
    if (stamp) {
      const unsigned long long tb = __builtin_amdgcn_s_memtime();
      if (t_prev) t_cmp += ta - t_prev;
      t_bar += tb - ta;
      t_prev = tb;
    }
#endif
    const char* xb = smem + ((ph - ph0) % R) * STAGE;
    const char* wb = xb + XBYTES;
    i32x8 wf[NT];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int j = wave * (16 * NT) + tn * 16 + r16;
      const uint4 w0 = frag(wb, j, q), w1 = frag(wb, j, 4 + q);
      wf[tn] = i32x8{(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
    }
    uint4 x0[MT], x1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      x0[t] = frag(xb, t * 16 + r16, q);
      x1[t] = frag(xb, t * 16 + r16, 4 + q);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const i32x8 xf = {(int)x0[t].x, (int)x0[t].y, (int)x0[t].z, (int)x0[t].w,
                        (int)x1[t].x, (int)x1[t].y, (int)x1[t].z, (int)x1[t].w};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
        acc[tn][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[tn], xf, acc[tn][t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    // one wave per SIMD: nothing else hides the LDS latency, so the x fragments run PF m tiles ahead of their MFMAs
    constexpr int PF = MT < 2 ? MT : 2;
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 2 * PF, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (t + PF < MT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    }
  }
#ifdef MI_TUNING
  if (stamp && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) {
    unsigned long long* o = mi_xd_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
    o[0] += t_cmp; o[2] += t_bar; o[3] += (unsigned long long)(ph1 - ph0);
    o[4] += __builtin_amdgcn_s_memtime() - t_k0;          // kernel entry -> end of the main loop
    o[5] += __builtin_amdgcn_s_memrealtime() - r_k0;      // the same in 100 MHz ticks
    o[6] = r_k0;                                          // absolute (chip-wide 100 MHz counter), last launch
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every stage consumed: the epilogue may reuse LDS
  }

#ifdef MI_TUNING
  if (p.var & 64) return;                                // diagnostic: no epilogue at all (bounds what the stores cost)
#endif
  // ------------------------------------------------------------------ epilogue through LDS (`staged`): the consumers
  // park their [MT*16 rows m][128 block columns j] fp32 tile in the free stages (rows padded to 528 B: conflict-free
  // ds_write_b128), then ALL 8 waves write it out in whole rows.  Measured before this form existed (per-lane stores
  // straight from the MFMA layout: 16 rows x 64 B -- or 16 B with EPI -- per instruction, and with EPI the SiLU
  // arithmetic of a 128 x 64 tile on two waves): 4.3-8.1 us of a 13-37 us kernel went to the epilogue.
  if (staged) {
    constexpr int TP = WR * 4 + 16;
    const bool to_slab = !EPI && (S > 1 || force_slab);
    if (wave < 4) {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        float sbv[4] = {1.f, 1.f, 1.f, 1.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (!to_slab) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t n = min(col_of(wave * (16 * NT) + tn * 16 + 4 * q + r), p.N - 1);
            sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
            bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
          }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = acc[tn][t];
          if (!to_slab) {
            const int64_t m = min((int64_t)t * 16 + r16, p.M - 1);
            const float sav = p.sa_row ? p.sa[m] : p.sa[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float y = v[r] * sav * sbv[r] + bv[r];
              v[r] = EPI ? round_to<OutT>(y) : y;
            }
          }
          *(f32x4*)(smem + (t * 16 + r16) * TP + (wave * (16 * NT) + tn * 16 + 4 * q) * 4) = v;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (EPI == 1) {
      // lane -> row 16*wave + lane/4, output columns 16*(lane%4) .. +15 of the workgroup's 64: one 16-byte store
      const int row = wave * 16 + (lane >> 2), c = lane & 3;
      if (row < ROWS && row < p.M) {
        const float qs = *epi.q_scale;
        const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
        const char* gp = smem + row * TP + c * 64;
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 g4 = *(const f32x4*)(gp + k * 16), u4 = *(const f32x4*)(gp + 256 + k * 16);
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = g4[r];
            o[r] = round_to<OutT>(round_to<OutT>(silu_f32(g)) * u4[r]);
            o[r] = fmaxf(fminf(o[r] * qinv, 448.0f), -448.0f);
          }
          w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
          w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w[k], true);
        }
        *(uint4*)(epi.q_out + (int64_t)row * Ihalf + (int64_t)blockIdx.x * 64 + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else if (to_slab) {
      float* sbp = slab + (int64_t)sp * p.M * p.N + (int64_t)blockIdx.x * WR;
      constexpr int LPR = WR / 4, RPW = 64 / LPR;        // lanes per row (16 B each), rows per wave instruction
#pragma unroll
      for (int i = 0; i < (ROWS + 8 * RPW - 1) / (8 * RPW); ++i) {   // 8 RPW rows per pass: wave -> RPW whole rows of 4 WR bytes
        const int row = (i * 8 + wave) * RPW + lane / LPR, c = lane % LPR;
        if (row < ROWS && row < p.M) *(f32x4*)(sbp + (int64_t)row * p.N + c * 4) = *(const f32x4*)(smem + row * TP + c * 16);
      }
    } else {
      OutT* op = (OutT*)p.out + (int64_t)blockIdx.x * WR;
      constexpr int LPR = WR / 8, RPW = 64 / LPR;        // 8 columns per lane
#pragma unroll
      for (int i = 0; i < (ROWS + 8 * RPW - 1) / (8 * RPW); ++i) {   // wave -> RPW whole rows of 2 WR bytes
        const int row = (i * 8 + wave) * RPW + lane / LPR, c = lane % LPR;
        if (row < ROWS && row < p.M) {
          const f32x4 a = *(const f32x4*)(smem + row * TP + c * 32), b = *(const f32x4*)(smem + row * TP + c * 32 + 16);
          *(uint4*)(op + (int64_t)row * p.ldo + c * 8) =
              make_uint4(pack2<OutT>(a[0], a[1]), pack2<OutT>(a[2], a[3]), pack2<OutT>(b[0], b[1]), pack2<OutT>(b[2], b[3]));
        }
      }
    }
    return;
  }
  // ------------------------------------------------------------------ epilogue (consumer `wave`, tile tn: columns
  // col_of(32*wave + 16*tn + 4*q + r), rows 16*t + r16)
  if constexpr (EPI == 1) {
    f32x4* xch = (f32x4*)smem;                           // [2 up consumers][2][MT][64] f32x4
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      float sbv[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = min(col_of(wave * 32 + tn * 16 + 4 * q + r), p.N - 1);
        sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
        bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int64_t m = min((int64_t)t * 16 + r16, p.M - 1);
        const float sav = p.sa_row ? p.sa[m] : p.sa[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tn][t][r] = round_to<OutT>(acc[tn][t][r] * sav * sbv[r] + bv[r]);
        if (wave >= 2) xch[(((wave - 2) * 2 + tn) * MT + t) * 64 + lane] = acc[tn][t];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 2) return;
    const float qs = *epi.q_scale;
    const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int64_t c = (int64_t)blockIdx.x * 64 + wave * 32 + tn * 16 + 4 * q;     // output column (of Ihalf)
      if (c >= Ihalf) continue;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int64_t m = (int64_t)t * 16 + r16;
        if (m >= p.M) continue;
        const f32x4 u = xch[((wave * 2 + tn) * MT + t) * 64 + lane];
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = acc[tn][t][r];
          o[r] = round_to<OutT>(round_to<OutT>(silu_f32(g)) * u[r]);
          o[r] = fmaxf(fminf(o[r] * qinv, 448.0f), -448.0f);
        }
        uint32_t w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w, true);
        *(uint32_t*)(epi.q_out + m * Ihalf + c) = w;
      }
    }
    return;
  } else {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int64_t nb = col_of(wave * (16 * NT) + tn * 16 + 4 * q);
      if (nb >= p.N) continue;
      if (S > 1 || force_slab) {                         // fp32 partial tile -> slab[sp][m][n]
        float* sbp = slab + (int64_t)sp * p.M * p.N;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int64_t m = (int64_t)t * 16 + r16;
          if (m >= p.M) continue;
          float* o = sbp + m * p.N + nb;
          if (nb + 3 < p.N && (p.N & 3) == 0) {
            *(f32x4*)o = acc[tn][t];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (nb + r < p.N) o[r] = acc[tn][t][r];
          }
        }
        continue;
      }
      float sbv[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = min(nb + r, p.N - 1);
        sbv[r] = p.sb_row ? p.sb[n] : p.sb[0];
        bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int64_t m = (int64_t)t * 16 + r16;
        if (m >= p.M) continue;
        const float sav = p.sa_row ? p.sa[m] : p.sa[0];
        OutT* o = (OutT*)p.out + m * p.ldo + nb;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[tn][t][r] * sav * sbv[r] + bv[r];
        if (nb + 3 < p.N && (p.ldo & 3) == 0) {
          *(uint2*)o = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < p.N) o[r] = (OutT)v[r];
        }
      }
    }
  }
}
// LDS of the role kernel: its ring, and never less than the staged output tile (MT*16 rows of 528 B)
static size_t xw_lds(int mt, int r, int width = 128) {
  const size_t ring = (size_t)r * ((size_t)mt * 16 * 128 + (size_t)width * 128), tile = (size_t)mt * 16 * ((size_t)width * 4 + 16);
  return ring > tile ? ring : tile;
}
// LDS of the deep-ring kernel: 4 stages of (x [M x 128 B] + 8 x 2 KiB of weights); never below the EPI exchange buffer
static size_t xd_lds(int mt, int r) { return (size_t)r * ((size_t)mt * 16 * 128 + 8 * 16 * 128); }

// sum the S split-K slabs and apply the epilogue: one thread per 4 consecutive n
template <typename OutT>
__global__ __launch_bounds__(256) void fp8_gemm_reduce_kernel(const GemmParams p, const float* __restrict__ slab, int S) {
  const int64_t nq = (p.N + 3) / 4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= p.M * nq) return;
  const int64_t m = gid / nq, nb = (gid % nq) * 4;
  const bool vec = (nb + 3 < p.N) && ((p.N & 3) == 0);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) {
    const float* src = slab + ((int64_t)s * p.M + m) * p.N + nb;
    if (vec) {
      const f32x4 x = *(const f32x4*)src;
      v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
    } else {
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.N) v[r] += src[r];
    }
  }
  const float sav = p.sa_row ? p.sa[m] : p.sa[0];
  OutT* o = (OutT*)p.out + m * p.ldo + nb;
  for (int r = 0; r < 4; ++r) {
    const int64_t n = min(nb + r, p.N - 1);
    const float sbv = p.sb_row ? p.sb[n] : p.sb[0];
    const float bv = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
    v[r] = v[r] * sav * sbv + bv;
  }
  if (vec && (p.ldo & 3) == 0) {
    *(uint2*)o = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
  } else {
    for (int r = 0; r < 4; ++r)
      if (nb + r < p.N) o[r] = (OutT)v[r];
  }
}

static int gemm_rotate() {   // bit 0: k rotation (skinny kernel); bits 8..15: tile-kernel group size override (tuning);
                             // bits 16+: tile-kernel L2 prefetch distance in k-steps (0 = off)
  static const int r = mi_tune("MI_GEMM_ROTATE", 1) | ((mi_tune("MI_GEMM_TILE_GROUP_M", 0) & 0xff) << 8) |
                       (mi_tune("MI_GEMM_TILE_PF", 3) << 16);
  return r;
}
// waves per workgroup: 8 (two per SIMD: one wave's DMA issue overlaps the other's MFMAs) unless
// MI_GEMM_XS_NW=4 or the problem is too narrow to give 128-row blocks
static int xs_waves(int64_t N) {
  static const int big = mi_tune("MI_GEMM_XS_NW", 8);
  return (big == 8 && N >= 1024) ? 8 : 4;
}

// Split-K plan of the decode kernels: S slices of `ppw` 256-byte phases, and (role kernel only) the width of a
// workgroup's column block.  128 columns minimise what a CU ingests per output ((M + width) x k bytes at ~45 GB/s,
// tools/xd_stamps.py) but need the deepest split to fill the chip, and every slab byte is written (~4 TB/s as measured:
// 16.8 MB of o_proj slabs = 4.4 us of a 14.9-us kernel) and read again by the consumer kernel; 64 columns halve the
// split.  The cheaper of the two by that model: 64 for the 4096 x 4096 o_proj (S 8 -> 4), 128 for qkv / down / gate_up.
static void xs_plan(int64_t M, int64_t N, int64_t K, int* S, int* ppw, int* width = nullptr) {
  static const int target = mi_tune("MI_GEMM_XS_TARGET", 256);   // one round of workgroups on 256 CUs
  static const int roles = mi_tune("MI_GEMM_XW", 1), narrow = mi_tune("MI_GEMM_XW_NARROW", 1);
  const int64_t nph = cdiv64(K / 128, 2);
  auto plan = [&](int64_t w, int* s_out, int* per_out) -> double {
    const int64_t nblk = cdiv64(N, w);
    int64_t want = nblk >= 200 ? 1 : target / nblk;
    if (want < 1) want = 1;
    if (want > nph) want = nph;
    const int64_t per = cdiv64(nph, want), s = cdiv64(nph, per);
    *s_out = (int)s;
    *per_out = (int)per;
    const double ingest_us = (double)(128 + w) * (double)(per * 256) / 45e3;
    const double slab_us = s > 1 ? 2.0 * (double)s * 128.0 * (double)N * 4.0 / 4.5e6 : 0.0;
    const double rounds = (double)cdiv64(nblk * s, 256);
    return rounds * ingest_us + slab_us;
  };
  int s128, per128;
  const double c128 = plan(16 * xs_waves(N), &s128, &per128);
  *S = s128; *ppw = per128;
  if (width) *width = 128;
  if (width && roles && narrow && xs_waves(N) == 8 && M <= 128 && N % 64 == 0) {
    int s64, per64;
    const double c64 = plan(64, &s64, &per64);
    if (c64 < 0.9 * c128) { *S = s64; *ppw = per64; *width = 64; }
  }
}

// LDS of the split ring (8 waves): 2 x stages + 3 weight stages = 160 KiB at M = 128 (all of a gfx950 CU's LDS)
static size_t xs_split_lds(int mt) { return 2 * (size_t)mt * 16 * 256 + 3 * (size_t)8 * 16 * 256; }

template <typename OutT, int MT>
static void launch_xs(const GemmParams& p, float* slab, int S, int ppw, hipStream_t st, bool partial = false, int width = 128) {
  const int nw = xs_waves(p.N);
  [[maybe_unused]] static const int nst4 = mi_tune("MI_GEMM_XS_STAGES", 3);
  const size_t stage = (size_t)(MT * 16 + nw * 16) * 256;
  dim3 grid((unsigned)cdiv64(p.N, width == 64 ? 64 : 16 * nw), (unsigned)S);
  const int fs = partial ? 1 : 0;
  [[maybe_unused]] static const int split = mi_tune("MI_GEMM_XS_SPLIT", 1);
  [[maybe_unused]] static const int deep = mi_tune("MI_GEMM_XD", 1);
  static const int roles = mi_tune("MI_GEMM_XW", 1);    // 0: fp8_gemm_xd_kernel; bit 1 set: per-lane epilogue
  if constexpr (MT == 16) {
    // 129..256 rows in one pass over the weights: 48-KiB stages (x 32 KiB + weights 16 KiB), ring of 3
    fp8_gemm_xd_kernel<OutT, 16, 0, 3><<<grid, 512, xd_lds(16, 3), st>>>(p, slab, S, 2 * ppw, fs, SiluEpi{nullptr, nullptr});
  } else
  if (nw == 8 && roles) {
    // whole-row epilogue: every block column exists and the rows of the destination take 16-byte stores
    const bool to_slab = S > 1 || partial;
    const int staged = (roles & 2) == 0 && p.N % width == 0 &&
                       (to_slab ? ((uintptr_t)slab & 15) == 0 : (p.ldo % 8 == 0 && ((uintptr_t)p.out & 15) == 0));
    if (width == 64) fp8_gemm_xw_kernel<OutT, MT, 0, 4, 1><<<grid, 512, xw_lds(MT, 4, 64), st>>>(p, slab, S, 2 * ppw, fs, SiluEpi{nullptr, nullptr}, staged);
    else fp8_gemm_xw_kernel<OutT, MT, 0, 4><<<grid, 512, xw_lds(MT, 4), st>>>(p, slab, S, 2 * ppw, fs, SiluEpi{nullptr, nullptr}, staged);
  }
#ifdef MI_TUNING   // the kernels the role kernel replaced, selectable in tuning builds only (MI_GEMM_XW=0 ...)
  else if (nw == 8 && deep == 5) fp8_gemm_xd_kernel<OutT, MT, 0, 5><<<grid, 512, xd_lds(MT, 5), st>>>(p, slab, S, 2 * ppw, fs, SiluEpi{nullptr, nullptr});
  else if (nw == 8 && deep) fp8_gemm_xd_kernel<OutT, MT, 0, 4><<<grid, 512, xd_lds(MT, 4), st>>>(p, slab, S, 2 * ppw, fs, SiluEpi{nullptr, nullptr});
  else if (nw == 8 && split) fp8_gemm_xs_kernel<OutT, MT, 8, 5><<<grid, 512, xs_split_lds(MT), st>>>(p, slab, S, ppw, fs);
  else if (nw == 8) fp8_gemm_xs_kernel<OutT, MT, 8, 2><<<grid, 512, 2 * stage, st>>>(p, slab, S, ppw, fs);
  else if (nst4 != 3) fp8_gemm_xs_kernel<OutT, MT, 4, 2><<<grid, 256, 2 * stage, st>>>(p, slab, S, ppw, fs);
#endif
  else fp8_gemm_xs_kernel<OutT, MT, 4, 3><<<grid, 256, 3 * stage, st>>>(p, slab, S, ppw, fs);   // N < 1024: 4-wave blocks
  if (S > 1 && !partial) {
    const int64_t total = p.M * cdiv64(p.N, 4);
    fp8_gemm_reduce_kernel<OutT><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(p, slab, S);
  }
}

// Decode batches beyond 128 rows (graph batch sizes up to 512, C5's batch 256): the 256 x 256 tile kernel would put
// cdiv(M,256) * cdiv(N,256) workgroups on the chip -- 16 for a 4096-wide o/down projection at M = 256, each
// streaming (256 + 256) x K bytes through ONE CU's ~42 GB/s LDS-DMA path (measured: 199 us for 256 x 4096 x 14336,
// against 24 us at M = 128).  Until the tile grid fills the chip it is cheaper to run the x-stationary decode
// kernel once per 128-row chunk: every chunk streams the weights again (the second pass mostly from the 256-MB
// Infinity Cache), all 256 CUs busy each time.  Cost model in us, from the measured rates.
// Decode batches beyond 128 rows (graph batch sizes up to 512, C5's batch 256): the 256 x 256 tile kernel would put
// cdiv(M,256) * cdiv(N,256) workgroups on the chip -- 16 for a 4096-wide o/down projection at M = 256, each
// streaming (256 + 256) x K bytes through ONE CU's ~42 GB/s LDS-DMA path (measured: 199 us for 256 x 4096 x 14336,
// against 24 us at M = 128).  Until the tile grid fills the chip it is cheaper to run the x-stationary decode
// kernel once per chunk of rows, all 256 CUs busy each time.  Cost model in us, from the measured rates.
static double tile_time(int64_t M, int64_t N, int64_t K, int* S_out);
// rows one pass of the decode kernels takes: 256 with the 8-wave deep-ring kernel (MT = 16), else 128
MI_INTERNAL int64_t mi_fp8_gemm_partial_max_rows(int64_t N) {
  [[maybe_unused]] static const int deep = mi_tune("MI_GEMM_XD", 1), wide = mi_tune("MI_GEMM_XD16", 1);
  return (xs_waves(N) == 8 && deep && wide) ? 256 : 128;
}
static bool mid_m_chunked(int64_t M, int64_t N, int64_t K) {
  if (M <= 128 || M > 1024 || K % 128 != 0) return false;
  const int64_t rows = mi_fp8_gemm_partial_max_rows(N);
  const double t_tile = tile_time(M, N, K, nullptr);   // with its own split-K where that pays
  const double t_chunk = (double)cdiv64(M, rows) * ((double)N * (double)K / 3.5e6 + 10.0) * (rows == 256 ? 1.5 : 1.0);
  return t_chunk < t_tile;
}

static int tile_splits(int64_t M, int64_t N, int64_t K);
extern "C" int64_t mi_fp8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || K % 128 != 0) return 0;
  if (M > 128 && !mid_m_chunked(M, N, K)) {
    const int S = tile_splits(M, N, K);
    return S > 1 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
  }
  if (M > mi_fp8_gemm_partial_max_rows(N)) M = mi_fp8_gemm_partial_max_rows(N);   // chunks reuse the same slabs
  int S, spw, width;
  xs_plan(M, N, K, &S, &spw, &width);
  return S > 1 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
}


// ---------------------------------------------------------------------------------------------
// Prefill-shaped GEMM (M > 128): 256 x 256 output tile per 8-wave workgroup, K in steps of 128 B.
//   * both operands are staged global -> LDS by LDS-DMA (64 KiB per k-step, two buffers = 128 KiB);
//     a 1-KiB DMA piece = 8 rows x 128 B, the LDS image is lane-linear; two consecutive tile rows
//     share one 256-B LDS line and the 16-byte position inside the line pair is XOR-swizzled with
//     (line & 15) on the SOURCE side, which makes every ds_read_b128 fragment read conflict-free;
//   * wave (wm, wn) of the 2 x 4 wave grid owns 128 (m) x 64 (n): 8 x 4 MFMA tiles, weights as the
//     A operand, activations as B (each lane ends with 4 consecutive n of one output row);
//   * v_mfma_scale_f32_16x16x128_f8f6f4, unit scales: one MFMA per (m-tile, n-tile, k-step);
//   * one DMA-wait + barrier per k-step (stage k+1 is issued before computing k);
//   * blocks are renumbered so that the 8 XCDs each walk a contiguous range of tiles, in groups of
//     8 m-blocks x all n: the ~32 tiles resident on one XCD form an 8 x 4 patch and share BOTH operands
//     in that XCD's L2 (TCC hit rate 76-81 % vs 49-72 % with m fastest over all of M).
// Bound (measured, gate_up shape 16384 x 28672 x 4096: 2.27 ms = 1.70 PFLOP/s): the stage-ahead DMA.  With the
// MFMAs removed the kernel still takes 1.94 ms: one 64-KiB stage per CU takes ~2.2 us to land under full
// load (L2 hits + 24 % from the Infinity Cache), the MMA of a k-step ~0.85 us, and 160 KiB of LDS hold only
// two 64-KiB stages -- one stage in flight.  Next: finer ring units (quarter stages) or a 4-wave x 2 form.
// EPI = 1 (gate_up of a gated MLP, N = 2*I, I % 128 == 0, no split-K): the tile's 256 weight rows are 128 GATE rows
// n0.. and the 128 UP rows I + n0.. of the same output columns; waves wn = 0,1 end with gate values, wn = 2,3 with the
// up values of the same (row, column) in the same registers, exchanged through the (free) stage buffers: the
// epilogue writes fp8(silu(gate) * up) [M, I] directly -- bit-identical to the GEMM followed by mi_silu_and_mul_fp8,
// without writing and re-reading the [M, 2I] intermediate (0.94 GB per gate_up at 16 k tokens).
// 16-byte streaming (non-temporal) store: the output tile is not read again by this kernel, and written the ordinary way
// it displaces the operand slabs the other workgroups of the XCD are about to share in L2 (probe, 16384 x 28672 x 4096,
// persistent loop: 1.57 ms with plain stores, 1.23 ms non-temporal, 1.10 ms with no output at all)
__device__ __forceinline__ void stream_store16(void* dst, uint4 v) {
  typedef __attribute__((ext_vector_type(4))) uint32_t st_u32x4;
  __builtin_nontemporal_store(st_u32x4{v.x, v.y, v.z, v.w}, (st_u32x4*)dst);
}

template <typename OutT, int EPI = 0>
__global__ __launch_bounds__(512) void fp8_gemm_tile_kernel(const GemmParams p, int mblocks, int nblocks,
                                                            float* __restrict__ slab, int S,
                                                            const SiluEpi epi = SiluEpi{nullptr, nullptr}) {
  constexpr int BM = 256, BN = 256, BK = 128;
  constexpr int TILE = BM * BK;            // 32 KiB per operand per stage
  // SPLIT ring over all 160 KiB of LDS: two activation stages (requested one k-step ahead) at [0, 64 KiB), THREE weight
  // stages (requested TWO k-steps ahead) behind them.  The weights are what comes from HBM; the activation block is
  // re-read by every n-block and sits in L2 / the Infinity Cache.  With one stage ahead for both, every k-step exposed
  // the HBM latency of its weight lines: 1.50 ms for 16384 x 28672 x 4096 with fresh weights against 1.21 ms with the
  // weights left in the Infinity Cache by the previous launch (probe tools/src/lds_fill.hip); with the weights two
  // k-steps ahead the same probe runs fresh weights in 1.23 ms (and 16384 x 4096 x 14336 in 0.57 instead of 0.72 ms).
#define TL_XST(s_) ((uint32_t)((s_) & 1) * (uint32_t)TILE)
#define TL_WST(s_) ((uint32_t)(2 * TILE) + (uint32_t)((s_) % 3) * (uint32_t)TILE)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // PERSISTENT over tiles (grid.x = min(tiles, CUs) when S == 1; with split-K every workgroup has one item).
  // XCD-aware order: hardware deals workgroup ids round-robin to the 8 XCDs, so XCD x owns a contiguous range of the
  // tile list and its (up to 32) workgroups walk that range together, 32 tiles at a time.  Inside the range the tiles
  // are grouped GM m-blocks x all n: the ~32 tiles in flight on an XCD form an 8 x 4 patch and share BOTH operands
  // in that XCD's L2 (12 operand blocks per 32 tiles instead of 33 with m fastest over all of M).
  const int nwg = mblocks * nblocks;
  const int orig = blockIdx.x, G = gridDim.x;
  const int xcd = orig & 7, slot = orig >> 3, gx = (G - xcd + 7) >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int xstart = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd;
  const int xcnt = qd + (xcd < rm ? 1 : 0);
  const int GM = (p.rotate >> 8) & 0xff ? (p.rotate >> 8) & 0xff : 8;
  const int per_group = GM * nblocks;
  const int64_t Ihalf = p.N / 2;   // EPI only
  // split-K (grid.y = S > 1): this workgroup walks k-steps [kt0, KT) of its tile and leaves raw fp32 partials in
  // slab[blockIdx.y]; fp8_gemm_reduce_kernel sums them and applies the epilogue.  For grids that leave most CUs idle
  // (1-2 k tokens on a 4096-wide, K = 14336 down projection: 64-128 tiles, each a 112-step serial walk)
  const int64_t KT_all = p.K / BK;
  const int64_t kt_per = (KT_all + S - 1) / S;
  const int64_t kt0 = (int64_t)blockIdx.y * kt_per;
  const int64_t KT = min(KT_all, kt0 + kt_per);
  if (kt0 >= KT) return;        // (host never launches an empty split)

  // DMA geometry: piece i (0..31 per operand) covers tile rows 8i..8i+7; lane L -> LDS byte i*1024 + L*16.
  // LDS line pair l = row>>1, position P = L & 15 within it holds logical (rowbit, slot) = P ^ (l & 15).
  // Wave w issues pieces 4w .. 4w+3 of both operands: a scalar base per operand (block start + k offset) and one
  // 32-bit lane offset per piece, no vector address arithmetic in the loop.
  const int dline = lane >> 4;             // 0..3: which 256-B line of the piece
  const int dpos = lane & 15;
  const uint32_t lds_base = lds_addr_of(smem);
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_base + wave * 4 * 1024);
  int prow[4], pslot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int line = (wave * 4 + i) * 4 + dline;
    const int logical = dpos ^ (line & 15);
    prow[i] = line * 2 + (logical >> 3);
    pslot[i] = (logical & 7) * 16;
  }
  // per tile only scalars: block bases and the last valid row of each operand block (rows past the matrix edge are
  // clamped onto it); the lane offsets are rebuilt from (prow, pslot) where they are used -- two VALU ops per piece
  struct TileAddr {
    int64_t m0, n0;
    const uint8_t* xblk;
    const uint8_t* wblk;
    int xclamp, wclamp;
  };
  auto tile_addr = [&](int li, TileAddr& t) __attribute__((always_inline)) {
    const int tid = xstart + li;
    const int grp = tid / per_group, in_grp = tid % per_group;
    const int first_m = grp * GM;
    const int gsz = min(mblocks - first_m, GM);
    const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
    t.m0 = (int64_t)mb * BM;
    t.n0 = (int64_t)nb * (EPI ? BN / 2 : BN);
    const int64_t w0 = EPI ? (wave < 4 ? t.n0 : Ihalf + t.n0 - 128) : t.n0;     // EPI: waves 4..7 stage the UP rows
    t.xblk = uniform_ptr(p.a + t.m0 * p.lda);
    t.wblk = uniform_ptr(p.b + w0 * p.ldb);
    t.xclamp = __builtin_amdgcn_readfirstlane((int)min((int64_t)255, p.M - 1 - t.m0));
    t.wclamp = __builtin_amdgcn_readfirstlane((int)min((int64_t)255, p.N - 1 - w0));
  };
  const uint32_t lda32 = (uint32_t)p.lda, ldb32 = (uint32_t)p.ldb;      // < 2^24 (host check): 24-bit multiplies
#define TL_XOFF(t_, i_) (__umul24((uint32_t)min(prow[i_], (t_).xclamp), lda32) + (uint32_t)pslot[i_])
#define TL_WOFF(t_, i_) (__umul24((uint32_t)min(prow[i_], (t_).wclamp), ldb32) + (uint32_t)pslot[i_])
  // fragment address of (tile row, 16-byte slot): line pair = row>>1, physical pos = ((row&1)*8+slot) ^ (line&15)
#define TL_FRAG(base_, row_, slot_) \
  (*(const uint4*)((base_) + ((row_) >> 1) * 256 + (((((row_) & 1) << 3) | (slot_)) ^ (((row_) >> 1) & 15)) * 16))

  const int PF = p.rotate >> 16;
  uint32_t pf_sink = 0;
  const float sa0 = p.sa ? p.sa[0] : 1.f, sb0 = p.sb ? p.sb[0] : 1.f;     // per-tensor scales: read once, not per tile
  int step = 0;                 // k-steps done by this workgroup: the stage buffer of a k-step is step & 1
  bool first_issued = false;    // the current tile's first stage was requested during the previous tile's last k-step
  TileAddr cur, nxt;
  if (slot < xcnt) tile_addr(slot, cur);
  for (int li = slot; li < xcnt; li += gx) {
#ifdef MI_TUNING
  const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
#endif
  const bool more = li + gx < xcnt;
  // the next tile's first stage lands during this tile's epilogue (which only uses the other stage buffer); the last
  // tile's last k-step requests its own first stage again (never read): the loop body stays one basic block
  const bool ahead = more && KT - kt0 >= 2;     // (a one-step k range cannot request two steps ahead across tiles)
  if (more) tile_addr(li + gx, nxt);
  else nxt = cur;
  const int64_t m0 = cur.m0, n0 = cur.n0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // L2 prefetch: a stage completes when its SLOWEST line is back (the DMA pieces of a wave retire in order), and ~20 %
  // of the lines of every stage miss the XCD's L2.  Each wave therefore touches, PF k-steps ahead, 64 of the tile's 512
  // operand rows with ONE global_load_dword (one lane = one 128-byte line; waves 0..3 activation rows, 4..7 weight
  // rows); the value is never read.  It is issued AFTER the stage's DMA, so the counted wait at the end of the k-step
  // leaves exactly this one load outstanding and only forces the previous one (a full k-step older).
  const uint8_t* pf_base;
  uint32_t pf_off;
  {
    const int r = (wave & 3) * 64 + lane;
    if (wave < 4) {
      pf_base = cur.xblk;
      pf_off = (uint32_t)((min(m0 + r, p.M - 1) - m0) * p.lda);
    } else {
      const int64_t pw0 = EPI ? (r < 128 ? n0 : Ihalf + n0 - 128) : n0;   // uniform per wave (64-row groups)
      pf_base = p.b + pw0 * p.ldb;
      pf_off = (uint32_t)((min(pw0 + r, p.N - 1) - pw0) * p.ldb);
    }
    pf_base = uniform_ptr(pf_base);
  }
  if (!first_issued) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_s(TL_XOFF(cur, i), cur.xblk + kt0 * BK, lds_piece + TL_XST(step) + i * 1024);
      glds16_s(TL_WOFF(cur, i), cur.wblk + kt0 * BK, lds_piece + TL_WST(step) + i * 1024);
      glds16_s(TL_WOFF(cur, i), cur.wblk + min(kt0 + 1, KT - 1) * BK, lds_piece + TL_WST(step + 1) + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#ifdef MI_TUNING
  unsigned long long ks_issue = 0, ks_vm = 0, ks_bar = 0;
#endif
  for (int64_t kt = kt0; kt < KT; ++kt, ++step) {
#ifdef MI_TUNING
    const unsigned long long ks0 = __builtin_amdgcn_s_memtime();
#endif
    // the requests are placed piece by piece BETWEEN the first MFMA groups: both waves of a SIMD leave the barrier
    // together, and with all eight requests up front they spent ~0.3 us issuing DMAs side by side before the first
    // MFMA -- now one wave's requests run under the other's (and its own) MFMAs.  A wave requests its 4 activation
    // pieces of k-step kt + 1 FIRST, then its 4 weight pieces of k-step kt + 2: the wait at the end of the k-step
    // leaves those 4 (and the prefetch) in flight.  Past the end of the tile the requests go to the next tile's first
    // k-steps (`ahead`), or repeat this tile's (never read): the loop body stays one basic block.
    const bool xnext = kt + 1 >= KT, wnext = kt + 2 >= KT;
    const int64_t kx = xnext ? kt0 : kt + 1;
    const int64_t kw = wnext ? min(kt0 + (kt + 2 - KT), KT - 1) : kt + 2;
    const uint8_t* xs = ((xnext && ahead) ? nxt.xblk : cur.xblk) + kx * BK;
    const uint8_t* ws = ((wnext && ahead) ? nxt.wblk : cur.wblk) + kw * BK;
    const int xcl = (xnext && ahead) ? nxt.xclamp : cur.xclamp, wcl = (wnext && ahead) ? nxt.wclamp : cur.wclamp;
    const uint32_t xdst = lds_piece + TL_XST(step + 1), wdst = lds_piece + TL_WST(step + 2);
    const char* xb = smem + TL_XST(step);
    const char* wb = smem + TL_WST(step);
    i32x8 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wn * 64 + j * 16 + r16;
      const uint4 a0 = TL_FRAG(wb, row, q), a1 = TL_FRAG(wb, row, 4 + q);
      wf[j] = i32x8{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    }
    // activation fragments one m-tile AHEAD in a second register set: read right after the MFMAs of m-tile i were
    // issued and waited for in front of the MFMAs of i + 1, a fragment had 128 cycles (4 MFMAs) to arrive -- with the
    // CU's LDS half busy (192 KiB of fragment reads per k-step) it took longer, ~100 exposed cycles per m-tile
    // (tools/tile_stamps.py: 3160 cycles per k-step for 2048 of MFMA)
    i32x8 xfb[2];
    {
      const int row = wm * 128 + r16;
      const uint4 b0 = TL_FRAG(xb, row, q), b1 = TL_FRAG(xb, row, 4 + q);
      xfb[0] = i32x8{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) {
        const int row = wm * 128 + (i + 1) * 16 + r16;
        const uint4 b0 = TL_FRAG(xb, row, q), b1 = TL_FRAG(xb, row, 4 + q);
        xfb[(i + 1) & 1] = i32x8{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
      }
      __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the read back behind this m-tile's MFMAs)
      const i32x8 xf = xfb[i & 1];
      if (i < 2) {
        glds16_s(__umul24((uint32_t)min(prow[2 * i], xcl), lda32) + (uint32_t)pslot[2 * i], xs, xdst + (2 * i) * 1024);
        glds16_s(__umul24((uint32_t)min(prow[2 * i + 1], xcl), lda32) + (uint32_t)pslot[2 * i + 1], xs, xdst + (2 * i + 1) * 1024);
      } else if (i < 4) {
        glds16_s(__umul24((uint32_t)min(prow[2 * i - 4], wcl), ldb32) + (uint32_t)pslot[2 * i - 4], ws, wdst + (2 * i - 4) * 1024);
        glds16_s(__umul24((uint32_t)min(prow[2 * i - 3], wcl), ldb32) + (uint32_t)pslot[2 * i - 3], ws, wdst + (2 * i - 3) * 1024);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], xf, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      if (i == 3)   // `pf_sink` is tied in and out: its register stays reserved for the whole loop (the load lands late)
        asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(pf_off), "s"(pf_base + min(kt + PF, KT - 1) * BK) : "memory");
    }
#ifdef MI_TUNING
    const unsigned long long ks1 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // all but the 4 weight pieces and the prefetch just issued
#ifdef MI_TUNING
    const unsigned long long ks2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef MI_TUNING
    { const unsigned long long ks3 = __builtin_amdgcn_s_memtime(); ks_issue += ks1 - ks0; ks_vm += ks2 - ks1; ks_bar += ks3 - ks2; }
#endif
  }
#ifdef MI_TUNING
  if (lane == 0 && blockIdx.x < 512) {     // per k-step: reads + MFMA issue / DMA wait / barrier, and the k-steps counted
    unsigned long long* o = mi_xd_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
    o[4] += ks_issue; o[5] += ks_vm; o[6] += ks_bar; o[7] += (unsigned long long)(KT - kt0);
  }
#endif
  first_issued = ahead;
#ifdef MI_TUNING
  const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
#endif
  // the activation stage and the weight stage of the last k-step are free for the epilogue (32 KiB each; the others
  // hold or are receiving the next tile's first k-steps): waves 0..3 use the one, waves 4..7 the other
  char* const free_x = smem + TL_XST(step - 1);
  char* const free_w = smem + TL_WST(step - 1);

  // ---- epilogue: lane holds out[m = m0 + wm*128 + 16i + r16][n = n0 + wn*64 + 16j + 4q + r]
  if constexpr (EPI == 1) {
    // Waves (wm, c) [gate] and (wm, c + 2) [up] hold the two factors of the same 128 x 64 outputs in the same
    // registers.  Each finishes HALF of them: the gate wave rows i = 0..3 (it needs the partner's up values), the up
    // wave rows i = 4..7 (it needs the partner's gate values) -- all 8 waves share the SiLU arithmetic.  The values
    // travel through the free 64-KiB stage buffer in two rounds of 8 KiB per wave (2 i x 4 j fragments), so the other
    // buffer can take the next tile's first stage meanwhile.  Bits as mi_silu_and_mul_fp8 on the unfused GEMM.
    const bool up = wn >= 2;
    const int c64 = (wn & 1) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + c64 + j * 16 + 4 * q;                    // output column (0 .. I)
      const int64_t nsrc = (up ? Ihalf : 0) + col;                      // weight row the scale belongs to
      float sbv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sbv[r] = p.sb_row ? p.sb[min(nsrc + r, p.N - 1)] : sb0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = min(m0 + wm * 128 + i * 16 + r16, p.M - 1);
        const float sav = p.sa_row ? p.sa[m] : sa0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = round_to<OutT>(acc[i][j][r] * sav * sbv[r]);
      }
    }
    const float qs = *epi.q_scale;
    const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
    f32x4* mine = (f32x4*)((wave < 4 ? free_x : free_w) + (wave & 3) * 8192);           // [8 fragments][64 lanes]
    const f32x4* theirs = (const f32x4*)((wave < 4 ? free_x : free_w) + ((wave ^ 2) & 3) * 8192);
    const int ib = up ? 4 : 0;                                // the rows this wave finishes: i = ib .. ib + 3
    uint32_t wq[4][4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      // publish what the partner finishes in this round: its rows are (4 - ib) + 2 rd + {0, 1}
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (4 - ib) + 2 rd + ii with a compile-time index on both sides of the select
          const f32x4 v = up ? acc[2 * rd + ii][j] : acc[4 + 2 * rd + ii][j];
          mine[(ii * 4 + j) * 64 + lane] = v;
        }
      __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 o2 = theirs[(ii * 4 + j) * 64 + lane];
          const f32x4 own = up ? acc[4 + 2 * rd + ii][j] : acc[2 * rd + ii][j];
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = up ? o2[r] : own[r], u = up ? own[r] : o2[r];
            o[r] = round_to<OutT>(round_to<OutT>(silu_f32(g)) * u);
            o[r] = fmaxf(fminf(o[r] * qinv, 448.0f), -448.0f);
          }
          uint32_t w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w, true);
          wq[2 * rd + ii][j] = w;
        }
      __syncthreads();
    }
    if (n0 + 128 <= Ihalf && (Ihalf & 15) == 0 && ((uintptr_t)epi.q_out & 15) == 0) {
      // full tiles leave as whole 128-byte lines (the tile's 128 fp8 columns of a row), staged per 128-row half:
      // [128 rows][128 B], 16-byte chunk c of row R at position c ^ (R & 7).  Straight from the MFMA layout it was
      // 4-byte stores, 16 rows x 16 B per instruction.
      char* stg = wm == 0 ? free_x : free_w;        // 16 KiB per 128-row half
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = (ib + ii) * 16 + r16;
          *(uint32_t*)(stg + row * 128 + ((((wn & 1) * 4 + j) ^ (row & 7)) << 4) + q * 4) = wq[ii][j];
        }
      __syncthreads();
      int lane_e = lane;                     // (opaque: see the plain epilogue below)
      asm volatile("" : "+v"(lane_e));
      const int pr = lane_e >> 3, pc = lane_e & 7;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {       // the four waves of this half take 32 rows each
        const int row = wn * 32 + ps * 8 + pr;
        const uint4 v = *(const uint4*)(stg + row * 128 + ((pc ^ (row & 7)) << 4));
        const int64_t m = m0 + wm * 128 + row;
        if (m < p.M) stream_store16(epi.q_out + m * Ihalf + n0 + pc * 16, v);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + c64 + j * 16 + 4 * q;
        if (col >= Ihalf) continue;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int64_t m = m0 + wm * 128 + (ib + ii) * 16 + r16;
          if (m >= p.M) continue;
          *(uint32_t*)(epi.q_out + m * Ihalf + col) = wq[ii][j];
        }
      }
    }
  } else if (S > 1) {
    float* sb = slab + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t nbase = n0 + wn * 64 + j * 16 + 4 * q;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + wm * 128 + i * 16 + r16;
        if (m >= p.M) continue;
        float* o = sb + m * p.N + nbase;
        if (nbase + 3 < p.N && (p.N & 3) == 0) {
          *(f32x4*)o = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nbase + r < p.N) o[r] = acc[i][j][r];
        }
      }
    }
  } else if (n0 + BN <= p.N && (p.ldo & 7) == 0 && ((uintptr_t)p.out & 15) == 0) {
    // Full tiles: the wave's 128 x 64 outputs go through 8 KiB of the free stage buffer, 64 rows at a time, and leave
    // as whole 128-byte lines, 8 rows per (non-temporal) store instruction; the stores drain under the next tile's
    // first k-steps.  Straight from the MFMA layout a lane owns 4 consecutive n of one row -- 8-byte stores, 16 rows x
    // 32 B per instruction, and with one workgroup per tile every epilogue also paid a workgroup launch and an exposed
    // first stage: 0.74 of the 1.95 ms of a 16384 x 28672 x 4096 GEMM (probe: tools/src/lds_fill.hip; k-loop alone
    // 1.21 ms, persistent with these stores 1.23 ms).  LDS image: [64 rows][128 B], 16-byte chunk c of row R at
    // position c ^ (R & 7) (the 8-byte writes of the MFMA layout then spread over all banks).
    char* stg = (wave < 4 ? free_x : free_w) + (wave & 3) * 8192;
    // (an opaque copy of the lane id: the lane-constant parts of the store addresses below are then computed here
    // instead of being hoisted to kernel entry, spilled around the k-loop and reloaded from scratch per tile)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int pr = lane_e >> 3, pc = lane_e & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float sav[4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
        sav[ii] = p.sa_row ? p.sa[min(m0 + wm * 128 + (h * 4 + ii) * 16 + r16, p.M - 1)] : sa0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t nbase = n0 + wn * 64 + j * 16 + 4 * q;
        float sbv[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sbv[r] = p.sb_row ? p.sb[nbase + r] : sb0;
          bv[r] = p.bias ? (float)((const OutT*)p.bias)[nbase + r] : 0.f;
        }
        const int c = 2 * j + (q >> 1);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = h * 4 + ii, row = ii * 16 + r16;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * sav[ii] * sbv[r] + bv[r];
          *(uint2*)(stg + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
        }
      }
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 8 + pr;
        const uint4 v = *(const uint4*)(stg + row * 128 + ((pc ^ (row & 7)) << 4));
        const int64_t m = m0 + wm * 128 + h * 64 + row;
        if (m < p.M) stream_store16((OutT*)p.out + m * p.ldo + n0 + wn * 64 + pc * 8, v);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t nbase = n0 + wn * 64 + j * 16 + 4 * q;
      float sbv[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = min(nbase + r, p.N - 1);
        sbv[r] = p.sb_row ? p.sb[n] : sb0;
        bv[r] = p.bias ? (float)((const OutT*)p.bias)[n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + wm * 128 + i * 16 + r16;
        if (m >= p.M) continue;
        const float sav = p.sa_row ? p.sa[m] : sa0;
        OutT* o = (OutT*)p.out + m * p.ldo + nbase;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * sav * sbv[r] + bv[r];
        if (nbase + 3 < p.N && (p.ldo & 3) == 0) {
          *(uint2*)o = make_uint2(pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nbase + r < p.N) o[r] = (OutT)v[r];
        }
      }
    }
  }
  // every wave is done with the LDS of this tile (staging reads included) before the next tile requests into it
#ifdef MI_TUNING
  const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef MI_TUNING
  if (lane == 0 && blockIdx.x < 512) {     // tools/tile_stamps.py: k-loop / epilogue / end barrier ticks per wave, tiles
    unsigned long long* o = mi_xd_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
    o[0] += tt1 - tt0; o[1] += tt2 - tt1; o[2] += __builtin_amdgcn_s_memtime() - tt2; o[3] += 1;
  }
#endif
  cur = nxt;
  }   // tiles of this workgroup
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink)::"memory");   // the last prefetch has landed: its register is free again
#undef TL_FRAG
#undef TL_XOFF
#undef TL_WOFF
#undef TL_XST
#undef TL_WST
}


// split-K factor of the tile kernel, from the measured rates (us): a k-step of one workgroup costs ~1.3 us of LDS-DMA
// ingest whatever else runs, the slabs cost a write and a read of S * M * N fp32 at ~4 TB/s
static double tile_time(int64_t M, int64_t N, int64_t K, int* S_out) {
  static const int enable = mi_tune("MI_GEMM_TILE_SPLITK", 1);
  const int64_t tiles = cdiv64(M, 256) * cdiv64(N, 256), steps = K / 128;
  int best = 1;
  double best_t = (double)cdiv64(tiles, 256) * (double)steps * 1.3 + 5.0;
  if (enable && tiles < 192 && steps >= 32) {
    for (int s = 2; s <= 4 && tiles * s <= 256 && steps / s >= 16; ++s) {
      const double t = (double)cdiv64(steps, s) * 1.3 + (double)s * (double)M * (double)N * 8.0 / 4e6 + 10.0;
      if (t < best_t) { best_t = t; best = s; }
    }
  }
  if (S_out) *S_out = best;
  return best_t;
}
static int tile_splits(int64_t M, int64_t N, int64_t K) {
  int S;
  tile_time(M, N, K, &S);
  return S;
}

// workgroups of a persistent tile launch: one per CU (LDS and VGPRs allow one 8-wave workgroup per CU)
static int tile_grid(int tiles) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  static const int persist = mi_tune("MI_GEMM_TILE_PERSIST", 1);
  return persist && tiles > cus ? cus : tiles;
}

template <typename OutT> static void launch_tile(const GemmParams& p, hipStream_t st, void* workspace, int64_t workspace_bytes) {
  const int mblocks = (int)cdiv64(p.M, 256), nblocks = (int)cdiv64(p.N, 256);
  int S = tile_splits(p.M, p.N, p.K);
  if (S > 1 && (!workspace || workspace_bytes < (int64_t)S * p.M * p.N * (int64_t)sizeof(float))) S = 1;
  // S == 1: persistent, one workgroup per CU walks its share of the tiles; split-K: one workgroup per (tile, split)
  const unsigned gx = S == 1 ? (unsigned)tile_grid(mblocks * nblocks) : (unsigned)(mblocks * nblocks);
  fp8_gemm_tile_kernel<OutT><<<dim3(gx, (unsigned)S), 512, 5 * 256 * 128, st>>>(p, mblocks, nblocks, (float*)workspace, S);
  if (S > 1) {
    const int64_t total = p.M * cdiv64(p.N, 4);
    fp8_gemm_reduce_kernel<OutT><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(p, (const float*)workspace, S);
  }
}

template <typename OutT>
static void launch_fp8_gemm(const GemmParams& p, hipStream_t st, void* workspace, int64_t workspace_bytes) {
  const int64_t pass_rows = p.M > 128 && p.M <= 256 && p.K % 128 == 0 ? mi_fp8_gemm_partial_max_rows(p.N) : 128;
  if (p.M <= pass_rows && p.K % 128 == 0 && (p.M <= 128 || mid_m_chunked(p.M, p.N, p.K))) {
    // decode shapes: x-stationary, weights streamed once
    int S, spw, width;
    xs_plan(p.M, p.N, p.K, &S, &spw, &width);
    const int64_t need = S > 1 ? (int64_t)S * p.M * p.N * (int64_t)sizeof(float) : 0;
    if (need > workspace_bytes || (need > 0 && workspace == nullptr)) {  // no room for slabs: no split-K
      S = 1;
      spw = (int)cdiv64(p.K / 128, 2);
      width = 128;
    }
    float* slab = (float*)workspace;
    if (p.M <= 16) launch_xs<OutT, 1>(p, slab, S, spw, st, false, width);
    else if (p.M <= 32) launch_xs<OutT, 2>(p, slab, S, spw, st, false, width);
    else if (p.M <= 64) launch_xs<OutT, 4>(p, slab, S, spw, st, false, width);
    else if (p.M <= 128) launch_xs<OutT, 8>(p, slab, S, spw, st, false, width);
    else launch_xs<OutT, 16>(p, slab, S, spw, st);
    return;
  }
  if (p.M <= 128) {  // K % 128 != 0: fragment-streaming kernel
    if (p.M <= 16) launch_skinny<OutT, 1>(p, st);
    else if (p.M <= 32) launch_skinny<OutT, 2>(p, st);
    else if (p.M <= 64) launch_skinny<OutT, 4>(p, st);
    else launch_skinny<OutT, 8>(p, st);
    return;
  }
  if (mid_m_chunked(p.M, p.N, p.K)) {  // decode batches of 129..1024 rows on narrow outputs: chunks of rows
    const int64_t rows = mi_fp8_gemm_partial_max_rows(p.N);
    for (int64_t m0 = 0; m0 < p.M; m0 += rows) {
      GemmParams q = p;
      q.M = p.M - m0 < rows ? p.M - m0 : rows;
      q.a = p.a + m0 * p.lda;
      q.out = (char*)p.out + m0 * p.ldo * (int64_t)sizeof(OutT);
      if (p.sa_row) q.sa = p.sa + m0;
      launch_fp8_gemm<OutT>(q, st, workspace, workspace_bytes);
    }
    return;
  }
  if (p.K % 128 == 0 && p.lda < (1 << 24) && p.ldb < (1 << 24)) {  // prefill shapes: 256 x 256 LDS-tiled MFMA kernel
    launch_tile<OutT>(p, st, workspace, workspace_bytes);             // (row pitches in its 24-bit offset multiplies)
    return;
  }
  const unsigned gx = (unsigned)cdiv64(p.N, 64);
  if (p.M <= 16) fp8_gemm_kernel<OutT, 1><<<dim3(gx, (unsigned)cdiv64(p.M, 16)), 256, 0, st>>>(p);
  else if (p.M <= 32) fp8_gemm_kernel<OutT, 2><<<dim3(gx, (unsigned)cdiv64(p.M, 32)), 256, 0, st>>>(p);
  else if (p.M <= 64) fp8_gemm_kernel<OutT, 4><<<dim3(gx, (unsigned)cdiv64(p.M, 64)), 256, 0, st>>>(p);
  else fp8_gemm_kernel<OutT, 8><<<dim3(gx, (unsigned)cdiv64(p.M, 128)), 256, 0, st>>>(p);
}

extern "C" int mi_fp8_gemm(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                           const void* bias, void* out, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldo, int scale_a_mode, int scale_b_mode, int out_dtype,
                           void* workspace, int64_t workspace_bytes, void* stream) {
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
  {
    static const int rot_env = gemm_rotate();
    p.rotate = rot_env;
    p.var = xd_var(); p.rot_step = xw_rot();
  }
  hipStream_t st = (hipStream_t)stream;
  MI_CHECK_ARG(((uintptr_t)workspace & 15) == 0 && workspace_bytes >= 0);
  if (out_dtype == MI_BF16) launch_fp8_gemm<bf16_t>(p, st, workspace, workspace_bytes);
  else launch_fp8_gemm<f16_t>(p, st, workspace, workspace_bytes);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---- split-K partial form: raw fp32 accumulators, no epilogue; a fused consumer (fused_glue.hip)
// sums the slabs and applies scales / residual / norm / rope itself.
MI_INTERNAL int mi_fp8_gemm_plan_splits(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || M > mi_fp8_gemm_partial_max_rows(N) || K % 128 != 0) return 0;   // 0: the partial form does not apply
  int S, ppw, width;
  xs_plan(M, N, K, &S, &ppw, &width);
  return S;
}

MI_INTERNAL int mi_fp8_gemm_partial(const void* a, const void* b_nk, float* slabs, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldb, void* stream) {
  MI_CHECK_ARG(a && b_nk && slabs && M > 0 && N > 0 && K > 0 && M <= mi_fp8_gemm_partial_max_rows(N));
  if (K % 128 != 0 || lda % 16 != 0 || ldb % 16 != 0 || N % 4 != 0)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_fp8_gemm_partial: need K%%128==0, N%%4==0, lda/ldb%%16==0");
  MI_CHECK_ARG((((uintptr_t)a | (uintptr_t)b_nk | (uintptr_t)slabs) & 15) == 0);
  GemmParams p;
  p.a = (const uint8_t*)a; p.b = (const uint8_t*)b_nk; p.sa = nullptr; p.sb = nullptr; p.bias = nullptr; p.out = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldo = N; p.sa_row = 0; p.sb_row = 0; p.rotate = gemm_rotate(); p.var = xd_var(); p.rot_step = xw_rot();
  int S, ppw, width;
  xs_plan(M, N, K, &S, &ppw, &width);
  hipStream_t st = (hipStream_t)stream;
  if (M <= 16) launch_xs<bf16_t, 1>(p, slabs, S, ppw, st, true, width);
  else if (M <= 32) launch_xs<bf16_t, 2>(p, slabs, S, ppw, st, true, width);
  else if (M <= 64) launch_xs<bf16_t, 4>(p, slabs, S, ppw, st, true, width);
  else if (M <= 128) launch_xs<bf16_t, 8>(p, slabs, S, ppw, st, true, width);
  else launch_xs<bf16_t, 16>(p, slabs, S, ppw, st, true);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---- gate_up GEMM with the SiLU*mul + static FP8 quant epilogue (EPI = 1).  Returns 1 when the shape is not
// eligible (split-K wanted, narrow N, M tile) so that the caller takes the slab route instead.
MI_INTERNAL int mi_fp8_gemm_silu_epilogue(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                                          void* q_out, const float* q_scale, int64_t M, int64_t I, int64_t K, int64_t lda,
                                          int64_t ldb, int dtype, void* stream) {
  const int64_t N = 2 * I;
  if (M > 512 && K % 128 == 0 && I % 128 == 0 && lda % 16 == 0 && ldb % 16 == 0 && lda < (1 << 24) && ldb < (1 << 24) &&
      !(((uintptr_t)a | (uintptr_t)b_nk) & 15) && !((uintptr_t)q_out & 3)) {
    // prefill: the 256 x 256 tile kernel with the same epilogue (a tile = 128 gate + 128 up rows of the weights)
    GemmParams p;
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b_nk; p.sa = scale_a; p.sb = scale_b; p.bias = nullptr; p.out = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldo = N; p.sa_row = 0; p.sb_row = 0; p.rotate = gemm_rotate(); p.var = xd_var(); p.rot_step = xw_rot();
    const SiluEpi epi{(uint8_t*)q_out, q_scale};
    const int mblocks = (int)cdiv64(M, 256), nblocks = (int)(I / 128);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI_BF16)
      fp8_gemm_tile_kernel<bf16_t, 1><<<(unsigned)tile_grid((int)(mblocks * nblocks)), 512, 5 * 256 * 128, st>>>(p, mblocks, nblocks, nullptr, 1, epi);
    else
      fp8_gemm_tile_kernel<f16_t, 1><<<(unsigned)tile_grid((int)(mblocks * nblocks)), 512, 5 * 256 * 128, st>>>(p, mblocks, nblocks, nullptr, 1, epi);
    MI_CHECK_LAUNCH();
    return MI_OK;
  }
  if (M <= 0 || M > 256 || K % 128 != 0 || I % 64 != 0 || lda % 16 != 0 || ldb % 16 != 0) return 1;
  if ((((uintptr_t)a | (uintptr_t)b_nk) & 15) || ((uintptr_t)q_out & 3)) return 1;
  if (xs_waves(N) != 8 || M > mi_fp8_gemm_partial_max_rows(N)) return 1;
  int S, ppw;
  xs_plan(M, N, K, &S, &ppw);
  if (S != 1) return 1;
  GemmParams p;
  p.a = (const uint8_t*)a; p.b = (const uint8_t*)b_nk; p.sa = scale_a; p.sb = scale_b; p.bias = nullptr; p.out = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldo = N; p.sa_row = 0; p.sb_row = 0; p.rotate = gemm_rotate(); p.var = xd_var(); p.rot_step = xw_rot();
  const SiluEpi epi{(uint8_t*)q_out, q_scale};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(I / 64), 1);
  [[maybe_unused]] static const int deep = mi_tune("MI_GEMM_XD", 1);
  static const int roles = mi_tune("MI_GEMM_XW", 1);    // 0: fp8_gemm_xd_kernel; bit 1 set: per-lane epilogue
  const int staged = (roles & 2) == 0 && I % 16 == 0 && ((uintptr_t)q_out & 15) == 0;
#ifdef MI_TUNING
#define LAUNCH_EPI_LEGACY(TT, MTV)                                                                             \
  if (deep == 5) fp8_gemm_xd_kernel<TT, MTV, 1, 5><<<grid, 512, xd_lds(MTV, 5), st>>>(p, nullptr, 1, 2 * ppw, 0, epi); \
  else if (deep) fp8_gemm_xd_kernel<TT, MTV, 1, 4><<<grid, 512, xd_lds(MTV, 4), st>>>(p, nullptr, 1, 2 * ppw, 0, epi); \
  else fp8_gemm_xs_kernel<TT, MTV, 8, 5, 1><<<grid, 512, xs_split_lds(MTV), st>>>(p, nullptr, 1, ppw, 0, epi)
#else
#define LAUNCH_EPI_LEGACY(TT, MTV) return 1
#endif
#define LAUNCH_EPI(TT, MTV)                                                                                    \
  if (roles) fp8_gemm_xw_kernel<TT, MTV, 1, 4><<<grid, 512, xw_lds(MTV, 4), st>>>(p, nullptr, 1, 2 * ppw, 0, epi, staged); \
  else LAUNCH_EPI_LEGACY(TT, MTV)
  if (dtype == MI_BF16) {
    if (M <= 16) LAUNCH_EPI(bf16_t, 1); else if (M <= 32) LAUNCH_EPI(bf16_t, 2); else if (M <= 64) LAUNCH_EPI(bf16_t, 4); else if (M <= 128) LAUNCH_EPI(bf16_t, 8);
    else fp8_gemm_xd_kernel<bf16_t, 16, 1, 3><<<grid, 512, xd_lds(16, 3), st>>>(p, nullptr, 1, 2 * ppw, 0, epi);
  } else {
    if (M <= 16) LAUNCH_EPI(f16_t, 1); else if (M <= 32) LAUNCH_EPI(f16_t, 2); else if (M <= 64) LAUNCH_EPI(f16_t, 4); else if (M <= 128) LAUNCH_EPI(f16_t, 8);
    else fp8_gemm_xd_kernel<f16_t, 16, 1, 3><<<grid, 512, xd_lds(16, 3), st>>>(p, nullptr, 1, 2 * ppw, 0, epi);
  }
#undef LAUNCH_EPI
#undef LAUNCH_EPI_LEGACY
  MI_CHECK_LAUNCH();
  return MI_OK;
}
