// Split-KV token (decode) attention over the paged KV pool -- gfx950, wave64.
//
// One WAVE owns one (request b, kv head hk, kv split s) and all `group` query heads that
// share that kv head; the W waves of a workgroup are W consecutive kv heads of the same
// (b, s), so the 256-B per-head segments of one 2-KB token slot are fetched together.
// No LDS, no barriers: waves are independent.
//
//   S^T[token][qhead] = K_tile[16 tok x D] . Q^T        MFMA 16x16x32 (K rows straight from HBM
//                                                        into the A fragment: lane -> token l&15,
//                                                        dims 32*ks + 8*(l>>4) .. +8)
//   online softmax in the MFMA C layout (lane: qhead = l&15, tokens 4*(l>>4)+r), fp32,
//   deferred rescale (only when the running max grows by > 2^8)
//   O[qhead][d] += P . V                                  VALU fp32 FMA; V rows read fully
//                                                        coalesced (16 lanes x 16 B = one 256-B
//                                                        row), P broadcast inside each 16-lane
//                                                        row by DPP row_newbcast
// HBM-bound: algorithmic bytes = 2 * tokens * Hkv * D * sizeof(T) (+ indices, q, o).
#include "common.h"
#include <type_traits>

struct DecodeParams {
  const void* q;
  const void* k_buf;
  const void* v_buf;
  void* o;
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  float* ws_o;   // [B][Hq][splits][D]
  float* ws_ml;  // [B][Hq][splits][2]
  int32_t num_q_heads, num_kv_heads, group, num_splits;
  const int32_t* plan;   // optional DEVICE-side plan {num_work, num_splits, split_chunk}: overrides the three scalars
                         // below, so a captured launch (grid = capacity of the work list) follows a plan the host
                         // rewrites before every replay; workgroups beyond num_work exit
  const int32_t* work;   // optional work list [num_work][2] = (request, split): launch order and non-empty splits only
  int32_t num_work;
  int32_t split_chunk;   // > 0: every split covers this many keys (multiple of 16) unless the request needs more
  int64_t stride_q_tok, stride_o_tok, stride_k_slot, stride_v_slot;
  float scale_log2;  // sm_scale * log2(e)   (logit_cap == 0)
  float sm_scale, logit_cap;
  float v_scale;         // fp8 KV: output multiplier (k_scale is folded into sm_scale / scale_log2 by the host)
  uint8_t* o_q;          // optional fp8 copy of o, [B][Hq*D] contiguous, = quant(T-rounded o, *o_qscale)
  const float* o_qscale;
  // page-granular indices (page_size = 1 << page_shift >= 16 slots, page-aligned allocation): kv_indices holds ONE
  // entry per page (request b's pages start at page_indptr[b]); token t of the request lives in slot
  // (page_id[t >> shift] << shift) | (t & (page - 1)).  kv_indptr still counts tokens.  page_indptr == nullptr: the
  // token-granular form above.
  const int32_t* page_indptr;
  int32_t page_shift;
};

// T-rounded values -> e4m3fn with a static scale (same arithmetic as quant_tensor_kernel mode 1)
template <typename T> __device__ __forceinline__ uint32_t quant4_static(float a, float b, float c, float d, float inv) {
  auto f = [inv](float v) { return fmaxf(fminf(round_to<T>(v) * inv, 448.0f), -448.0f); };
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f(a), f(b), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f(c), f(d), w, true);
  return w;
}

#define RESCALE_THR 8.0f

// 16-byte global load; NT = non-temporal (streamed-once KV rows should not displace q / indices / partials)
typedef __attribute__((ext_vector_type(4))) uint32_t ldg_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t ldg_u32x2;
template <bool NT> __device__ __forceinline__ uint2 ldg8(const void* p) {
  if constexpr (NT) {
    const ldg_u32x2 v = __builtin_nontemporal_load((const ldg_u32x2*)p);
    return make_uint2(v[0], v[1]);
  } else {
    return *(const uint2*)p;
  }
}
// 8 fp8 (e4m3fn) bytes -> 8 floats / one 8-element MFMA fragment of T (exact: every e4m3 value is a bf16 / fp16 value)
typedef __attribute__((ext_vector_type(2))) float cvt_f32x2;
__device__ __forceinline__ void fp8x8_to_f32(uint32_t lo, uint32_t hi, float (&f)[8]) {
  const cvt_f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
  const cvt_f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
  f[0] = a[0]; f[1] = a[1]; f[2] = b[0]; f[3] = b[1]; f[4] = c[0]; f[5] = c[1]; f[6] = d[0]; f[7] = d[1];
}
// two fp32 values that ARE values of T (every e4m3 number is one) -> one packed dword, one instruction
template <typename T> __device__ __forceinline__ uint32_t pack2_exact(float a, float b) {
  if constexpr (__is_same(T, bf16_t)) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
  }
}
template <typename T> __device__ __forceinline__ typename Elem<T>::vec8 fp8x8_to_frag(uint32_t lo, uint32_t hi) {
  float f[8];
  fp8x8_to_f32(lo, hi, f);
  const uint4 u = make_uint4(pack2_exact<T>(f[0], f[1]), pack2_exact<T>(f[2], f[3]), pack2_exact<T>(f[4], f[5]),
                             pack2_exact<T>(f[6], f[7]));
  return __builtin_bit_cast(typename Elem<T>::vec8, u);
}
template <bool NT> __device__ __forceinline__ uint4 ldg16(const void* p) {
  if constexpr (NT) {
    const ldg_u32x4 v = __builtin_nontemporal_load((const ldg_u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    return *(const uint4*)p;
  }
}

template <int N> struct IntC { static constexpr int value = N; };
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

// At least 2 waves per SIMD (<= 256 VGPRs) whatever W is: two 4-wave workgroups or one 8-wave workgroup
// are then co-resident per CU.  G = 16 (128 accumulator VGPRs) is the exception.
// KV8: the pool holds fp8 e4m3fn (1 byte per element, strides in bytes).  K rows are fetched as two 16-byte loads
// per lane (lane row r: dims 16r .. 16r+15 and 64+16r .. 64+16r+15, so each load instruction reads 64 contiguous
// bytes per token) and converted to T fragments; MFMA step j contracts dims 64(j>>1) + 16r + 8(j&1) .. +8, and Q
// uses the same permuted dim order (any order gives the same sum).
// V rows are fetched 8 bytes per lane (16 lanes = one 128-byte head row) and converted straight to fp32.
template <typename T, int D, int G, int W, bool KV8>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(G >= 16 ? 1 : 2)))
void decode_attn_kernel(const DecodeParams p) {
  constexpr bool NT = true;        // streamed-once KV rows: non-temporal loads (+3 % measured)
  static_assert(!KV8 || D == 128, "fp8 KV: head_dim 128 only");
  constexpr int KS = D / 32;       // MFMA k-steps over the head dim
  constexpr int LPT = D / 8;       // lanes per V token row (8 elements per lane)
  constexpr int TPR = 16 / LPT;    // tokens per 16-lane row per V load (1: D=128, 2: D=64)
  constexpr int NLOAD = 4 / TPR;   // V loads per 16-token tile
  typedef typename Elem<T>::vec8 vec8;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = lane >> 4, col = lane & 15;
  // request index fastest: consecutive workgroups (which the hardware deals round-robin to the 8 XCDs) are
  // different requests of the SAME split, so the non-empty splits of a ragged batch spread over all XCDs
  // (with the split index fastest, fixed-size splits s = 0..k of every request landed on XCDs 0..k only)
  // work list (ragged batches): the host hands out only non-empty (request, split) pairs, full chunks first and the
  // short remainders last, so the tail of the launch is filled with small pieces (longest-first packing)
  if (p.plan && (int)blockIdx.x >= p.plan[0]) return;
  const int b = p.work ? p.work[2 * blockIdx.x] : (int)blockIdx.x;
  const int hk = blockIdx.y * W + wave;
  const int s = p.work ? p.work[2 * blockIdx.x + 1] : (int)blockIdx.z;
  if (hk >= p.num_kv_heads) return;
  const int group = p.group;
  const int nsplit = p.num_splits;                       // workspace stride (the capacity when a plan is given)
  const int nsplit_eff = p.plan ? p.plan[1] : nsplit;    // splits of this launch
  const int32_t chunk_eff = p.plan ? p.plan[2] : p.split_chunk;

  const int32_t base = p.kv_indptr[b];
  const int32_t S = p.kv_indptr[b + 1] - base;
  int32_t per = (S + nsplit_eff - 1) / nsplit_eff;
  per = (per + 15) & ~15;
  // fixed-size splits balance RAGGED batches (every non-empty workgroup walks <= split_chunk keys, short requests
  // leave their trailing splits empty); a request longer than nsplit * split_chunk falls back to S / nsplit
  per = max(per, chunk_eff);
  const int32_t start = s * per;
  const int32_t end = min(S, start + per);
  const int hq0 = hk * group;

  if (start >= end) {  // empty split (or empty request)
    if (nsplit > 1) {
      if (row == 0 && col < group) {
        float* ml = p.ws_ml + (((int64_t)b * p.num_q_heads + hq0 + col) * nsplit + s) * 2;
        ml[0] = -INFINITY;
        ml[1] = 0.f;
      }
    } else if (row == 0 && col < LPT) {
      for (int g = 0; g < group; ++g) {
        T* o = (T*)p.o + (int64_t)b * p.stride_o_tok + (int64_t)(hq0 + g) * D + col * 8;
        if (p.o) *(uint4*)o = make_uint4(0, 0, 0, 0);
        if (p.o_q) *(uint2*)(p.o_q + ((int64_t)b * p.num_q_heads + hq0 + g) * D + col * 8) = make_uint2(0, 0);
      }
    }
    return;
  }

  // ---- Q fragments (B operand): lane -> qhead col, dims 32*ks + 8*row .. +8
  vec8 qf[KS];
  {
    const T* q = (const T*)p.q + (int64_t)b * p.stride_q_tok + (int64_t)(hq0 + col) * D + (KV8 ? row * 16 : row * 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 z = make_uint4(0, 0, 0, 0);
      if (col < group) z = *(const uint4*)(q + (KV8 ? (ks >> 1) * 64 + (ks & 1) * 8 : ks * 32));
      qf[ks] = __builtin_bit_cast(vec8, z);
    }
  }

  const int32_t* idx = p.kv_indices + (p.page_indptr ? p.page_indptr[b] : base);
  const int32_t pshift = p.page_indptr ? p.page_shift : 0, pmask = (1 << pshift) - 1;
  typedef typename std::conditional<KV8, uint8_t, T>::type TKV;
  const TKV* kb = (const TKV*)p.k_buf + (int64_t)hk * D + (KV8 ? row * 16 : row * 8);
  // slot strides fit 31 bits (checked on the host): slot index x stride is ONE v_mad_u64_u32 per address
  const uint32_t kstride = (uint32_t)p.stride_k_slot, vstride = (uint32_t)p.stride_v_slot;
  const TKV* vb = (const TKV*)p.v_buf + (int64_t)hk * D + (col % LPT) * 8;
  const int vtok = 4 * row + (col / LPT);  // + TPR * i

  float acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  struct Tile {
    uint4 kf[KV8 ? 2 : KS];     // KV8: 32 raw bytes; else KS fragments of 8 T
    typename std::conditional<KV8, uint2, uint4>::type vv[NLOAD];   // KV8: 8 bytes per lane and load
  };
  // kv_indices: ONE coalesced load covers 64 tokens (= 4 tiles), lane l holds idx[64*blk + l]; the lanes of
  // a tile pick theirs with ds_bpermute (LDS crossbar, no LDS memory).  Block b+2 is requested while block b
  // computes, so waiting for an index block never drains K/V loads issued after it.  (vmcnt retires in
  // order: the previous form loaded 5 index dwords per tile AFTER the tile ahead of it and had to drain
  // that tile before the next could be requested -- one tile in flight per wave.)
  auto load_idx_block = [&](int32_t blk) __attribute__((always_inline)) -> int32_t {
    const int32_t t = min(start + blk * 64 + lane, end - 1);
    return (idx[t >> pshift] << pshift) | (t & pmask);      // token-granular: shift 0, mask 0
  };
  auto load_tile = [&](Tile& t, int32_t vblk, int j) __attribute__((always_inline)) {   // tile j (0..3) of the block whose indices are vblk
    const int32_t ik = __shfl(vblk, 16 * j + col);
    const TKV* kp = kb + (uint64_t)(uint32_t)ik * kstride;
    if constexpr (KV8) {
      t.kf[0] = ldg16<NT>(kp);
      t.kf[1] = ldg16<NT>(kp + 64);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) t.kf[ks] = ldg16<NT>(kp + ks * 32);
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int32_t iv = __shfl(vblk, 16 * j + vtok + TPR * i);
      if constexpr (KV8) {
        t.vv[i] = ldg8<NT>(vb + (uint64_t)(uint32_t)iv * vstride);
      } else {
        t.vv[i] = ldg16<NT>(vb + (uint64_t)(uint32_t)iv * vstride);
      }
    }
  };

  auto compute = [&](const Tile& t, int32_t t0) __attribute__((always_inline)) {
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KV8) {
      sacc = Elem<T>::mfma16(fp8x8_to_frag<T>(t.kf[0].x, t.kf[0].y), qf[0], sacc);
      sacc = Elem<T>::mfma16(fp8x8_to_frag<T>(t.kf[0].z, t.kf[0].w), qf[1], sacc);
      sacc = Elem<T>::mfma16(fp8x8_to_frag<T>(t.kf[1].x, t.kf[1].y), qf[2], sacc);
      sacc = Elem<T>::mfma16(fp8x8_to_frag<T>(t.kf[1].z, t.kf[1].w), qf[3], sacc);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sacc = Elem<T>::mfma16(__builtin_bit_cast(vec8, t.kf[ks]), qf[ks], sacc);
    }
    float sc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x;
      if (p.logit_cap > 0.f) {
        float y = sacc[r] * p.sm_scale / p.logit_cap;
        float e = __expf(2.f * y);
        x = p.logit_cap * (1.f - 2.f / (e + 1.f)) * 1.4426950408889634f;
      } else {
        x = sacc[r] * p.scale_log2;
      }
      sc[r] = x;
    }
    if (t0 + 16 > end) {   // only the last live tile (and the masked ones behind it) cross the end: a scalar branch
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[r] = (t0 + 4 * row + r < end) ? sc[r] : -INFINITY;
    }
    float tm = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    tm = fmaxf(tm, __shfl_xor(tm, 16));
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    const bool grow = (col < group) && (tm > m + RESCALE_THR);
    if (__any(grow)) {
      const float mn = fmaxf(m, tm);
      const float alpha = fast_exp2(m - mn);  // m = -inf on the first tile -> 0
      m = mn;
      lsum *= alpha;
      static_for<0, G>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        const float a = row_bcast<g>(alpha);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] *= a;
      });
    }
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = fast_exp2(sc[r] - m);
      lsum += pr[r];
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      float vf[8];
      if constexpr (KV8) {
        fp8x8_to_f32(t.vv[i].x, t.vv[i].y, vf);
      } else {
        const uint32_t w[4] = {t.vv[i].x, t.vv[i].y, t.vv[i].z, t.vv[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[2 * j] = Elem<T>::lo(w[j]);
          vf[2 * j + 1] = Elem<T>::hi(w[j]);
        }
      }
      static_for<0, G>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        float pg;
        if constexpr (TPR == 1) {
          pg = row_bcast<g>(pr[i]);
        } else {
          const float p0 = row_bcast<g>(pr[2 * i]);
          const float p1 = row_bcast<g>(pr[2 * i + 1]);
          pg = (col / LPT) ? p1 : p0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = fmaf(pg, vf[j], acc[g][j]);
      });
    }
  };

  // ---- main loop: a ring of NB tile buffers (16 tokens each).  Tile n+NB is requested right after tile n
  // has been consumed, so NB-1 tiles (8 KB each at D=128) stay in flight per wave while one computes.
  constexpr int TILE_VGPRS = KV8 ? (2 + NLOAD) * 4 : (KS + NLOAD) * 4;
  constexpr int NB = (G * 8 + TILE_VGPRS * 4 <= 176) ? 4 : 2;   // acc + ring VGPRs; NB in {2,4} tiles a block
  const int32_t ntiles = (end - start + 15) >> 4;
  Tile tl[NB];
  int32_t vcur = load_idx_block(0);
  int32_t vnext = load_idx_block(1);
  static_for<0, NB>([&](auto ji) {
    constexpr int j = decltype(ji)::value;
    load_tile(tl[j], vcur, j);
  });
  // The body is branch-free: every iteration computes 4 tiles and refills 4 slots.  Tiles past the end are fully
  // masked (p = 0; the first tile of a non-empty split always holds a live key, so m is finite by then) and their
  // loads read the split's LAST row (load_idx_block clamps), one cached line.  With `if (tile < ntiles)` around the
  // refill the slot registers became a phi of "old" and "loaded": the compiler parked the loads in spare registers
  // and copied them into the slot behind `s_waitcnt vmcnt(0)` -- the ring drained once per tile (fp8 pool) or
  // every other tile (bf16 pool), one tile in flight per wave instead of NB - 1.
  for (int32_t tb = 0; tb < ntiles; tb += 4) {   // one 64-token index block per iteration
    const int32_t vafter = load_idx_block((tb >> 2) + 2);
    static_for<0, 4>([&](auto ji) {
      constexpr int j = decltype(ji)::value;
      constexpr int slot = j % NB;
      if constexpr (KV8) {
        compute(tl[slot], start + (tb + j) * 16);
        __builtin_amdgcn_sched_barrier(0);   // refill after the tile has been consumed, straight into its registers
        if constexpr (j + NB < 4) load_tile(tl[slot], vcur, j + NB);
        else load_tile(tl[slot], vnext, j + NB - 4);
      } else {     // bf16 / fp16 pool: the branch-free body needs more than 256 VGPRs (spills); kept as it was
        const int32_t tn = tb + j;
        if (tn < ntiles) {
          compute(tl[slot], start + tn * 16);
          if (tn + NB < ntiles) {
            if constexpr (j + NB < 4) load_tile(tl[slot], vcur, j + NB);
            else load_tile(tl[slot], vnext, j + NB - 4);
          }
        }
      }
    });
    vcur = vnext;
    vnext = vafter;
  }

  // ---- combine the partial sums held by the 4 lane rows (and the TPR token sub-rows)
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = acc[g][j];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      if constexpr (TPR == 2) x += __shfl_xor(x, 8);
      acc[g][j] = x;
    }
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);

  static_for<0, G>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float lg = row_bcast<g>(lsum);
    const float mg = row_bcast<g>(m);
    if (g < group && row == 0 && col < LPT) {
      const int64_t hq = hq0 + g;
      if (nsplit == 1) {
        const float inv = KV8 ? p.v_scale / lg : 1.f / lg;
        uint4 out;
        out.x = pack2<T>(acc[g][0] * inv, acc[g][1] * inv);
        out.y = pack2<T>(acc[g][2] * inv, acc[g][3] * inv);
        out.z = pack2<T>(acc[g][4] * inv, acc[g][5] * inv);
        out.w = pack2<T>(acc[g][6] * inv, acc[g][7] * inv);
        if (p.o) *(uint4*)((T*)p.o + (int64_t)b * p.stride_o_tok + hq * D + col * 8) = out;
        if (p.o_q) {
          const float qs = *p.o_qscale;
          const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
          uint2 w;
          w.x = quant4_static<T>(acc[g][0] * inv, acc[g][1] * inv, acc[g][2] * inv, acc[g][3] * inv, qinv);
          w.y = quant4_static<T>(acc[g][4] * inv, acc[g][5] * inv, acc[g][6] * inv, acc[g][7] * inv, qinv);
          *(uint2*)(p.o_q + ((int64_t)b * p.num_q_heads + hq) * D + col * 8) = w;
        }
      } else {
        const int64_t slot = ((int64_t)b * p.num_q_heads + hq) * nsplit + s;
        float4* wo = (float4*)(p.ws_o + slot * D + col * 8);
        wo[0] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
        wo[1] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
        if (col == 0) {
          p.ws_ml[slot * 2] = mg;
          p.ws_ml[slot * 2 + 1] = lg;
        }
      }
    }
  });
}

// ---- stage 2: merge the per-split partials (one wave per (b, qhead)) -------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_merge_kernel(const float* __restrict__ ws_o,
                                                           const float* __restrict__ ws_ml, T* o,
                                                           int64_t n_bh, int32_t num_q_heads,
                                                           int32_t nsplit, int64_t stride_o_tok,
                                                           uint8_t* __restrict__ o_q, const float* __restrict__ o_qscale,
                                                           float out_scale, const int32_t* __restrict__ kv_indptr,
                                                           int32_t split_chunk, const int32_t* __restrict__ plan) {
  constexpr int EPL = D / 64 > 0 ? D / 64 : 1;  // elements per lane
  const int lane = threadIdx.x & 63;
  const int64_t bh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= n_bh) return;
  const float* ml = ws_ml + bh * nsplit * 2;
  // splits that exist for this request (the same rule as the split kernel): with a work list the others were never
  // launched and their workspace entries are undefined
  int nvalid = nsplit;   // no kv_indptr (extend split-KV): every split wrote its (m, l), empty ones l = 0
  if (kv_indptr) {
    const int64_t bq = bh / num_q_heads;
    const int32_t S = kv_indptr[bq + 1] - kv_indptr[bq];
    const int ne = plan ? plan[1] : nsplit;
    int32_t per = (S + ne - 1) / ne;
    per = max((per + 15) & ~15, plan ? plan[2] : split_chunk);
    nvalid = per > 0 ? min(ne, (S + per - 1) / per) : 0;
  }
  // Many splits (a few long requests take up to 64): a serial walk pays one memory latency per split (~0.75 us each,
  // 48 of the 81 us of a B=1, S=32768 call).  The maximum is taken with one (m, l) pair per lane, and the partials are
  // fetched MB splits at a time before the first of them is used; the sums run in split order as before (same bits).
  float M = -INFINITY;
  for (int base = 0; base < nvalid; base += 64) {
    const int s = base + lane;
    M = fmaxf(M, s < nvalid ? ml[2 * s] : -INFINITY);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
  float L = 0.f, acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
  const bool active = lane * EPL < D;
  constexpr int MB = 8;
  for (int s0 = 0; s0 < nvalid; s0 += MB) {
    float ms[MB], ls[MB], pv[MB][EPL];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int s = min(s0 + i, nvalid - 1);
      ms[i] = ml[2 * s];
      ls[i] = s0 + i < nvalid ? ml[2 * s + 1] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      // an empty split (l = 0) never wrote its ws_o slice: do not read it
      const float* po = ws_o + (bh * nsplit + min(s0 + i, nvalid - 1)) * D + lane * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) pv[i][e] = (active && ls[i] > 0.f) ? po[e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (!(ls[i] > 0.f)) continue;
      const float w = fast_exp2(ms[i] - M);
      L += ls[i] * w;
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[e] += pv[i][e] * w;
    }
  }
  if (active) {
    const int64_t b = bh / num_q_heads, h = bh % num_q_heads;
    // a zero-length request has no valid split: write zeros, as the unsplit path does (not 0 * inf = NaN)
    const float inv = !(L > 0.f) ? 0.f : out_scale == 1.f ? 1.f / L : out_scale / L;
    if (o) {
      T* out = o + b * stride_o_tok + h * D + lane * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) out[e] = (T)round_to<T>(acc[e] * inv);
    }
    if (o_q) {
      const float qs = *o_qscale;
      const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
      uint8_t* out = o_q + bh * D + lane * EPL;
      if constexpr (EPL == 2) {
        const uint32_t w = quant4_static<T>(acc[0] * inv, acc[1] * inv, 0.f, 0.f, qinv);
        *(uint16_t*)out = (uint16_t)(w & 0xffffu);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) out[e] = (uint8_t)(quant4_static<T>(acc[e] * inv, 0.f, 0.f, 0.f, qinv) & 0xffu);
      }
    }
  }
}

// ---------------------------------------------------------------------------- host side
MI_INTERNAL int mi_attn_merge_splits(const float* ws_o, const float* ws_ml, void* o, int64_t rows, int64_t num_q_heads,
                                     int64_t num_splits, int64_t stride_o_tok, int64_t head_dim, int dtype, void* stream) {
  const int64_t n_bh = rows * num_q_heads;
  const unsigned blocks = (unsigned)cdiv64(n_bh, 4);
  hipStream_t st = (hipStream_t)stream;
#define MERGE_X(TT, DD) decode_merge_kernel<TT, DD><<<blocks, 256, 0, st>>>(ws_o, ws_ml, (TT*)o, n_bh, (int)num_q_heads, (int)num_splits, stride_o_tok, nullptr, nullptr, 1.f, nullptr, 0, nullptr)
  if (dtype == MI_BF16) { if (head_dim == 128) MERGE_X(bf16_t, 128); else MERGE_X(bf16_t, 64); }
  else { if (head_dim == 128) MERGE_X(f16_t, 128); else MERGE_X(f16_t, 64); }
#undef MERGE_X
  MI_CHECK_LAUNCH();
  return MI_OK;
}

extern "C" int64_t mi_decode_attn_workspace_bytes(int64_t batch, int64_t num_q_heads,
                                                  int64_t v_head_dim, int64_t num_splits) {
  if (num_splits <= 1) return 0;
  return batch * num_q_heads * num_splits * (v_head_dim + 2) * (int64_t)sizeof(float);
}

template <typename T, int D, int G, int W, bool KV8>
static void launch_decode(const DecodeParams& p, int64_t batch, hipStream_t st) {
  dim3 grid((unsigned)(p.work ? p.num_work : batch), (unsigned)((p.num_kv_heads + W - 1) / W), (unsigned)(p.work ? 1 : p.num_splits));
  decode_attn_kernel<T, D, G, W, KV8><<<grid, W * 64, 0, st>>>(p);
}

template <typename T, int D, int G, bool KV8>
static void launch_decode_w(const DecodeParams& p, int64_t batch, hipStream_t st) {
  static const int wenv = mi_tune("MI_DECODE_W", 0);
  const int h = p.num_kv_heads;
  // kv heads (= waves) per workgroup: 8 fills a CU with one workgroup, but a small batch x few splits (long-context
  // decode of a few requests) then leaves most CUs idle -- halve W until the launch has ~one workgroup per CU
  // (measured B=8, S=8192, 16 splits: W=8 90 us, W=4 65 us; B=1, S=32768, 64 splits: 107 / 86 / 81 us for W=8/4/2)
  int w = h % 8 == 0 && G < 16 ? 8 : h % 4 == 0 ? 4 : h % 2 == 0 ? 2 : 1;   // G=16: 128 accumulator VGPRs, <= 4 waves
  const int64_t items = p.work ? p.num_work : batch * p.num_splits;
  while (w > 2 && items * (h / w) < 256) w >>= 1;
  if (wenv == 8 && h % 8 == 0 && G < 16) w = 8;
  else if (wenv == 4 && h % 4 == 0) w = 4;
  else if (wenv == 2 && h % 2 == 0) w = 2;
  if (w == 8) launch_decode<T, D, G, 8, KV8>(p, batch, st);
  else if (w == 4) launch_decode<T, D, G, 4, KV8>(p, batch, st);
  else if (w == 2) launch_decode<T, D, G, 2, KV8>(p, batch, st);
  else launch_decode<T, D, G, 1, KV8>(p, batch, st);
}

template <typename T, int D, bool KV8 = false>
static int launch_decode_g(const DecodeParams& p, int64_t batch, hipStream_t st) {
  const int g = p.group;
  if (g == 1) launch_decode_w<T, D, 1, KV8>(p, batch, st);
  else if (g == 2) launch_decode_w<T, D, 2, KV8>(p, batch, st);
  else if (g <= 4) launch_decode_w<T, D, 4, KV8>(p, batch, st);
  else if (g <= 8) launch_decode_w<T, D, 8, KV8>(p, batch, st);
  else if (g <= 16) launch_decode_w<T, D, 16, KV8>(p, batch, st);
  else return MI_ERR_UNSUPPORTED;
  return MI_OK;
}

static int decode_attn_impl(const void* q, const void* k_buf, const void* v_buf, void* o,
                            const int32_t* kv_indptr, const int32_t* kv_indices, void* workspace,
                            int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                            int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                            int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                            float logit_cap, int64_t num_splits, int64_t split_chunk, const int32_t* work, int64_t num_work,
                            const int32_t* plan, int dtype, void* stream, void* o_fp8, const float* o_scale, bool kv8 = false,
                            float k_scale = 1.f, float v_scale = 1.f, const int32_t* page_indptr = nullptr,
                            int64_t page_size = 1) {
  MI_CHECK_ARG(batch >= 0);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(q && k_buf && v_buf && (o || o_fp8) && kv_indptr && kv_indices);
  MI_CHECK_ARG(!o_fp8 || (o_scale && ((uintptr_t)o_fp8 & 7) == 0));
  MI_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0);
  MI_CHECK_ARG(num_splits >= 1 && num_splits <= 65535 && batch <= 65535);
  MI_CHECK_ARG(num_splits == 1 || workspace != nullptr);
  MI_CHECK_ARG(split_chunk >= 0 && split_chunk % 16 == 0 && split_chunk < (1 << 30));
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (head_dim != 64 && head_dim != 128)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_decode_attn: head_dim %lld not supported (64, 128)", (long long)head_dim);
  if (kv8 && head_dim != 128) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_decode_attn_fp8kv: head_dim 128 only");
  MI_CHECK_ARG(k_scale > 0.f && v_scale > 0.f);
  MI_CHECK_ARG(!kv8 || (stride_k_slot % 16 == 0 && stride_v_slot % 16 == 0));
  // 16-byte vector accesses on q/k/v/o rows
  MI_CHECK_ARG(stride_q_tok % 8 == 0 && stride_o_tok % 8 == 0 && stride_k_slot % 8 == 0 &&
               stride_v_slot % 8 == 0);
  MI_CHECK_ARG(stride_k_slot > 0 && stride_v_slot > 0 && stride_k_slot < (1ll << 31) && stride_v_slot < (1ll << 31));
  MI_CHECK_ARG((((uintptr_t)q | (uintptr_t)k_buf | (uintptr_t)v_buf | (uintptr_t)o) & 15) == 0);
  MI_CHECK_ARG(((uintptr_t)workspace & 15) == 0);

  DecodeParams p;
  p.q = q; p.k_buf = k_buf; p.v_buf = v_buf; p.o = o;
  p.kv_indptr = kv_indptr; p.kv_indices = kv_indices;
  p.ws_o = (float*)workspace;
  p.ws_ml = p.ws_o ? p.ws_o + batch * num_q_heads * num_splits * head_dim : nullptr;
  p.num_q_heads = (int32_t)num_q_heads; p.num_kv_heads = (int32_t)num_kv_heads;
  p.group = (int32_t)(num_q_heads / num_kv_heads); p.num_splits = (int32_t)num_splits;
  p.split_chunk = (int32_t)split_chunk;
  MI_CHECK_ARG(!work || (num_work > 0 && num_work <= 0x7fffffff && num_splits > 1 && (split_chunk > 0 || plan)));
  MI_CHECK_ARG(!plan || (work && num_splits > 1));
  p.work = work; p.num_work = (int32_t)num_work; p.plan = plan;
  p.stride_q_tok = stride_q_tok; p.stride_o_tok = stride_o_tok;
  p.stride_k_slot = stride_k_slot; p.stride_v_slot = stride_v_slot;
  // fp8 KV: logits = sm_scale * k_scale * (q . k8), out = v_scale * softmax . v8
  p.sm_scale = kv8 ? sm_scale * k_scale : sm_scale; p.logit_cap = logit_cap;
  p.scale_log2 = p.sm_scale * 1.4426950408889634f;
  p.v_scale = kv8 ? v_scale : 1.f;
  p.o_q = (uint8_t*)o_fp8; p.o_qscale = o_scale;
  p.page_indptr = page_indptr; p.page_shift = 0;
  if (page_indptr) {
    MI_CHECK_ARG(page_size >= 1 && page_size <= (1 << 20) && (page_size & (page_size - 1)) == 0);
    while ((1ll << p.page_shift) < page_size) ++p.page_shift;
  }
  hipStream_t st = (hipStream_t)stream;

  int rc;
  if (kv8)
    rc = dtype == MI_BF16 ? launch_decode_g<bf16_t, 128, true>(p, batch, st) : launch_decode_g<f16_t, 128, true>(p, batch, st);
  else if (dtype == MI_BF16)
    rc = head_dim == 128 ? launch_decode_g<bf16_t, 128>(p, batch, st) : launch_decode_g<bf16_t, 64>(p, batch, st);
  else
    rc = head_dim == 128 ? launch_decode_g<f16_t, 128>(p, batch, st) : launch_decode_g<f16_t, 64>(p, batch, st);
  if (rc != MI_OK) MI_FAIL(rc, "mi_decode_attn: group size %d not supported (<= 16)", p.group);
  MI_CHECK_LAUNCH();

  if (num_splits > 1) {
    const int64_t n_bh = batch * num_q_heads;
    const unsigned blocks = (unsigned)cdiv64(n_bh, 4);
    // without a work list every (request, split) workgroup ran and wrote its (m, l) -- empty splits l = 0 -- so the
    // merge need not derive the live splits from kv_indptr (a dependent load in front of its first read)
    if (!p.work) kv_indptr = nullptr;
    if (dtype == MI_BF16) {
      if (head_dim == 128)
        decode_merge_kernel<bf16_t, 128><<<blocks, 256, 0, st>>>(p.ws_o, p.ws_ml, (bf16_t*)o, n_bh, p.num_q_heads, p.num_splits, stride_o_tok, p.o_q, p.o_qscale, p.v_scale, kv_indptr, p.split_chunk, p.plan);
      else
        decode_merge_kernel<bf16_t, 64><<<blocks, 256, 0, st>>>(p.ws_o, p.ws_ml, (bf16_t*)o, n_bh, p.num_q_heads, p.num_splits, stride_o_tok, p.o_q, p.o_qscale, p.v_scale, kv_indptr, p.split_chunk, p.plan);
    } else {
      if (head_dim == 128)
        decode_merge_kernel<f16_t, 128><<<blocks, 256, 0, st>>>(p.ws_o, p.ws_ml, (f16_t*)o, n_bh, p.num_q_heads, p.num_splits, stride_o_tok, p.o_q, p.o_qscale, p.v_scale, kv_indptr, p.split_chunk, p.plan);
      else
        decode_merge_kernel<f16_t, 64><<<blocks, 256, 0, st>>>(p.ws_o, p.ws_ml, (f16_t*)o, n_bh, p.num_q_heads, p.num_splits, stride_o_tok, p.o_q, p.o_qscale, p.v_scale, kv_indptr, p.split_chunk, p.plan);
    }
    MI_CHECK_LAUNCH();
  }
  return MI_OK;
}

extern "C" int mi_decode_attn(const void* q, const void* k_buf, const void* v_buf, void* o,
                              const int32_t* kv_indptr, const int32_t* kv_indices, void* workspace,
                              int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                              int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                              int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                              float logit_cap, int64_t num_splits, int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan, int dtype, void* stream) {
  MI_CHECK_ARG(o != nullptr);
  return decode_attn_impl(q, k_buf, v_buf, o, kv_indptr, kv_indices, workspace, batch, num_q_heads, num_kv_heads, head_dim,
                          stride_q_tok, stride_o_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, num_splits, split_chunk, work, num_work, plan, dtype,
                          stream, nullptr, nullptr);
}

extern "C" int mi_decode_attn_fp8out(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                                     void* o_fp8, const float* o_scale, const int32_t* kv_indptr,
                                     const int32_t* kv_indices, void* workspace, int64_t batch, int64_t num_q_heads,
                                     int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                                     int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale, float logit_cap,
                                     int64_t num_splits, int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan,
                                     int dtype, void* stream) {
  MI_CHECK_ARG(o_fp8 != nullptr && o_scale != nullptr);
  return decode_attn_impl(q, k_buf, v_buf, o, kv_indptr, kv_indices, workspace, batch, num_q_heads, num_kv_heads, head_dim,
                          stride_q_tok, stride_o_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, num_splits, split_chunk, work, num_work, plan, dtype,
                          stream, o_fp8, o_scale);
}

extern "C" int mi_decode_attn_fp8kv(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                                    void* o_fp8 /* nullable */, const float* o_scale, float k_scale, float v_scale,
                                    const int32_t* kv_indptr, const int32_t* kv_indices, void* workspace, int64_t batch,
                                    int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                                    int64_t stride_o_tok, int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                                    float logit_cap, int64_t num_splits, int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan, int dtype, void* stream) {
  return decode_attn_impl(q, k_buf, v_buf, o, kv_indptr, kv_indices, workspace, batch, num_q_heads, num_kv_heads, head_dim,
                          stride_q_tok, stride_o_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, num_splits, split_chunk, work, num_work, plan, dtype,
                          stream, o_fp8, o_scale, true, k_scale, v_scale);
}

// Page-granular form of the three entry points above: `page_indices` holds one page id per page of every
// request (request b: page_indptr[b] .. ), kv_indptr still counts tokens; o_fp8 / o_scale and kv8 / k_scale / v_scale as
// in mi_decode_attn_fp8out / mi_decode_attn_fp8kv (o_fp8 nullable, kv8 = 0 for a bf16 / fp16 pool).
extern "C" int mi_decode_attn_paged(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                                    void* o_fp8 /* nullable */, const float* o_scale, int kv8, float k_scale,
                                    float v_scale, const int32_t* kv_indptr, const int32_t* page_indptr,
                                    const int32_t* page_indices, int64_t page_size, void* workspace, int64_t batch,
                                    int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                                    int64_t stride_o_tok, int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                                    float logit_cap, int64_t num_splits, int64_t split_chunk, const int32_t* work,
                                    int64_t num_work, const int32_t* plan, int dtype, void* stream) {
  MI_CHECK_ARG(page_indptr != nullptr && page_size >= 1);
  return decode_attn_impl(q, k_buf, v_buf, o, kv_indptr, page_indices, workspace, batch, num_q_heads, num_kv_heads, head_dim,
                          stride_q_tok, stride_o_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, num_splits,
                          split_chunk, work, num_work, plan, dtype, stream, o_fp8, o_scale, kv8 != 0, k_scale, v_scale,
                          page_indptr, page_size);
}
