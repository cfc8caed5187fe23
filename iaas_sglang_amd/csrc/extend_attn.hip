// Ragged extend (prefill-with-prefix) attention -- gfx950, wave64, MFMA 16x16x32.
//
// Workgroup = 4 waves = 128 query rows (64 for short extends) that share ONE kv head of ONE request:
//   group >= 4 : 32 tokens x 4 q heads      (wave w -> q head 4*hg + w, 32 tokens = two 16-row MFMA tiles)
//   group == 2 : 64 tokens x 2 q heads      group == 1 : 128 tokens x 1 q head
// so every K/V tile staged in LDS is reused by 128 rows.  Keys are walked in tiles of 64 (double-buffered):
// first the cached prefix (gathered slot by slot through kv_indices from the paged pool),
// then the new tokens (contiguous k_ext/v_ext), i.e. the new tokens never round-trip
// through the pool.  Per tile and wave:
//   S^T[key][row]  = K_tile . Q^T      A = K rows from LDS (ds_read_b128), B = Q (registers)
//   online softmax in the C layout (lane: row = l&15, keys 4*(l>>4)+r), fp32, exp2 domain
//   O^T[d][row]   += V^T . P^T         A = V^T via ds_read_b64_tr_b16 (hardware transpose of the
//                                      row-major V tile), B = P^T straight from the S^T registers
// Global->LDS staging is register-staged and split (issue loads for tile t+1 before computing
// tile t, write them into the other LDS stage after; one barrier per tile).  LDS rows are padded by 32 B:
// conflict-free for both read kinds.
// Bound: MFMA for long extends, HBM gather for long prefixes.
#include "common.h"

// decode_attn.hip: merge [rows][heads][splits] partials (unnormalised O, m, l) into o [rows, heads, D]
MI_INTERNAL int mi_attn_merge_splits(const float* ws_o, const float* ws_ml, void* o, int64_t rows, int64_t num_q_heads,
                                     int64_t num_splits, int64_t stride_o_tok, int64_t head_dim, int dtype, void* stream);

struct ExtendParams {
  const void* q;
  const void* k_ext;
  const void* v_ext;
  void* o;
  const void* k_buf;
  const void* v_buf;
  const int32_t* qo_indptr;
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  int32_t num_q_heads, num_kv_heads, group, causal;
  int64_t stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok, stride_k_slot, stride_v_slot;
  int64_t sliding_window;
  float scale_log2, sm_scale, logit_cap;
  float k_scale, v_scale;   // fp8 pool (KV8): prefix keys are k8 * k_scale, prefix values v8 * v_scale
  // speculative-decode tree mask (extend_attention.py:93-94,168-178,245-257): request i's mask is a row-major
  // [ext_len_i, prefix_i + ext_len_i] byte matrix at custom_mask + mask_indptr[i]; it REPLACES the causal rule on
  // the new-token keys; prefix keys are all visible when skip_prefix_mask (the reference's default)
  const uint8_t* custom_mask;
  const int64_t* mask_indptr;
  int32_t skip_prefix_mask;
  // split-KV (short extends over long prefixes: speculative verify, chunk tails): grid.z = batch * num_splits, split s
  // walks key tiles [s * tps, (s+1) * tps) of its block and leaves (unnormalised O, m, l) in the decode kernel's
  // workspace layout [token][head][split]; the decode merge kernel combines them
  int32_t num_splits;
  float* ws_o;
  float* ws_ml;
  // page-granular prefix (extend_attn32_kernel only; page_shift = log2(page size), 0 = token-granular): prefix key j of
  // request i is slot page_indices[page_indptr[i] + (j >> page_shift)] * page + (j & (page - 1)) -- one index per page of
  // a page-aligned pool (PagedTokenToKVPoolAllocator, allocator.py:407-543) instead of one kv_indices entry per key
  const int32_t* page_indptr;
  const int32_t* page_indices;
  int32_t page_shift;
  // optional fp8 copy of the output (extend_attn32_kernel only): o_q [tokens][Hq * D] contiguous e4m3fn =
  // quant(T-rounded o, *o_qscale), the static input scale of the following FP8 linear (o_proj); `o` may then be null
  uint8_t* o_q;
  const float* o_qscale;
  // optional NeoX RoPE of Q on load (extend_attn32_kernel only; the caller then runs mi_rope_neox on k alone):
  // q_positions [tokens] and the rotary cache [max_pos][128] rounded to T
  const int64_t* q_positions;
  const void* q_rope_t;
};

// T-rounded values -> e4m3fn with a static scale (the arithmetic of quant_tensor_kernel mode 1, fp8_quant.hip, and of
// the decode kernels' fp8 output, decode_attn.hip)
template <typename T> __device__ __forceinline__ uint32_t x_quant4_static(float a, float b, float c, float d, float inv) {
  auto f = [inv](float v) { return fmaxf(fminf(round_to<T>(v) * inv, 448.0f), -448.0f); };
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f(a), f(b), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f(c), f(d), w, true);
  return w;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 8 fp8 e4m3fn bytes -> 8 T elements (exact: every e4m3 value is a bf16 / fp16 value)
typedef __attribute__((ext_vector_type(2))) float ext_f32x2;
template <typename T> __device__ __forceinline__ uint4 fp8x8_to_T(uint2 u) {
  const ext_f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, true);
  const ext_f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, true);
  return make_uint4(pack2<T>(a[0], a[1]), pack2<T>(b[0], b[1]), pack2<T>(c[0], c[1]), pack2<T>(d[0], d[1]));
}

// RT = 16-row MFMA tiles per wave (a wave owns 16*RT query rows of ONE q head), KH = 16-key halves per key
// tile (tile = 16*KH keys).  <RT=2, KH=4>: 128 query rows per workgroup share every 64-key tile, each K / V^T
// fragment read from LDS feeds two MFMAs, and the tile buffers are double-buffered: ONE barrier per 64 keys
// (the first form, <1, 2> single-buffered, paid two barriers per 32 keys and reached 0.25-0.32 PFLOP/s).
// KV8: the POOL (cached prefix) holds fp8 e4m3fn rows (strides in bytes); they are converted to T while being
// staged (exact), k_scale multiplies the prefix logits and v_scale the prefix probabilities fed to the PV MFMA
// (never the softmax denominator).  The new tokens (k_ext / v_ext) stay T-typed and unscaled.
template <typename T, int D, int HG, int RT, int KH, bool KV8>  // HG q heads per workgroup (1, 2 or 4)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))   // <= 256 registers: two workgroups per CU
void extend_attn_kernel(const ExtendParams p, int nqb, int gy) {
  constexpr int KS = D / 32;
  constexpr int DB = D / 16;
  constexpr int ROW = D * 2 + 32;       // padded LDS row, bytes
  constexpr int KT = 16 * KH;           // keys per tile
  constexpr int BQ = 64 * RT / HG;      // tokens per workgroup
  constexpr int TPR = 256 / (D / 8);    // key rows staged per pass by 256 threads (16 B each)
  constexpr int STAGE = 2 * KT * ROW;   // K tile + V tile
  typedef typename Elem<T>::vec8 vec8;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, c16 = lane & 15;
  // one-dimensional launch, q-block-major with the blocks that see the most keys first (see extend_attn32_kernel):
  // block index = ((nqb - 1 - qb) * gz + bz) * gy + by
  const int per_level = (int)(gridDim.x / (unsigned)nqb);
  const int qb = nqb - 1 - (int)blockIdx.x / per_level;
  const int rem = (int)blockIdx.x % per_level;
  const int by = rem % gy, bz = rem / gy;
  const int hgroups = p.group / HG;
  const int hk = by / hgroups;
  const int hg = by % hgroups;
  const int req = bz / p.num_splits;
  const int split = bz % p.num_splits;

  const int32_t q_start = p.qo_indptr[req];
  const int32_t ext_len = p.qo_indptr[req + 1] - q_start;
  const int32_t kv_base = p.kv_indptr[req];
  const int32_t prefix = p.kv_indptr[req + 1] - kv_base;
  if (qb * BQ >= ext_len) return;  // whole workgroup: no barrier reached yet

  const int head = hk * p.group + hg * HG + (wave % HG);
  const int tok0 = qb * BQ + (wave / HG) * (16 * RT);  // first token (within the extend) of this wave
  const bool wave_active = tok0 < ext_len;

  // ---- Q fragments (B operand): lane -> row c16 of row tile rt, dims 32*ks + 8*g4
  vec8 qf[RT][KS];
  int32_t q_pos[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int my_tok = min(tok0 + 16 * rt + c16, ext_len - 1);
    q_pos[rt] = prefix + my_tok;     // absolute position of this lane's query row
    const T* qp = (const T*)p.q + (int64_t)(q_start + my_tok) * p.stride_q_tok + (int64_t)head * D + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[rt][ks] = __builtin_bit_cast(vec8, *(const uint4*)(qp + ks * 32));
  }

  // keys this workgroup needs: prefix + (causal ? tokens up to the block's last : all)
  const int32_t blk_last = min(ext_len, (qb + 1) * BQ);
  const bool masked = p.custom_mask != nullptr;
  const bool causal = p.causal && !masked;
  const int32_t n_keys = prefix + (causal ? blk_last : ext_len);
  const int64_t mask_base = masked ? p.mask_indptr[req] : 0;
  const int32_t seq_len = prefix + ext_len;
  const int32_t n_tiles_all = (n_keys + KT - 1) / KT;
  const int32_t tps = (n_tiles_all + p.num_splits - 1) / p.num_splits;   // tiles per split
  const int32_t t_lo = split * tps;
  const int32_t n_tiles = min(n_tiles_all, t_lo + tps);                  // this split walks tiles [t_lo, n_tiles)

  // ---- staging: thread -> (row srow + TPR*pass, 16-byte chunk schunk); registers between global and LDS
  const int srow = tid / (D / 8), schunk = tid % (D / 8);
  // The next tile is staged in NSUB pieces of 32 keys, each loaded before and written after one 32-key
  // sub-tile of compute (the other LDS stage is free for the whole iteration), so only 32 keys' worth of
  // staging registers is live.  Named registers + macros: register ARRAYS here ended up in scratch.
  constexpr int NSUB = KH / 2;          // 32-key sub-tiles per LDS tile
  constexpr int SPASS = 32 / TPR;       // staging passes per sub-tile (2 for D=128, 1 for D=64)
  static_assert(SPASS == 1 || SPASS == 2, "staging passes");
  uint4 kreg0, vreg0, kreg1, vreg1;
#define STAGE_LOAD_ONE(tile_, sub_, ps_, KR, VR)                                                                \
  {                                                                                                             \
    int32_t kp_ = (tile_) * KT + (sub_) * 32 + srow + TPR * (ps_);                                              \
    kp_ = min(kp_, n_keys - 1);                                                                                 \
    const bool in_pool_ = kp_ < prefix;                                                                         \
    const int64_t slot_ = in_pool_ ? (int64_t)p.kv_indices[kv_base + min(kp_, max(prefix - 1, 0))] : 0;        \
    const int64_t t_ = q_start + max(kp_ - prefix, 0);                                                          \
    if (KV8 && in_pool_) {                                                                                      \
      const uint2 k8_ = *(const uint2*)((const uint8_t*)p.k_buf + slot_ * p.stride_k_slot + (int64_t)hk * D + schunk * 8); \
      const uint2 v8_ = *(const uint2*)((const uint8_t*)p.v_buf + slot_ * p.stride_v_slot + (int64_t)hk * D + schunk * 8); \
      KR = fp8x8_to_T<T>(k8_);                                                                                  \
      VR = fp8x8_to_T<T>(v8_);                                                                                  \
    } else {                                                                                                    \
      const T* kr_ = (!KV8 && in_pool_) ? (const T*)p.k_buf + slot_ * p.stride_k_slot : (const T*)p.k_ext + t_ * p.stride_kx_tok; \
      const T* vr_ = (!KV8 && in_pool_) ? (const T*)p.v_buf + slot_ * p.stride_v_slot : (const T*)p.v_ext + t_ * p.stride_vx_tok; \
      KR = *(const uint4*)(kr_ + (int64_t)hk * D + schunk * 8);                                                 \
      VR = *(const uint4*)(vr_ + (int64_t)hk * D + schunk * 8);                                                 \
    }                                                                                                           \
  }
#define STAGE_LOAD(tile_, sub_)                                                  \
  {                                                                              \
    STAGE_LOAD_ONE(tile_, sub_, 0, kreg0, vreg0);                                \
    if constexpr (SPASS == 2) STAGE_LOAD_ONE(tile_, sub_, 1, kreg1, vreg1);      \
  }
#define STAGE_WRITE(st_, sub_)                                                               \
  {                                                                                          \
    char* ksw_ = smem + (st_) * STAGE + ((sub_) * 32 + srow) * ROW + schunk * 16;            \
    char* vsw_ = ksw_ + KT * ROW;                                                            \
    *(uint4*)ksw_ = kreg0;                                                                   \
    *(uint4*)vsw_ = vreg0;                                                                   \
    if constexpr (SPASS == 2) {                                                              \
      *(uint4*)(ksw_ + TPR * ROW) = kreg1;                                                   \
      *(uint4*)(vsw_ + TPR * ROW) = vreg1;                                                   \
    }                                                                                        \
  }

  f32x4 acc[RT][DB];
  float m[RT], lsum[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m[rt] = -INFINITY;
    lsum[rt] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) acc[rt][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // last key (exclusive) this WAVE can see: lets early waves skip fully masked sub-tiles
  const int32_t wave_keys = causal ? prefix + min(ext_len, tok0 + 16 * RT) : n_keys;

#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub) {
    STAGE_LOAD(t_lo, sub);
    STAGE_WRITE(0, sub);
  }
  __syncthreads();

  for (int32_t tile = t_lo; tile < n_tiles; ++tile) {
    const int st = (tile - t_lo) & 1;
    const bool has_next = tile + 1 < n_tiles;
    const char* ks_lds = smem + st * STAGE;
    const char* vs_lds = ks_lds + KT * ROW;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (has_next) STAGE_LOAD(tile + 1, sub);
      const int32_t kbase = tile * KT + sub * 32;
      if (wave_active && kbase < wave_keys) {
        // every key of the sub-tile visible to every row of the wave?  (then no mask arithmetic at all)
        const bool all_visible = kbase + 31 < n_keys && (!causal || kbase + 31 <= prefix + tok0) && p.sliding_window <= 0 &&
                                 (!masked || (p.skip_prefix_mask && kbase + 31 < prefix));
        // ---- S^T = K . Q^T for the two 16-key halves of the sub-tile; each K fragment feeds the RT row tiles
        f32x4 s[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) s[rt][0] = s[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const char* kr = ks_lds + (32 * sub + 16 * h + c16) * ROW + g4 * 16;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const vec8 kf = __builtin_bit_cast(vec8, *(const uint4*)(kr + ks * 64));
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) s[rt][h] = Elem<T>::mfma16(kf, qf[rt][ks], s[rt][h]);
          }
        }
        // ---- scale, cap, mask, online softmax; P^T packed as the B operand of the PV MFMAs:
        // k-slot (g4, j): j<4 -> key 4*g4+j of the first half, j>=4 -> of the second
        vec8 pf[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float sc[2][4];
          float tm = -INFINITY;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x;
              if (p.logit_cap > 0.f) {
                const float y = s[rt][h][r] * p.sm_scale / p.logit_cap;
                const float e = __expf(2.f * y);
                x = p.logit_cap * (1.f - 2.f / (e + 1.f)) * 1.4426950408889634f;
              } else {
                x = s[rt][h][r] * p.scale_log2;
              }
              if constexpr (KV8) {   // prefix keys come from the fp8 pool: logits * k_scale (cap path: before the cap)
                if (p.logit_cap > 0.f) {
                  const float sk = (kbase + 16 * h + 4 * g4 + r < prefix) ? p.k_scale : 1.f;
                  const float y = s[rt][h][r] * sk * p.sm_scale / p.logit_cap;
                  const float e = __expf(2.f * y);
                  x = p.logit_cap * (1.f - 2.f / (e + 1.f)) * 1.4426950408889634f;
                } else if (kbase + 16 * h + 4 * g4 + r < prefix) {
                  x *= p.k_scale;
                }
              }
              if (!all_visible) {   // wave-uniform: only diagonal / last / windowed sub-tiles pay for the mask
                const int32_t kp = kbase + 16 * h + 4 * g4 + r;
                bool ok = kp < n_keys;
                if (causal) ok = ok && (kp <= q_pos[rt]);
                if (masked && ok && !(p.skip_prefix_mask && kp < prefix))
                  ok = p.custom_mask[mask_base + (int64_t)(q_pos[rt] - prefix) * seq_len + kp] != 0;
                if (p.sliding_window > 0) ok = ok && ((int64_t)q_pos[rt] <= (int64_t)kp + p.sliding_window);
                x = ok ? x : -INFINITY;
              }
              sc[h][r] = x;
              tm = fmaxf(tm, x);
            }
          tm = fmaxf(tm, __shfl_xor(tm, 16));
          tm = fmaxf(tm, __shfl_xor(tm, 32));
          // deferred rescale: the running max only moves when it grows by more than 2^8 (P <= 2^8 is exact enough
          // in fp32 / the MFMA input type; the final 1/lsum normalisation uses the same reference max)
          if (__any(tm > m[rt] + 8.0f)) {
            const float mn = fmaxf(m[rt], tm);
            const float mns = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = (m[rt] == -INFINITY) ? 0.f : fast_exp2(m[rt] - mns);
            m[rt] = mn;
            lsum[rt] *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db) acc[rt][db] *= alpha;
          }
          const float msafe = (m[rt] == -INFINITY) ? 0.f : m[rt];
          float pr[2][4], psum = 0.f;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pr[h][r] = fast_exp2(sc[h][r] - msafe);
              psum += pr[h][r];
            }
          lsum[rt] += psum;
          if constexpr (KV8) {   // prefix values are v8 * v_scale: scale their probabilities for the PV MFMA only
            if (kbase < prefix) {
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (kbase + 16 * h + 4 * g4 + r < prefix) pr[h][r] *= p.v_scale;
            }
          }
          const u32x4 pw = {pack2<T>(pr[0][0], pr[0][1]), pack2<T>(pr[0][2], pr[0][3]),
                            pack2<T>(pr[1][0], pr[1][1]), pack2<T>(pr[1][2], pr[1][3])};
          pf[rt] = __builtin_bit_cast(vec8, pw);
        }
        // ---- O^T += V^T . P^T ; V^T fragments by transposed LDS reads, each feeds the RT row tiles
        const int qq = c16 >> 2, pp = c16 & 3;
        const char* vr0 = vs_lds + (32 * sub + 4 * g4 + qq) * ROW + pp * 8;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr0 + db * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr0 + 16 * ROW + db * 32));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][db] = Elem<T>::mfma16(__builtin_bit_cast(vec8, both), pf[rt], acc[rt][db]);
        }
      }
      // the other stage was last read in the previous iteration (every wave is past that iteration's barrier)
      if (has_next) STAGE_WRITE(st ^ 1, sub);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds O[row c16][d = 16*db + 4*g4 + r]
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float l = lsum[rt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (p.num_splits > 1) {
      if (wave_active && tok0 + 16 * rt + c16 < ext_len) {
        const int64_t slot = ((int64_t)(q_start + tok0 + 16 * rt + c16) * p.num_q_heads + head) * p.num_splits + split;
        float* wo = p.ws_o + slot * D + 4 * g4;
#pragma unroll
        for (int db = 0; db < DB; ++db) *(f32x4*)(wo + db * 16) = acc[rt][db];
        if (g4 == 0) {
          p.ws_ml[slot * 2] = m[rt];
          p.ws_ml[slot * 2 + 1] = l;
        }
      }
      continue;
    }
    if (wave_active && tok0 + 16 * rt + c16 < ext_len) {
      const float inv = 1.f / l;
      T* op = (T*)p.o + (int64_t)(q_start + tok0 + 16 * rt + c16) * p.stride_o_tok + (int64_t)head * D + 4 * g4;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        *(uint2*)(op + db * 16) = make_uint2(pack2<T>(acc[rt][db][0] * inv, acc[rt][db][1] * inv),
                                             pack2<T>(acc[rt][db][2] * inv, acc[rt][db][3] * inv));
      }
    }
  }
}

#undef STAGE_LOAD
#undef STAGE_WRITE
#undef STAGE_LOAD_ONE

// ---------------------------------------------------------------------------------------------
// Long extends, head_dim 128, bf16 / fp16 pool, plain causal / full attention: the MFMA 32x32x16 form.
//   Workgroup = 8 waves = HG q heads of ONE kv head x (8 / HG) blocks of 32 tokens; every wave owns a 32-row query
//   block of one q head, Q in registers (8 B-operand fragments).  Keys are walked in tiles of 128 (four 32-key
//   sub-tiles), K and V double-buffered in LDS, staged through registers (loads for tile t+1 issued before tile t is
//   computed, written to the other stage after).  Per 32-key sub-tile and wave:
//     S^T[key][row] = K . Q^T          8 MFMA 32x32x16, A = K rows by ds_read_b128: each 1-KiB fragment feeds 32 query
//                                      rows (the 16x16x32 form read one per 16), i.e. half the LDS traffic per flop
//     softmax                          a lane holds 16 of the 32 keys of ONE query row (column = lane & 31): the row max
//                                      and sum are in-lane over 16 registers + one exchange with lane ^ 32
//     O^T[d][row] += V^T . P^T         8 MFMA, A = V^T by ds_read_b64_tr_b16 (two 4-key blocks per fragment), B = P^T
//                                      straight from the S^T accumulator registers: element j of lane half h of k-step s
//                                      is key 16 s + 8 (j >> 2) + 4 h + (j & 3), and the V^T blocks are fetched in that
//                                      same key order (any consistent order gives the same sum)
//   LDS image of both tiles: plain 256-byte rows, 16-byte chunk c of row r at 16 * (c ^ (((r & 3) << 2) | ((r >> 2) & 3)))
//   -- conflict-free for the row reads and for the transposed reads (cdna_hip_programming.md T10, image (b)).
//   Deferred rescale (threshold 2^8, decided before a sub-tile's P is exponentiated), mask arithmetic only on
//   sub-tiles that cross the diagonal or the end of the key range.
// Everything else (head_dim 64, fp8 pool, tree mask, sliding window, logit cap, split-KV, short extends) stays on
// extend_attn_kernel above.  Bound: MFMA (bf16 32x32x16), 1024 MFMA cycles per 64 keys and wave.
// two floats -> one packed dword of T, one instruction where the ISA has it (round to nearest even, as pack2)
template <typename T> __device__ __forceinline__ uint32_t pack2_fast(float a, float b) {
  if constexpr (__is_same(T, bf16_t)) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    return pack2<T>(a, b);
  }
}

#ifdef MI_TUNING
// diagnostic (tools/x32_stamps.py): cycles of wave 0, summed over work items: [0] items, [1] entry -> Q / first-tile
// requests issued, [2] -> first tile landed (barrier passed), [3] -> key-tile loop done, [4] -> outputs stored,
// [5] key tiles walked, [6] -> end-of-item barrier passed
__device__ unsigned long long mi_x32_stamps[8];
extern "C" int mi_debug_x32_stamps(unsigned long long* host_out, int reset) {
  if (reset) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(mi_x32_stamps)) != hipSuccess) return -1;
    return hipMemset(d, 0, sizeof(mi_x32_stamps)) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mi_x32_stamps), sizeof(mi_x32_stamps)) == hipSuccess ? 0 : -1;
}
#define X32_STAMP(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#else
#define X32_STAMP(var_)
#endif

// max of three scores (MFMA outputs or -inf, never NaN).  Through HIP's fmaxf every score first passed a canonicalising
// v_max_f32 x, x, x and no v_max3_f32 was formed: 104 + 17 max instructions per tile and wave in a loop whose VALU is the
// bound; the builtin (llvm.maxnum) gives 20 + 31.
__device__ __forceinline__ float x32_fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <typename T, int HG, int NW>   // NW waves per workgroup: 8 (128-key tiles, one workgroup per CU) or 4 (64-key tiles, two)
__global__ __launch_bounds__(NW * 64) void extend_attn32_kernel(const ExtendParams p, int nqb, int nreq) {
  constexpr int D = 128, KT = 16 * NW, ROWB = 256;   // four staging passes of NW * 4 key rows
  constexpr int RP = NW * 4;
  constexpr int TB = NW / HG;           // 32-token blocks per workgroup
  constexpr int BQ = 32 * TB;
  constexpr int TILE = KT * ROWB;
  constexpr int STAGE = 2 * TILE;       // K tile, V tile
  typedef typename Elem<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // static priority for the second-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4): the younger wave of
  // every SIMD loses the issue arbitration on every segment otherwise
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  const int r = lane & 31, h = lane >> 5;
  const int hgroups = p.group / HG;
  // Items = (q block, head group, request), numbered q-block-MAJOR with the blocks that see the most keys first: the
  // launch starts with the heaviest block of every (head group, request) and ends with the lightest ones, so the tail
  // of the launch is filled with short items.  (Until round 2 the q block was the FASTEST grid index: each (head
  // group, request) ran heavy-to-light on its own, heavy blocks kept arriving until the very end and the last CUs
  // finished long after the first -- 16 x 2048 causal 1.11 ms against 0.95 ms in this order, 8 x 2048 0.59 / 0.43.)
  // A workgroup takes items blockIdx.x, + gridDim.x, ...: the launch is normally one workgroup per item.
  const int gy = p.num_kv_heads * hgroups;
  const int per_level = gy * nreq;
  const int total = nqb * per_level;
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
  // (the item's coordinates come out of integer divisions, i.e. in VGPRs; forced back into SGPRs so that the four
  // indptr reads below are scalar loads issued together -- as vector loads each one was waited for in turn, three
  // dependent memory round trips in front of every work item: in-kernel stamps, tools/x32_stamps.py)
  const int qb = __builtin_amdgcn_readfirstlane(nqb - 1 - item / per_level);
  const int rem = item % per_level;
  const int by = rem % gy;
  const int hk = __builtin_amdgcn_readfirstlane(by / hgroups), hg = __builtin_amdgcn_readfirstlane(by % hgroups);
  const int req = __builtin_amdgcn_readfirstlane(rem / gy);
  // qo_indptr[req], [req + 1], kv_indptr[req], [req + 1] and page_indptr[req] in ONE vector load (lanes 0..4 read one
  // word each, handed out by v_readlane): written as five reads they were two or three dependent round trips
  const int pshift = p.page_shift;
  int32_t meta;
  {
    int lp = lane;
    asm volatile("" : "+v"(lp));     // (keeps the lane-dependent pointer below from being hoisted to kernel entry and spilled)
    const int32_t* src = lp < 2 ? p.qo_indptr + req + lp : lp < 4 ? p.kv_indptr + req + (lp - 2)
                                                         : (pshift ? p.page_indptr + req : p.kv_indptr + req);
    meta = lp < 5 ? *src : 0;
  }
  const int32_t q_start = __builtin_amdgcn_readlane(meta, 0);
  const int32_t ext_len = __builtin_amdgcn_readlane(meta, 1) - q_start;
  const int32_t kv_base = __builtin_amdgcn_readlane(meta, 2);
  const int32_t prefix = __builtin_amdgcn_readlane(meta, 3) - kv_base;
  const int32_t pg_base = pshift ? __builtin_amdgcn_readlane(meta, 4) : 0;
  if (qb * BQ >= ext_len) continue;     // whole workgroup, before any barrier of this item
  X32_STAMP(ts0);

  const int head = hk * p.group + hg * HG + (wave % HG);
  const int tok0 = qb * BQ + (wave / HG) * 32;
  const bool wave_active = tok0 < ext_len;
  const int my_tok = min(tok0 + r, ext_len - 1);
  const int32_t q_pos = prefix + my_tok;

  // Q fragments (B operand of S^T = K . Q^T): lane -> query row r, dims 16 kk + 8 h .. + 8
  vec8 qf[8];
  // (requested with the Q rows: back by the time the first K / V tile's loads have been issued, see the rotation below)
  // (unconditional: inside a branch the load was waited for on the spot, a round trip in front of the Q loads; without
  // rope it reads two words of kv_indptr, which has at least two)
  const int64_t rope_pos = *(p.q_rope_t ? p.q_positions + q_start + my_tok : (const int64_t*)p.kv_indptr);
  {
    const T* qp = (const T*)p.q + (int64_t)(q_start + my_tok) * p.stride_q_tok + (int64_t)head * D + 8 * h;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(vec8, *(const uint4*)(qp + 16 * kk));
  }

  const bool causal = p.causal != 0;
  const int32_t blk_last = min(ext_len, (qb + 1) * BQ);
  const int32_t n_keys = prefix + (causal ? blk_last : ext_len);
  const int32_t n_tiles = (n_keys + KT - 1) / KT;
  const int32_t wave_keys = causal ? prefix + min(ext_len, tok0 + 32) : n_keys;   // keys this wave can see (exclusive)

  // ---- staging by LDS-DMA (round 3; was: 8 register loads + 8 ds_write_b128 per thread and tile, 32 VGPRs of a kernel
  // that sits at the 256-VGPR cap).  Wave w fills rows RP * ps + 4 w .. + 3 of both tiles with ONE 1-KiB piece per
  // pass ps: lane L = (row 4 w + (L >> 4), 16-byte POSITION L & 15); the LDS image keeps chunk c of row r at position
  // c ^ x(r), so the lane fetches chunk (L & 15) ^ x(r) -- the swizzle moves to the source side, and x(r) depends on the
  // low four bits of r only (RP * ps is a multiple of 32): one constant per thread.
  //   fast form: a tile that lies wholly in the NEW tokens (no index, no clamp: every tile of a prompt without cached
  //   prefix except the last) is a scalar base + four constant 32-bit lane offsets -- no vector address arithmetic at
  //   all (the per-load min / compare / 64-bit multiply-add of the general form was ~120 VALU per thread and tile in a
  //   kernel whose softmax VALU, not its MFMA, is the bound);
  //   general form: per-lane 64-bit addresses as before (prefix rows through kv_indices / page_indices, the clamped
  //   last tile).
  // The caller of a stage waits `vmcnt(0)` and meets the workgroup barrier before anybody reads it.
  const int srow = tid >> 4, spos = tid & 15;
  const int sch = spos ^ (((srow & 3) << 2) | ((srow >> 2) & 3));      // the chunk of its row this lane fetches
  const bool same_strides = p.stride_k_slot == p.stride_v_slot && p.stride_kx_tok == p.stride_vx_tok;   // the usual case
  const int64_t hd_off = (int64_t)hk * D + sch * 8;
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)wave * 1024u);
  uint32_t xoffk[4], xoffv[4];          // byte offsets of this lane's four rows from the tile's first new-token row
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    xoffk[ps] = (uint32_t)(((int64_t)(RP * ps + srow) * p.stride_kx_tok + hd_off) * (int64_t)sizeof(T));
    xoffv[ps] = (uint32_t)(((int64_t)(RP * ps + srow) * p.stride_vx_tok + hd_off) * (int64_t)sizeof(T));
  }
#define X32_DMA_ONE(tile_, st_, ps_)                                                                                \
  {                                                                                                                 \
    const int32_t kp_ = min((tile_) * KT + RP * (ps_) + srow, n_keys - 1);                                          \
    const bool in_pool_ = kp_ < prefix;                                                                             \
    const int64_t slot_ = !in_pool_ ? 0 : pshift == 0 ? (int64_t)p.kv_indices[kv_base + kp_]                        \
        : ((int64_t)p.page_indices[pg_base + (kp_ >> pshift)] << pshift) + (kp_ & ((1 << pshift) - 1));             \
    const int64_t t_ = q_start + max(kp_ - prefix, 0);                                                              \
    const int64_t ko_ = (in_pool_ ? slot_ * p.stride_k_slot : t_ * p.stride_kx_tok) + hd_off;                       \
    const int64_t vo_ = same_strides ? ko_ : (in_pool_ ? slot_ * p.stride_v_slot : t_ * p.stride_vx_tok) + hd_off;  \
    glds16((in_pool_ ? (const T*)p.k_buf : (const T*)p.k_ext) + ko_,                                                \
           lds_piece + (uint32_t)((st_) * STAGE + (ps_) * RP * ROWB));                                              \
    glds16((in_pool_ ? (const T*)p.v_buf : (const T*)p.v_ext) + vo_,                                                \
           lds_piece + (uint32_t)((st_) * STAGE + TILE + (ps_) * RP * ROWB));                                       \
  }
  // a tile wholly inside the cached prefix: one index (or page) lookup + one 64-bit multiply-add per row, without the
  // clamp and the prefix / new-token selects of the general form
#define X32_DMA_POOL(tile_, st_, ps_)                                                                               \
  {                                                                                                                 \
    const int32_t kp_ = (tile_) * KT + RP * (ps_) + srow;                                                           \
    const int64_t slot_ = pshift == 0 ? (int64_t)p.kv_indices[kv_base + kp_]                                        \
        : ((int64_t)p.page_indices[pg_base + (kp_ >> pshift)] << pshift) + (kp_ & ((1 << pshift) - 1));             \
    glds16((const T*)p.k_buf + slot_ * p.stride_k_slot + hd_off,                                                    \
           lds_piece + (uint32_t)((st_) * STAGE + (ps_) * RP * ROWB));                                              \
    glds16((const T*)p.v_buf + slot_ * p.stride_v_slot + hd_off,                                                    \
           lds_piece + (uint32_t)((st_) * STAGE + TILE + (ps_) * RP * ROWB));                                       \
  }
#define X32_STAGE_DMA(tile_, st_)                                                                                   \
  {                                                                                                                 \
    const int32_t t0_ = (tile_) * KT;                                                                               \
    if (t0_ + KT <= prefix) {                         /* wave-uniform */                                            \
      X32_DMA_POOL(tile_, st_, 0);                                                                                  \
      X32_DMA_POOL(tile_, st_, 1);                                                                                  \
      X32_DMA_POOL(tile_, st_, 2);                                                                                  \
      X32_DMA_POOL(tile_, st_, 3);                                                                                  \
    } else if (t0_ >= prefix && t0_ + KT <= n_keys) {        /* wave-uniform: scalars only */                       \
      const uint8_t* kb_ = uniform_ptr((const T*)p.k_ext + (int64_t)(q_start + t0_ - prefix) * p.stride_kx_tok);     \
      const uint8_t* vb_ = uniform_ptr((const T*)p.v_ext + (int64_t)(q_start + t0_ - prefix) * p.stride_vx_tok);     \
      _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                            \
        glds16_s(xoffk[ps], kb_, lds_piece + (uint32_t)((st_) * STAGE + ps * RP * ROWB));                           \
        glds16_s(xoffv[ps], vb_, lds_piece + (uint32_t)((st_) * STAGE + TILE + ps * RP * ROWB));                    \
      }                                                                                                             \
    } else {                                                                                                        \
      X32_DMA_ONE(tile_, st_, 0);                                                                                   \
      X32_DMA_ONE(tile_, st_, 1);                                                                                   \
      X32_DMA_ONE(tile_, st_, 2);                                                                                   \
      X32_DMA_ONE(tile_, st_, 3);                                                                                   \
    }                                                                                                               \
  }
#define X32_STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  f32x16 acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  // lane-constant LDS byte offsets of the two read kinds, computed once: everything that changes inside the tile loop
  // (stage, sub-tile, k-step, d block) is a compile-time constant added to them (the scheduler otherwise re-derives
  // the XOR swizzle per read: 12 VALU instructions per MFMA measured, twice the MFMA time)
  const int xr = ((r & 3) << 2) | ((r >> 2) & 3);                 // K row reads: row 32 sub + r
  uint32_t kofs[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) kofs[kk] = (uint32_t)(256 * r + 16 * ((2 * kk + h) ^ xr));
  const int ti = lane & 15, tq = ti >> 2, tp = ti & 3;            // transposed reads: lane 4 tq + tp of its 16-lane group
  const int tcol = 2 * ((lane >> 4) & 1) + (tp >> 1);             // chunk inside the 32-d block (+ 4 db)
  uint32_t vlo[4], vhi[4];                                        // blocks at keys 4 h + tq (+ 16 s2 + 32 sub) and 8 on
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    vlo[db] = (uint32_t)(256 * (4 * h + tq) + 16 * ((4 * db + tcol) ^ ((tq << 2) | h)) + 8 * (tp & 1));
    vhi[db] = (uint32_t)(256 * (4 * h + tq + 8) + 16 * ((4 * db + tcol) ^ ((tq << 2) | (h + 2))) + 8 * (tp & 1));
  }

  X32_STAGE_DMA(0, 0);
  // (Q rotation sits here so that its position -> cos / sin round trips run under the first K / V tile's loads; in front
  // of them it added ~3 us of exposed latency to every work item: 16 x 2048 causal 0.78 -> 0.86 ms)
  if (p.q_rope_t) {
    // NeoX RoPE in registers, rope_neox_kernel's arithmetic (elementwise.hip: cos / sin as T, every product and sum
    // rounded to T): fragment kk < 4 holds x1 = dims 16 kk + 8 h .. + 8 and fragment kk + 4 their partners x2 = + 64.
    // Once per (query block, head): the rope kernel's read + write of q (536 MB per 32 k tokens) disappears.
    const T* cs = (const T*)p.q_rope_t + rope_pos * 128 + 8 * h;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 cw4 = *(const uint4*)(cs + 16 * kk), sw4 = *(const uint4*)(cs + 64 + 16 * kk);
      const uint4 a4 = __builtin_bit_cast(uint4, qf[kk]), b4 = __builtin_bit_cast(uint4, qf[kk + 4]);
      const uint32_t cw[4] = {cw4.x, cw4.y, cw4.z, cw4.w}, sw[4] = {sw4.x, sw4.y, sw4.z, sw4.w};
      const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
      uint32_t r1[4], r2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float o1[2], o2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float x1 = t ? Elem<T>::hi(aw[e]) : Elem<T>::lo(aw[e]), x2 = t ? Elem<T>::hi(bw[e]) : Elem<T>::lo(bw[e]);
          const float co = t ? Elem<T>::hi(cw[e]) : Elem<T>::lo(cw[e]), si = t ? Elem<T>::hi(sw[e]) : Elem<T>::lo(sw[e]);
          o1[t] = round_to<T>(round_to<T>(x1 * co) - round_to<T>(x2 * si));
          o2[t] = round_to<T>(round_to<T>(x2 * co) + round_to<T>(x1 * si));
        }
        r1[e] = pack2<T>(o1[0], o1[1]);
        r2[e] = pack2<T>(o2[0], o2[1]);
      }
      qf[kk] = __builtin_bit_cast(vec8, make_uint4(r1[0], r1[1], r1[2], r1[3]));
      qf[kk + 4] = __builtin_bit_cast(vec8, make_uint4(r2[0], r2[1], r2[2], r2[3]));
    }
  }

  X32_STAMP(ts1);
  X32_STAGE_WAIT();
  __syncthreads();
  X32_STAMP(ts2);

  for (int32_t tile = 0; tile < n_tiles; ++tile) {
    const int st = tile & 1;
    const bool has_next = tile + 1 < n_tiles;
    if (has_next) X32_STAGE_DMA(tile + 1, st ^ 1);
    const char* kl = smem + st * STAGE;
    const char* vl = kl + TILE;
    const int32_t tbase = tile * KT;
    // A wave walks the four 32-key sub-tiles of the tile as a two-stage software pipeline, written out by hand: the
    // in-order issue of one wave otherwise serialises QK^T (8 dependent MFMAs) -> max -> exp -> PV, and with two waves
    // per SIMD nothing hides those latencies (measured: MFMA 18 % busy, VALU 25 %, LDS 13 %; two thirds of the wave
    // cycles waiting).
    //   stage 1 of sub-tile j:  QK^T MFMAs of sub-tile j + 1, one per step, each followed by the exp / row-sum / pack
    //                           of two scores of sub-tile j
    //   stage 2 of sub-tile j:  PV MFMAs of sub-tile j, one per step, each followed by the running max of two
    //                           scores of sub-tile j + 1
    // LDS fragments are requested four MFMAs ahead of their use.  A sub-tile no row of the wave can see is still
    // computed (masked: p = 0) -- only whole invisible TILES are skipped; that costs at most three sub-tiles per wave.
    // The interleave is expressed by source order only.  Pinning it with `__builtin_amdgcn_sched_barrier(0)` after
    // every MFMA + chunk pair produced WRONG results on hipcc 7.2 (rows of S read by the mask / max before the last
    // MFMA of the chain had written them: the hazard wait states were missing next to the barriers) -- do not add them.
    // Measured 16 x 2048 causal: 1.31 -> 1.17 ms; 4 x 8192: 3.17 -> 2.94 ms.
    if (wave_active && tbase < wave_keys) {
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      f32x16 sA, sB;
      const f32x16 x32_zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      uint4 kq[4];
      s16x4 vq_lo[4], vq_hi[4];
      uint32_t pw[2][4];
      float tmx, nmsafe;
      // D_ = A_ . B_ + C_; the first MFMA of a score chain takes the constant zero as C (an inline operand: the 16
      // v_mov per chain that zeroed the accumulator -- ~40 VALU slots per tile and wave -- are gone)
#define X32_MFMA(A_, B_, C_, D_)                                                         \
  if constexpr (__is_same(T, bf16_t)) D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0); \
  else D_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0);
#define X32_KREAD(sub_, kk_) (*(const uint4*)(kl + 256 * 32 * (sub_) + kofs[kk_]))
#define X32_VREAD(sub_, s2_, db_)                                                                                       \
  {                                                                                                                     \
    vq_lo[db_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vl + 256 * (32 * (sub_) + 16 * (s2_)) + vlo[db_])); \
    vq_hi[db_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vl + 256 * (32 * (sub_) + 16 * (s2_)) + vhi[db_])); \
  }
      // mask (diagonal / tail sub-tiles only), row max over the 32 keys, deferred rescale decision for sub-tile sub_
#define X32_MASK(sub_, S_)                                                                       \
  {                                                                                              \
    const int32_t kb_ = tbase + 32 * (sub_);                                                     \
    if (!(kb_ + 31 < n_keys && (!causal || kb_ + 31 <= prefix + tok0))) {                        \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                           \
        const int32_t kp = kb_ + (i & 3) + 8 * (i >> 2) + 4 * h;                                 \
        const bool ok = kp < n_keys && (!causal || kp <= q_pos);                                 \
        S_[i] = ok ? S_[i] : -INFINITY;                                                          \
      }                                                                                          \
    }                                                                                            \
  }
#define X32_DECIDE()                                                                             \
  {                                                                                              \
    float tm_ = fmaxf(tmx, __shfl_xor(tmx, 32)) * p.scale_log2;   /* scale > 0: max(c x) = c max(x) */ \
    if (__any(tm_ > m + 8.0f)) {   /* deferred rescale, decided before this sub-tile's P exists */ \
      const float mn = fmaxf(m, tm_);                                                            \
      const float mns = (mn == -INFINITY) ? 0.f : mn;                                            \
      const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mns);                           \
      m = mn;                                                                                    \
      lsum *= alpha;                                                                             \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                           \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[db][i] *= alpha;                      \
    }                                                                                            \
    nmsafe = (m == -INFINITY) ? 0.f : -m;                                                        \
  }
      // exp, row sum and pack of scores 2c, 2c + 1 of S_ (c = 0..7)
#define X32_EXP(S_, c_)                                                                          \
  {                                                                                              \
    const float p0 = fast_exp2(fmaf(S_[2 * (c_)], p.scale_log2, nmsafe));                        \
    const float p1 = fast_exp2(fmaf(S_[2 * (c_) + 1], p.scale_log2, nmsafe));                    \
    lsum += p0 + p1;                                                                             \
    pw[(c_) >> 2][(c_) & 3] = pack2_fast<T>(p0, p1);                                             \
  }
      // one pipelined sub-tile: CUR_ holds S of sub-tile sub_ (masked, its max decided), NXT_ receives sub-tile sub_ + 1
#define X32_STEP(sub_, CUR_, NXT_)                                                               \
  {                                                                                              \
    constexpr bool more_ = (sub_) + 1 < KT / 32;                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                           \
      if constexpr (more_) {                                                                     \
        const vec8 kf_ = __builtin_bit_cast(vec8, kq[kk & 3]);                                   \
        if (kk == 0) { X32_MFMA(kf_, qf[kk], x32_zero, NXT_); } else { X32_MFMA(kf_, qf[kk], NXT_, NXT_); } \
        if (kk + 4 < 8) kq[kk & 3] = X32_KREAD((sub_) + 1, kk + 4);                              \
      }                                                                                          \
      if (kk >= 4) X32_VREAD(sub_, 0, kk - 4);                                                   \
      X32_EXP(CUR_, kk);                                                                         \
    }                                                                                            \
    if constexpr (more_) X32_MASK((sub_) + 1, NXT_);                                             \
    tmx = -INFINITY;                                                                             \
    _Pragma("unroll") for (int n = 0; n < 8; ++n) {                                              \
      const vec8 pf_ = __builtin_bit_cast(vec8, u32x4{pw[n >> 2][0], pw[n >> 2][1], pw[n >> 2][2], pw[n >> 2][3]}); \
      const s16x8 both_ = __builtin_shufflevector(vq_lo[n & 3], vq_hi[n & 3], 0, 1, 2, 3, 4, 5, 6, 7); \
      X32_MFMA(__builtin_bit_cast(vec8, both_), pf_, acc[n & 3], acc[n & 3]);                    \
      if (n < 4) X32_VREAD(sub_, 1, n)                                                           \
      else if constexpr ((sub_) + 2 < KT / 32) kq[n & 3] = X32_KREAD((sub_) + 2, n & 3);         \
      if constexpr (more_) tmx = x32_fmax3(tmx, NXT_[2 * n], NXT_[2 * n + 1]);                   \
    }                                                                                            \
    if constexpr (more_) X32_DECIDE();                                                           \
  }
      // ---- prologue: S of sub-tile 0 (plain), its mask / max / decision; K fragments of sub-tile 1
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kq[kk] = X32_KREAD(0, kk);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const vec8 kf = __builtin_bit_cast(vec8, kq[kk & 3]);
        if (kk == 0) { X32_MFMA(kf, qf[kk], x32_zero, sA); } else { X32_MFMA(kf, qf[kk], sA, sA); }
        if (kk + 4 < 8) kq[kk & 3] = X32_KREAD(0, kk + 4);
        else kq[kk & 3] = X32_KREAD(1, kk & 3);
      }
      X32_MASK(0, sA);
      tmx = x32_fmax3(x32_fmax3(sA[0], sA[1], sA[2]), sA[3], sA[4]);
#pragma unroll
      for (int i = 5; i < 15; i += 2) tmx = x32_fmax3(tmx, sA[i], sA[i + 1]);
      tmx = x32_fmax3(tmx, sA[15], sA[15]);
      X32_DECIDE();
      X32_STEP(0, sA, sB);
      X32_STEP(1, sB, sA);
      X32_STEP(2, sA, sB);
      X32_STEP(3, sB, sA);
#undef X32_MFMA
#undef X32_KREAD
#undef X32_VREAD
#undef X32_MASK
#undef X32_DECIDE
#undef X32_EXP
#undef X32_STEP
    }
    X32_STAGE_WAIT();      // this wave's pieces of the next tile have landed; the barrier publishes everybody's
    __syncthreads();
  }
#undef X32_DMA_ONE
#undef X32_DMA_POOL
#undef X32_STAGE_DMA
#undef X32_STAGE_WAIT

  // ---- epilogue: lane holds O[row r][d = 32 db + (i & 3) + 8 (i >> 2) + 4 h].  The wave's 32 x 128 outputs go through
  // 8 KiB of the (free) stage buffers and leave as whole 256-byte head rows, 4 rows per non-temporal store instruction;
  // straight from the MFMA layout it was 16 stores of 32 rows x 16 B each (partial lines: the pattern that cost the
  // tile GEMM a third of its time).  LDS image: [32 rows][256 B], 16-byte chunk c of row R at position c ^ (R & 15).
  X32_STAMP(ts3);
  lsum += __shfl_xor(lsum, 32);
  if (wave_active) {
    const float inv = 1.f / lsum;
    // The 16 staging addresses below are lane constants: hipcc hoisted them to kernel entry, spilled them around the
    // tile loop (256-VGPR cap) and reloaded each one from SCRATCH behind an `s_waitcnt vmcnt(0)` right here -- 16
    // dependent memory round trips, 17 % of a 2048-token work item (in-kernel stamps, tools/x32_stamps.py).  Derived
    // from an opaque copy of the lane's row they are recomputed after the loop (three VALU each).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r = lane_e & 31, h = lane_e >> 5;
    char* stg = smem + wave * 8192;
    typedef __attribute__((ext_vector_type(4))) uint32_t st_u32x4;
    if (p.o) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(uint2*)(stg + 256 * r + (((4 * db + g) ^ (r & 15)) << 4) + 8 * h) =
              make_uint2(pack2<T>(acc[db][4 * g] * inv, acc[db][4 * g + 1] * inv),
                         pack2<T>(acc[db][4 * g + 2] * inv, acc[db][4 * g + 3] * inv));
      const int pr = lane_e >> 4, pc = lane_e & 15;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 4 + pr;
        const uint4 v = *(const uint4*)(stg + 256 * row + ((pc ^ (row & 15)) << 4));
        if (tok0 + row < ext_len)
          __builtin_nontemporal_store(st_u32x4{v.x, v.y, v.z, v.w},
                                      (st_u32x4*)((T*)p.o + (int64_t)(q_start + tok0 + row) * p.stride_o_tok + (int64_t)head * D + 8 * pc));
      }
    }
    if (p.o_q) {
      // fp8 copy for the following FP8 linear (static input scale): the same T-rounded values, quantised as
      // mi_fp8_quant_per_tensor(static) would -- the separate quant launch and its read of `o` disappear.  Staged in
      // the same wave-private 8 KiB (LDS operations of one wave execute in order: the reads above are done) as
      // [32 rows][128 B], row pitch 144 B, and stored as whole 128-byte head rows, 8 rows per instruction.
      const float qs = *p.o_qscale;
      const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(uint32_t*)(stg + 144 * r + 32 * db + 8 * g + 4 * h) =
              x_quant4_static<T>(acc[db][4 * g] * inv, acc[db][4 * g + 1] * inv, acc[db][4 * g + 2] * inv,
                                 acc[db][4 * g + 3] * inv, qinv);
      const int qr = lane_e >> 3, qc = lane_e & 7;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + qr;
        const uint4 v = *(const uint4*)(stg + 144 * row + 16 * qc);
        if (tok0 + row < ext_len)
          __builtin_nontemporal_store(st_u32x4{v.x, v.y, v.z, v.w},
                                      (st_u32x4*)(p.o_q + ((int64_t)(q_start + tok0 + row) * p.num_q_heads + head) * D + 16 * qc));
      }
    }
  }
  X32_STAMP(ts4);
  __syncthreads();     // the staged outputs have been read: the next item may fill the stage buffers
#ifdef MI_TUNING
  if (tid == 0) {
    const unsigned long long ts5 = __builtin_amdgcn_s_memtime();
    atomicAdd(&mi_x32_stamps[0], 1ull); atomicAdd(&mi_x32_stamps[1], ts1 - ts0); atomicAdd(&mi_x32_stamps[2], ts2 - ts1);
    atomicAdd(&mi_x32_stamps[3], ts3 - ts2); atomicAdd(&mi_x32_stamps[4], ts4 - ts3); atomicAdd(&mi_x32_stamps[5], (unsigned long long)n_tiles);
    atomicAdd(&mi_x32_stamps[6], ts5 - ts4);
  }
#endif
  }   // items of this workgroup
}

template <typename T, int D, int HG, bool KV8>
static void launch_extend(const ExtendParams& p, int64_t batch, int64_t max_extend_len, hipStream_t st) {
  constexpr int ROW = D * 2 + 32;
  // measured (8 x 2048 causal, Llama-3-8B heads): <1,2> 0.656 ms (419 TFLOP/s, 112 VGPRs: 4 workgroups per CU),
  // <2,4> 0.873 ms (209 VGPRs: 2 per CU) -- occupancy beats fragment reuse here; the big form stays selectable
  static const int big = mi_tune("MI_EXTEND_BIG", 0);
  const int gy = (int)(p.num_kv_heads * (p.group / HG));
  const int64_t gz = batch * p.num_splits;
  if (big && max_extend_len > 16) {       // 128 rows x 64-key tiles, double-buffered
    constexpr int BQ = 128 / HG;
    const int64_t nqb = cdiv64(max_extend_len, BQ);
    extend_attn_kernel<T, D, HG, 2, 4, KV8><<<(unsigned)(nqb * gy * gz), 256, 2 * 2 * 64 * ROW, st>>>(p, (int)nqb, gy);
  } else {                                // short extends (speculative verify, chunk tails): 64 rows x 32-key tiles
    constexpr int BQ = 64 / HG;
    const int64_t nqb = cdiv64(max_extend_len, BQ);
    extend_attn_kernel<T, D, HG, 1, 2, KV8><<<(unsigned)(nqb * gy * gz), 256, 2 * 2 * 32 * ROW, st>>>(p, (int)nqb, gy);
  }
}

static int x32_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  // measured 16 x 2048 causal: one workgroup per item in this order 0.95 ms, persistent (one workgroup per CU walking
  // its items) 0.94 ms; 4 x 8192: 2.74 / 2.81 ms -- the hardware dispatcher keeps ragged batches balanced, so the
  // one-item-per-workgroup launch is the default and the persistent walk stays selectable
  static const int persist = mi_tune("MI_EXTEND_PERSIST", 0);
  return persist ? cus : 0x7fffffff;
}

// the 32x32x16 form: its preconditions (see extend_attn32_kernel) and its launch
template <typename T>
static bool try_launch_extend32(const ExtendParams& p, int64_t batch, int64_t max_extend_len, hipStream_t st) {
  static const int enable = mi_tune("MI_EXTEND_32", 1);
  if (!enable || max_extend_len < 64 || p.custom_mask || p.sliding_window > 0 || p.logit_cap > 0.f || p.num_splits != 1)
    return false;
  if (p.stride_o_tok % 8 != 0 || ((uintptr_t)p.o & 15) != 0) return false;     // its 16-byte output stores
  if (p.stride_kx_tok >= (1 << 23) || p.stride_vx_tok >= (1 << 23)) return false;   // 32-bit lane offsets of its LDS-DMA
  const int g = p.group;
#define X32(HGV, NWV)                                                                                                     \
  {                                                                                                                       \
    const int64_t nqb_ = cdiv64(max_extend_len, 32 * NWV / HGV);                                                          \
    const int64_t items_ = nqb_ * p.num_kv_heads * (g / HGV) * batch;                                                     \
    if (items_ > 0x7fffffff) return false;                                                                                \
    const unsigned grid_ = (unsigned)(items_ < (int64_t)x32_cus() ? items_ : (int64_t)x32_cus());                         \
    extend_attn32_kernel<T, HGV, NWV><<<grid_, NWV * 64, 2 * 2 * 16 * NWV * 256, st>>>(p, (int)nqb_, (int)batch);         \
  }
  // (a 4-wave / 64-key-tile form, two workgroups per CU, was 1.6x slower: 1.09 vs 0.67 ms on 8 x 2048 causal)
  if (g % 8 == 0) X32(8, 8) else if (g % 4 == 0) X32(4, 8) else if (g % 2 == 0) X32(2, 8) else X32(1, 8)
#undef X32
  return true;
}

template <typename T, int D, bool KV8 = false>
static int launch_extend_g(const ExtendParams& p, int64_t batch, int64_t max_extend_len, hipStream_t st) {
  if constexpr (D == 128 && !KV8) {
    if (try_launch_extend32<T>(p, batch, max_extend_len, st)) return MI_OK;
  }
  const int g = p.group;
  if (g % 4 == 0) launch_extend<T, D, 4, KV8>(p, batch, max_extend_len, st);
  else if (g % 2 == 0) launch_extend<T, D, 2, KV8>(p, batch, max_extend_len, st);
  else launch_extend<T, D, 1, KV8>(p, batch, max_extend_len, st);
  return MI_OK;
}

static int extend_attn_impl(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                            const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                            const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                            int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                            int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                            int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                            int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                            int64_t sliding_window, int dtype, void* stream, bool kv8, float k_scale, float v_scale,
                            const uint8_t* custom_mask = nullptr, const int64_t* mask_indptr = nullptr,
                            int skip_prefix_mask = 1, void* workspace = nullptr, int64_t total_tokens = 0,
                            int64_t num_splits = 1, const int32_t* page_indptr = nullptr,
                            const int32_t* page_indices = nullptr, int64_t page_size = 1, void* o_fp8 = nullptr,
                            const float* o_scale = nullptr, const int64_t* q_positions = nullptr,
                            const void* q_rope_t = nullptr) {
  MI_CHECK_ARG(batch >= 0 && max_extend_len >= 0);
  if (batch == 0 || max_extend_len == 0) return MI_OK;
  MI_CHECK_ARG(q_ext && k_ext && v_ext && (o_ext || o_fp8) && qo_indptr && kv_indptr);
  MI_CHECK_ARG(!o_fp8 || (o_scale && ((uintptr_t)o_fp8 & 15) == 0 && num_splits == 1));
  MI_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0);
  MI_CHECK_ARG(num_splits >= 1 && num_splits <= 64 && batch * num_splits <= 65535);
  // one-dimensional launches: (q blocks of >= 16 rows) x heads x (requests x splits) workgroups
  MI_CHECK_ARG(cdiv64(max_extend_len, 16) * num_q_heads * batch * num_splits < (1ll << 31));
  MI_CHECK_ARG(num_splits == 1 || (workspace && total_tokens > 0 && ((uintptr_t)workspace & 15) == 0));
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  if (head_dim != 64 && head_dim != 128)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_extend_attn: head_dim %lld not supported (64, 128)", (long long)head_dim);
  MI_CHECK_ARG(stride_q_tok % 8 == 0 && stride_o_tok % 4 == 0 && stride_kx_tok % 8 == 0 &&
               stride_vx_tok % 8 == 0 && stride_k_slot % 8 == 0 && stride_v_slot % 8 == 0);
  MI_CHECK_ARG((((uintptr_t)q_ext | (uintptr_t)k_ext | (uintptr_t)v_ext | (uintptr_t)k_buf | (uintptr_t)v_buf) & 15) == 0);
  MI_CHECK_ARG(((uintptr_t)o_ext & 7) == 0);
  ExtendParams p;
  p.q = q_ext; p.k_ext = k_ext; p.v_ext = v_ext; p.o = o_ext; p.k_buf = k_buf; p.v_buf = v_buf;
  p.qo_indptr = qo_indptr; p.kv_indptr = kv_indptr; p.kv_indices = kv_indices;
  p.num_q_heads = (int32_t)num_q_heads; p.num_kv_heads = (int32_t)num_kv_heads;
  p.group = (int32_t)(num_q_heads / num_kv_heads); p.causal = causal;
  p.stride_q_tok = stride_q_tok; p.stride_o_tok = stride_o_tok; p.stride_kx_tok = stride_kx_tok;
  p.stride_vx_tok = stride_vx_tok; p.stride_k_slot = stride_k_slot; p.stride_v_slot = stride_v_slot;
  p.sliding_window = sliding_window; p.sm_scale = sm_scale; p.logit_cap = logit_cap;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  p.k_scale = k_scale; p.v_scale = v_scale;
  MI_CHECK_ARG(!custom_mask || mask_indptr);
  p.custom_mask = custom_mask; p.mask_indptr = mask_indptr; p.skip_prefix_mask = skip_prefix_mask;
  p.num_splits = (int32_t)num_splits;
  p.page_indptr = page_indptr; p.page_indices = page_indices; p.page_shift = 0;
  if (page_indptr && page_indices && page_size > 1) {
    MI_CHECK_ARG((page_size & (page_size - 1)) == 0 && page_size <= (1 << 20));
    while ((1ll << p.page_shift) < page_size) ++p.page_shift;
  }
  p.ws_o = (float*)workspace;
  p.ws_ml = p.ws_o ? p.ws_o + total_tokens * num_q_heads * num_splits * head_dim : nullptr;
  p.o_q = (uint8_t*)o_fp8; p.o_qscale = o_scale;
  MI_CHECK_ARG((q_positions == nullptr) == (q_rope_t == nullptr) && (!q_rope_t || (o_fp8 && ((uintptr_t)q_rope_t & 15) == 0)));
  p.q_positions = q_positions; p.q_rope_t = q_rope_t;
  hipStream_t st = (hipStream_t)stream;
  if (o_fp8) {
    // the fp8 output exists in the long-extend kernel only; its caller checked the preconditions (mi_extend_attn_fp8out)
    const bool ok = dtype == MI_BF16 ? try_launch_extend32<bf16_t>(p, batch, max_extend_len, st)
                                     : try_launch_extend32<f16_t>(p, batch, max_extend_len, st);
    if (!ok) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_extend_attn_fp8out: shape not served by the long-extend kernel");
    MI_CHECK_LAUNCH();
    return MI_OK;
  }
  if (kv8) {
    MI_CHECK_ARG(k_scale > 0.f && v_scale > 0.f);
    if (head_dim != 128) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_extend_attn_fp8kv: head_dim 128 only");
    if (dtype == MI_BF16) launch_extend_g<bf16_t, 128, true>(p, batch, max_extend_len, st);
    else launch_extend_g<f16_t, 128, true>(p, batch, max_extend_len, st);
  } else if (dtype == MI_BF16) {
    if (head_dim == 128) launch_extend_g<bf16_t, 128>(p, batch, max_extend_len, st);
    else launch_extend_g<bf16_t, 64>(p, batch, max_extend_len, st);
  } else {
    if (head_dim == 128) launch_extend_g<f16_t, 128>(p, batch, max_extend_len, st);
    else launch_extend_g<f16_t, 64>(p, batch, max_extend_len, st);
  }
  MI_CHECK_LAUNCH();
  if (num_splits > 1)
    return mi_attn_merge_splits(p.ws_o, p.ws_ml, o_ext, total_tokens, num_q_heads, num_splits, stride_o_tok, head_dim, dtype,
                                stream);
  return MI_OK;
}

extern "C" int mi_extend_attn_splitkv(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                                      const void* k_buf, const void* v_buf, int kv_fp8, float k_scale, float v_scale,
                                      const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                      const uint8_t* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask,
                                      int64_t batch, int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                                      int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_kx_tok,
                                      int64_t stride_vx_tok, int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                                      float logit_cap, int causal, int64_t sliding_window, void* workspace,
                                      int64_t total_tokens, int64_t num_splits, int dtype, void* stream) {
  return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, max_extend_len,
                          num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok,
                          stride_k_slot, stride_v_slot, sm_scale, logit_cap, custom_mask ? 0 : causal, sliding_window, dtype,
                          stream, kv_fp8 != 0, kv_fp8 ? k_scale : 1.f, kv_fp8 ? v_scale : 1.f, custom_mask, mask_indptr,
                          skip_prefix_custom_mask, workspace, total_tokens, num_splits);
}

extern "C" int mi_extend_attn(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                              const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                              const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                              int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                              int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                              int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                              int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                              int64_t sliding_window, int dtype, void* stream) {
  return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, max_extend_len,
                          num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok,
                          stride_k_slot, stride_v_slot, sm_scale, logit_cap, causal, sliding_window, dtype, stream, false,
                          1.f, 1.f);
}

// mi_extend_attn on a page-aligned pool: the 32x32x16 kernel (head_dim 128, extends >= 64 tokens, no mask / window /
// cap) takes ONE index per page of the cached prefix; every other case runs the token-granular kernels on kv_indices.
extern "C" int mi_extend_attn_paged(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                                    const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                                    const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* page_indptr,
                                    const int32_t* page_indices, int64_t page_size, int64_t batch,
                                    int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                                    int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                                    int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                                    int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                                    int64_t sliding_window, int dtype, void* stream) {
  MI_CHECK_ARG(kv_indices && page_indptr && page_indices && page_size >= 1 && (page_size & (page_size - 1)) == 0);
  return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, max_extend_len,
                          num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok,
                          stride_k_slot, stride_v_slot, sm_scale, logit_cap, causal, sliding_window, dtype, stream, false,
                          1.f, 1.f, nullptr, nullptr, 1, nullptr, 0, 1, page_indptr, page_indices, page_size);
}

extern "C" int mi_extend_attn_fp8kv(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                                    const void* k_buf8, const void* v_buf8, float k_scale, float v_scale,
                                    const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                    int64_t batch, int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                                    int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_kx_tok,
                                    int64_t stride_vx_tok, int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                                    float logit_cap, int causal, int64_t sliding_window, int dtype, void* stream) {
  return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf8, v_buf8, qo_indptr, kv_indptr, kv_indices, batch, max_extend_len,
                          num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok,
                          stride_k_slot, stride_v_slot, sm_scale, logit_cap, causal, sliding_window, dtype, stream, true,
                          k_scale, v_scale);
}

extern "C" int mi_extend_attn_masked(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                                     const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                                     const int32_t* kv_indptr, const int32_t* kv_indices, const uint8_t* custom_mask,
                                     const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t batch,
                                     int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                     int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_kx_tok,
                                     int64_t stride_vx_tok, int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale,
                                     float logit_cap, int64_t sliding_window, int dtype, void* stream) {
  MI_CHECK_ARG(custom_mask != nullptr && mask_indptr != nullptr);
  return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, max_extend_len,
                          num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok, stride_vx_tok,
                          stride_k_slot, stride_v_slot, sm_scale, logit_cap, 0, sliding_window, dtype, stream, false, 1.f,
                          1.f, custom_mask, mask_indptr, skip_prefix_custom_mask);
}

// mi_extend_attn (token- or page-granular prefix) with the output ALSO / ONLY as e4m3fn for the FP8 linear that follows
// (o_proj with a static input scale; prefill's form of mi_decode_attn_fp8out): o_fp8 [tokens, Hq * D] contiguous =
// mi_fp8_quant_per_tensor(static, *o_scale) of the T-typed result, bit for bit; o_ext may be null.  The long-extend
// kernel (head_dim 128, bf16 / fp16 pool, extends >= 64 tokens, no mask / window / cap) writes it from its epilogue;
// for every other shape the T-typed kernel runs into o_ext (then required) and the quantisation is a second launch.
// q_positions + cos_sin_cache_t (both or neither; long-extend kernel only, else MI_ERR_UNSUPPORTED): q_ext is UNROTATED
// and the kernel applies NeoX RoPE to the Q fragments as it loads them (position q_positions[token], cache rounded to T)
// (the caller rotates only k: mi_rope_neox with num_q_heads = 0); same bits as rotating q with mi_rope_neox first.
extern "C" int mi_extend_attn_fp8out(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext /* nullable */,
                                     void* o_fp8, const float* o_scale, const void* k_buf, const void* v_buf,
                                     const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                     const int32_t* page_indptr /* nullable */, const int32_t* page_indices /* nullable */,
                                     int64_t page_size, int64_t batch, int64_t total_tokens, int64_t max_extend_len,
                                     int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                                     int64_t stride_o_tok, int64_t stride_kx_tok, int64_t stride_vx_tok,
                                     int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale, float logit_cap,
                                     int causal, int64_t sliding_window, const int64_t* q_positions /* nullable */,
                                     const void* cos_sin_cache_t /* nullable */, int dtype, void* stream) {
  MI_CHECK_ARG(o_fp8 != nullptr && o_scale != nullptr && total_tokens >= 0);
  MI_CHECK_ARG((q_positions == nullptr) == (cos_sin_cache_t == nullptr));
  MI_CHECK_ARG((page_indptr == nullptr) == (page_indices == nullptr));
  static const int enable32 = mi_tune("MI_EXTEND_32", 1);
  const bool fused = enable32 && head_dim == 128 && max_extend_len >= 64 && sliding_window <= 0 && !(logit_cap > 0.f) &&
                     stride_o_tok % 8 == 0 && ((uintptr_t)o_ext & 15) == 0 && stride_kx_tok < (1 << 23) &&
                     stride_vx_tok < (1 << 23);      // (the preconditions of try_launch_extend32)
  if (fused)
    return extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch,
                            max_extend_len, num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok, stride_kx_tok,
                            stride_vx_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, causal, sliding_window, dtype,
                            stream, false, 1.f, 1.f, nullptr, nullptr, 1, nullptr, 0, 1, page_indptr, page_indices,
                            page_indptr ? page_size : 1, o_fp8, o_scale, q_positions, cos_sin_cache_t);
  if (q_positions) MI_FAIL(MI_ERR_UNSUPPORTED, "mi_extend_attn_fp8out: Q rotation on load exists in the long-extend kernel only");
  if (!o_ext) MI_FAIL(MI_ERR_INVALID, "mi_extend_attn_fp8out: this shape needs the T-typed output buffer o_ext as well");
  const int rc = extend_attn_impl(q_ext, k_ext, v_ext, o_ext, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch,
                                  max_extend_len, num_q_heads, num_kv_heads, head_dim, stride_q_tok, stride_o_tok,
                                  stride_kx_tok, stride_vx_tok, stride_k_slot, stride_v_slot, sm_scale, logit_cap, causal,
                                  sliding_window, dtype, stream, false, 1.f, 1.f, nullptr, nullptr, 1, nullptr, 0, 1,
                                  page_indptr, page_indices, page_indptr ? page_size : 1);
  if (rc != MI_OK) return rc;
  return mi_fp8_quant_per_tensor(o_ext, o_fp8, const_cast<float*>(o_scale), total_tokens, num_q_heads * head_dim,
                                 stride_o_tok, 1, dtype, stream);
}
