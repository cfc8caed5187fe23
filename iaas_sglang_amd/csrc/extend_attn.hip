// Ragged extend (prefill-with-prefix) attention -- gfx950, wave64, MFMA 16x16x32.
//
// Workgroup = 4 waves = 64 query rows that share ONE kv head of ONE request:
//   group >= 4 : 16 tokens x 4 q heads      (wave w -> q head 4*hg + w)
//   group == 2 : 32 tokens x 2 q heads      group == 1 : 64 tokens x 1 q head
// so every K/V tile staged in LDS is reused by 64 rows.  Keys are walked in tiles of 32:
// first the cached prefix (gathered slot by slot through kv_indices from the paged pool),
// then the new tokens (contiguous k_ext/v_ext), i.e. the new tokens never round-trip
// through the pool.  Per tile and wave:
//   S^T[key][row]  = K_tile . Q^T      A = K rows from LDS (ds_read_b128), B = Q (registers)
//   online softmax in the C layout (lane: row = l&15, keys 4*(l>>4)+r), fp32, exp2 domain
//   O^T[d][row]   += V^T . P^T         A = V^T via ds_read_b64_tr_b16 (hardware transpose of the
//                                      row-major V tile), B = P^T straight from the S^T registers
// Global->LDS staging is register-staged and split (issue loads for tile t+1 before computing
// tile t, write them after).  LDS rows are padded by 32 B: conflict-free for both read kinds.
// Bound: MFMA for long extends, HBM gather for long prefixes.
#include "common.h"

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
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T, int D, int HG>  // HG q heads per workgroup (1, 2 or 4)
__global__ __launch_bounds__(256) void extend_attn_kernel(const ExtendParams p) {
  constexpr int KS = D / 32;
  constexpr int DB = D / 16;
  constexpr int ROW = D * 2 + 32;       // padded LDS row, bytes
  constexpr int BQ = 64 / HG;           // tokens per workgroup
  constexpr int TPR = 256 / (D / 8);    // key rows staged per pass by 256 threads (16 B each)
  constexpr int NPASS = 32 / TPR;       // passes per 32-key tile (2 for D=128, 1 for D=64)
  typedef typename Elem<T>::vec8 vec8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_lds = smem;
  char* vs_lds = smem + 32 * ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, c16 = lane & 15;
  const int qb = blockIdx.x;
  const int hgroups = p.group / HG;
  const int hk = blockIdx.y / hgroups;
  const int hg = blockIdx.y % hgroups;
  const int req = blockIdx.z;

  const int32_t q_start = p.qo_indptr[req];
  const int32_t ext_len = p.qo_indptr[req + 1] - q_start;
  const int32_t kv_base = p.kv_indptr[req];
  const int32_t prefix = p.kv_indptr[req + 1] - kv_base;
  if (qb * BQ >= ext_len) return;  // whole workgroup: no barrier reached yet

  const int head = hk * p.group + hg * HG + (wave % HG);
  const int tok0 = qb * BQ + (wave / HG) * 16;  // first token (within the extend) of this wave
  const int my_tok = min(tok0 + c16, ext_len - 1);
  const bool wave_active = tok0 < ext_len;

  // ---- Q fragments (B operand): lane -> row c16, dims 32*ks + 8*g4
  vec8 qf[KS];
  {
    const T* qp = (const T*)p.q + (int64_t)(q_start + my_tok) * p.stride_q_tok + (int64_t)head * D + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(vec8, *(const uint4*)(qp + ks * 32));
  }

  // keys this workgroup needs: prefix + (causal ? tokens up to the block's last : all)
  const int32_t blk_last = min(ext_len, (qb + 1) * BQ);
  const int32_t n_keys = prefix + (p.causal ? blk_last : ext_len);
  const int32_t n_tiles = (n_keys + 31) / 32;

  // ---- staging: thread -> (row srow + TPR*pass, 16-byte chunk schunk)
  const int srow = tid / (D / 8), schunk = tid % (D / 8);
  // named registers + macros (arrays captured by lambdas / indexed in loops ended up in scratch)
  uint4 kreg0, vreg0, kreg1, vreg1;
#define STAGE_LOAD_ONE(tile_, ps_, KR, VR)                                                    \
  {                                                                                           \
    int32_t kp_ = (tile_) * 32 + srow + TPR * (ps_);                                          \
    kp_ = min(kp_, n_keys - 1);                                                               \
    const bool in_pool_ = kp_ < prefix;                                                       \
    const int64_t slot_ = in_pool_ ? (int64_t)p.kv_indices[kv_base + min(kp_, max(prefix - 1, 0))] : 0; \
    const int64_t t_ = q_start + max(kp_ - prefix, 0);                                        \
    const T* kr_ = in_pool_ ? (const T*)p.k_buf + slot_ * p.stride_k_slot : (const T*)p.k_ext + t_ * p.stride_kx_tok; \
    const T* vr_ = in_pool_ ? (const T*)p.v_buf + slot_ * p.stride_v_slot : (const T*)p.v_ext + t_ * p.stride_vx_tok; \
    KR = *(const uint4*)(kr_ + (int64_t)hk * D + schunk * 8);                                 \
    VR = *(const uint4*)(vr_ + (int64_t)hk * D + schunk * 8);                                 \
  }
#define STAGE_LOAD(tile_)                                       \
  {                                                             \
    STAGE_LOAD_ONE(tile_, 0, kreg0, vreg0);                     \
    if constexpr (NPASS == 2) STAGE_LOAD_ONE(tile_, 1, kreg1, vreg1); \
  }
#define STAGE_WRITE()                                                         \
  {                                                                           \
    *(uint4*)(ks_lds + srow * ROW + schunk * 16) = kreg0;                     \
    *(uint4*)(vs_lds + srow * ROW + schunk * 16) = vreg0;                     \
    if constexpr (NPASS == 2) {                                               \
      *(uint4*)(ks_lds + (srow + TPR) * ROW + schunk * 16) = kreg1;           \
      *(uint4*)(vs_lds + (srow + TPR) * ROW + schunk * 16) = vreg1;           \
    }                                                                         \
  }

  f32x4 acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  const int32_t q_pos = prefix + my_tok;  // absolute position of this lane's query row
  // last key (exclusive) this WAVE can see: lets early waves skip fully masked tiles
  const int32_t wave_keys = p.causal ? prefix + min(ext_len, tok0 + 16) : n_keys;

  STAGE_LOAD(0);
  STAGE_WRITE();
  __syncthreads();

  for (int32_t tile = 0; tile < n_tiles; ++tile) {
    const bool has_next = tile + 1 < n_tiles;
    if (has_next) STAGE_LOAD(tile + 1);

    if (wave_active && tile * 32 < wave_keys) {
      // ---- S^T = K . Q^T for the two 16-key halves
      f32x4 s[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        s[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* kr = ks_lds + (16 * h + c16) * ROW + g4 * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const vec8 kf = __builtin_bit_cast(vec8, *(const uint4*)(kr + ks * 64));
          s[h] = Elem<T>::mfma16(kf, qf[ks], s[h]);
        }
      }
      // ---- scale, cap, mask
      float sc[2][4];
      float tm = -INFINITY;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x;
          if (p.logit_cap > 0.f) {
            const float y = s[h][r] * p.sm_scale / p.logit_cap;
            const float e = __expf(2.f * y);
            x = p.logit_cap * (1.f - 2.f / (e + 1.f)) * 1.4426950408889634f;
          } else {
            x = s[h][r] * p.scale_log2;
          }
          const int32_t kp = tile * 32 + 16 * h + 4 * g4 + r;
          bool ok = kp < n_keys;
          if (p.causal) ok = ok && (kp <= q_pos);
          if (p.sliding_window > 0) ok = ok && ((int64_t)q_pos <= (int64_t)kp + p.sliding_window);
          x = ok ? x : -INFINITY;
          sc[h][r] = x;
          tm = fmaxf(tm, x);
        }
      tm = fmaxf(tm, __shfl_xor(tm, 16));
      tm = fmaxf(tm, __shfl_xor(tm, 32));
      const float mn = fmaxf(m, tm);
      const float msafe = (mn == -INFINITY) ? 0.f : mn;
      const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - msafe);
      m = mn;
      float pr[2][4], psum = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[h][r] = fast_exp2(sc[h][r] - msafe);
          psum += pr[h][r];
        }
      lsum = lsum * alpha + psum;
#pragma unroll
      for (int db = 0; db < DB; ++db) acc[db] *= alpha;
      // ---- P^T as the B operand: k-slot (g4, j): j<4 -> key 4*g4+j, j>=4 -> key 16+4*g4+(j-4)
      const u32x4 pw = {pack2<T>(pr[0][0], pr[0][1]), pack2<T>(pr[0][2], pr[0][3]),
                        pack2<T>(pr[1][0], pr[1][1]), pack2<T>(pr[1][2], pr[1][3])};
      const vec8 pf = __builtin_bit_cast(vec8, pw);
      // ---- O^T += V^T . P^T ; V^T fragments by transposed LDS reads
      const int qq = c16 >> 2, pp = c16 & 3;
      const char* vr0 = vs_lds + (4 * g4 + qq) * ROW + pp * 8;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr0 + db * 32));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr0 + 16 * ROW + db * 32));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[db] = Elem<T>::mfma16(__builtin_bit_cast(vec8, both), pf, acc[db]);
      }
    }
    __syncthreads();  // everyone is done reading this tile
    if (has_next) {
      STAGE_WRITE();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds O[row c16][d = 16*db + 4*g4 + r]
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  if (wave_active && tok0 + c16 < ext_len) {
    const float inv = 1.f / lsum;
    T* op = (T*)p.o + (int64_t)(q_start + tok0 + c16) * p.stride_o_tok + (int64_t)head * D + 4 * g4;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      *(uint2*)(op + db * 16) = make_uint2(pack2<T>(acc[db][0] * inv, acc[db][1] * inv),
                                           pack2<T>(acc[db][2] * inv, acc[db][3] * inv));
    }
  }
}

template <typename T, int D, int HG>
static void launch_extend(const ExtendParams& p, int64_t batch, int64_t max_extend_len, hipStream_t st) {
  constexpr int BQ = 64 / HG;
  constexpr int ROW = D * 2 + 32;
  const size_t lds = 2 * 32 * ROW;
  dim3 grid((unsigned)cdiv64(max_extend_len, BQ), (unsigned)(p.num_kv_heads * (p.group / HG)), (unsigned)batch);
  extend_attn_kernel<T, D, HG><<<grid, 256, lds, st>>>(p);
}

template <typename T, int D>
static int launch_extend_g(const ExtendParams& p, int64_t batch, int64_t max_extend_len, hipStream_t st) {
  const int g = p.group;
  if (g % 4 == 0) launch_extend<T, D, 4>(p, batch, max_extend_len, st);
  else if (g % 2 == 0) launch_extend<T, D, 2>(p, batch, max_extend_len, st);
  else launch_extend<T, D, 1>(p, batch, max_extend_len, st);
  return MI_OK;
}

extern "C" int mi_extend_attn(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                              const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                              const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                              int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                              int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                              int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                              int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                              int64_t sliding_window, int dtype, void* stream) {
  MI_CHECK_ARG(batch >= 0 && max_extend_len >= 0);
  if (batch == 0 || max_extend_len == 0) return MI_OK;
  MI_CHECK_ARG(q_ext && k_ext && v_ext && o_ext && qo_indptr && kv_indptr);
  MI_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0);
  MI_CHECK_ARG(batch <= 65535);
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
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16) {
    if (head_dim == 128) launch_extend_g<bf16_t, 128>(p, batch, max_extend_len, st);
    else launch_extend_g<bf16_t, 64>(p, batch, max_extend_len, st);
  } else {
    if (head_dim == 128) launch_extend_g<f16_t, 128>(p, batch, max_extend_len, st);
    else launch_extend_g<f16_t, 64>(p, batch, max_extend_len, st);
  }
  MI_CHECK_LAUNCH();
  return MI_OK;
}
