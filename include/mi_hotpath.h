/*
 * mi_hotpath.h -- C ABI of the MI355X (gfx950) hot-path library  libmi_hotpath.so
 *
 * Drop-in boundary for the reference's batched prefill/decode hot path (SURVEY.md
 * section 8b).  Every entry point replaces one native/Triton op the reference's
 * Python calls; the citation after "replaces:" is path:line under /root/reference.
 *
 * Conventions (all functions):
 *   - plain C: raw DEVICE pointers, explicit element strides, int64 sizes, a
 *     hipStream_t passed as void*; no torch / ATen types;
 *   - return 0 on success, <0 on error (MI_ERR_*); mi_last_error() gives the text
 *     (thread-local); nothing is thrown;
 *   - never allocate, never synchronise, never touch the default stream: work is
 *     enqueued on `stream` only, so every call is hipGraph-capturable;
 *   - all buffers (outputs, workspaces) are owned by the caller.
 * dtype codes: MI_BF16 / MI_FP16 for activations and KV; FP8 is OCP e4m3fn (gfx950).
 */
#ifndef MI_HOTPATH_H
#define MI_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 9

enum { MI_BF16 = 0, MI_FP16 = 1, MI_F32 = 2 };
enum {
  MI_OK = 0,
  MI_ERR_INVALID = -1,     /* bad argument (null pointer, bad size/stride/alignment) */
  MI_ERR_UNSUPPORTED = -2, /* shape / dtype combination has no kernel */
  MI_ERR_LAUNCH = -3       /* hipLaunch reported an error */
};
/* scale granularity for mi_fp8_gemm */
enum { MI_SCALE_TENSOR = 0, MI_SCALE_ROW = 1 /* per token (A) / per channel (B) */ };
/* weight packing for mi_w4a16_gemm */
enum { MI_W4_AWQ = 0, MI_W4_GPTQ = 1 };

int mi_abi_version(void);
const char* mi_last_error(void);
/* number of compute units of the current device (used by split heuristics) */
int mi_device_cu_count(void);

/* ---------------------------------------------------------------- integer path */

/* kv_indptr[0]=0, kv_indptr[i+1]=sum_{j<=i} lens[j]  (int32 out; lens int32 or int64).
 * replaces: torch.cumsum at triton_backend.py:173-174,303-306. */
int mi_kv_indptr(const void* lens, int lens_is_i64, int32_t* kv_indptr, int64_t batch, void* stream);

/* Page-granular index tables (page_size > 1, page-aligned allocation -- PagedTokenToKVPoolAllocator,
 * mem_cache/allocator.py:407-543): page_indptr[i+1] = sum_{j<=i} ceil(lens[j] / page_size);
 * page_indices[page_indptr[b] + j] = req_to_token[row_b, j * page_size] / page_size.  One entry per PAGE instead of
 * one per token (the role of kv_indices for mi_decode_attn_paged). */
int mi_kv_page_indptr(const void* lens, int lens_is_i64, int64_t page_size, int32_t* page_indptr, int64_t batch,
                      void* stream);
int mi_kv_page_indices(const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                       const void* lens, int lens_is_i64, const int32_t* page_indptr, int32_t* page_indices,
                       int64_t batch, int64_t page_size, void* stream);

/* kv_indices[kv_indptr[b] + j] = req_to_token[req_pool_indices[b]*stride + start_b + j], j < lens[b]
 * replaces: create_flashinfer_kv_indices_triton, layers/attention/utils.py:5-41 (bit-exact). */
int mi_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                  const int64_t* req_pool_indices, const void* lens, int lens_is_i64,
                  const int32_t* kv_indptr, const int32_t* kv_start_idx /* nullable */,
                  int32_t* kv_indices, int64_t batch, void* stream);

/* k_cache[loc[t]] = k[t]; v_cache[loc[t]] = v[t]   (rows of Hkv*D / Hkv*Dv elements, same dtype)
 * replaces: MHATokenToKVPool.set_kv_buffer index_put, mem_cache/memory_pool.py:454-455. */
int mi_kv_write(void* k_cache, void* v_cache, const int64_t* loc, const void* k, const void* v,
                int64_t tokens, int64_t row_elems_k, int64_t row_elems_v,
                int64_t cache_stride_k, int64_t cache_stride_v, /* elements between slots   */
                int64_t src_stride_k, int64_t src_stride_v,     /* elements between tokens  */
                int dtype, void* stream);

/* The same scatter into an fp8 (e4m3fn, stored as bytes) pool: row = sat(round_T(row * (1/scale))) -> fp8.
 * k_scale / v_scale = 1 means a plain cast (what the reference's Triton backend stores).
 * replaces: MHATokenToKVPool.set_kv_buffer with an fp8 cache dtype, mem_cache/memory_pool.py:432-455
 * (cache_k.div_(k_scale); cache_k.to(fp8); index_put).  Values beyond +-448 saturate (torch: NaN). */
int mi_kv_write_fp8(void* k_cache, void* v_cache, const int64_t* loc, const void* k, const void* v,
                    int64_t tokens, int64_t row_elems_k, int64_t row_elems_v,
                    int64_t cache_stride_k, int64_t cache_stride_v, /* BYTES between slots      */
                    int64_t src_stride_k, int64_t src_stride_v,     /* elements between tokens  */
                    float k_scale, float v_scale, int dtype /* of k, v */, void* stream);

/* Paged slot allocation (scheduler side, page_size >= 1).  Request i receives seq_lens[i] - prefix_lens[i] slot
 * indices at out_indices[sum_{j<i} extend_j ...]: the rest of its old partial page (last_loc[i] + 1 ...), whole
 * new pages, then the head of one more new page; new pages come from the front of free_pages in request order.
 * *ret_value = (num_new_pages << 32) | sum_extend_lens (extend) / num_new_pages (decode): the caller checks it
 * against len(free_pages) and drops that many pages.  scratch: int64 [2 * batch].  All tensors int64.
 * replaces: alloc_extend_kernel / alloc_decode_kernel, mem_cache/allocator.py:278-404 (bit-exact). */
int mi_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                    const int64_t* free_pages, int64_t* out_indices, int64_t* ret_value, int64_t* scratch,
                    int64_t batch, int64_t page_size, void* stream);
int mi_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                    int64_t* out_indices, int64_t* ret_value, int64_t* scratch, int64_t batch,
                    int64_t page_size, void* stream);

/* Scheduler-side request bookkeeping of an EXTEND batch (one launch each, integer, bit-exact).
 *  mi_write_req_to_token: req_to_token[req_pool_indices[i], pre_lens[i] : seq_lens[i]] =
 *      out_cache_loc[sum_{j<i} extend_lens[j] ...]  (int64 slot ids stored as int32).
 *      replaces: write_req_to_token_pool_triton, managers/schedule_batch.py:1848-1882 (call site :1290-1301; the
 *      python fallback loop :1303-1309 is what an unmodified scheduler runs for a non-Triton backend)
 *  mi_get_last_loc: result[i] = prefix_lens[i] > 0 ? req_to_token[req_pool_indices[i], prefix_lens[i] - 1] : -1
 *      replaces: get_last_loc_triton / get_last_loc_torch, managers/schedule_batch.py:1885-1956
 *  mi_compute_position: positions[start_i + j] = prefix_i + j (j < extend_i), extend_start_loc[i] = start_i =
 *      sum_{j<i} extend_j; extend_prefix_lens == NULL means no prefixes.  lens int32 (as ForwardBatch.init_new
 *      builds them) or int64.
 *      replaces: compute_position_triton / compute_position_torch, model_executor/forward_batch_info.py:678-750 */
int mi_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                          const int64_t* pre_lens, const int64_t* seq_lens, const int64_t* extend_lens,
                          const int64_t* out_cache_loc, int64_t batch, void* stream);
int mi_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                    const int64_t* prefix_lens, int64_t* result, int64_t batch, void* stream);
int mi_compute_position(const void* extend_prefix_lens, const void* extend_seq_lens, int lens_is_i64,
                        int64_t* positions, int32_t* extend_start_loc, int64_t batch, void* stream);

/* ------------------------------------------------------------------- attention */

/* bytes of fp32 workspace mi_decode_attn needs for (batch, Hq, Dv, num_splits) */
int64_t mi_decode_attn_workspace_bytes(int64_t batch, int64_t num_q_heads, int64_t v_head_dim,
                                       int64_t num_splits);

/* Split-KV token (decode) attention over a paged KV pool, one query token per request.
 *   q [B,Hq,D] (stride_q_tok elements between tokens, heads contiguous D apart)
 *   k_buf/v_buf [slots,Hkv,D] (stride_*_slot elements between slots, heads D apart; 0 < stride < 2^31:
 *   slot index x stride is one 32 x 32 -> 64-bit multiply per row address)
 *   o [B,Hq,D] ; kv_indptr int32 [B+1] ; kv_indices int32 [kv_indptr[B]]
 *   o[b,h] = softmax_j(sm_scale * q[b,h].k_buf[kv_indices[kv_indptr[b]+j], h/group]) . v_buf[...]
 *   logit_cap > 0 applies cap*tanh(x/cap) to the scaled logits.
 *   num_splits >= 1; workspace needed when num_splits > 1.  split_chunk (multiple of 16, 0 = off): every split
 *   covers split_chunk keys -- balanced workgroups on RAGGED batches, short requests leave trailing splits empty;
 *   a request longer than num_splits * split_chunk falls back to ceil(S / num_splits) keys per split.
 *   work (nullable, needs split_chunk > 0): int32 [num_work][2] = (request, split) pairs to launch, in launch order --
 *   only the non-empty splits of a ragged batch, full chunks first and short remainders last (longest-first
 *   packing of the last round of workgroups); splits not listed must be empty.
 *   plan (nullable, needs work): DEVICE int32 {num_work, num_splits, split_chunk} read by the kernels instead of the
 *   scalar arguments; the launch then covers the whole capacity `num_work` of the list (extra workgroups exit) and
 *   `num_splits` is the workspace stride, so a hipGraph-captured launch follows a plan the host rewrites before each
 *   replay (init_forward_metadata_replay_cuda_graph hands the host-side lengths over, triton_backend.py:544-566).
 * replaces: decode_attention_fwd (stage1 + stage2), triton_ops/decode_attention.py:677-728;
 * oracle: TorchNativeAttnBackend._run_sdpa_forward_decode, torch_native_backend.py:112-180. */
int mi_decode_attn(const void* q, const void* k_buf, const void* v_buf, void* o,
                   const int32_t* kv_indptr, const int32_t* kv_indices, void* workspace,
                   int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                   int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_k_slot,
                   int64_t stride_v_slot, float sm_scale, float logit_cap, int64_t num_splits,
                   int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan, int dtype,
                   void* stream);

/* Decode attention with PAGE-granular indices: token t of request b lives in slot
 * page_indices[page_indptr[b] + t / page_size] * page_size + t % page_size (page_size a power of two); kv_indptr still
 * counts tokens.  Same kernel, same arithmetic and tile order as the token-granular entry points -- identical bits --
 * with 1/page_size of the index traffic.  o_fp8/o_scale, kv8/k_scale/v_scale: the _fp8out / _fp8kv variants in one.
 * replaces: decode over a page_size > 1 pool (the reference expands pages to token indices first,
 * triton_backend.py:173-186 with allocator.py:407-543). */
int mi_decode_attn_paged(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                         void* o_fp8 /* nullable */, const float* o_scale, int kv8, float k_scale, float v_scale,
                         const int32_t* kv_indptr, const int32_t* page_indptr, const int32_t* page_indices,
                         int64_t page_size, void* workspace, int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                         int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_k_slot,
                         int64_t stride_v_slot, float sm_scale, float logit_cap, int64_t num_splits,
                         int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan, int dtype,
                         void* stream);

/* mi_decode_attn with the static per-tensor FP8 quantisation of the FOLLOWING linear (o_proj) fused into the
 * output stage: o_fp8 [B, Hq*D] contiguous = quant(o rounded to `dtype`, *o_scale), bit-identical to
 * mi_decode_attn followed by mi_fp8_quant_per_tensor(mode 1).  `o` may be null (fp8 only).
 * replaces: decode_attention_fwd + the scaled_fp8_quant in front of o_proj (fp8_utils.py:654-674). */
int mi_decode_attn_fp8out(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                          void* o_fp8, const float* o_scale, const int32_t* kv_indptr,
                          const int32_t* kv_indices, void* workspace, int64_t batch, int64_t num_q_heads,
                          int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                          int64_t stride_o_tok, int64_t stride_k_slot, int64_t stride_v_slot,
                          float sm_scale, float logit_cap, int64_t num_splits, int64_t split_chunk,
                          const int32_t* work, int64_t num_work, const int32_t* plan, int dtype, void* stream);

/* mi_decode_attn over an fp8 (e4m3fn) KV pool: k_buf/v_buf hold bytes, stride_*_slot are in BYTES, head_dim 128.
 *   o[b,h] = v_scale * softmax_j(sm_scale * k_scale * q[b,h] . k8[...]) . v8[...]   (K is converted up to the q
 *   dtype for the MFMA, V to fp32; P stays fp32).  o and/or o_fp8 (+ o_scale) as in mi_decode_attn_fp8out.
 * replaces: decode attention over an fp8 MHATokenToKVPool (memory_pool.py:113-117,389-405) with the k/v scale
 * convention of flashinfer_backend.py:474-555; halves the algorithmic bytes of the decode step. */
int mi_decode_attn_fp8kv(const void* q, const void* k_buf, const void* v_buf, void* o /* nullable */,
                         void* o_fp8 /* nullable */, const float* o_scale, float k_scale, float v_scale,
                         const int32_t* kv_indptr, const int32_t* kv_indices, void* workspace,
                         int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                         int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_k_slot,
                         int64_t stride_v_slot, float sm_scale, float logit_cap, int64_t num_splits,
                         int64_t split_chunk, const int32_t* work, int64_t num_work, const int32_t* plan, int dtype,
                         void* stream);

/* Ragged extend (prefill-with-prefix) attention.
 *   q_ext [E,Hq,D], k_ext/v_ext [E,Hkv,D] : the new tokens, request i owns rows
 *   qo_indptr[i]..qo_indptr[i+1]; its cached prefix is kv_indices[kv_indptr[i]..kv_indptr[i+1])
 *   into k_buf/v_buf.  Row j of request i attends the whole prefix plus new tokens 0..j
 *   (causal) or all new tokens (non-causal).  sliding_window > 0 additionally requires
 *   q_pos <= k_pos + sliding_window.
 * replaces: extend_attention_fwd, triton_ops/extend_attention.py:306-438;
 * oracle: TorchNativeAttnBackend._run_sdpa_forward_extend, torch_native_backend.py:27-110. */
int mi_extend_attn(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                   const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                   const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                   int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                   int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                   int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                   int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                   int64_t sliding_window, int dtype, void* stream);

/* mi_extend_attn over a page-aligned pool (page_size a power of two; PagedTokenToKVPoolAllocator, allocator.py:407-543):
 * page_indices[page_indptr[i] .. page_indptr[i+1]) = the page ids of request i's cached prefix (mi_kv_page_indptr /
 * mi_kv_page_indices on the prefix lengths); prefix key j is slot page * page_size + j % page_size.  The long-extend
 * kernel reads one index per page; shapes it does not take run on kv_indices (must be given too).  Same bits as
 * mi_extend_attn.  replaces: the prefix loop of extend_attention_fwd (extend_attention.py:118-196) on a paged pool. */
int mi_extend_attn_paged(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                         const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                         const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* page_indptr,
                         const int32_t* page_indices, int64_t page_size, int64_t batch, int64_t max_extend_len,
                         int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                         int64_t stride_o_tok, int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                         int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                         int64_t sliding_window, int dtype, void* stream);

/* mi_extend_attn / mi_extend_attn_paged with the output ALSO (o_ext given) or ONLY (o_ext null) as e4m3fn for the FP8
 * linear that follows -- prefill's form of mi_decode_attn_fp8out: o_fp8 [total_tokens, num_q_heads * head_dim] contiguous =
 * mi_fp8_quant_per_tensor(mode 1, *o_scale) of the T-typed result, bit for bit.  The long-extend kernel (head_dim 128,
 * bf16 / fp16 pool, max_extend_len >= 64, no window / cap) writes it from its epilogue; every other shape runs the
 * T-typed kernel into o_ext (then required: MI_ERR_INVALID without it) and quantises in a second launch.  page_indptr /
 * page_indices: both null (token-granular prefix) or both given (as mi_extend_attn_paged).
 * q_positions [total_tokens] + cos_sin_cache_t (the rotary cache [max_pos][128] rounded to T; both or neither, long-extend
 * kernel only, MI_ERR_UNSUPPORTED otherwise): q_ext is UNROTATED and NeoX RoPE is applied to Q as it is loaded (the
 * caller then rotates only k: mi_rope_neox with num_q_heads = 0); same bits as mi_rope_neox on q first.
 * replaces: extend_attention_fwd (extend_attention.py:306-438) followed by static_quant_fp8 in apply_fp8_linear
 * (fp8_utils.py:654-660) of the o_proj RowParallelLinear (models/llama.py:186-190). */
int mi_extend_attn_fp8out(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext /* nullable */,
                          void* o_fp8, const float* o_scale, const void* k_buf, const void* v_buf,
                          const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                          const int32_t* page_indptr /* nullable */, const int32_t* page_indices /* nullable */,
                          int64_t page_size, int64_t batch, int64_t total_tokens, int64_t max_extend_len,
                          int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t stride_q_tok,
                          int64_t stride_o_tok, int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                          int64_t stride_v_slot, float sm_scale, float logit_cap, int causal, int64_t sliding_window,
                          const int64_t* q_positions /* nullable */, const void* cos_sin_cache_t /* nullable */,
                          int dtype, void* stream);

/* mi_extend_attn whose cached PREFIX lives in an fp8 (e4m3fn) pool: k_buf8/v_buf8 hold bytes, stride_*_slot in
 * BYTES, head_dim 128; prefix keys are k8 * k_scale, prefix values v8 * v_scale (converted while being staged);
 * the new tokens k_ext/v_ext stay T-typed.  replaces: extend_attention_fwd over an fp8 MHATokenToKVPool. */
int mi_extend_attn_fp8kv(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                         const void* k_buf8, const void* v_buf8, float k_scale, float v_scale,
                         const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                         int64_t batch, int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads,
                         int64_t head_dim, int64_t stride_q_tok, int64_t stride_o_tok,
                         int64_t stride_kx_tok, int64_t stride_vx_tok, int64_t stride_k_slot,
                         int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                         int64_t sliding_window, int dtype, void* stream);

/* The general extend call -- bf16/fp16 or fp8 (kv_fp8 != 0: k_scale / v_scale as in mi_extend_attn_fp8kv) pool, optional
 * tree mask (custom_mask nullable; when given it replaces the causal rule as in mi_extend_attn_masked) -- with SPLIT-KV:
 * the key range of every query block is walked by num_splits (1..64) workgroups and merged, for short extends over
 * long prefixes (speculative verify, chunk tails), where the unsplit launch has only batch * heads workgroups.
 * workspace: mi_decode_attn_workspace_bytes(total_tokens, num_q_heads, head_dim, num_splits) bytes (nullable when
 * num_splits == 1); total_tokens = qo_indptr[batch].  Same math as the unsplit call up to fp32 reassociation.
 * replaces: extend_attention_fwd, triton_ops/extend_attention.py:306-438 (which has no split form: the reference
 * runs verify batches with one program per (sequence, head, 64-row block)). */
int mi_extend_attn_splitkv(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext, const void* k_buf,
                           const void* v_buf, int kv_fp8, float k_scale, float v_scale, const int32_t* qo_indptr,
                           const int32_t* kv_indptr, const int32_t* kv_indices, const uint8_t* custom_mask,
                           const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t batch,
                           int64_t max_extend_len, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                           int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_kx_tok, int64_t stride_vx_tok,
                           int64_t stride_k_slot, int64_t stride_v_slot, float sm_scale, float logit_cap, int causal,
                           int64_t sliding_window, void* workspace, int64_t total_tokens, int64_t num_splits, int dtype,
                           void* stream);

/* mi_extend_attn with a per-request visibility mask on the NEW-token keys instead of the causal rule
 * (speculative-decode tree verification): request i's mask is a row-major byte matrix
 * [ext_len_i, prefix_i + ext_len_i] at custom_mask + mask_indptr[i] (non-zero = visible); prefix keys are all
 * visible when skip_prefix_custom_mask (the reference's default), else masked by the same matrix.
 * replaces: extend_attention_fwd(custom_mask, mask_indptr, skip_prefix_custom_mask),
 * triton_ops/extend_attention.py:93-94,168-178,245-257,306-438 as called for TARGET_VERIFY,
 * triton_backend.py:226-263,632-685. */
int mi_extend_attn_masked(const void* q_ext, const void* k_ext, const void* v_ext, void* o_ext,
                          const void* k_buf, const void* v_buf, const int32_t* qo_indptr,
                          const int32_t* kv_indptr, const int32_t* kv_indices,
                          const uint8_t* custom_mask, const int64_t* mask_indptr,
                          int skip_prefix_custom_mask, int64_t batch, int64_t max_extend_len,
                          int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                          int64_t stride_q_tok, int64_t stride_o_tok, int64_t stride_kx_tok,
                          int64_t stride_vx_tok, int64_t stride_k_slot, int64_t stride_v_slot,
                          float sm_scale, float logit_cap, int64_t sliding_window, int dtype,
                          void* stream);

/* out = (a*e^{la} + b*e^{lb}) / (e^{la}+e^{lb}); out_lse = log(e^{la}+e^{lb})  (lse fp32, may be null)
 *   a,b,out [n,h,d] contiguous; lse [n,h].
 * replaces: merge_state_v2, sgl-kernel/csrc/attention/merge_attn_states.cu:31-106,182-204. */
int mi_merge_state(const void* o_a, const float* lse_a, const void* o_b, const float* lse_b,
                   void* out, float* out_lse, int64_t n, int64_t h, int64_t d, int dtype,
                   void* stream);

/* ------------------------------------------------------------- FP8 (OCP e4m3fn) */

/* Per-tensor quant.  x [M,K] row stride ldx elements, q contiguous [M,K]; `mode`:
 *   0 dynamic activation : scale[0] = absmax(x)/448 (written), q = sat(x * (1/scale))
 *   1 static             : scale given,                         q = sat(x * (1/scale))
 *   2 dynamic weight     : q = sat(x * (448/max(absmax,1e-12))), scale[0] = 1/(448/amax) (written)
 *   Modes 0/2 are two launches (per-block maxima into an internal 8-KB scratch, then reduce +
 *   quantise; no atomics, no memset); calls must be stream-ordered with respect to each other.
 * replaces: 0/1 sgl_per_tensor_quant_fp8, sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:96-123
 * (and vllm ops.scaled_fp8_quant per-tensor as called at fp8_utils.py:669-674);
 * 2 input_to_float8, quantization/fp8_utils.py:310-326 (weights of bf16 checkpoints, fp8.py:359). */
int mi_fp8_quant_per_tensor(const void* x, void* q, float* scale, int64_t M, int64_t K,
                            int64_t ldx, int mode, int dtype, void* stream);

/* Per-token quant: scales[m] = absmax(x[m,:])/448.
 * replaces: sgl_per_token_quant_fp8, sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:78-106. */
int mi_fp8_quant_per_token(const void* x, void* q, float* scales, int64_t M, int64_t K,
                           int64_t ldx, int dtype, void* stream);

/* out[M,N] = (A[M,K] . B[K,N]) * sa * sb (+ bias) -> out_dtype, fp32 accumulate.
 *   a   fp8 [M,K] row-major (lda), b_nk fp8 stored [N,K] row-major (ldb) == the column-major
 *   [K,N] view the reference passes (fp8.py:364,406 `weight.t()`).
 *   scale_a: 1 value (MI_SCALE_TENSOR) or M values (MI_SCALE_ROW); scale_b: 1 or N values.
 * replaces: torch._scaled_mm at fp8_utils.py:715-723 / fallback :479-507, and
 * fp8_scaled_mm, sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146. */
int mi_fp8_gemm(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                const void* bias /* nullable, out_dtype */, void* out, int64_t M, int64_t N,
                int64_t K, int64_t lda, int64_t ldb, int64_t ldo, int scale_a_mode,
                int scale_b_mode, int out_dtype, void* workspace /* nullable */,
                int64_t workspace_bytes, void* stream);
/* bytes of split-K scratch mi_fp8_gemm can use for this shape (0: none needed).  Passing less
 * (or null) is always correct, only slower for small-N decode shapes. */
int64_t mi_fp8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* -------------------------------------------------------- int4 weight-only GEMM */

/* One-time weight repack (called from process_weights_after_loading) of a 4-bit checkpoint into
 * the MFMA-native layout consumed by mi_w4a16_gemm:
 *   MI_W4_AWQ : qweight [K,N/8] i32 (logical column 8c+j is nibble {0,4,1,5,2,6,3,7}[j] of word
 *               c), qzeros [K/g,N/8] packed the same way, scales [K/g,N]; W = (w - z) * s
 *   MI_W4_GPTQ: qweight [K/8,N] i32 (packed along K), qzeros [K/g,N/8] (sequential nibbles,
 *               stored minus one), scales [K/g,N]; W = (w - (z+1)) * s; `perm` (i32 [K], nullable) = argsort(g_idx)
 *               for act-order checkpoints: native row k' holds checkpoint row perm[k'].
 *   out: qw_native u32 [N/16][K/128][64][4] (N*K/2 bytes), zs_native u32 [K/g][N].
 * replaces: the weight layout prepared in AWQLinearMethod.process_weights_after_loading,
 * quantization/awq.py:183-186 (and vllm's gptq shuffle). */
int mi_w4_repack(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                 const int32_t* perm /* nullable */, void* qw_native, void* zs_native, int64_t N,
                 int64_t K, int64_t group_size, int layout, int dtype, void* stream);

/* out[M,N] = x[M,K] . dequant(W)[K,N] (+ bias); dequant W = (w - z) * s is fused into the MFMA
 * main loop and is bit-identical to the reference's dequantised fp16/bf16 weight.
 *   x [M,K] (ldx), qw_native/zs_native from mi_w4_repack, perm as given to mi_w4_repack.
 * replaces: awq_dequantize + torch.matmul, quantization/awq.py:199-203 with
 * sgl-kernel/csrc/gemm/awq_kernel.cu:126-221; GPTQ: vllm gptq_gemm (parity unpinned). */
int mi_w4a16_gemm(const void* x, const void* qw_native, const void* zs_native,
                  const int32_t* perm /* nullable */, const void* bias /* nullable */, void* out,
                  int64_t M, int64_t N, int64_t K, int64_t group_size, int64_t ldx, int64_t ldo,
                  int dtype, void* workspace /* nullable */, int64_t workspace_bytes, void* stream);
/* bytes of split-K scratch mi_w4a16_gemm can use (0: none needed); less is correct but slower */
int64_t mi_w4a16_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* out[m][k'] = x[m][perm[k']] (2-byte elements): the activations of a GPTQ act-order layer in the native k order of
 * mi_w4_repack, so that mi_w4a16_gemm can be called with perm == NULL (the tile kernel of prefill batches needs that;
 * decode batches may pass perm instead).  K % 4 == 0, ldo % 4 == 0.
 * replaces: the `x[:, g_idx_sort_indices]` reorder of vllm's gptq path (absent here; AutoGPTQ convention). */
int mi_gather_columns(const void* x, const int32_t* perm, void* out, int64_t M, int64_t K, int64_t ldx, int64_t ldo,
                      void* stream);

/* W^T [N,K] (K contiguous) = dequant(native-layout weights) in `dtype` -- for inspection and tests; no product path
 * materialises it (prefill batches run the fused tile kernel inside mi_w4a16_gemm).  Same bits as the fused kernels'
 * fragments.  With GPTQ act-order the rows are the PERMUTED ones.
 * replaces: awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:186-221 (on our load-time layout). */
int mi_w4_dequantize_native(const void* qw_native, const void* zs_native, void* w_nk, int64_t N, int64_t K,
                            int64_t group_size, int dtype, void* stream);

/* W[K,N] = dequant(checkpoint-layout qweight) in `dtype` (unfused form, for tests and tools).
 * replaces: awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:186-221. */
int mi_w4_dequantize(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                     const int32_t* g_idx /* nullable */, void* w_out, int64_t N, int64_t K,
                     int64_t group_size, int layout, int dtype, void* stream);

/* --------------------------------------------------- TP all-reduce over xGMI (hipIpc peers) */

/* Native 1-stage / 2-stage all-reduce for the latency-bound decode messages ([tokens, hidden],
 * <= max_bytes), one process per GPU.  Every rank owns one shared allocation
 * [signals | staging | tmp] whose hipIpc handle all peers open.  Setup functions (alloc / ipc /
 * create / destroy / error) are init-time and DO allocate or synchronise; mi_ar_all_reduce follows
 * the usual rules (stream-ordered, capturable, no allocation).  Sums are fp32 in rank order, so all
 * ranks get identical bits.  Barrier spins are bounded: a missing peer sets an error flag
 * (mi_ar_error) instead of hanging the GPU.
 * replaces: init_custom_ar / register_buffer / all_reduce_unreg / dispose / meta_size,
 * sgl-kernel/include/sgl_kernel_ops.h:51-68 with csrc/allreduce/custom_all_reduce.hip and
 * custom_all_reduce_hip.cuh:155-345,541-551 (policy); Python side custom_all_reduce.py:360-497. */
int64_t mi_ar_shared_bytes(int64_t max_bytes);
int mi_ar_alloc_shared(int64_t bytes, void** ptr);       /* uncached device memory, zeroed */
int mi_ar_free_shared(void* ptr);
int mi_ar_ipc_get(void* ptr, void* handle64);            /* 64-byte hipIpcMemHandle_t */
int mi_ar_ipc_open(const void* handle64, void** ptr);
int mi_ar_ipc_close(void* ptr);
void* mi_ar_create(void** shared_ptrs /* [world], own entry = local pointer */, int64_t max_bytes,
                   int rank, int world);
int mi_ar_destroy(void* ctx);
int mi_ar_error(void* ctx);                               /* 0 ok, 1 a barrier timed out */
int mi_ar_all_reduce(void* ctx, const void* inp, void* out, int64_t bytes, int dtype, void* stream);
/* The rank's own IPC-mapped staging buffer (max_bytes).  A producer that writes its output there and passes this
 * pointer as `inp` skips the staging copy.
 * replaces: register_buffer / get_graph_buffer_ipc_meta / register_graph_buffers, sgl_kernel_ops.h:58-68 with
 * custom_all_reduce.py:387-412 (one persistent registered buffer instead of re-registration after capture). */
void* mi_ar_staging(void* ctx);
/* All-reduce of inp [rows, H] fused with its consumer: x32 = sum_ranks(inp) (rounded to T) + residual;
 * residual <- x32 (in place, nullable); out = x32 * rsqrt(mean(x32^2) + eps) * weight (T, nullable); q_out (nullable)
 * = fp8 e4m3fn of out with the static per-tensor *q_scale.  Bit-identical to mi_ar_all_reduce followed by
 * mi_rmsnorm / mi_rmsnorm_fp8; the reduced tensor is never written.  bf16 / fp16; rows*H*2 <= max_bytes.
 * replaces: tensor_model_parallel_all_reduce + RMSNorm, layers/communicator.py with
 * layers/flashinfer_comm_fusion.py (fused allreduce + residual + rmsnorm) and layers/layernorm.py:128-146. */
int mi_ar_all_reduce_add_rmsnorm(void* ctx, const void* inp, void* residual /* nullable, in/out */,
                                 const void* weight, void* out /* nullable */, void* q_out /* nullable */,
                                 const float* q_scale, int64_t rows, int64_t H, int64_t ldr, int64_t ldo, float eps,
                                 int dtype, void* stream);

/* ------------------------------------------ layer glue (SURVEY 8f "next" rows 1-2) */

/* (fused add +) RMSNorm, one row per token: x32 = x (+ residual); residual <- x32 (in place,
 * nullable); out = x32 * rsqrt(mean(x32^2)+eps) * weight.
 * replaces: RMSNorm.forward_native, layers/layernorm.py:128-146 (sgl-kernel
 * fused_add_rmsnorm / rmsnorm, csrc/elementwise/fused_add_rms_norm_kernel.cu). */
int mi_rmsnorm(const void* x, void* residual /* nullable, in/out */, const void* weight, void* out,
               int64_t M, int64_t H, int64_t ldx, int64_t ldr, int64_t ldo, float eps, int dtype,
               void* stream);

/* The same two producers with a fused static per-tensor FP8 output: q_out fp8
 * [M,H] / [M,I] contiguous = quantise(T-rounded result, *q_scale), bit-identical to running
 * mi_fp8_quant_per_tensor(mode 1) on `out`; `out` (nullable) is still written when given.
 * replaces: norm/activation + scaled_fp8_quant pairs in front of an FP8 linear (fp8_utils.py:654-674). */
int mi_rmsnorm_fp8(const void* x, void* residual, const void* weight, void* out /* nullable */,
                   void* q_out, const float* q_scale, int64_t M, int64_t H, int64_t ldx, int64_t ldr,
                   int64_t ldo, float eps, int dtype, void* stream);
int mi_silu_and_mul_fp8(const void* x, void* out /* nullable */, void* q_out, const float* q_scale,
                        int64_t M, int64_t I, int64_t ldx, int64_t ldo, int dtype, void* stream);

/* In-place NeoX rotary embedding on q [tokens,Hq,D] and k [tokens,Hkv,D] (rotary_dim == D);
 * cos_sin_cache fp32 [max_pos, D] = [cos(D/2) | sin(D/2)], positions int64 [tokens].
 * replaces: RotaryEmbedding.forward_native, layers/rotary_embedding.py:49-74,138-166
 * (sgl-kernel csrc/elementwise/rope.cu). */
int mi_rope_neox(void* q, void* k, const int64_t* positions, const float* cos_sin_cache,
                 int64_t tokens, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                 int64_t ldq, int64_t ldk, int dtype, void* stream);

/* out[M,I] = silu(x[:, :I]) * x[:, I:2I].
 * replaces: SiluAndMul.forward_native, layers/activation.py:56-58 (csrc/elementwise/activation.cu). */
int mi_silu_and_mul(const void* x, void* out, int64_t M, int64_t I, int64_t ldx, int64_t ldo,
                    int dtype, void* stream);

/* ------------------------- decode-shaped FP8 linears fused with their consumer
 *
 * M <= 512 (beyond one GEMM pass -- 256 rows, 128 when N < 1024 -- run per chunk through the same workspace; every
 * row comes out exactly as from a <= 128-row call),
 * K % 128 == 0, per-tensor scales.  The GEMM leaves raw fp32 split-K partials in `workspace`
 * (mi_fp8_gemm_fused_workspace_bytes(M,N,K) bytes, 16-byte aligned) and ONE consumer kernel sums them,
 * applies the GEMM epilogue x = round_T(acc * sa * sb) and the next op(s) of the decoder layer.  Every
 * rounding of the unfused call sequence is reproduced: results are bit-identical to it (tests/test_fused_gpu.py).
 * Each call replaces 3-5 launches of the unfused sequence at decode batch sizes. */
int64_t mi_fp8_gemm_fused_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* x = a.b (o_proj / down_proj);  x32 = x (+ residual, updated in place, nullable);
 * y = rmsnorm(x32) * norm_weight -> out (T, nullable) and/or q_out = fp8(y / *q_scale) (nullable).
 * replaces: apply_fp8_linear (fp8_utils.py:715-723) + RMSNorm.forward_native with residual
 * (layernorm.py:128-146) + the static scaled_fp8_quant of the next linear (fp8_utils.py:654-658). */
int mi_fp8_gemm_add_rmsnorm_fp8(const void* a, const void* b_nk, const float* scale_a,
                                const float* scale_b, void* residual, const void* norm_weight,
                                void* out, void* q_out, const float* q_scale, int64_t M, int64_t N,
                                int64_t K, int64_t lda, int64_t ldb, float eps, int dtype,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* qkv = a.b_nk with N = (Hq + 2*Hkv)*D; NeoX RoPE on q and k; q -> q_out [M, Hq*D] (row stride ldq);
 * k -> k_cache[loc[t]], v -> v_cache[loc[t]] (slot strides in elements).
 * replaces: apply_fp8_linear + RotaryEmbedding.forward_native (rotary_embedding.py:49-166) +
 * MHATokenToKVPool.set_kv_buffer (memory_pool.py:454-455). */
int mi_fp8_gemm_rope_kvwrite(const void* a, const void* b_nk, const float* scale_a,
                             const float* scale_b, const int64_t* positions,
                             const float* cos_sin_cache, void* q_out, void* k_cache, void* v_cache,
                             const int64_t* loc, int64_t M, int64_t num_q_heads, int64_t num_kv_heads,
                             int64_t head_dim, int64_t K, int64_t lda, int64_t ldb, int64_t ldq,
                             int64_t cache_stride_k, int64_t cache_stride_v, int dtype,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* gate_up = a.b_nk with N = 2*I; q_out [M, I] = fp8(round_T(silu(gate) * up) / *q_scale).  When the GEMM
 * needs no split-K (N large enough to fill the chip) the activation runs in the GEMM's own epilogue and
 * the [M, 2I] intermediate never exists; otherwise through the workspace as above.
 * replaces: apply_fp8_linear + SiluAndMul.forward_native (activation.py:56-58) + static scaled_fp8_quant. */
int mi_fp8_gemm_silu_mul_fp8(const void* a, const void* b_nk, const float* scale_a,
                             const float* scale_b, void* q_out, const float* q_scale, int64_t M,
                             int64_t I, int64_t K, int64_t lda, int64_t ldb, int dtype,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------- decode-shaped int4 (AWQ / GPTQ without act-order) linears fused with their consumer
 *
 * M <= 128, K % 128 == 0, N % 64 == 0, group_size % 128 == 0; 16-bit activations in and out.  The GEMM leaves raw fp32
 * split-K partials in `workspace` (mi_w4a16_fused_workspace_bytes() bytes, 16-byte aligned) and ONE consumer kernel sums
 * them, rounds to T (what mi_w4a16_gemm returns without a bias) and applies the next op(s) of the decoder layer:
 * bit-identical to mi_w4a16_gemm followed by the unfused kernels.
 * replaces: AWQLinearMethod.apply (awq.py:199-203: awq_dequantize + torch.matmul) / vllm gptq_gemm, followed by
 * RMSNorm.forward_native with residual (layernorm.py:128-146); RotaryEmbedding.forward_native
 * (rotary_embedding.py:49-166) + set_kv_buffer (memory_pool.py:454-455); SiluAndMul.forward_native (activation.py:56-58). */
int64_t mi_w4a16_fused_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t group_size);
int mi_w4a16_gemm_add_rmsnorm(const void* x, const void* qw_native, const void* zs_native, void* residual,
                              const void* norm_weight, void* out, int64_t M, int64_t N, int64_t K, int64_t group_size,
                              int64_t ldx, float eps, int dtype, void* workspace, int64_t workspace_bytes, void* stream);
int mi_w4a16_gemm_rope_kvwrite(const void* x, const void* qw_native, const void* zs_native, const int64_t* positions,
                               const float* cos_sin_cache, void* q_out, void* k_cache, void* v_cache, const int64_t* loc,
                               int64_t M, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim, int64_t K,
                               int64_t group_size, int64_t ldx, int64_t ldq, int64_t cache_stride_k, int64_t cache_stride_v,
                               int dtype, void* workspace, int64_t workspace_bytes, void* stream);
int mi_w4a16_gemm_silu_mul(const void* x, const void* qw_native, const void* zs_native, void* out, int64_t M, int64_t I,
                           int64_t K, int64_t group_size, int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_HOTPATH_H */
