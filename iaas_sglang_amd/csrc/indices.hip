// Integer / byte-moving kernels of the hot path: kv_indptr scan, kv_indices gather,
// KV-pool scatter.  All bit-exact; HBM/latency-bound, no MFMA.
#include "common.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" int mi_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

// ---------------------------------------------------------------- kv_indptr
// One workgroup of 1024 threads scans up to any batch in chunks (batch is <= a few thousand).
template <typename L>
__global__ __launch_bounds__(1024) void kv_indptr_kernel(const L* __restrict__ lens,
                                                         int32_t* __restrict__ indptr, int64_t batch, int32_t page = 1) {
  __shared__ int32_t wave_sum[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    indptr[0] = 0;
  }
  __syncthreads();
  for (int64_t base = 0; base < batch; base += 1024) {
    int64_t i = base + tid;
    int32_t v = (i < batch) ? (int32_t)lens[i] : 0;
    if (page > 1) v = (v + page - 1) / page;   // pages instead of tokens
    int32_t x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    int32_t prefix = carry_s;
    for (int w = 0; w < wave; ++w) prefix += wave_sum[w];
    if (i < batch) indptr[i + 1] = prefix + x;
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
}

extern "C" int mi_kv_indptr(const void* lens, int lens_is_i64, int32_t* kv_indptr, int64_t batch,
                            void* stream) {
  MI_CHECK_ARG(kv_indptr != nullptr && batch >= 0);
  MI_CHECK_ARG(batch == 0 || lens != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (lens_is_i64)
    kv_indptr_kernel<int64_t><<<1, 1024, 0, s>>>((const int64_t*)lens, kv_indptr, batch);
  else
    kv_indptr_kernel<int32_t><<<1, 1024, 0, s>>>((const int32_t*)lens, kv_indptr, batch);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// page_indptr[i+1] = sum_{j<=i} ceil(lens[j] / page_size): where request i's page ids start in the page-index array
extern "C" int mi_kv_page_indptr(const void* lens, int lens_is_i64, int64_t page_size, int32_t* page_indptr,
                                 int64_t batch, void* stream) {
  MI_CHECK_ARG(page_indptr != nullptr && batch >= 0 && page_size >= 1 && page_size <= (1 << 20));
  MI_CHECK_ARG(batch == 0 || lens != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (lens_is_i64)
    kv_indptr_kernel<int64_t><<<1, 1024, 0, s>>>((const int64_t*)lens, page_indptr, batch, (int32_t)page_size);
  else
    kv_indptr_kernel<int32_t><<<1, 1024, 0, s>>>((const int32_t*)lens, page_indptr, batch, (int32_t)page_size);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// page_indices[page_indptr[b] + j] = req_to_token[row_b, j * page] / page: the page id of the request's j-th page
// (a paged allocator hands out page-aligned runs: allocator.py:407-543, so the first slot of a page names it)
template <typename L>
__global__ __launch_bounds__(256) void kv_page_indices_kernel(const int32_t* __restrict__ req_to_token, int64_t stride,
                                                              const int64_t* __restrict__ req_pool_indices,
                                                              const L* __restrict__ lens,
                                                              const int32_t* __restrict__ page_indptr,
                                                              int32_t* __restrict__ page_indices, int32_t page) {
  const int b = blockIdx.y;
  const int64_t npages = ((int64_t)lens[b] + page - 1) / page;
  const int32_t* src = req_to_token + req_pool_indices[b] * stride;
  int32_t* dst = page_indices + page_indptr[b];
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < npages; j += (int64_t)gridDim.x * 256)
    dst[j] = src[j * page] / page;
}

extern "C" int mi_kv_page_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                                  const int64_t* req_pool_indices, const void* lens, int lens_is_i64,
                                  const int32_t* page_indptr, int32_t* page_indices, int64_t batch, int64_t page_size,
                                  void* stream) {
  MI_CHECK_ARG(batch >= 0 && batch <= 65535 && page_size >= 1 && page_size <= (1 << 20));
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(req_to_token && req_pool_indices && lens && page_indptr && page_indices && req_to_token_stride > 0);
  hipStream_t s = (hipStream_t)stream;
  int chunks = (int)(cdiv64(cdiv64(req_to_token_stride, page_size), 256) < 8 ? cdiv64(cdiv64(req_to_token_stride, page_size), 256) : 8);
  if (chunks < 1) chunks = 1;
  dim3 grid(chunks, (unsigned)batch);
  if (lens_is_i64)
    kv_page_indices_kernel<int64_t><<<grid, 256, 0, s>>>(req_to_token, req_to_token_stride, req_pool_indices,
                                                         (const int64_t*)lens, page_indptr, page_indices, (int32_t)page_size);
  else
    kv_page_indices_kernel<int32_t><<<grid, 256, 0, s>>>(req_to_token, req_to_token_stride, req_pool_indices,
                                                         (const int32_t*)lens, page_indptr, page_indices, (int32_t)page_size);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---------------------------------------------------------------- kv_indices
// grid (chunks, batch): each workgroup copies a 1024-int slice of one request's row.
template <typename L>
__global__ __launch_bounds__(256) void kv_indices_kernel(const int32_t* __restrict__ req_to_token,
                                                         int64_t stride,
                                                         const int64_t* __restrict__ req_pool_indices,
                                                         const L* __restrict__ lens,
                                                         const int32_t* __restrict__ kv_indptr,
                                                         const int32_t* __restrict__ kv_start_idx,
                                                         int32_t* __restrict__ kv_indices) {
  const int b = blockIdx.y;
  const int64_t len = (int64_t)lens[b];
  const int64_t start = kv_start_idx ? (int64_t)kv_start_idx[b] : 0;
  const int32_t* src = req_to_token + req_pool_indices[b] * stride + start;
  int32_t* dst = kv_indices + kv_indptr[b];
  for (int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x; j < len; j += (int64_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t jj = j + u * 256;
      if (jj < len) dst[jj] = src[jj];
    }
  }
}

extern "C" int mi_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                             const int64_t* req_pool_indices, const void* lens, int lens_is_i64,
                             const int32_t* kv_indptr, const int32_t* kv_start_idx,
                             int32_t* kv_indices, int64_t batch, void* stream) {
  MI_CHECK_ARG(batch >= 0 && batch <= 65535);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(req_to_token && req_pool_indices && lens && kv_indptr && kv_indices);
  MI_CHECK_ARG(req_to_token_stride > 0);
  hipStream_t s = (hipStream_t)stream;
  // enough chunks that a 128 x 2048 batch fills the chip; longer rows grid-stride.
  int chunks = (int)(cdiv64(req_to_token_stride, 1024) < 8 ? cdiv64(req_to_token_stride, 1024) : 8);
  if (chunks < 1) chunks = 1;
  dim3 grid(chunks, (unsigned)batch);
  if (lens_is_i64)
    kv_indices_kernel<int64_t><<<grid, 256, 0, s>>>(req_to_token, req_to_token_stride, req_pool_indices,
                                                    (const int64_t*)lens, kv_indptr, kv_start_idx,
                                                    kv_indices);
  else
    kv_indices_kernel<int32_t><<<grid, 256, 0, s>>>(req_to_token, req_to_token_stride, req_pool_indices,
                                                    (const int32_t*)lens, kv_indptr, kv_start_idx,
                                                    kv_indices);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---------------------------------------------------------------- kv_write
// One wave per (token, k|v) row; 16 B per lane when rows are 16-B granular, else 2 B elements.
__global__ __launch_bounds__(256) void kv_write_kernel(uint16_t* __restrict__ k_cache,
                                                       uint16_t* __restrict__ v_cache,
                                                       const int64_t* __restrict__ loc,
                                                       const uint16_t* __restrict__ k,
                                                       const uint16_t* __restrict__ v, int64_t tokens,
                                                       int64_t row_k, int64_t row_v, int64_t cs_k,
                                                       int64_t cs_v, int64_t ss_k, int64_t ss_v,
                                                       int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= tokens * 2) return;
  const int64_t t = w >> 1;
  const bool is_v = w & 1;
  const int64_t slot = loc[t];
  const uint16_t* src = is_v ? v + t * ss_v : k + t * ss_k;
  uint16_t* dst = is_v ? v_cache + slot * cs_v : k_cache + slot * cs_k;
  const int64_t n = is_v ? row_v : row_k;
  if (vec_ok) {
    const uint4* s4 = (const uint4*)src;
    uint4* d4 = (uint4*)dst;
    for (int64_t i = lane; i < n / 8; i += 64) d4[i] = s4[i];
  } else {
    for (int64_t i = lane; i < n; i += 64) dst[i] = src[i];
  }
}

extern "C" int mi_kv_write(void* k_cache, void* v_cache, const int64_t* loc, const void* k,
                           const void* v, int64_t tokens, int64_t row_elems_k, int64_t row_elems_v,
                           int64_t cache_stride_k, int64_t cache_stride_v, int64_t src_stride_k,
                           int64_t src_stride_v, int dtype, void* stream) {
  MI_CHECK_ARG(tokens >= 0);
  if (tokens == 0) return MI_OK;
  MI_CHECK_ARG(k_cache && v_cache && loc && k && v);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  MI_CHECK_ARG(row_elems_k > 0 && row_elems_v > 0);
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  int vec_ok = (row_elems_k % 8 == 0) && (row_elems_v % 8 == 0) && (cache_stride_k % 8 == 0) &&
               (cache_stride_v % 8 == 0) && (src_stride_k % 8 == 0) && (src_stride_v % 8 == 0) &&
               al16(k_cache) && al16(v_cache) && al16(k) && al16(v);
  int64_t waves = tokens * 2;
  kv_write_kernel<<<(unsigned)cdiv64(waves, 4), 256, 0, (hipStream_t)stream>>>(
      (uint16_t*)k_cache, (uint16_t*)v_cache, loc, (const uint16_t*)k, (const uint16_t*)v, tokens,
      row_elems_k, row_elems_v, cache_stride_k, cache_stride_v, src_stride_k, src_stride_v, vec_ok);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---------------------------------------------------------------- kv_write into an fp8 (e4m3fn) pool
// set_kv_buffer for an fp8 KV cache (memory_pool.py:432-440): cache_k.div_(k_scale) on the T-typed rows
// (on the GPU torch multiplies by the fp32 reciprocal, BinaryDivTrueKernel.cu; the result is rounded to T),
// then .to(fp8).  Deviation: values beyond +-448 saturate (torch's cast produces NaN).
// One wave per (token, k|v) row, 8 elements (16 B in, 8 B out) per lane.
template <typename T>
__global__ __launch_bounds__(256) void kv_write_fp8_kernel(uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache,
                                                           const int64_t* __restrict__ loc, const T* __restrict__ k,
                                                           const T* __restrict__ v, int64_t tokens, int64_t row_k,
                                                           int64_t row_v, int64_t cs_k, int64_t cs_v, int64_t ss_k,
                                                           int64_t ss_v, float inv_k, float inv_v) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= tokens * 2) return;
  const int64_t t = w >> 1;
  const bool is_v = w & 1;
  const int64_t slot = loc[t];
  const T* src = is_v ? v + t * ss_v : k + t * ss_k;
  uint8_t* dst = is_v ? v_cache + slot * cs_v : k_cache + slot * cs_k;
  const int64_t n = is_v ? row_v : row_k;
  const float inv = is_v ? inv_v : inv_k;
  for (int64_t i = lane; i < n / 8; i += 64) {
    const uint4 u = *(const uint4*)(src + i * 8);
    const uint32_t wd[4] = {u.x, u.y, u.z, u.w};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = Elem<T>::lo(wd[j]);
      f[2 * j + 1] = Elem<T>::hi(wd[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (inv != 1.0f) f[j] = round_to<T>(f[j] * inv);
      f[j] = fmaxf(fminf(f[j], 448.0f), -448.0f);
    }
    uint32_t lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *(uint2*)(dst + i * 8) = make_uint2(lo, hi);
  }
}

extern "C" int mi_kv_write_fp8(void* k_cache, void* v_cache, const int64_t* loc, const void* k, const void* v,
                               int64_t tokens, int64_t row_elems_k, int64_t row_elems_v, int64_t cache_stride_k,
                               int64_t cache_stride_v, int64_t src_stride_k, int64_t src_stride_v, float k_scale,
                               float v_scale, int dtype, void* stream) {
  MI_CHECK_ARG(tokens >= 0);
  if (tokens == 0) return MI_OK;
  MI_CHECK_ARG(k_cache && v_cache && loc && k && v);
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  MI_CHECK_ARG(k_scale > 0.f && v_scale > 0.f);
  if (row_elems_k % 8 || row_elems_v % 8 || cache_stride_k % 8 || cache_stride_v % 8 || src_stride_k % 8 || src_stride_v % 8 ||
      (((uintptr_t)k_cache | (uintptr_t)v_cache) & 7) || (((uintptr_t)k | (uintptr_t)v) & 15))
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_kv_write_fp8: rows must be multiples of 8 elements and 16-byte aligned");
  const float inv_k = k_scale == 1.0f ? 1.0f : 1.0f / k_scale, inv_v = v_scale == 1.0f ? 1.0f : 1.0f / v_scale;
  const int64_t waves = tokens * 2;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_BF16)
    kv_write_fp8_kernel<bf16_t><<<(unsigned)cdiv64(waves, 4), 256, 0, st>>>((uint8_t*)k_cache, (uint8_t*)v_cache, loc, (const bf16_t*)k, (const bf16_t*)v, tokens, row_elems_k, row_elems_v, cache_stride_k, cache_stride_v, src_stride_k, src_stride_v, inv_k, inv_v);
  else
    kv_write_fp8_kernel<f16_t><<<(unsigned)cdiv64(waves, 4), 256, 0, st>>>((uint8_t*)k_cache, (uint8_t*)v_cache, loc, (const f16_t*)k, (const f16_t*)v, tokens, row_elems_k, row_elems_v, cache_stride_k, cache_stride_v, src_stride_k, src_stride_v, inv_k, inv_v);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---------------------------------------------------------------- paged slot allocation (scheduler side)
// alloc_extend / alloc_decode of PagedTokenToKVPoolAllocator (mem_cache/allocator.py:278-404, Triton there):
// request i gets (seq_lens[i] - prefix_lens[i]) slot indices: first the rest of its old partial page
// (last_loc + 1 ...), then whole new pages, then the head of one more new page; new pages are taken from the
// front of free_pages in request order.  Integer, bit-exact.  The reference recomputes the prefix sums in
// every program (O(bs^2)); here ONE workgroup scans once into `scratch`, then one workgroup per request fills.
__global__ __launch_bounds__(256) void alloc_scan_kernel(const int64_t* __restrict__ pre_lens /* null: decode */,
                                                         const int64_t* __restrict__ seq_lens, int64_t bs, int64_t ps,
                                                         int64_t* __restrict__ scratch /* [2*bs]: out start, page start */,
                                                         int64_t* __restrict__ ret, int merged_ret) {
  __shared__ int64_t wsum[2][4];
  __shared__ int64_t carry[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
  __syncthreads();
  for (int64_t base = 0; base < bs; base += 256) {
    const int64_t i = base + threadIdx.x;
    int64_t ext = 0, pages = 0;
    if (i < bs) {
      const int64_t s = seq_lens[i], p = pre_lens ? pre_lens[i] : s - 1;
      ext = s - p;
      pages = (s + ps - 1) / ps - (p + ps - 1) / ps;
    }
    int64_t ie = ext, ip = pages;   // inclusive scans inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t te = __shfl_up(ie, off), tp = __shfl_up(ip, off);
      if (lane >= off) { ie += te; ip += tp; }
    }
    if (lane == 63) { wsum[0][wave] = ie; wsum[1][wave] = ip; }
    __syncthreads();
    int64_t oe = carry[0], op = carry[1];
    for (int w = 0; w < wave; ++w) { oe += wsum[0][w]; op += wsum[1][w]; }
    if (i < bs) { scratch[i] = oe + ie - ext; scratch[bs + i] = op + ip - pages; }
    __syncthreads();
    if (threadIdx.x == 255) { carry[0] = oe + ie; carry[1] = op + ip; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *ret = merged_ret ? ((carry[1] << 32) | carry[0]) : carry[1];
}

__global__ __launch_bounds__(256) void alloc_extend_fill_kernel(const int64_t* __restrict__ pre_lens,
                                                                const int64_t* __restrict__ seq_lens,
                                                                const int64_t* __restrict__ last_loc,
                                                                const int64_t* __restrict__ free_pages,
                                                                const int64_t* __restrict__ scratch, int64_t bs, int64_t ps,
                                                                int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x;
  const int64_t p = pre_lens[i], s = seq_lens[i];
  const int64_t pos = scratch[i], page = scratch[bs + i];
  const int64_t ceil_pre = (p + ps - 1) / ps;
  const int64_t new_pages = (s + ps - 1) / ps - ceil_pre;
  const int64_t n1 = min(s, ceil_pre * ps) - p;
  const int64_t n2 = (p + n1 != s) ? s / ps * ps - ceil_pre * ps : 0;
  const int64_t n3 = (p + n1 != s && p + n1 + n2 != s) ? s - s / ps * ps : 0;
  const int64_t ll = last_loc[i];
  for (int64_t j = threadIdx.x; j < s - p; j += 256) {
    int64_t v;
    if (j < n1) v = ll + 1 + j;
    else if (j < n1 + n2) v = free_pages[page + (j - n1) / ps] * ps + (j - n1) % ps;
    else v = free_pages[page + new_pages - 1] * ps + (j - n1 - n2);
    out[pos + j] = v;
  }
  (void)n3;
}

__global__ __launch_bounds__(256) void alloc_decode_fill_kernel(const int64_t* __restrict__ seq_lens,
                                                                const int64_t* __restrict__ last_loc,
                                                                const int64_t* __restrict__ free_pages,
                                                                const int64_t* __restrict__ scratch, int64_t bs, int64_t ps,
                                                                int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= bs) return;
  const int64_t s = seq_lens[i];
  const int64_t need = (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  out[i] = need == 0 ? last_loc[i] + 1 : free_pages[scratch[bs + i]] * ps;
}

extern "C" int mi_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                               const int64_t* free_pages, int64_t* out_indices, int64_t* ret_value, int64_t* scratch,
                               int64_t batch, int64_t page_size, void* stream) {
  MI_CHECK_ARG(batch >= 0 && page_size >= 1);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(prefix_lens && seq_lens && last_loc && free_pages && out_indices && ret_value && scratch);
  MI_CHECK_ARG(batch <= 0x7fffffff);
  hipStream_t st = (hipStream_t)stream;
  alloc_scan_kernel<<<1, 256, 0, st>>>(prefix_lens, seq_lens, batch, page_size, scratch, ret_value, 1);
  MI_CHECK_LAUNCH();
  alloc_extend_fill_kernel<<<(unsigned)batch, 256, 0, st>>>(prefix_lens, seq_lens, last_loc, free_pages, scratch, batch, page_size, out_indices);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

extern "C" int mi_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                               int64_t* out_indices, int64_t* ret_value, int64_t* scratch, int64_t batch,
                               int64_t page_size, void* stream) {
  MI_CHECK_ARG(batch >= 0 && page_size >= 1);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(seq_lens && last_loc && free_pages && out_indices && ret_value && scratch);
  hipStream_t st = (hipStream_t)stream;
  alloc_scan_kernel<<<1, 256, 0, st>>>(nullptr, seq_lens, batch, page_size, scratch, ret_value, 0);
  MI_CHECK_LAUNCH();
  alloc_decode_fill_kernel<<<(unsigned)cdiv64(batch, 256), 256, 0, st>>>(seq_lens, last_loc, free_pages, scratch, batch, page_size, out_indices);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// ---------------------------------------------------------------- scheduler-side request bookkeeping ("K10")
// Replace the Triton helpers the scheduler runs once per batch:
//   write_req_to_token_pool_triton  python/sglang/srt/managers/schedule_batch.py:1848-1882
//   get_last_loc_kernel             python/sglang/srt/managers/schedule_batch.py:1912-1932
//   compute_position_kernel         python/sglang/srt/model_executor/forward_batch_info.py:704-732
// Integer, bit-exact, latency-bound.  One workgroup per request; the request's offset into the flat per-token arrays
// (the exclusive prefix sum of the extend lengths) is a coalesced 256-thread reduction over the requests before it,
// where the Triton programs each walk them serially.
template <typename L>
__device__ __forceinline__ int64_t block_prefix_sum(const L* __restrict__ lens, int64_t n, int64_t* red) {
  int64_t acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += (int64_t)lens[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  int64_t total = 0;
  for (unsigned w = 0; w < blockDim.x / 64; ++w) total += red[w];
  return total;
}

__global__ __launch_bounds__(256) void write_req_to_token_kernel(int32_t* __restrict__ req_to_token, int64_t stride,
                                                                 const int64_t* __restrict__ req_pool_indices,
                                                                 const int64_t* __restrict__ pre_lens,
                                                                 const int64_t* __restrict__ seq_lens,
                                                                 const int64_t* __restrict__ extend_lens,
                                                                 const int64_t* __restrict__ out_cache_loc) {
  __shared__ int64_t red[4];
  const int64_t b = blockIdx.x;
  const int64_t start = block_prefix_sum(extend_lens, b, red);
  const int64_t pre = pre_lens[b], n = seq_lens[b] - pre;
  int32_t* dst = req_to_token + req_pool_indices[b] * stride + pre;
  const int64_t* src = out_cache_loc + start;
  for (int64_t j = threadIdx.x; j < n; j += 256) dst[j] = (int32_t)src[j];
}

extern "C" int mi_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride,
                                     const int64_t* req_pool_indices, const int64_t* pre_lens,
                                     const int64_t* seq_lens, const int64_t* extend_lens,
                                     const int64_t* out_cache_loc, int64_t batch, void* stream) {
  MI_CHECK_ARG(batch >= 0 && batch <= 0x7fffffff);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(req_to_token && req_pool_indices && pre_lens && seq_lens && extend_lens && out_cache_loc);
  MI_CHECK_ARG(req_to_token_stride > 0);
  write_req_to_token_kernel<<<(unsigned)batch, 256, 0, (hipStream_t)stream>>>(
      req_to_token, req_to_token_stride, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

__global__ __launch_bounds__(256) void get_last_loc_kernel(const int32_t* __restrict__ req_to_token, int64_t stride,
                                                           const int64_t* __restrict__ req_pool_indices,
                                                           const int64_t* __restrict__ prefix_lens,
                                                           int64_t* __restrict__ result, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t pre = prefix_lens[i];
  result[i] = pre > 0 ? (int64_t)req_to_token[req_pool_indices[i] * stride + pre - 1] : -1;
}

extern "C" int mi_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride,
                               const int64_t* req_pool_indices, const int64_t* prefix_lens, int64_t* result,
                               int64_t batch, void* stream) {
  MI_CHECK_ARG(batch >= 0);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(req_to_token && req_pool_indices && prefix_lens && result && req_to_token_stride > 0);
  get_last_loc_kernel<<<(unsigned)cdiv64(batch, 256), 256, 0, (hipStream_t)stream>>>(
      req_to_token, req_to_token_stride, req_pool_indices, prefix_lens, result, batch);
  MI_CHECK_LAUNCH();
  return MI_OK;
}

template <typename L>
__global__ __launch_bounds__(256) void compute_position_kernel(const L* __restrict__ extend_prefix_lens,
                                                               const L* __restrict__ extend_seq_lens,
                                                               int64_t* __restrict__ positions,
                                                               int32_t* __restrict__ extend_start_loc) {
  __shared__ int64_t red[4];
  const int64_t b = blockIdx.x;
  const int64_t start = block_prefix_sum(extend_seq_lens, b, red);
  const int64_t pre = extend_prefix_lens ? (int64_t)extend_prefix_lens[b] : 0;
  const int64_t n = (int64_t)extend_seq_lens[b];
  for (int64_t j = threadIdx.x; j < n; j += 256) positions[start + j] = pre + j;
  if (threadIdx.x == 0) extend_start_loc[b] = (int32_t)start;
}

extern "C" int mi_compute_position(const void* extend_prefix_lens /* nullable: no prefixes */,
                                   const void* extend_seq_lens, int lens_is_i64, int64_t* positions,
                                   int32_t* extend_start_loc, int64_t batch, void* stream) {
  MI_CHECK_ARG(batch >= 0 && batch <= 0x7fffffff);
  if (batch == 0) return MI_OK;
  MI_CHECK_ARG(extend_seq_lens && positions && extend_start_loc);
  hipStream_t s = (hipStream_t)stream;
  if (lens_is_i64)
    compute_position_kernel<int64_t><<<(unsigned)batch, 256, 0, s>>>(
        (const int64_t*)extend_prefix_lens, (const int64_t*)extend_seq_lens, positions, extend_start_loc);
  else
    compute_position_kernel<int32_t><<<(unsigned)batch, 256, 0, s>>>(
        (const int32_t*)extend_prefix_lens, (const int32_t*)extend_seq_lens, positions, extend_start_loc);
  MI_CHECK_LAUNCH();
  return MI_OK;
}
