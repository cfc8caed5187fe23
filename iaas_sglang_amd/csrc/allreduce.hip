// Tensor-parallel all-reduce over xGMI for the small, latency-bound messages of decode
// ([tokens, hidden] bf16, 1-4 MiB): one process per GPU, peers' buffers mapped through hipIpc.
//
// Each rank owns ONE shared allocation  [ ArSignal | staging (max_bytes) | tmp (max_bytes) ]
// (uncached device memory), whose IPC handle every peer opens.
//   1-stage (world 2, or small messages): every rank reads all peers' staging buffers and reduces
//            the whole message itself:      1 hop, N-1 remote reads of the full message per rank.
//   2-stage: reduce-scatter (rank r reduces slice r from all peers' staging into its own tmp),
//            barrier, all-gather (every rank reads slice k from peer k's tmp).  Per stage each GPU
//            moves bytes/N to/from each of its N-1 peers concurrently -- the all-to-all pattern that
//            keeps all 7 xGMI links busy (a ring would serialise 2(N-1) steps on one link each).
// Sums are accumulated in fp32 in rank order 0..N-1 by exactly one owner per element, so every
// rank ends with bit-identical results.  Barriers are monotonically increasing per-block flags
// written with system-scope atomics (no reset, no host involvement); every spin is bounded (20 s)
// and reports a timeout in ArSignal::error instead of hanging the GPU.
// Mirrors the protocol of sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh (Appendix C of SURVEY.md).
#include "common.h"
#include <string.h>

#define AR_MAX_BLOCKS 64
#define AR_MAX_RANKS 8
#define AR_THREADS 512
#define AR_TIMEOUT_TICKS (20ull * 100000000ull)  // 20 s of the 100 MHz s_memrealtime clock

struct alignas(128) ArSignal {
  uint32_t start[AR_MAX_BLOCKS][AR_MAX_RANKS];
  uint32_t end[AR_MAX_BLOCKS][AR_MAX_RANKS];
  uint32_t flag[AR_MAX_BLOCKS];
  uint32_t error;  // != 0: a barrier timed out (peer missing); results are invalid
  uint32_t pad[31];
};

struct ArPeers {
  ArSignal* sig[AR_MAX_RANKS];
  const char* stage[AR_MAX_RANKS];
  char* tmp[AR_MAX_RANKS];
};

struct ArCtx {
  int rank, world;
  int64_t max_bytes;
  ArPeers peers;
};

__device__ __forceinline__ void ar_store_flag(uint32_t* p, uint32_t v, bool release) {
  if (release) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// all ranks arrive; `slots` = start or end array of every rank's signal
template <bool ACQ_REL>
__device__ __forceinline__ void ar_barrier(const ArPeers& pr, int rank, int world, bool use_end, uint32_t flag) {
  // an end barrier says "this workgroup is done reading": every wave must have finished its loads before the flag
  // goes out (custom_all_reduce_hip.cuh:205 does the same, final sync included); a start barrier has nothing before it
  if (use_end) __syncthreads();
  if (ACQ_REL) __threadfence_system();
  ArSignal* self = pr.sig[rank];
  if ((int)threadIdx.x < world) {
    ArSignal* peer = pr.sig[threadIdx.x];
    uint32_t* dst = use_end ? &peer->end[blockIdx.x][rank] : &peer->start[blockIdx.x][rank];
    ar_store_flag(dst, flag, ACQ_REL);
    uint32_t* src = use_end ? &self->end[blockIdx.x][threadIdx.x] : &self->start[blockIdx.x][threadIdx.x];
    // bounded spin: a peer that never arrives (crashed rank) flags an error after 20 s instead of
    // hanging the GPU; a merely late peer (host-side skew) is waited for
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t spins = 0;
    while (__hip_atomic_load(src, ACQ_REL ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < flag) {
      if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > AR_TIMEOUT_TICKS) {
        __hip_atomic_store(&self->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

template <typename T> struct ArPack;  // 16-byte pack <-> fp32 lanes
template <> struct ArPack<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void up(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  static __device__ __forceinline__ uint4 down(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <typename T> struct ArPack16 {
  static constexpr int N = 8;
  static __device__ __forceinline__ void up(const uint4& u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = Elem<T>::lo(w[j]); f[2 * j + 1] = Elem<T>::hi(w[j]); }
  }
  static __device__ __forceinline__ uint4 down(const float* f) {
    return make_uint4(pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7]));
  }
};
template <> struct ArPack<bf16_t> : ArPack16<bf16_t> {};
template <> struct ArPack<f16_t> : ArPack16<f16_t> {};

// fp32 sum over ranks 0..world-1 of pack `idx` (fixed order => identical bits wherever it is computed)
template <typename T>
__device__ __forceinline__ uint4 ar_reduce_pack(const ArPeers& pr, int world, int64_t idx) {
  constexpr int N = ArPack<T>::N;
  float acc[N], tmp[N];
  ArPack<T>::up(((const uint4*)pr.stage[0])[idx], acc);
  for (int r = 1; r < world; ++r) {
    ArPack<T>::up(((const uint4*)pr.stage[r])[idx], tmp);
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] += tmp[j];
  }
  return ArPack<T>::down(acc);
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void ar_1stage_kernel(const ArPeers pr, uint4* __restrict__ out, int rank,
                                                               int world, int64_t packs) {
  ArSignal* self = pr.sig[rank];
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  ar_barrier<false>(pr, rank, world, false, flag);  // everybody's staging copy is in place
  for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < packs; i += (int64_t)gridDim.x * AR_THREADS)
    out[i] = ar_reduce_pack<T>(pr, world, i);
  ar_barrier<false>(pr, rank, world, true, flag);   // nobody overwrites staging before all have read it
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void ar_2stage_kernel(const ArPeers pr, uint4* __restrict__ out, int rank,
                                                               int world, int64_t packs) {
  ArSignal* self = pr.sig[rank];
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  const int64_t part = packs / world;
  const int64_t lo = rank * part, hi = (rank == world - 1) ? packs : lo + part;
  ar_barrier<false>(pr, rank, world, false, flag);
  // stage 1: reduce-scatter -- my slice, from every rank's staging, into my tmp
  uint4* mytmp = (uint4*)pr.tmp[rank];
  for (int64_t i = lo + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < hi; i += (int64_t)gridDim.x * AR_THREADS)
    mytmp[i - lo] = ar_reduce_pack<T>(pr, world, i);
  ar_barrier<true>(pr, rank, world, true, flag);  // slices are visible to peers
  // stage 2: all-gather -- slice k from rank k's tmp
  for (int k = 0; k < world; ++k) {
    const int src = (rank + k) % world;  // start with the local slice, spread peers over time
    const int64_t slo = src * part, shi = (src == world - 1) ? packs : slo + part;
    const uint4* t = (const uint4*)pr.tmp[src];
    for (int64_t i = slo + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < shi; i += (int64_t)gridDim.x * AR_THREADS)
      out[i] = t[i - slo];
  }
  // no third barrier: a peer can only write its tmp / staging again after the NEXT call's start
  // barrier, which every rank reaches only after finishing the reads above
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}

// ------------------------------------------------- all-reduce fused with (add +) RMSNorm (+ static fp8 quant)
// The consumer of every TP all-reduce in a Llama layer is `x32 = h + residual; residual <- x32; out = norm(x32) * w`
// (layers/layernorm.py:128-146; the reference fuses it on the flashinfer path, layers/flashinfer_comm_fusion.py and
// layers/communicator.py).  Here the LAST phase of the all-reduce -- the pass in which a rank reads the reduced
// values anyway -- is that row-wise consumer: 2-stage: reduce-scatter as above, then the gather phase reads each
// row's packs from their owners' tmp straight into registers, adds the residual, norms, and writes
// residual / out / fp8(out); the reduced tensor itself never goes to memory.  1-stage: the same with the packs
// reduced on the fly.  Each 256-thread half of the workgroup owns one row with exactly the vector->thread mapping
// and reduction order of rmsnorm_kernel (elementwise.hip), and the reduced sum is rounded to T first, so the
// results are bit-identical to mi_ar_all_reduce followed by mi_rmsnorm[_fp8].
template <typename T> __device__ __forceinline__ void ar_unpack8(const uint4& u, float (&f)[8]) { ArPack16<T>::up(u, f); }

__device__ __forceinline__ uint2 ar_quant8_static(const float (&f)[8], float inv) {
  uint32_t lo = 0, hi = 0;
  auto c = [](float v) { return fmaxf(fminf(v, 448.0f), -448.0f); };
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[0] * inv), c(f[1] * inv), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[2] * inv), c(f[3] * inv), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[4] * inv), c(f[5] * inv), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c(f[6] * inv), c(f[7] * inv), hi, true);
  return make_uint2(lo, hi);
}

template <typename T, int VPT, bool TWO_STAGE>
__global__ __launch_bounds__(AR_THREADS) void ar_add_rmsnorm_kernel(const ArPeers pr, int rank, int world, int64_t rows,
                                                                    int64_t H, T* __restrict__ residual, int64_t ldr,
                                                                    const T* __restrict__ w, float eps,
                                                                    T* __restrict__ out, int64_t ldo,
                                                                    uint8_t* __restrict__ q_out,
                                                                    const float* __restrict__ q_scale) {
  static_assert(AR_THREADS == 512, "two 256-thread row groups per workgroup");
  __shared__ float red[2][4];
  ArSignal* self = pr.sig[rank];
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  const int64_t nvec = H / 8;
  // 2-stage ownership: rank r reduces COLUMN slice r (cpart packs, the last rank takes the remainder) of every row,
  // and workgroup b of every rank handles the same rows (2b, 2b+1, then + 2*gridDim.x ...) in both phases.  The
  // packs workgroup b gathers from an owner were therefore written by workgroup b of that owner, which is exactly
  // what the per-workgroup end barrier orders.  (A flat slice of the message -- what ar_2stage_kernel uses with a
  // matching gather stride -- would make this row-wise gather read packs other workgroups of the owner wrote.)
  const int64_t cpart = nvec / world;
  ar_barrier<false>(pr, rank, world, false, flag);
  if (TWO_STAGE) {
    const int64_t clo = rank * cpart, cw = (rank == world - 1) ? nvec - clo : cpart;
    const int64_t rstep = (int64_t)gridDim.x * 2;
    const int64_t rb = (int64_t)blockIdx.x * 2;
    const int64_t npair = rb < rows ? (rows - rb + rstep - 1) / rstep : 0;   // row pairs of this workgroup
    uint4* mytmp = (uint4*)pr.tmp[rank];
    for (int64_t i = threadIdx.x; i < npair * 2 * cw; i += AR_THREADS) {
      const int64_t j = i / cw, c = clo + (i - j * cw);
      const int64_t row = rb + (j >> 1) * rstep + (j & 1);
      if (row < rows) mytmp[row * nvec + c] = ar_reduce_pack<T>(pr, world, row * nvec + c);
    }
    ar_barrier<true>(pr, rank, world, true, flag);
  }
  const int half = threadIdx.x >> 8, t = threadIdx.x & 255;
  for (int64_t row0 = (int64_t)blockIdx.x * 2; row0 < rows; row0 += (int64_t)gridDim.x * 2) {
    const int64_t row = row0 + half;
    const bool valid = row < rows;
    float v[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int64_t c = t + i * 256;
      if (valid && c < nvec) {
        const int64_t idx = row * nvec + c;
        uint4 u;
        if (TWO_STAGE) {
          const int64_t o = c / cpart;
          const int owner = o < world ? (int)o : world - 1;
          u = ((const uint4*)pr.tmp[owner])[idx];
        } else {
          u = ar_reduce_pack<T>(pr, world, idx);
        }
        ar_unpack8<T>(u, v[i]);
        if (residual) {
          float r[8];
          ar_unpack8<T>(*(const uint4*)(residual + row * ldr + c * 8), r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] += r[j];
          *(uint4*)(residual + row * ldr + c * 8) = ArPack16<T>::down(v[i]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[half][(threadIdx.x >> 6) & 3] = ss;
    __syncthreads();
    ss = red[half][0] + red[half][1] + red[half][2] + red[half][3];
    const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int64_t c = t + i * 256;
      if (valid && c < nvec) {
        float wf[8], o[8];
        ar_unpack8<T>(*(const uint4*)(w + c * 8), wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[i][j] * inv * wf[j];
        if (out) *(uint4*)(out + row * ldo + c * 8) = ArPack16<T>::down(o);
        if (q_out) {
          const float qs = *q_scale;
          const float qinv = qs > 0.f ? 1.0f / qs : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = round_to<T>(o[j]);
          *(uint2*)(q_out + row * H + c * 8) = ar_quant8_static(o, qinv);
        }
      }
    }
    __syncthreads();  // red[] is reused by the next pair of rows
  }
  // 1-stage: nobody overwrites staging before all have read it.  2-stage: as in ar_2stage_kernel, a peer can write
  // its tmp again only after the next call's start barrier, which every rank reaches after the reads above
  if (!TWO_STAGE) ar_barrier<false>(pr, rank, world, true, flag);
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}

// --------------------------------------------------------------------------- host side
extern "C" int64_t mi_ar_shared_bytes(int64_t max_bytes) {
  return (int64_t)sizeof(ArSignal) + 2 * ((max_bytes + 255) & ~(int64_t)255);
}

extern "C" int mi_ar_alloc_shared(int64_t bytes, void** ptr) {
  MI_CHECK_ARG(ptr && bytes >= (int64_t)sizeof(ArSignal));
  if (hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess)
    MI_FAIL(MI_ERR_LAUNCH, "mi_ar_alloc_shared: hipExtMallocWithFlags(%lld) failed", (long long)bytes);
  if (hipMemset(*ptr, 0, (size_t)bytes) != hipSuccess) MI_FAIL(MI_ERR_LAUNCH, "mi_ar_alloc_shared: memset failed");
  return MI_OK;
}
extern "C" int mi_ar_free_shared(void* ptr) {
  if (ptr && hipFree(ptr) != hipSuccess) MI_FAIL(MI_ERR_LAUNCH, "mi_ar_free_shared: hipFree failed");
  return MI_OK;
}
extern "C" int mi_ar_ipc_get(void* ptr, void* handle64) {
  MI_CHECK_ARG(ptr && handle64);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
  if (hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, ptr) != hipSuccess)
    MI_FAIL(MI_ERR_LAUNCH, "mi_ar_ipc_get: hipIpcGetMemHandle failed");
  return MI_OK;
}
extern "C" int mi_ar_ipc_open(const void* handle64, void** ptr) {
  MI_CHECK_ARG(ptr && handle64);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
    MI_FAIL(MI_ERR_LAUNCH, "mi_ar_ipc_open: hipIpcOpenMemHandle failed");
  return MI_OK;
}
extern "C" int mi_ar_ipc_close(void* ptr) {
  if (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) MI_FAIL(MI_ERR_LAUNCH, "mi_ar_ipc_close failed");
  return MI_OK;
}

extern "C" void* mi_ar_create(void** shared_ptrs, int64_t max_bytes, int rank, int world) {
  if (!shared_ptrs || world < 2 || world > AR_MAX_RANKS || rank < 0 || rank >= world || max_bytes <= 0) {
    mi_set_error("mi_ar_create: invalid argument (world 2..8, 0 <= rank < world)");
    return nullptr;
  }
  ArCtx* c = new ArCtx();
  c->rank = rank; c->world = world; c->max_bytes = max_bytes;
  const int64_t cap = (max_bytes + 255) & ~(int64_t)255;
  for (int r = 0; r < world; ++r) {
    char* base = (char*)shared_ptrs[r];
    c->peers.sig[r] = (ArSignal*)base;
    c->peers.stage[r] = base + sizeof(ArSignal);
    c->peers.tmp[r] = base + sizeof(ArSignal) + cap;
  }
  return c;
}
extern "C" int mi_ar_destroy(void* ctx) {
  delete (ArCtx*)ctx;
  return MI_OK;
}

// 0 = no timeout recorded so far (reads the rank's own signal block; synchronises the stream's device)
extern "C" int mi_ar_error(void* ctx) {
  ArCtx* c = (ArCtx*)ctx;
  if (!c) return -1;
  uint32_t e = 0;
  if (hipMemcpy(&e, &c->peers.sig[c->rank]->error, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

extern "C" int mi_ar_all_reduce(void* ctx, const void* inp, void* out, int64_t bytes, int dtype, void* stream) {
  ArCtx* c = (ArCtx*)ctx;
  MI_CHECK_ARG(c && inp && out && bytes >= 0);
  if (bytes == 0) return MI_OK;
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16 || dtype == MI_F32);
  if (bytes % 16 != 0 || bytes > c->max_bytes)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_ar_all_reduce: %lld bytes (need a multiple of 16, <= %lld)", (long long)bytes,
            (long long)c->max_bytes);
  MI_CHECK_ARG((((uintptr_t)inp | (uintptr_t)out) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  // eager path: stage the input in the IPC-mapped buffer (custom_all_reduce.py:446-450) -- unless the producer
  // already wrote it there (mi_ar_staging: the role of the reference's registered / graph buffers)
  if (inp != (const void*)c->peers.stage[c->rank] &&
      hipMemcpyAsync((void*)c->peers.stage[c->rank], inp, (size_t)bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    MI_FAIL(MI_ERR_LAUNCH, "mi_ar_all_reduce: staging copy failed");
  const int64_t packs = bytes / 16;
  // policy of custom_all_reduce_hip.cuh:541-551: world 2 -> 1-stage; else 1-stage below 256 KiB (512 KiB for <= 4)
  const bool one_stage = c->world == 2 || (c->world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024 ||
                         packs < c->world;
  int64_t work = one_stage ? packs : packs / c->world;
  int blocks = (int)cdiv64(work, AR_THREADS);
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
#define AR_LAUNCH(KERNEL, TT) KERNEL<TT><<<blocks, AR_THREADS, 0, st>>>(c->peers, (uint4*)out, c->rank, c->world, packs)
  if (one_stage) {
    if (dtype == MI_BF16) AR_LAUNCH(ar_1stage_kernel, bf16_t);
    else if (dtype == MI_FP16) AR_LAUNCH(ar_1stage_kernel, f16_t);
    else AR_LAUNCH(ar_1stage_kernel, float);
  } else {
    if (dtype == MI_BF16) AR_LAUNCH(ar_2stage_kernel, bf16_t);
    else if (dtype == MI_FP16) AR_LAUNCH(ar_2stage_kernel, f16_t);
    else AR_LAUNCH(ar_2stage_kernel, float);
  }
#undef AR_LAUNCH
  MI_CHECK_LAUNCH();
  return MI_OK;
}

// The rank's own IPC-mapped staging buffer (max_bytes): a producer (the row-parallel GEMM) that writes its output
// here and passes this pointer as `inp` skips the staging copy -- what register_buffer / register_graph_buffers
// (sgl_kernel_ops.h:58-68, custom_all_reduce.py:387-412) achieve in the reference by registering the producer's
// own buffers after capture; here one persistent buffer is registered once, so there is nothing to re-register.
extern "C" void* mi_ar_staging(void* ctx) {
  ArCtx* c = (ArCtx*)ctx;
  return c ? (void*)c->peers.stage[c->rank] : nullptr;
}

extern "C" int mi_ar_all_reduce_add_rmsnorm(void* ctx, const void* inp, void* residual, const void* weight, void* out,
                                            void* q_out, const float* q_scale, int64_t rows, int64_t H, int64_t ldr,
                                            int64_t ldo, float eps, int dtype, void* stream) {
  ArCtx* c = (ArCtx*)ctx;
  MI_CHECK_ARG(c && inp && weight && (out || q_out) && rows >= 0 && H > 0);
  MI_CHECK_ARG(!q_out || q_scale);
  if (rows == 0) return MI_OK;
  MI_CHECK_ARG(dtype == MI_BF16 || dtype == MI_FP16);
  const int64_t bytes = rows * H * 2;
  if (H % 8 != 0 || H > 256 * 8 * 8 || ldo % 8 || (residual && ldr % 8) || bytes > c->max_bytes)
    MI_FAIL(MI_ERR_UNSUPPORTED, "mi_ar_all_reduce_add_rmsnorm: H %% 8 == 0, H <= 16384, rows*H*2 <= %lld (rows=%lld H=%lld)",
            (long long)c->max_bytes, (long long)rows, (long long)H);
  MI_CHECK_ARG((((uintptr_t)inp | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)out) & 15) == 0);
  MI_CHECK_ARG(((uintptr_t)q_out & 7) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (inp != (const void*)c->peers.stage[c->rank] &&
      hipMemcpyAsync((void*)c->peers.stage[c->rank], inp, (size_t)bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    MI_FAIL(MI_ERR_LAUNCH, "mi_ar_all_reduce_add_rmsnorm: staging copy failed");
  const int64_t packs = bytes / 16;
  const bool one_stage = c->world == 2 || (c->world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024 ||
                         packs < c->world || H / 8 < c->world;
  int blocks = (int)cdiv64(rows, 2);
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  const int vpt = (int)cdiv64(H / 8, 256);
#define ARN_LAUNCH(TT, V, TS)                                                                                         \
  ar_add_rmsnorm_kernel<TT, V, TS><<<blocks, AR_THREADS, 0, st>>>(c->peers, c->rank, c->world, rows, H, (TT*)residual, \
                                                                  ldr, (const TT*)weight, eps, (TT*)out, ldo,          \
                                                                  (uint8_t*)q_out, q_scale)
#define ARN_VPT(TT, TS)                                                                             \
  do {                                                                                              \
    if (vpt <= 1) ARN_LAUNCH(TT, 1, TS); else if (vpt <= 2) ARN_LAUNCH(TT, 2, TS);                   \
    else if (vpt <= 4) ARN_LAUNCH(TT, 4, TS); else ARN_LAUNCH(TT, 8, TS);                            \
  } while (0)
  if (dtype == MI_BF16) { if (one_stage) ARN_VPT(bf16_t, false); else ARN_VPT(bf16_t, true); }
  else { if (one_stage) ARN_VPT(f16_t, false); else ARN_VPT(f16_t, true); }
#undef ARN_VPT
#undef ARN_LAUNCH
  MI_CHECK_LAUNCH();
  return MI_OK;
}
