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
  if (ACQ_REL) {
    __syncthreads();
    __threadfence_system();
  }
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
  // eager path: stage the input in the IPC-mapped buffer (custom_all_reduce.py:446-450)
  if (hipMemcpyAsync((void*)c->peers.stage[c->rank], inp, (size_t)bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
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
