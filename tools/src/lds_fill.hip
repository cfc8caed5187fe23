// Ceiling probe (not product code): how fast can ONE 8-wave workgroup per CU fill LDS from L2-resident memory,
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// Each workgroup streams `iters` x 64 KiB (8 waves x 8 instructions x 1 KiB) out of a shared 8 MiB region,
// one barrier per 64 KiB (the structure of the GEMM kernels' stages).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int MODE, int DEPTH>   // DEPTH = 64-KiB stages in flight (1 or 2)
__global__ __launch_bounds__(512) void fill(const uint4* __restrict__ src, int iters, size_t region_vec, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  float acc = 0.f;
  size_t pos = ((size_t)blockIdx.x * 4096) % region_vec;       // start offset, in uint4
  for (int it = 0; it < iters; ++it) {
    const int st = it % DEPTH;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        glds16(src + (pos + (wave * 8 + i) * 64 + lane) % region_vec, lds_base + st * 65536 + (wave * 8 + i) * 1024);
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(pos + (wave * 8 + i) * 64 + lane) % region_vec];
#pragma unroll
      for (int i = 0; i < 8; ++i) *(uint4*)(smem + st * 65536 + (wave * 8 + i) * 1024 + lane * 16) = v[i];
    }
    __syncthreads();
    acc += *(const float*)(smem + st * 65536 + ((lane * 67 + wave * 131 + it) & 16383) * 4);   // keep the stores alive
    pos = (pos + 4096) % region_vec;
  }
  if (acc == 1.2345f) out[0] = acc;
}

// mode 2/3: the decode GEMM's traffic mix -- per 64-KiB stage, half from a shared 512-KiB block that every workgroup
// reads in lockstep (x, L2-resident), half from the workgroup's own stream (weights, HBM).  DEPTH stages in flight.
template <int DEPTH, bool SHARED_HALF>
__global__ __launch_bounds__(512) void fill_mix(const uint4* __restrict__ xsrc, const uint4* __restrict__ wsrc, int iters,
                                                float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  float acc = 0.f;
  const uint4* wmine = wsrc + (size_t)blockIdx.x * iters * 2048;          // 32 KiB (2048 uint4) of weights per stage
  for (int it = 0; it < iters; ++it) {
    const int st = it % DEPTH;
#pragma unroll
    for (int i = 0; i < 4; ++i)     // x half: the same 32 KiB for every workgroup
      glds16((SHARED_HALF ? xsrc + ((size_t)(it % 16) * 2048) : wmine + (size_t)it * 2048) + (wave * 4 + i) * 64 + lane,
             lds_base + st * 65536 + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i)     // weight half: private stream
      glds16(wmine + (size_t)it * 2048 + (wave * 4 + i) * 64 + lane, lds_base + st * 65536 + 32768 + (wave * 4 + i) * 1024);
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (DEPTH - 1)) : "memory");
    __syncthreads();
    acc += *(const float*)(smem + st * 65536 + ((lane * 67 + wave * 131 + it) & 16383) * 4);
  }
  if (acc == 1.2345f) out[0] = acc;
}

// mode 4: as fill_mix but with HALF-size stages (16 KiB shared x + 16 KiB private weights = one 128-byte k-phase of the
// M=128 decode GEMM), `iters` = 32 of them, DEPTH stages in the ring (DEPTH-1 in flight across the barrier)
template <int DEPTH>
__global__ __launch_bounds__(512) void fill_mix_half(const uint4* __restrict__ xsrc, const uint4* __restrict__ wsrc, int iters,
                                                     float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  float acc = 0.f;
  const uint4* wmine = wsrc + (size_t)blockIdx.x * iters * 1024;          // 16 KiB (1024 uint4) of weights per stage
  auto issue = [&](int it) {
    const int st = it % DEPTH;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(xsrc + ((size_t)(it % 32) * 1024) + (wave * 2 + i) * 64 + lane, lds_base + st * 32768 + (wave * 2 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(wmine + (size_t)it * 1024 + (wave * 2 + i) * 64 + lane, lds_base + st * 32768 + 16384 + (wave * 2 + i) * 1024);
  };
  for (int d = 0; d < DEPTH - 1 && d < iters; ++d) issue(d);
  for (int it = 0; it < iters; ++it) {
    if (it + DEPTH - 1 < iters) issue(it + DEPTH - 1);
    // stage `it` must have landed; the DEPTH-1 younger stages (4 entries each) may fly -- fewer near the end
    const int younger = min(DEPTH - 1, iters - 1 - it);
    if (younger >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *(const float*)(smem + (it % DEPTH) * 32768 + ((lane * 67 + wave * 131 + it) & 8191) * 4);
    __syncthreads();   // the stage is free again (stands for the end-of-phase barrier of the GEMM)
  }
  if (acc == 1.2345f) out[0] = acc;
}

int main() {
  const size_t region = 8u << 20;
  uint4* src; float* out;
  CK(hipMalloc(&src, region)); CK(hipMemset(src, 1, region)); CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048, wgs = 256;
  auto run = [&](const char* name, auto kern) {
    kern(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * iters * 65536;
    printf("%-56s %8.3f ms  %7.1f GB/s total  %6.1f GB/s per CU\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
  };
  run("LDS-DMA, 1 stage (issue 64 KiB, wait, barrier)", [&] { fill<0, 1><<<wgs, 512, 65536>>>(src, iters, region / 16, out); });
  run("LDS-DMA, 2 stages (one 64-KiB stage in flight)", [&] { fill<0, 2><<<wgs, 512, 131072>>>(src, iters, region / 16, out); });
  run("load -> VGPR -> ds_write_b128, 1 stage", [&] { fill<1, 1><<<wgs, 512, 65536>>>(src, iters, region / 16, out); });
  run("load -> VGPR -> ds_write_b128, 2 stages", [&] { fill<1, 2><<<wgs, 512, 131072>>>(src, iters, region / 16, out); });
  {
    const int it2 = 16;                                   // 16 stages = the gate_up GEMM's 16 k-phases
    uint4* wbuf; CK(hipMalloc(&wbuf, (size_t)4 * 256 * it2 * 32768)); CK(hipMemset(wbuf, 2, (size_t)4 * 256 * it2 * 32768));
    int rep = 0;
    auto runm = [&](const char* name, auto kern) {
      kern(rep++ % 4); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 4; ++r) kern(rep++ % 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
      const double bytes = 256.0 * it2 * 65536;
      printf("%-56s %8.1f us  %7.1f GB/s total  %6.1f GB/s per CU\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 256);
    };
#define WB(r) (wbuf + (size_t)(r) * 256 * it2 * 2048)
    runm("mix: 32K shared x + 32K private HBM per stage, 1 in flight", [&](int r) { fill_mix<1, true><<<256, 512, 65536>>>(src, WB(r), it2, out); });
    runm("mix: same, 2 stages (one in flight)", [&](int r) { fill_mix<2, true><<<256, 512, 131072>>>(src, WB(r), it2, out); });
    runm("private HBM only (32K + 32K of the same stream), 1", [&](int r) { fill_mix<1, false><<<256, 512, 65536>>>(src, WB(r), it2, out); });
    runm("private HBM only, 2 stages", [&](int r) { fill_mix<2, false><<<256, 512, 131072>>>(src, WB(r), it2, out); });
    runm("half stages (16K x + 16K w) x 32, ring 2", [&](int r) { fill_mix_half<2><<<256, 512, 2 * 32768>>>(src, WB(r), 32, out); });
    runm("half stages x 32, ring 3", [&](int r) { fill_mix_half<3><<<256, 512, 3 * 32768>>>(src, WB(r), 32, out); });
    runm("half stages x 32, ring 4", [&](int r) { fill_mix_half<4><<<256, 512, 4 * 32768>>>(src, WB(r), 32, out); });
    runm("half stages x 32, ring 5", [&](int r) { fill_mix_half<5><<<256, 512, 5 * 32768>>>(src, WB(r), 32, out); });
  }
  return 0;
}
