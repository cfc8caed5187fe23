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
  return 0;
}
