// Ceiling probe (not product code): how fast can gfx950 READ HBM with plain 16-B/lane loads?
//   mode 0: every workgroup streams a contiguous chunk (coalesced 1 KiB per wave instruction)
//   mode 1: 2-KiB rows at random positions, one 256-B segment per wave (the decode-attention V pattern)
//   mode 2: as 1 but each wave instruction touches 16 rows x 64 B (the decode-attention K pattern)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UNROLL, int MODE>
__global__ __launch_bounds__(512) void rd(const uint4* __restrict__ buf, const int* __restrict__ rowidx, int rows_per_wg,
                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  const int* ri = rowidx + (size_t)blockIdx.x * rows_per_wg;
  for (int r0 = 0; r0 < rows_per_wg; r0 += 16 * UNROLL) {
    uint4 v[UNROLL][4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        size_t off;   // in uint4 units; a row = 2 KiB = 128 uint4, wave w owns segment w (16 uint4)
        if (MODE == 0) off = ((size_t)blockIdx.x * rows_per_wg + r0 + u * 16 + i * 4 + (lane >> 4)) * 128 + wave * 16 + (lane & 15);
        else if (MODE == 1) off = (size_t)ri[r0 + u * 16 + i * 4 + (lane >> 4)] * 128 + wave * 16 + (lane & 15);
        else if (MODE == 2) off = (size_t)ri[r0 + u * 16 + (lane & 15)] * 128 + wave * 16 + i * 4 + (lane >> 4);
        else if (MODE == 3) off = ((size_t)((r0 + u * 16 + i * 4 + (lane >> 4)) & 255)) * 128 + wave * 16 + (lane & 15);   // every WG: the same 512 KiB, in lockstep
        else {   // MODE 4: even instructions stream the WG's own chunk, odd ones read the shared 512 KiB
          if (i & 1) off = ((size_t)((r0 + u * 16 + i * 4 + (lane >> 4)) & 255)) * 128 + wave * 16 + (lane & 15);
          else off = ((size_t)blockIdx.x * rows_per_wg + r0 + u * 16 + i * 4 + (lane >> 4)) * 128 + wave * 16 + (lane & 15);
        }
        v[u][i] = buf[off];
      }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += __uint_as_float(v[u][i].x ^ v[u][i].y ^ v[u][i].z ^ v[u][i].w);
  }
  if (acc == 1.2345f) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t rows = 524288;           // 1 GiB of 2-KiB rows per buffer
  const int nbuf = 4, wgs = 2048, rpw = rows / wgs;
  uint4* buf[nbuf];
  for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&buf[i], rows * 2048)); CK(hipMemset(buf[i], i + 1, rows * 2048)); }
  std::vector<int> perm(rows);
  for (size_t i = 0; i < rows; ++i) perm[i] = (int)i;
  std::mt19937 g(0);
  std::shuffle(perm.begin(), perm.end(), g);
  int* d_perm; float* d_out;
  CK(hipMalloc(&d_perm, rows * 4)); CK(hipMemcpy(d_perm, perm.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < nbuf; ++i) kern(buf[i]);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) for (int i = 0; i < nbuf; ++i) kern(buf[i]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * nbuf);
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, us, rows * 2048.0 / us / 1e3);
  };
#define RUN(U, M, label) run(label, [&](uint4* b) { rd<U, M><<<wgs, 512>>>(b, d_perm, rpw, d_out); })
  // one 8-wave workgroup per CU (128 KiB of LDS each): does the bandwidth follow the bytes a CU keeps in flight?
  auto run1 = [&](const char* name, auto kern) { run(name, kern); };
#define RUN1(U, label) run1(label, [&](uint4* b) { rd<U, 0><<<256, 512, 131072>>>(b, d_perm, (int)(rows / 256), d_out); })
  RUN1(1, "1 WG/CU (8 waves), 4 KiB/wave in flight");
  RUN1(2, "1 WG/CU (8 waves), 8 KiB/wave in flight");
  RUN1(4, "1 WG/CU (8 waves), 16 KiB/wave in flight");
  RUN1(8, "1 WG/CU (8 waves), 32 KiB/wave in flight");
#define RUN3(U, M, label) run1(label, [&](uint4* b) { rd<U, M><<<256, 512, 131072>>>(b, d_perm, (int)(rows / 256), d_out); })
  RUN3(2, 3, "1 WG/CU, all WGs re-read one 512 KiB (L2)");
  RUN3(4, 3, "1 WG/CU, all WGs re-read one 512 KiB (L2), 16K/wave");
  RUN3(2, 4, "1 WG/CU, half stream / half shared 512 KiB");
  RUN3(4, 4, "1 WG/CU, half stream / half shared, 16K/wave");
  RUN(1, 0, "contiguous, 4 KiB/wave in flight");
  RUN(2, 0, "contiguous, 8 KiB/wave in flight");
  RUN(4, 0, "contiguous, 16 KiB/wave in flight");
  RUN(1, 1, "random 2K rows, V pattern, 4 KiB/wave");
  RUN(2, 1, "random 2K rows, V pattern, 8 KiB/wave");
  RUN(4, 1, "random 2K rows, V pattern, 16 KiB/wave");
  RUN(2, 2, "random 2K rows, K pattern, 8 KiB/wave");
  RUN(4, 2, "random 2K rows, K pattern, 16 KiB/wave");
  return 0;
}
