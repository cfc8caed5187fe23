// Shared device/host helpers for the gfx950 hot-path kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mi_hotpath.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MI_WAVE 64
// cross-file helpers that are NOT part of the C ABI (not exported from libmi_hotpath.so)
#define MI_INTERNAL extern "C" __attribute__((visibility("hidden")))

// ---- error plumbing (host) --------------------------------------------------
void mi_set_error(const char* fmt, ...);
#define MI_FAIL(code, ...)        \
  do {                            \
    mi_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)
#define MI_CHECK_ARG(cond)                                                          \
  do {                                                                              \
    if (!(cond)) MI_FAIL(MI_ERR_INVALID, "%s: invalid argument: %s", __func__, #cond); \
  } while (0)
#define MI_CHECK_LAUNCH()                                                            \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess)                                                           \
      MI_FAIL(MI_ERR_LAUNCH, "%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
  } while (0)

// ---- element traits ---------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<f16_t> {
  typedef f16x8 vec8;
  static __device__ __forceinline__ float lo(uint32_t w) {
    return (float)__builtin_bit_cast(f16_t, (uint16_t)(w & 0xffffu));
  }
  static __device__ __forceinline__ float hi(uint32_t w) {
    return (float)__builtin_bit_cast(f16_t, (uint16_t)(w >> 16));
  }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (__is_same(T, bf16_t)) {
    // one v_cvt_pk_bf16_f32 for the pair: the instruction hipcc itself uses for each `(bf16)x` (round to nearest even),
    // without the two single-lane converts + shift + or around it (5 instructions per pair in the GEMM epilogues)
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    asm volatile("" : "+v"(a), "+v"(b));   // fp32 values first, then ONE conversion each (see round_to below)
    T x = (T)a, y = (T)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
  }
}

// fp32 -> T -> fp32 with the fp32 value materialised first: without the (empty) asm hipcc folds a preceding
// fp32 multiply into v_fma_mixlo_f16, which rounds the exact product ONCE to f16 -- the unfused sequence
// (and the reference's torch ops) round to fp32 first, then to f16; the two differ in rare double-rounding
// cases (measured: 28 of 524,288 q values after RoPE).  Fused kernels that promise bit-identity use this.
template <typename T> __device__ __forceinline__ float round_to(float v) {
  asm volatile("" : "+v"(v));
  return (float)(T)v;
}

// broadcast lane `N` of each 16-lane row to the whole row (gfx90a+ DPP row_newbcast)
template <int N> __device__ __forceinline__ float row_bcast(float v) {
  // bound_ctrl: every lane has a source under row_newbcast, and with it the compiler need not materialise `old` (a v_mov 0
  // in front of every DPP move otherwise)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xf, 0xf, true));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// SiLU in fp32, g / (1 + e^-g) as g * rcp(1 + 2^(-g log2 e)): v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul instead of the
// ~37 instructions of expf() + an IEEE division.  Every SiLU of the library goes through this one function, so the
// fused forms stay bit-identical to the unfused kernels; against the reference's activation.py:56-58 (torch's fp32
// silu, then one rounding to the model dtype) the fp32 value differs by a few ulp (|g| * 6e-8 relative from the scaled
// exponent, 1 ulp each from v_exp_f32 / v_rcp_f32), i.e. far inside the half-ulp of bf16 / fp16 it is rounded to
// (tests/test_elementwise_gpu.py::test_silu_and_mul_golden).  Large negative g: 2^(+big) = inf, rcp = 0, g * 0 = -0.
// Measured: the SiLU epilogue of the decode gate_up GEMM (16 elements per lane on all 8 waves) and of the prefill tile
// GEMM (~10 us per 256 x 256 tile) were issue-bound on this arithmetic.
__device__ __forceinline__ float silu_f32(float g) {
  return g * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Tuning constants.  The shipped library has them compiled in; a developer build (`make TUNING=1`, -DMI_TUNING) lets
// the environment variable of the same name override each one for A/B runs.  Nothing else in csrc/ reads the
// environment.
#ifdef MI_TUNING
#include <stdlib.h>
static inline int mi_tune(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
static inline constexpr int mi_tune(const char*, int dflt) { return dflt; }
#endif

// LDS-DMA (global -> LDS, 16 B per lane, LDS destination = wave-uniform byte address + lane*16) issued
// from inline asm so that hipcc does NOT know LDS is being written asynchronously: with the builtin it
// puts an `s_waitcnt vmcnt(0)` in front of every later ds_read that might alias, which drains the whole
// load pipeline (weights, next stage) many times per k-step.  The caller orders DMA -> ds_read itself:
// `s_waitcnt vmcnt(0)` (or a counted wait) followed by a workgroup barrier.  Hidden entries in the vmcnt
// queue can only make the compiler's own counted waits more conservative, never too weak.
__device__ __forceinline__ uint32_t lds_addr_of(const void* shared_ptr) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)shared_ptr;
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_byte_addr_uniform) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr_uniform);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// a wave-uniform pointer forced into SGPRs (values derived through integer division live in VGPRs even when uniform,
// and an "s" asm operand does not move them)
__device__ __forceinline__ const uint8_t* uniform_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)(uintptr_t)ptr;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// LDS-DMA, scalar base + 32-bit lane offset; lds_dst wave-uniform (already an SGPR value)
__device__ __forceinline__ void glds16_s(uint32_t lane_off, const uint8_t* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_off), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// the 4-byte form: 256 B per wave instruction, lane L -> LDS byte lds_dst + 4 L
__device__ __forceinline__ void glds4_s(uint32_t lane_off, const uint8_t* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_off), "s"(sbase), "s"(lds_dst)
               : "memory");
}
