// TEMPORARY: entry points not implemented yet report MI_ERR_UNSUPPORTED (loudly).
#include "common.h"
#define STUB(name) MI_FAIL(MI_ERR_UNSUPPORTED, #name ": not implemented yet")
extern "C" int mi_extend_attn(const void*, const void*, const void*, void*, const void*, const void*,
                              const int32_t*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t,
                              int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float,
                              float, int, int64_t, int, void*) { STUB(mi_extend_attn); }
extern "C" int mi_fp8_quant_per_tensor(const void*, void*, float*, int64_t, int64_t, int64_t, int, int, void*) { STUB(mi_fp8_quant_per_tensor); }
extern "C" int mi_fp8_quant_per_token(const void*, void*, float*, int64_t, int64_t, int64_t, int, void*) { STUB(mi_fp8_quant_per_token); }
extern "C" int mi_fp8_gemm(const void*, const void*, const float*, const float*, const void*, void*, int64_t,
                           int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, void*) { STUB(mi_fp8_gemm); }
extern "C" int mi_w4a16_gemm(const void*, const int32_t*, const int32_t*, const void*, const int32_t*,
                             const void*, void*, int64_t, int64_t, int64_t, int64_t, int, int, void*) { STUB(mi_w4a16_gemm); }
extern "C" int mi_w4_dequantize(const int32_t*, const int32_t*, const void*, const int32_t*, void*, int64_t,
                                int64_t, int64_t, int, int, void*) { STUB(mi_w4_dequantize); }
