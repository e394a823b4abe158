// Split twins of channel-last activations (include/priorflow_hip.h, pf_conv_desc): per pixel row and 32-channel chunk the
// 128 bytes {bf16 hi[32], bf16 lo[32]}, hi = bf16(x) (round to nearest even), lo = bf16(x - hi) -- the operand format of the
// PF_PREC_BF16X3 GEMMs, written by the producer so that the consumer's operands go global -> LDS by DMA.  Device only.
#pragma once
#include "pf_common.h"
#if defined(__HIPCC__)
// address of the hi half of element (row, ch); the lo half sits 64 bytes further
__device__ __forceinline__ char* pf_split_ptr(void* base, long row, int lds, int ch) {
    return reinterpret_cast<char*>(base) + ((row * lds + (ch >> 5)) * 128 + 2 * (ch & 31));
}
// same arithmetic as the staging split of the fp32 kernels (v_cvt_pk_bf16_f32, exact fp32 subtraction)
__device__ __forceinline__ void pf_split_store(char* p, float v) {
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    *reinterpret_cast<__bf16*>(p) = hi;
    *reinterpret_cast<__bf16*>(p + 64) = lo;
}
// N = 4 | 8 consecutive channels starting at a multiple of N: one 2N-byte store per half
template <int N>
__device__ __forceinline__ void pf_split_store_n(char* p, const float (&v)[N]) {
    typedef __bf16 bfN __attribute__((ext_vector_type(N)));
    typedef float fN __attribute__((ext_vector_type(N)));
    fN x;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = v[i];
    const bfN hi = __builtin_convertvector(x, bfN);
    const fN rest = x - __builtin_convertvector(hi, fN);
    const bfN lo = __builtin_convertvector(rest, bfN);
    *reinterpret_cast<bfN*>(p) = hi;
    *reinterpret_cast<bfN*>(p + 64) = lo;
}
#endif
