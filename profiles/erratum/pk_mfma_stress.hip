// MI355X (gfx950) reproducer: packed-fp32 VALU instructions whose op_sel / op_sel_hi swaps or broadcasts the halves
// of a VGPR source return wrong results in lanes 48..63 while another wave on the same SIMD starts a burst of
// v_mfma_f32_32x32x16_bf16.
//
// One workgroup = 8 waves: waves 0..3 (one per SIMD) run a chain of packed / scalar fp32 operations on operands
// they re-load every iteration; waves 4..7 (same SIMDs) run the co-runner when `with_mfma` is set: bursts of four
// matrix instructions separated by s_sleep BURST (0 = back to back).  The results of waves 0..3 with and without
// the co-runner must be bit-identical.  Variants (compile time, the instruction mix is what matters):
//   -DPK_KIND=0  v_pk_mul/add_f32 with half SWAPS of VGPR sources   (op_sel:[0,1] op_sel_hi:[1,0])   -> FAILS
//   -DPK_KIND=1  v_pk_mul/add_f32 with half BROADCASTS of VGPR sources (q.xx, q.yy)                  -> FAILS
//   -DPK_KIND=2  v_pk_mul/add_f32, default selectors only                                             -> ok
//   -DPK_KIND=3  scalar v_mul/v_add_f32 (build with -fno-slp-vectorize)                               -> ok
//   -DPK_KIND=4  scalar VALU with DPP operands (quad_perm, row_shl; -fno-slp-vectorize)               -> ok
//   -DPK_KIND=5  v_cvt_pk_bf16_f32 + fp64 + 64-bit integer adds (the other wide VALU ops of the product) -> ok
//   -DMFMA_KIND=0 v_mfma_f32_32x32x16_bf16 (default)   =1 v_mfma_f32_32x32x2_f32 -> ok   =2 plain VALU loop -> ok
// Build + run the matrix: profiles/erratum/run.sh (results of this round: profiles/r1_pk_mfma_erratum.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef PK_KIND
#define PK_KIND 0
#endif
#ifndef MFMA_KIND
#define MFMA_KIND 0
#endif
#ifndef BURST
#define BURST 8
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) stress(const float* in, float* out, float* sink, int iters, int with_mfma) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {                                    // ---- co-runner: shares SIMD (wave - 4) with a packed wave
        if (!with_mfma) return;
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)in[lane * 8 + k]; b[k] = (__bf16)in[512 + lane * 8 + k]; }
        for (int it = 0; it < iters; ++it) {
            if (BURST > 0) __builtin_amdgcn_s_sleep(BURST);        // idle gap: the next four instructions are a burst onset
            for (int t = 0; t < 4; ++t) {
#if MFMA_KIND == 0
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
#elif MFMA_KIND == 1
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[t], 0, 0, 0);
#else
                for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] * 1.0001f + (float)a[r & 7];
#endif
            }
        }
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 12345.678f) sink[0] = s;               // keep the loop alive
        return;
    }
    // ---- waves 0..3: 8 accumulators, operands re-loaded every iteration (so nothing is hoisted out of the loop)
    v2f x[6], w[6], s[4];
    for (int k = 0; k < 4; ++k) s[k] = v2f{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const float* src = in + (it & 1) * 2048;
        for (int k = 0; k < 6; ++k) {
            x[k] = *reinterpret_cast<const v2f*>(src + (lane * 6 + k) * 2);
            w[k] = *reinterpret_cast<const v2f*>(src + 1024 + (lane * 6 + k) * 2);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const v2f q = x[p + k > 5 ? 5 : p + k];
#if PK_KIND == 0
                const v2f t0 = q * w[k], t1 = q.yx * w[k + 3];
                s[p] = s[p] + (t0 + t1.yx) * 0.001f;
#elif PK_KIND == 1
                const v2f t0 = q.xx * w[k], t1 = q.yy * w[k + 3];
                s[p] = s[p] + (t0 + t1) * 0.001f;
#elif PK_KIND == 2
                const v2f t0 = q * w[k], t1 = q * w[k + 3];
                s[p] = s[p] + (t0 + t1) * 0.001f;
#elif PK_KIND == 3
                const float a0 = q.x * w[k].x + q.y * w[k + 3].y, a1 = q.y * w[k].y + q.x * w[k + 3].x;
                s[p].x = s[p].x + a0 * 0.001f; s[p].y = s[p].y + a1 * 0.001f;
#elif PK_KIND == 5           // other wide VALU ops of the product's kernels: v_cvt_pk_bf16_f32, 64-bit integer and fp64 adds
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 hb = __builtin_convertvector(q * w[k], bf16x2);                 // v_cvt_pk_bf16_f32
                const unsigned hbits = __builtin_bit_cast(unsigned, hb);
                const float h0 = __builtin_bit_cast(float, hbits << 16), h1 = __builtin_bit_cast(float, hbits & 0xffff0000u);
                const double d = (double)s[p].x + (double)h0 * 0.001;                        // v_cvt_f64_f32, v_fma/add_f64
                const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, s[p].y) + (hbits & 0xffu);   // 64-bit add
                s[p].x = (float)d; s[p].y = __builtin_bit_cast(float, (unsigned)u & 0x3fffffffu) * 0.5f + h1 * 0.001f;
#else
                const float a0 = q.x * w[k].x, a1 = q.y * w[k + 3].y;
                const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0xB1, 0xf, 0xf, true));
                const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x104, 0xf, 0xf, true));
                s[p].x = s[p].x + (a0 + d0) * 0.001f; s[p].y = s[p].y + (a1 + d1) * 0.001f;
#endif
            }
    }
    for (int k = 0; k < 4; ++k) {
        out[((blockIdx.x * 4 + wave) * 64 + lane) * 8 + 2 * k] = s[k].x;
        out[((blockIdx.x * 4 + wave) * 64 + lane) * 8 + 2 * k + 1] = s[k].y;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 200, reps = argc > 3 ? atoi(argv[3]) : 50;
    std::vector<float> h(4096);
    unsigned st = 12345;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 65536.f - 0.5f; }
    float *in, *out, *sink;
    const size_t n = (size_t)blocks * 4 * 64 * 8;
    if (hipMalloc(&in, h.size() * 4) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 2;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ref(n), got(n);
    hipLaunchKernelGGL(stress, dim3(blocks), dim3(512), 0, 0, in, out, sink, iters, 0);
    hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode) {
        long bad_runs = 0, bad_vals = 0, rows[4] = {0, 0, 0, 0};
        for (int r = 0; r < reps; ++r) {
            hipMemset(out, 0, n * 4);
            hipLaunchKernelGGL(stress, dim3(blocks), dim3(512), 0, 0, in, out, sink, iters, mode);
            hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost);
            long b = 0;
            for (size_t i = 0; i < n; ++i)
                if (memcmp(&got[i], &ref[i], 4)) { ++b; ++rows[((i / 8) % 64) / 16]; }
            bad_vals += b; bad_runs += b != 0;
        }
        printf("PK_KIND=%d MFMA_KIND=%d BURST=%d co-runner=%s: %ld of %d launches differ (%ld of %zu values; lanes 0-15/16-31/32-47/48-63: %ld %ld %ld %ld)\n",
               PK_KIND, MFMA_KIND, BURST, mode ? "on " : "off", bad_runs, reps, bad_vals, n * (size_t)reps, rows[0], rows[1], rows[2], rows[3]);
    }
    return 0;
}
