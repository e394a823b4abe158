// Sustained issue rate of v_mfma_f32_32x32x16_bf16 on MI355X as a function of how many waves per SIMD issue it and of how many
// independent accumulators a wave rotates over.  One workgroup per CU (grid = 256 * k), every wave runs `iters` x UNROLL MFMAs.
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM: operands are four register sets of pseudo-random full-mantissa bf16 values, rotated per MFMA (every operand bit and
// most accumulator bits toggle between consecutive instructions, as in a real GEMM); else small constants (little switching).
template <int NACC, bool RANDOM = false>
__global__ void __launch_bounds__(1024) mfma_loop(float* out, int iters, int active_waves) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= active_waves) return;
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            const float va = RANDOM ? ((int)(h >> 8 & 0xffff) - 32768) * (1.f / 32768.f) : (float)(lane & 3);
            h = h * 1664525u + 1013904223u;
            const float vb = RANDOM ? ((int)(h >> 8 & 0xffff) - 32768) * (1.f / 32768.f) : (float)((lane >> 2) & 3);
            a[k][e] = (__bf16)va; b[k][e] = (__bf16)vb;
        }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u)
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[RANDOM ? u & 3 : 0], b[RANDOM ? (u >> 2) & 3 : 0], acc[u % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool RANDOM = false>
void run(int waves_per_wg, int active, int wgs, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, RANDOM>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, out, iters, active);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double mfmas = (double)wgs * active * iters * 24;
            const double flops = mfmas * 2.0 * 32 * 32 * 16;
            const double per_simd_ns = ms * 1e6 / ((double)iters * 24 * ((active + 3) / 4) * ((wgs + 255) / 256));
            printf("%s acc=%d  waves/WG=%d issuing=%d (%.1f per SIMD)  wgs=%d: %.3f ms  %.0f TFLOP/s  %.1f ns per MFMA per SIMD\n", RANDOM ? "random  " : "constant", NACC,
                   waves_per_wg, active, active / 4.0, wgs, ms, flops / ms * 1e-9, per_simd_ns);
        }
    }
}

// The same loop on v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS give-back item 7: on random data this shape holds a
// higher clock): 48 MFMAs per trip = the FLOPs of 24 32x32x16 ones, NACC accumulators of 4 registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool RANDOM = false>
__global__ void __launch_bounds__(1024) mfma_loop16(float* out, int iters, int active_waves) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= active_waves) return;
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            const float va = RANDOM ? ((int)(h >> 8 & 0xffff) - 32768) * (1.f / 32768.f) : (float)(lane & 3);
            h = h * 1664525u + 1013904223u;
            const float vb = RANDOM ? ((int)(h >> 8 & 0xffff) - 32768) * (1.f / 32768.f) : (float)((lane >> 2) & 3);
            a[k][e] = (__bf16)va; b[k][e] = (__bf16)vb;
        }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u)
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM ? u & 3 : 0], b[RANDOM ? (u >> 2) & 3 : 0], acc[u % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool RANDOM = false>
void run16(int waves_per_wg, int active, int wgs, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop16<NACC, RANDOM>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, out, iters, active);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double mfmas = (double)wgs * active * iters * 48;
            const double flops = mfmas * 2.0 * 16 * 16 * 32;
            const double per_simd_ns = ms * 1e6 / ((double)iters * 48 * ((active + 3) / 4) * ((wgs + 255) / 256));
            printf("16x16x32 %s acc=%d  waves/WG=%d issuing=%d (%.1f per SIMD)  wgs=%d: %.3f ms  %.0f TFLOP/s  %.1f ns per MFMA per SIMD\n", RANDOM ? "random  " : "constant", NACC,
                   waves_per_wg, active, active / 4.0, wgs, ms, flops / ms * 1e-9, per_simd_ns);
        }
    }
}

int main() {
    float* out; hipMalloc(&out, (size_t)1024 * 1024 * 4);
    const int iters = 4000;
    for (int wgs : {256}) {
        run<4>(4, 4, wgs, iters, out);      // one issuing wave per SIMD, 4 accumulators (the DMA kernel's NT=2 stream)
        run<2>(4, 4, wgs, iters, out);
        run<4>(8, 8, wgs, iters, out);      // two issuing waves per SIMD
        run<2>(8, 8, wgs, iters, out);
        run<4>(8, 4, wgs, iters, out);      // 8 waves resident, 4 issuing (the idle ones exit at once)
        run<4>(16, 16, wgs, iters, out);    // four per SIMD
    }
    run<4, true>(4, 4, 256, iters, out);    // the same with operands that toggle like real data
    run<4, true>(8, 8, 256, iters, out);
    run<4, true>(4, 4, 256, 10 * iters, out);   // ten times longer (15 ms): does the rate sag with time?
    run<4, true>(4, 4, 64, iters, out);
    run<4>(4, 4, 64, iters, out);           // a quarter of the CUs busy: does the per-SIMD rate change with chip-wide load?
    run<4>(8, 8, 64, iters, out);
    // ---- v_mfma_f32_16x16x32_bf16 beside the rows above (round 4) ----
    run16<16>(4, 4, 256, iters, out);
    run16<16>(8, 8, 256, iters, out);
    run16<16, true>(4, 4, 256, iters, out);
    run16<16, true>(8, 8, 256, iters, out);
    run16<4, true>(4, 4, 256, iters, out);
    run16<16, true>(4, 4, 256, 10 * iters, out);
    run16<16, true>(4, 4, 64, iters, out);
    run<4, true>(4, 4, 256, iters, out);        // 32x32x16 again, after the chip has been busy for a while
    run16<16, true>(4, 4, 256, iters, out);
    return 0;
}
