// Semantics check of ds_read_b64_tr_b16 (gfx950): tile [64 rows][32 cols] bf16; for the 32x32x16 MFMA operand a
// lane (r = L&31 -> column, h = L>>5) wants rows 8h .. 8h+7 of column r.  Two reads (rows +0..3, +4..7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const __bf16* g, float* o) {
    __shared__ __attribute__((aligned(16))) __bf16 sm[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) sm[i] = g[i];
    __syncthreads();
    const int L = threadIdx.x, grp = L >> 4, w = L & 15, q = w >> 2, p = w & 3;
    const int rb = 8 * (grp >> 1), cb = 16 * (grp & 1);
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    for (int rd = 0; rd < 2; ++rd) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(sm + (rb + 4 * rd + q) * 32 + cb + 4 * p));
        bf16x4 b = __builtin_bit_cast(bf16x4, v);
        for (int e = 0; e < 4; ++e) o[L * 8 + rd * 4 + e] = (float)b[e];
    }
}
int main() {
    std::vector<__bf16> h(64 * 32);
    int bad = 0;
    for (int mode = 0; mode < 2; ++mode) {
        for (int r = 0; r < 64; ++r) for (int c = 0; c < 32; ++c) h[r * 32 + c] = (__bf16)(float)(mode ? c : r);
        __bf16* dg; float* dout; hipMalloc(&dg, h.size() * 2); hipMalloc(&dout, 64 * 8 * 4);
        hipMemcpy(dg, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dg, dout);
        std::vector<float> out(64 * 8); hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        for (int L = 0; L < 64; ++L) for (int j = 0; j < 8; ++j) {
            const int want = mode ? (L & 31) : 8 * (L >> 5) + j;
            if ((int)out[L * 8 + j] != want) { if (bad < 8) printf("mode %d lane %d elem %d: got %g want %d\n", mode, L, j, out[L * 8 + j], want); ++bad; }
        }
    }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
