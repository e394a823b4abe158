// Private pieces shared by the pf_conv2d kernel files (pf_conv_mfma.hip: generic / halo / role-specialised kernels,
// pf_conv_dma.hip: the all-DMA kernel for pre-split activations): launch descriptors, geometry, and the fused tile
// epilogue (bias / activation / SepConvGRU gating, fp32 and bf16 hi|lo "split twin" outputs).
#pragma once
#include <type_traits>
#include "pf_common.h"
#include "../../include/priorflow_hip.h"

namespace pfconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: plain 16-byte loads, no struct memcpy
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32;        // channels per K-step
constexpr int LDS_LD = 36;    // padded row stride (floats) of the register-staged LDS images
constexpr int MAX_GROUPS = 4;

struct ConvGroups { pf_conv_desc d[MAX_GROUPS]; };

struct ConvGeom { int M, H, W, N; int taps, nchunks, cin_pad, kh, kw; int stride, Hin, Win, Nin;
                  int ntn, ntiles, xcd_map; };   // halo kernels: output-channel tiles, pixel tiles (all images), XCD-aware 1-D grid

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Gate nonlinearities of the fused epilogues on the hardware transcendental pipe: v_exp_f32 + v_rcp_f32 (1 ulp each)
// instead of ocml's expf / tanhf and an IEEE division -- ~5 instructions per element instead of ~25; absolute error
// < 3e-7 on outputs in (-1, 1), two orders below the bf16x3 GEMM's own rounding.
// (PF_PREC_F32, the exact validation mode, keeps expf / tanhf and the IEEE division.)
template <bool FAST>
__device__ __forceinline__ float pf_sigmoid(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
    else return 1.f / (1.f + expf(-x));
}
template <bool FAST>
__device__ __forceinline__ float pf_tanh(float x) {             // fast form: 1 - 2 / (1 + e^(2x)); saturates cleanly at +-1
    if constexpr (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
    else return tanhf(x);
}

// One value into a split twin: hi = bf16(v) (round to nearest even, v_cvt_pk_bf16_f32), lo = bf16(v - hi) -- the
// arithmetic of the staging split of the fp32 kernels, so a consumer of the twin sees the operand bits it would have made
// itself.  `p` points at the hi half of the element (the lo half sits 64 bytes further).
__device__ __forceinline__ void pf_store_split(char* p, float v) {
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    *reinterpret_cast<__bf16*>(p) = hi;
    *reinterpret_cast<__bf16*>(p + 64) = lo;
}

// Fused epilogue of a wave's NT 32x32 accumulators.  acc[t][r] is output channel jb + 32 t + li
// of pixel p0 + (r&3) + 8 (r>>2)  (p0 already holds the lane's +4*(lane>>5) row offset).
// The kind is tested once per tile, a lane keeps one base pointer per array and adds row * ld offsets, the channel-half
// decisions of the split epilogues are wave-uniform (jb is scalar), and the GRU operands of a tile are gathered before
// anything is stored.  Every output exists in up to two forms: fp32 rows (`out` / `aux_out`, may be NULL when the twin is
// given) and the bf16 hi|lo split twin (`out_split` / `aux_split`) the DMA-fed convolutions consume.
template <int NT, bool CHECK, bool FAST>
__device__ __forceinline__ void tile_epilogue_t(const pf_conv_desc& d, const f32x16 (&acc)[NT], int jb, int li,
                                                long p0, long plimit) {
    const int epi = d.epilogue;
    auto roff = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto live = [&](int r) { return !CHECK || p0 + roff(r) < plimit; };
    // rows of 16 values -> fp32 column (o, ld) and / or split twin (sp, chunks per row)
    auto put = [&](const float (&v)[16], float* base, int ld, int col, void* sbase, int lds) __attribute__((always_inline)) {
        if (base != nullptr) {
            float* o = base + p0 * ld + col;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (live(r)) o[(long)roff(r) * ld] = v[r];
        }
        if (sbase != nullptr) {
            char* sp = reinterpret_cast<char*>(sbase) + ((p0 * lds + (col >> 5)) * 128 + 2 * (col & 31));
            const long rs = (long)lds * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (live(r)) pf_store_split(sp + roff(r) * rs, v[r]);
        }
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int jt = jb + 32 * t;                 // wave-uniform
        const int j = jt + li;
        if (j >= d.cout) continue;
        const float bias = d.bias[j];
        float v[16];
        if (epi == PF_EPI_LINEAR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (acc[t][r] + bias) * d.scale;
            put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
        } else if (epi == PF_EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[t][r] + bias, 0.f);
            put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
        } else if (epi == PF_EPI_GRU_ZR) {
            if (jt < 128) {                                                       // z
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = pf_sigmoid<FAST>(acc[t][r] + bias);
                put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
            } else {                                                              // r * h
                const float* hp = d.h + p0 * d.ld_h + (j - 128);
                float hv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) hv[r] = live(r) ? hp[(long)roff(r) * d.ld_h] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = pf_sigmoid<FAST>(acc[t][r] + bias) * hv[r];
                put(v, d.aux_out, d.ld_aux, j - 128, d.aux_split, d.lds_aux);
            }
        } else if (epi == PF_EPI_TANH_RELU) {
            if (jt < 128) {                                                       // net
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = pf_tanh<FAST>(acc[t][r] + bias);
                put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
            } else {                                                              // inp
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[t][r] + bias, 0.f);
                put(v, d.aux_out, d.ld_aux, j - 128, d.aux_split, d.lds_aux);
            }
        } else {   // PF_EPI_GRU_Q:  h' = (1 - z) h + z tanh(v)
            const float* zp = d.z + p0 * d.ld_z + j;
            const float* hp = d.h + p0 * d.ld_h + j;
            float zv[16], hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zv[r] = live(r) ? zp[(long)roff(r) * d.ld_z] : 0.f;
                hv[r] = live(r) ? hp[(long)roff(r) * d.ld_h] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (1.f - zv[r]) * hv[r] + zv[r] * pf_tanh<FAST>(acc[t][r] + bias);
            put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
        }
    }
}

template <int NT, bool CHECK>
__device__ __forceinline__ void tile_epilogue(const pf_conv_desc& d, const f32x16 (&acc)[NT], int jb, int li,
                                              long p0, long plimit) {
    const bool gated = d.epilogue == PF_EPI_GRU_ZR || d.epilogue == PF_EPI_GRU_Q || d.epilogue == PF_EPI_TANH_RELU;
    if (gated && d.precision == PF_PREC_F32) tile_epilogue_t<NT, CHECK, false>(d, acc, jb, li, p0, plimit);
    else tile_epilogue_t<NT, CHECK, true>(d, acc, jb, li, p0, plimit);
}

}  // namespace pfconv

// pf_conv_mfma.hip (compiled as four units, PF_CONV_PART): kernels by pf_conv2d_tile code -- part 0: tiles 0..2 (generic
// kernel), part 1: tile 3, part 2: tile 4, part 3: tile 5 and the 256-px role-specialised tile; pf_conv_ws_choice: which
// wave organisation a TH = 4 launch takes (pf_conv2d_roles).
int pf_conv_part0_launch(int tile_id, const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout,
                         bool split, hipStream_t stream);
int pf_conv_part1_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_part2_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_part3_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_ws256_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_ws_choice(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout);

// pf_conv_dma.hip: launcher of the all-DMA kernel.  `roles` as in pf_conv2d_roles (1: 128-px tile, 2: 256 px x 64 channels).
int pf_conv_dma_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, int nt, int roles,
                       hipStream_t stream);
