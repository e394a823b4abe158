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

// n / d for a run-time d fixed per launch, without the ~25-instruction division expansion: t = mulhi(n, m);
// q = (t + ((n - t) >> s1)) >> s2  (round-up method; exact for every 32-bit n).  Host: fd_make(d).
struct FastDiv { unsigned m, s1, s2; };
inline FastDiv fd_make(unsigned d) {
    FastDiv f{0u, 0u, 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.m = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
    f.s1 = 1; f.s2 = s - 1;
    return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(n, f.m);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct ConvGeom { int M, H, W, N; int taps, nchunks, cin_pad, kh, kw; int stride, Hin, Win, Nin;
                  int ntn, ntiles, xcd_map;      // halo kernels: output-channel tiles, pixel tiles (all images), XCD-aware 1-D grid
                  int ngroups; FastDiv d_ntn, d_ntiles, d_tx, d_ty; };   // pf_conv_dma_kernel: work-sequence decode

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

// 16 accumulator rows of one output channel per lane -> split twin, TWO channels per store.  In the accumulator layout a
// lane owns one channel, so the natural twin store is a 2-byte store per value and half (32 per tile, 64-byte segments): the
// stamps / traces of the first pre-split build showed those stores costing more than the fp32 rows they replace (the fused
// combine + 1x1 went from 41 to 60 us).  Here lanes 2c / 2c+1 swap half of their rows (one DPP quad permute per pair of rows):
// the even lane ends up with channels (2c, 2c+1) of rows 0..7, the odd lane with the same channels of rows 8..15, each pair
// converted with one packed cvt and stored as ONE dword per half -- 16 stores per tile, same bits.
// `sp`: address of the hi half of this lane's EVEN channel at row p0; rs: row stride in bytes; CHECK: rows >= plimit are dead.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool CHECK, class V>
__device__ __forceinline__ void pf_store_split_pairs(char* sp, long rs, const V& v, bool odd, long p0, long plimit) {
    static_for<0, 8>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const float send = odd ? v[k] : v[k + 8];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));   // lane ^ 1
        const f32x2 ab = {odd ? recv : v[k], odd ? v[k + 8] : recv};            // (even channel, odd channel) of this lane's row
        constexpr int ro = (k & 3) + 8 * (k >> 2);                              // row offset of accumulator register k; k + 8 is 16 rows further
        const long row = ro + (odd ? 16 : 0);
        const bf16x2 hi = __builtin_convertvector(ab, bf16x2);
        const f32x2 rest = ab - __builtin_convertvector(hi, f32x2);
        const bf16x2 lo = __builtin_convertvector(rest, bf16x2);
        if (!CHECK || p0 + row < plimit) {
            char* q = sp + row * rs;
            *reinterpret_cast<bf16x2*>(q) = hi;
            *reinterpret_cast<bf16x2*>(q + 64) = lo;
        }
    });
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
                if (live(r)) o[(long)roff(r) * ld] = v[r];      // (nt stores here: no effect, profiles/r5_ab_epilogue_nt.txt)
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
                for (int r = 0; r < 16; ++r) v[r] = pf_sigmoid<FAST>(acc[t][r] + bias);
                if (d.save_gates && d.aux_out != nullptr) put(v, d.aux_out, d.ld_aux, j, nullptr, 0);      // r itself (training)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] * hv[r];
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
        } else if (epi == PF_EPI_RELU_RES || epi == PF_EPI_MASK || epi == PF_EPI_ADD) {
            const float* hp = d.h + p0 * d.ld_h + j;
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = live(r) ? hp[(long)roff(r) * d.ld_h] : 0.f;
            if (epi == PF_EPI_RELU_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(hv[r] + fmaxf(acc[t][r] + bias, 0.f), 0.f);
            } else if (epi == PF_EPI_MASK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = hv[r] > 0.f ? (acc[t][r] + bias) * d.scale : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = (acc[t][r] + bias) * d.scale + hv[r];
            }
            put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
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
            for (int r = 0; r < 16; ++r) v[r] = pf_tanh<FAST>(acc[t][r] + bias);
            if (d.save_gates && d.aux_out != nullptr) put(v, d.aux_out, d.ld_aux, j, nullptr, 0);          // q itself (training)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (1.f - zv[r]) * hv[r] + zv[r] * v[r];
            put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
        }
    }
}

// The same epilogue for the TWO pixel rows of an MFMA wave of the role-specialised kernels (acc[m][t], rows p0[m]), with
// every operand load of both rows -- bias, and h / z of the GRU kinds -- issued BEFORE the first store.  The per-tile
// form above alternates loads and stores, and since `out` may alias `h` as far as the compiler knows, each tile's loads
// wait behind the previous tile's stores: up to four dependent global round trips (s_memtime stamps of
// pf_conv_dma_kernel: 7 300 cycles from the end of the K loop to the last store retired, a tenth of a GRU launch).
template <int NT, bool CHECK, bool FAST>
__device__ __forceinline__ void tile_epilogue_pair_t(const pf_conv_desc& d, const f32x16 (&acc)[2][NT], int jb, int li,
                                                     const long (&p0)[2], const long (&plimit)[2]) {
    const int epi = d.epilogue;
    constexpr auto roff = [](int r) constexpr { return (r & 3) + 8 * (r >> 2); };
    // compile-time indices throughout (static_for): a runtime-indexed register array would live in scratch
    float bias[NT];
    f32x16 hv[2][NT], zv[2][NT];
    static_for<0, NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        const int jt = jb + 32 * t, j = jt + li;
        const bool jok = j < d.cout;
        bias[t] = jok ? d.bias[j] : 0.f;
        const bool need_h = jok && ((epi == PF_EPI_GRU_ZR && jt >= 128) || epi == PF_EPI_GRU_Q || epi == PF_EPI_RELU_RES ||
                                    epi == PF_EPI_MASK || epi == PF_EPI_ADD);
        const bool need_z = jok && epi == PF_EPI_GRU_Q;
        const int hc = epi == PF_EPI_GRU_ZR ? j - 128 : j;
        static_for<0, 2>([&](auto M) __attribute__((always_inline)) {
            constexpr int m = decltype(M)::value;
            if (need_h) {
                const float* hp = d.h + p0[m] * d.ld_h + hc;
                static_for<0, 16>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    hv[m][t][r] = (!CHECK || p0[m] + roff(r) < plimit[m]) ? hp[(long)roff(r) * d.ld_h] : 0.f;
                });
            }
            if (need_z) {
                const float* zp = d.z + p0[m] * d.ld_z + j;
                static_for<0, 16>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    zv[m][t][r] = (!CHECK || p0[m] + roff(r) < plimit[m]) ? zp[(long)roff(r) * d.ld_z] : 0.f;
                });
            }
        });
    });
    static_for<0, 2>([&](auto M) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value;
        const long pm = p0[m], pl = plimit[m];
        auto put = [&](const f32x16& v, float* base, int ld, int col, void* sbase, int lds) __attribute__((always_inline)) {
            if (base != nullptr) {
                float* o = base + pm * ld + col;
                static_for<0, 16>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    if (!CHECK || pm + roff(r) < pl) o[(long)roff(r) * ld] = v[r];
                });
            }
            if (sbase != nullptr) {
                char* sp = reinterpret_cast<char*>(sbase) + ((pm * lds + (col >> 5)) * 128 + 2 * (col & 31));
                const long rs = (long)lds * 128;
                static_for<0, 16>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    if (!CHECK || pm + roff(r) < pl) pf_store_split(sp + roff(r) * rs, v[r]);
                });
            }
        };
        static_for<0, NT>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            const int jt = jb + 32 * t, j = jt + li;
            if (j < d.cout) {
                f32x16 v;
                const float b = bias[t];
                if (epi == PF_EPI_LINEAR) {
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = (acc[m][t][r] + b) * d.scale; });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                } else if (epi == PF_EPI_RELU) {
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = fmaxf(acc[m][t][r] + b, 0.f); });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                } else if (epi == PF_EPI_GRU_ZR) {
                    if (jt < 128) {
                        static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = pf_sigmoid<FAST>(acc[m][t][r] + b); });
                        put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                    } else {
                        static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = pf_sigmoid<FAST>(acc[m][t][r] + b); });
                        if (d.save_gates && d.aux_out != nullptr) put(v, d.aux_out, d.ld_aux, j, nullptr, 0);      // r itself (training)
                        static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = v[r] * hv[m][t][r]; });
                        put(v, d.aux_out, d.ld_aux, j - 128, d.aux_split, d.lds_aux);
                    }
                } else if (epi == PF_EPI_TANH_RELU) {
                    if (jt < 128) {
                        static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = pf_tanh<FAST>(acc[m][t][r] + b); });
                        put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                    } else {
                        static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = fmaxf(acc[m][t][r] + b, 0.f); });
                        put(v, d.aux_out, d.ld_aux, j - 128, d.aux_split, d.lds_aux);
                    }
                } else if (epi == PF_EPI_RELU_RES) {
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = fmaxf(hv[m][t][r] + fmaxf(acc[m][t][r] + b, 0.f), 0.f); });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                } else if (epi == PF_EPI_MASK) {
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = hv[m][t][r] > 0.f ? (acc[m][t][r] + b) * d.scale : 0.f; });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                } else if (epi == PF_EPI_ADD) {
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = (acc[m][t][r] + b) * d.scale + hv[m][t][r]; });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                } else {   // PF_EPI_GRU_Q
                    static_for<0, 16>([&](auto R) { constexpr int r = decltype(R)::value; v[r] = pf_tanh<FAST>(acc[m][t][r] + b); });
                    if (d.save_gates && d.aux_out != nullptr) put(v, d.aux_out, d.ld_aux, j, nullptr, 0);          // q itself (training)
                    static_for<0, 16>([&](auto R) {
                        constexpr int r = decltype(R)::value;
                        v[r] = (1.f - zv[m][t][r]) * hv[m][t][r] + zv[m][t][r] * v[r];
                    });
                    put(v, d.out, d.ld_out, d.off_out + j, d.out_split, d.lds_out);
                }
            }
        });
    });
}

template <int NT, bool CHECK>
__device__ __forceinline__ void tile_epilogue_pair(const pf_conv_desc& d, const f32x16 (&acc)[2][NT], int jb, int li,
                                                   const long (&p0)[2], const long (&plimit)[2]) {
    const bool gated = d.epilogue == PF_EPI_GRU_ZR || d.epilogue == PF_EPI_GRU_Q || d.epilogue == PF_EPI_TANH_RELU;
    if (gated && d.precision == PF_PREC_F32) tile_epilogue_pair_t<NT, CHECK, false>(d, acc, jb, li, p0, plimit);
    else tile_epilogue_pair_t<NT, CHECK, true>(d, acc, jb, li, p0, plimit);
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
int pf_conv_part4_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_ws256_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t stream);
int pf_conv_ws_choice(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout);

// pf_enc_conv.hip: the weights-stationary kernel of the encoders' 3x3 64 -> 64 convolutions (round 5)
bool pf_enc_conv64_applies(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout);
int pf_enc_conv64_stats_blocks(const pfconv::ConvGeom& g);
int pf_enc_conv64_launch(const pfconv::ConvGroups& grp, const pfconv::ConvGeom& g, hipStream_t stream);

// pf_conv_dma.hip: launcher of the all-DMA kernel.  `roles` as in pf_conv2d_roles (1: 128-px tile, 2: 256 px x 64 channels).
int pf_conv_dma_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, int nt, int roles,
                       hipStream_t stream);
