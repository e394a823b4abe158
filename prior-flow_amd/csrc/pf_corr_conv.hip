// pf_dccl_combine_conv1x1: the tail of DCCL.__call__ fused with the first motion-encoder convolution
//   corr = own + img_rotate(raw, g_back)                 (core/corr.py:138, core/prior_raft.py:187-188)
//   c1   = relu(conv1x1_{324 -> 256}(corr))              (convc1_A / convc1, core/update.py:185,92)
// Round 1 ran this as pf_dccl_combine (16 us, writes the 324-channel tensor: 10.6 MB) + pf_conv2d 1x1 (28-37 us,
// reads it back) per branch and iteration, both on the critical path.  Here a workgroup (4 waves) owns 64 pixels:
//   stage 0  the four rotate-back taps of its pixels (constant per shape, from g_back) -> LDS
//   K loop   per 32-channel chunk: gather-combine (own row + 4 weighted raw rows, 16 bytes at a time, in the exact
//            operation order of pf_combine_vec4), split into bf16 hi|lo, written ONCE, to LDS (64 px x 128 B, XOR-swizzled
//            rows, double buffered) -- the combined tensor never exists in memory -- and a 64 x 256 x 32 GEMM step on
//            v_mfma_f32_32x32x16_bf16 (3-pass split, same accumulation order as the halo kernel's 1x1 path: results are
//            bit-identical to the two launches): wave w owns output channels [64 w, 64 w + 64) as a 64 x 64 register
//            tile (16 fragment reads per 24 MFMAs); its weight fragments come straight from L2 into registers three
//            chunks ahead (weights are 90 KB per branch, shared by all workgroups)
//   epilogue bias + ReLU, channel-last stores.
#include <type_traits>
#include "pf_common.h"
#include "pf_elem.h"
#include "pf_split.h"
#include "pf_conv_priv.h"
#include "../../include/priorflow_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CC_PX = 64;                 // pixels per workgroup
constexpr int CC_CIN = PF_CORR_CH;        // 324
constexpr int CC_NCH = (CC_CIN + 31) / 32;      // 11 chunks of 32 channels (352: the weight packing's Cin_pad)
constexpr int CC_COUT = 256;
constexpr int CC_LDS_A = 2 * CC_PX * 128;       // bytes: two 32-channel chunks of the combined tile
constexpr int CC_LDS = CC_LDS_A + CC_PX * 8 * 4;

struct CombGroups { pf_combine_conv_desc d[2]; };

template <int I, int N, class F>
__device__ __forceinline__ void cc_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cc_for<I + 1, N>(f);
    }
}

__global__ void __launch_bounds__(256, 1)
pf_combine_conv_kernel(const CombGroups groups, const int B, const int H, const int W) {
    extern __shared__ __attribute__((aligned(16))) char cc_smem[];
    char* const As = cc_smem;                                              // [chunk][pixel][128 B]
    int* const tap_idx = reinterpret_cast<int*>(cc_smem + CC_LDS_A);       // [pixel][4]
    float* const tap_w = reinterpret_cast<float*>(tap_idx + CC_PX * 4);    // [pixel][4]
    const pf_combine_conv_desc d = blockIdx.y == 0 ? groups.d[0] : groups.d[1];
    const int tid = threadIdx.x;
    const long N = (long)H * W, rows = (long)B * N;
    const long row0 = (long)blockIdx.x * CC_PX;

    // The launch moves 106 MB of gathered rows (every raw row is read by ~4 pixels) for 4 us of MFMA work, so the K loop
    // is built around the gather: per 32-channel chunk a thread has 10 gather loads (2 tasks x {4 raw rows, own row});
    // chunk c + 2 is in flight while chunk c + 1 is combined, split and written to LDS and chunk c is multiplied --
    // one barrier per chunk, the matrix work hides behind the memory time.  (Gathering all 352 channels first and
    // multiplying afterwards serialised the two phases chip-wide -- one workgroup per CU, all in lock step: 56-66 us.)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // ---- weight fragments: straight from L2 into registers, a ring of 4 chunks, 3 chunks ahead of their use -----------
    // weight rows n = 64 wave + 32 t + li; chunk row = {hi[32], lo[32]} bf16; pieces of K-half lh
    const char* wrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        wrow[t] = reinterpret_cast<const char*>(d.weight) + ((long)(64 * wave + 32 * t + li) * CC_NCH) * 128 + 32 * lh;
    bf16x8 fb[4][2][4];                                  // [ring slot][t][hi k0-7, hi k8-15, lo k0-7, lo k8-15]
    auto load_B = [&](auto SLOT, int chunk) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char* p = wrow[t] + chunk * 128;
            fb[slot][t][0] = *reinterpret_cast<const bf16x8*>(p);
            fb[slot][t][1] = *reinterpret_cast<const bf16x8*>(p + 16);
            fb[slot][t][2] = *reinterpret_cast<const bf16x8*>(p + 64);
            fb[slot][t][3] = *reinterpret_cast<const bf16x8*>(p + 80);
        }
    };
    load_B(std::integral_constant<int, 0>{}, 0);
    load_B(std::integral_constant<int, 1>{}, 1);
    load_B(std::integral_constant<int, 2>{}, 2);

    // ---- stage 0: rotate-back taps of the tile's pixels ---------------------------------------------------
    if (tid < CC_PX) {
        const long row = row0 + tid;
        PfTaps t;
        if (row < rows) {
            const long n = row % N;
            t = pf_taps0(pf_pymod(d.g_back[n], (float)W), d.g_back[N + n], H, W);
        } else {
            for (int j = 0; j < 4; ++j) { t.idx[j] = 0; t.w[j] = 0.f; }
        }
        for (int j = 0; j < 4; ++j) { tap_idx[tid * 4 + j] = t.idx[j]; tap_w[tid * 4 + j] = t.w[j]; }
    }
    __syncthreads();
    // ---- gather tasks of this thread: pixels px0, px0 + 32; 16-byte column kc of every chunk ---------------------------
    const int kc = tid & 7, px0 = tid >> 3;
    const float* gbase[2][5];                            // row pointers at channel 0: 4 raw rows + own row
    float gw[2][4];
    bool glive[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = px0 + 32 * i;
        const long row = row0 + px;
        glive[i] = row < rows;
        const long rr = glive[i] ? row : 0;
        const float* rb = d.raw + (rr / N) * N * d.ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) { gbase[i][j] = rb + (long)tap_idx[px * 4 + j] * d.ld; gw[i][j] = tap_w[px * 4 + j]; }
        gbase[i][4] = d.own + rr * d.ld;
    }
    f32x4 g[2][2][5];                                    // [register set][task][row]
    auto gather = [&](auto SET, int chunk) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const int c = 32 * chunk + 4 * kc;
        const int cc = c < CC_CIN ? c : 0;                // padded channels read a valid dummy column and are zeroed below
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) g[set][i][j] = *reinterpret_cast<const f32x4*>(gbase[i][j] + cc);
    };
    auto combine_to_lds = [&](auto SET, int chunk) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const bool real = 32 * chunk + 4 * kc < CC_CIN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = px0 + 32 * i;
            // pf_combine_vec4's order, component by component: vector * scalar arithmetic would compile to packed fp32 ops
            // with a broadcast selector, which MI355X gets wrong beside bf16 MFMA bursts
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float c = g[set][i][0][e] * gw[i][0];
                c = c + g[set][i][1][e] * gw[i][1];
                c = c + g[set][i][2][e] * gw[i][2];
                c = c + g[set][i][3][e] * gw[i][3];
                v[e] = (real && glive[i]) ? g[set][i][4][e] + c : 0.f;
            }
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const f32x4 rest = v - __builtin_convertvector(hi, f32x4);
            const bf16x4 lo = __builtin_convertvector(rest, bf16x4);
            const int c32 = 4 * kc, swz = (px >> 1) & 7;
            char* rowp = As + ((chunk & 1) * CC_PX + px) * 128;
            *reinterpret_cast<bf16x4*>(rowp + (((c32 >> 3) ^ swz) * 16) + (c32 & 7) * 2) = hi;
            *reinterpret_cast<bf16x4*>(rowp + ((((c32 >> 3) + 4) ^ swz) * 16) + (c32 & 7) * 2) = lo;
        }
    };
    unsigned a_piece[4];
    {
        const unsigned swz = (unsigned)((li >> 1) & 7), p0 = 2u * lh;
        a_piece[0] = ((p0 + 0) ^ swz) * 16; a_piece[1] = ((p0 + 1) ^ swz) * 16;
        a_piece[2] = ((p0 + 4) ^ swz) * 16; a_piece[3] = ((p0 + 5) ^ swz) * 16;
    }
    const char* const a_lane = As + li * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    auto compute = [&](auto SLOT, int chunk) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        bf16x8 fa[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                fa[m][q] = *reinterpret_cast<const bf16x8*>(a_lane + ((chunk & 1) * CC_PX + 32 * m) * 128 + a_piece[q]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][2 + ks], fb[slot][t][ks], acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][ks], fb[slot][t][2 + ks], acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][ks], fb[slot][t][ks], acc[m][t], 0, 0, 0);
                }
    };
    gather(std::integral_constant<int, 0>{}, 0);
    gather(std::integral_constant<int, 1>{}, 1);
    cc_for<0, CC_NCH>([&](auto CH) __attribute__((always_inline)) {
        constexpr int chunk = decltype(CH)::value;
        using SET = std::integral_constant<int, chunk & 1>;
        // LDS buffer chunk & 1 was last read by compute(chunk - 2): every wave has passed the barrier of chunk - 1 since
        combine_to_lds(SET{}, chunk);
        if constexpr (chunk + 2 < CC_NCH) gather(SET{}, chunk + 2);
        if constexpr (chunk + 3 < CC_NCH) load_B(std::integral_constant<int, (chunk + 3) & 3>{}, chunk + 3);
        __syncthreads();
        compute(std::integral_constant<int, chunk & 3>{}, chunk);
    });
    // ---- epilogue: acc[m][t][r] = pixel 32 m + (r&3) + 8 (r>>2) + 4 lh, channel 64 wave + 32 t + li --------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = 64 * wave + 32 * t + li;
        const float bias = d.bias[ch];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long prow = row0 + 32 * m + 4 * lh;
            if (d.out != nullptr) {
                float* o = d.out + prow * d.ld_out + d.off_out + ch;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pr = (r & 3) + 8 * (r >> 2);
                    if (prow + pr < rows) o[(long)pr * d.ld_out] = fmaxf(acc[m][t][r] + bias, 0.f);
                }
            }
            if (d.out_split != nullptr) {              // split twin for the DMA-fed 3x3 that follows (convc2): two channels per store
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[m][t][r] + bias, 0.f);
                char* sp = pf_split_ptr(d.out_split, prow, d.lds_out, (d.off_out + ch) & ~1);
                pfconv::pf_store_split_pairs<true>(sp, (long)d.lds_out * 128, v, (ch & 1) != 0, prow, rows);
            }
        }
    }
}

}  // namespace

extern "C" int pf_dccl_combine_conv1x1(const pf_combine_conv_desc* descs, int ngroups, int B, int H8, int W8, void* stream) {
    if (!descs || ngroups < 1 || ngroups > 2) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 1 || W8 <= 1) return PF_ERR_BAD_SHAPE;
    CombGroups g;
    for (int i = 0; i < ngroups; ++i) {
        const pf_combine_conv_desc& d = descs[i];
        if (!d.own || !d.raw || !d.g_back || !d.weight || !d.bias || (!d.out && !d.out_split)) return PF_ERR_BAD_ARG;
        if (d.cout != CC_COUT || d.ld < CC_CIN || (d.ld & 3) || d.off_out < 0 || (d.out && d.off_out + d.cout > d.ld_out)) return PF_ERR_BAD_SHAPE;
        if (d.out_split && ((d.off_out & 31) || d.off_out + d.cout > d.lds_out * 32)) return PF_ERR_BAD_SHAPE;
        g.d[i] = d;
    }
    if (ngroups == 1) g.d[1] = g.d[0];
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_combine_conv_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, CC_LDS);
    if (attr != hipSuccess) return (int)attr;
    const long rows = (long)B * H8 * W8;
    dim3 grid((unsigned)((rows + CC_PX - 1) / CC_PX), (unsigned)ngroups);
    hipLaunchKernelGGL(pf_combine_conv_kernel, grid, dim3(256), CC_LDS, (hipStream_t)stream, g, B, H8, W8);
    return (int)hipGetLastError();
}
