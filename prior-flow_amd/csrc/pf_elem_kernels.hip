// __global__ wrappers + C-ABI launchers of the per-element kernels (gfx950).
// The math lives in pf_elem.h; see include/priorflow_hip.h for the ABI contract.
#include "pf_elem.h"

namespace {

constexpr int kBlock = 256;
constexpr long kMaxBlocks = 256L * 64;   // 256 CUs x 64 blocks, grid-stride beyond that

template <class Args, void (*F)(long, const Args&)>
__global__ void __launch_bounds__(kBlock) pf_elem_kernel(const Args a, const long total) {
    long idx = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; idx < total; idx += stride) F(idx, a);
}

template <class Args, void (*F)(long, const Args&)>
int pf_launch_elem(const Args& a, long total, void* stream) {
    if (total <= 0) return PF_OK;
    long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL((pf_elem_kernel<Args, F>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, a, total);
    return (int)hipGetLastError();
}

// K5 on a wavefront: one pixel per wave, 4 channels per lane (C == 256), the four 64-channel
// group sums reduced with 16-lane butterfly shuffles.
__global__ void __launch_bounds__(kBlock) pf_warp_gcorr_wave(const PfWarpGcorrArgs a, const long rows) {
    const int lane = threadIdx.x & 63;
    const long N = (long)a.H * a.W;
    long row = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (kBlock / 64);
    for (; row < rows; row += stride) {
        const long b = row / N, n = row % N;
        const PfTaps t = pf_warp_taps(a, b, n);
        const float4 v1 = reinterpret_cast<const float4*>(a.f1 + row * 256)[lane];
        const float* f2b = a.f2 + b * N * 256;
        float4 s;
        {
            const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[0] * 256)[lane];
            s.x = v.x * t.w[0]; s.y = v.y * t.w[0]; s.z = v.z * t.w[0]; s.w = v.w * t.w[0];
        }
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[q] * 256)[lane];
            s.x = s.x + v.x * t.w[q]; s.y = s.y + v.y * t.w[q];
            s.z = s.z + v.z * t.w[q]; s.w = s.w + v.w * t.w[q];
        }
        float p = v1.x * s.x;
        p = p + v1.y * s.y;
        p = p + v1.z * s.z;
        p = p + v1.w * s.w;
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);
        p += __shfl_xor(p, 8);
        if ((lane & 15) == 0) a.dst.ptr[row * a.dst.ld + a.dst.c_off + (lane >> 4)] = p / 64.f;
    }
}

// Tiled direct convolution for the tiny-Cin layers (7x7 2->128, 3x3 8->32, 3x3 32->16).
// One workgroup = 32 consecutive pixels of one image row x all Cout.  The (KH) x (32+KW-1) x Cin
// input patch is staged once in LDS (zero padded); a thread owns ONE output channel and
// 32*Cout/256 pixels, so the weight stream is coalesced over lanes (packed [tap][cin][cout])
// and every LDS read is a broadcast (all lanes of a pixel group read the same address).
// Same arithmetic order as pf_direct_conv_elem (taps outer, channels inner).
template <int COUT>
__global__ void __launch_bounds__(256) pf_direct_conv_tile(const PfDirectConvArgs a) {
    constexpr int TP = 32;                       // pixels per tile
    constexpr int GROUPS = 256 / COUT;           // pixel groups
    constexpr int PPT = TP / GROUPS;             // pixels per thread
    static_assert(256 % COUT == 0 && TP % GROUPS == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float patch[];   // [KH][TP+KW-1][Cin]
    const int tiles_x = a.W / TP;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x;
    const int y = (tile / tiles_x) % a.H;
    const long b = tile / ((long)tiles_x * a.H);
    const int x0 = tx * TP;
    const int ph = a.KH / 2, pw = a.KW / 2;
    const int PW = TP + a.KW - 1;
    const long N = (long)a.H * a.W;
    const int total = a.KH * PW * a.Cin;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i % a.Cin;
        const int px = (i / a.Cin) % PW;
        const int ky = i / (a.Cin * PW);
        const int yy = y + ky - ph, xx = x0 + px - pw;
        float v = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
            v = a.in[(b * N + (long)yy * a.W + xx) * a.ld_in + a.c_in_off + c];
        patch[i] = v;
    }
    __syncthreads();
    const int co = threadIdx.x % COUT;
    const int grp = threadIdx.x / COUT;
    float acc[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) acc[i] = 0.f;
    const float* wp = a.w + co;
    for (int ky = 0; ky < a.KH; ++ky)
        for (int kx = 0; kx < a.KW; ++kx) {
            const float* prow = patch + (ky * PW + grp * PPT + kx) * a.Cin;
            for (int c = 0; c < a.Cin; ++c) {
                const float w = wp[((long)(ky * a.KW + kx) * a.Cin + c) * COUT];
#pragma unroll
                for (int i = 0; i < PPT; ++i) acc[i] = acc[i] + prow[i * a.Cin + c] * w;
            }
        }
    const float bias = a.bias[co];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        float v = acc[i] + bias;
        if (a.relu) v = fmaxf(v, 0.f);
        const long row = b * N + (long)y * a.W + x0 + grp * PPT + i;
        a.out[row * a.ld_out + a.c_out_off + co] = v;
    }
}

template <int COUT>
int launch_direct_tile(const PfDirectConvArgs& a, void* stream) {
    const size_t lds = (size_t)a.KH * (32 + a.KW - 1) * a.Cin * sizeof(float);
    const long tiles = (long)a.B * a.H * (a.W / 32);
    hipLaunchKernelGGL(pf_direct_conv_tile<COUT>, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace

#define PF_LAUNCH(name, args, total, stream) \
    pf_launch_elem<decltype(args), pf_##name##_elem>(args, total, stream)

// device build: route pf_conv2d_direct to the tiled kernel when the shape allows
#define PF_DIRECT_CONV_LAUNCH(a, total, stream)                                                  \
    (((a).W % 32 == 0 && (size_t)(a).KH * (32 + (a).KW - 1) * (a).Cin * 4 <= 48 * 1024)          \
         ? ((a).Cout == 128 ? launch_direct_tile<128>(a, stream)                                 \
            : (a).Cout == 32 ? launch_direct_tile<32>(a, stream)                                 \
            : (a).Cout == 16 ? launch_direct_tile<16>(a, stream)                                 \
                             : pf_launch_elem<PfDirectConvArgs, pf_direct_conv_elem>(a, total, stream)) \
         : pf_launch_elem<PfDirectConvArgs, pf_direct_conv_elem>(a, total, stream))

#include "pf_api_elem.inc"

extern "C" int pf_warp_gcorr(const float* f1, const float* f2, const float* coords, int add_grid,
                             float* dst, int dst_ld, int dst_off, int B, int H8, int W8, int C,
                             void* stream) {
    PfWarpGcorrArgs a;
    const int rc = pf_warp_gcorr_fill(a, f1, f2, coords, add_grid, dst, dst_ld, dst_off, B, H8, W8, C);
    if (rc != PF_OK) return rc;
    const long rows = (long)B * H8 * W8;
    if (C == 256) {
        long blocks = (rows + 3) / 4;
        if (blocks > kMaxBlocks) blocks = kMaxBlocks;
        hipLaunchKernelGGL(pf_warp_gcorr_wave, dim3((unsigned)blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, a, rows);
        return (int)hipGetLastError();
    }
    return PF_LAUNCH(warp_gcorr, a, rows * 4, stream);
}

extern "C" const char* pf_version(void) {
    return "priorflow-hip r1 gfx950 (fp32 MFMA 32x32x2 implicit-GEMM convs, fused corr+pyramid)";
}
