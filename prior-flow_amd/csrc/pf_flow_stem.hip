// pf_flow_stem: the motion encoders' 7x7 stride-1 convolutions over the TWO flow channels (core/update.py:87 convf1,
// :173 convf1_A, :175 convf1_B: 2 -> 128, ReLU; three per refinement iteration, one launch) on the matrix cores (round 4).
//
// Rounds 1-3 ran them on the vector ALUs (pf_stem7x7c2_valu: a lane = one pixel x 8 channels, weights by scalar loads): 24.6 us
// alone on the chip for the three stems of a 64 x 128 map -- 0.6 GFLOP, bound by the latency of ~800 scalar weight loads per
// wave and by staging the same input patch once per 16 output channels.  Here the K axis is the patch itself, one MFMA K-step per
// patch row: k = ky * 16 + kx * 2 + c (the 14 interleaved floats of a row of the 7x7x2 patch, padded to 16) -> K = 112 = 7 steps
// of v_mfma_f32_32x32x16_bf16, and a pixel's A fragment piece (8 consecutive k) is 8 CONTIGUOUS floats of the channel-last flow
// patch in LDS: four ds_read_b64 + the bf16 hi|lo split (same split arithmetic and pass order as every other PF_PREC_BF16X3
// kernel: x_lo * w_hi, x_hi * w_lo, x_hi * w_hi, fp32 accumulate).
//
// One workgroup = 4 waves = a 4-row x 32-column tile of output pixels x all 128 channels; wave w owns row w (four 32x32
// accumulators).  The fp32 weights [98][128] are split to bf16 hi|lo on the way into LDS (58 KB; every workgroup does this once:
// a 64 x 128 map is 64 tiles per stem, fewer workgroups than CUs, so nothing is persistent), the 10 x 38 x 2 input patch is
// 3 KB: 61 KB, two workgroups per CU.  Epilogue = the shared tile epilogue (bias, ReLU, fp32 rows and / or split twin).
#include <stdlib.h>
#include "pf_conv_priv.h"
#include "pf_flow_stem.h"

namespace {
using namespace pfconv;

constexpr int FS_STEPS = 7;               // MFMA K-steps = patch rows
constexpr int FS_WROW = 14 * 32;          // bytes per channel: 14 pieces of 8 k, each {bf16 hi[8], bf16 lo[8]}
constexpr int FS_WLDS = FS_WROW + 16;     // 464-byte LDS row stride (116 dwords = 52 mod 64: conflict-free ds_read_b128 over 32 rows)
constexpr int FS_TR = 4;                  // tile rows, one per wave
constexpr int FS_PROWS = FS_TR + 6;
constexpr int FS_PW = 80;                 // floats per patch row: 38 pixels x 2 channels = 76, padded (a piece may read up to float 77)
constexpr int FS_LDS = 128 * FS_WLDS + FS_PROWS * FS_PW * 4;
constexpr int FS_WPER = 98 * 128 / 256;   // 49 weights per thread

__global__ void __launch_bounds__(256, 2)
pf_flow_stem_kernel(const PfFlowStemMulti mm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PfFlowStemProblem& a = mm.p[blockIdx.y];
    char* const wl = smem;
    float* const patch = reinterpret_cast<float*>(smem + 128 * FS_WLDS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int H = mm.H, W = mm.W;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + FS_TR - 1) / FS_TR;
    const int tx = (int)(blockIdx.x % tiles_x), ty = (int)((blockIdx.x / tiles_x) % tiles_y);
    const long b = blockIdx.x / ((long)tiles_x * tiles_y);
    const int y0 = ty * FS_TR, x0 = tx * 32;

    // ---- all global loads of the prologue in flight together: 49 weights (element tid + 256 u = weight row 2 u + c, channel co
    //      with co = tid % 128, c = tid / 128: a thread owns ONE input channel of ONE output channel, tap u) and 3 patch floats
    float wv[FS_WPER];
#pragma unroll
    for (int u = 0; u < FS_WPER; ++u) wv[u] = a.w[tid + 256 * u];
    float pv[3];
    int pdst[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = tid + 256 * u;
        const int r = e / 76, t = e % 76;
        const int yy = y0 - 3 + r, xx = x0 - 3 + (t >> 1);
        const bool in = e < FS_PROWS * 76 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* src = a.in + ((b * H + (in ? yy : 0)) * (long)W + (in ? xx : 0)) * a.ld_in + a.c_in_off + (t & 1);
        const float v = *src;                                  // always a legal address: a branch-free load
        pv[u] = in ? v : 0.f;
        pdst[u] = e < FS_PROWS * 76 ? r * FS_PW + t : -1;
    }
    {   // weights -> LDS as bf16 hi | lo: k' = ky * 16 + kx * 2 + c, piece = k' / 8
        const int co = tid & 127, c = tid >> 7;
        char* const row = wl + co * FS_WLDS;
#pragma unroll
        for (int u = 0; u < FS_WPER; ++u) {                    // tap u = (ky, kx)
            const int ky = u / 7, kx = u % 7;
            const int kk = kx * 2 + c;
            char* const p = row + (2 * ky + (kk >> 3)) * 32 + (kk & 7) * 2;
            const __bf16 h = (__bf16)wv[u];
            *reinterpret_cast<__bf16*>(p) = h;
            *reinterpret_cast<__bf16*>(p + 16) = (__bf16)(wv[u] - (float)h);
        }
        if (tid < 128) {                                       // k' = 14, 15 of every patch row: zero weights
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                *reinterpret_cast<unsigned*>(row + (2 * ky + 1) * 32 + 12) = 0u;
                *reinterpret_cast<unsigned*>(row + (2 * ky + 1) * 32 + 28) = 0u;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (pdst[u] >= 0) patch[pdst[u]] = pv[u];
    if (tid < FS_PROWS * 4) patch[(tid >> 2) * FS_PW + 76 + (tid & 3)] = 0.f;      // the padding floats are read (x zero weights)
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // output pixel (row y0 + wave, column x0 + li) reads patch rows wave + ky, floats 2 li .. 2 li + 15; this lane's piece: + 8 lh
    const float* const ap0 = patch + wave * FS_PW + 2 * li + 8 * lh;
    const char* const bp0 = wl + li * FS_WLDS + lh * 32;
    static_for<0, FS_STEPS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const float* ap = ap0 + s * FS_PW;
        bf16x8 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(ap + 2 * q);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const __bf16 h = (__bf16)v[e];
                hi[2 * q + e] = h;
                lo[2 * q + e] = (__bf16)(v[e] - (float)h);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(bp0 + t * 32 * FS_WLDS + s * 64);
            const bf16x8 wo = *reinterpret_cast<const bf16x8*>(bp0 + t * 32 * FS_WLDS + s * 64 + 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, wh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, wo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, wh, acc[t], 0, 0, 0);
        }
    });

    // ---- epilogue: acc[t][r] = pixel (row y0 + wave, column x0 + (r & 3) + 8 (r >> 2) + 4 lh), channel 32 t + li
    pf_conv_desc d;
    d.bias = a.bias; d.out = a.out; d.ld_out = a.ld_out; d.off_out = a.c_out_off; d.cout = 128;
    d.epilogue = a.relu ? PF_EPI_RELU : PF_EPI_LINEAR;
    d.scale = 1.f; d.h = nullptr; d.ld_h = 0; d.z = nullptr; d.ld_z = 0; d.aux_out = nullptr; d.ld_aux = 0;
    d.precision = PF_PREC_BF16X3; d.out_split = a.out_split; d.lds_out = a.lds_out; d.aux_split = nullptr; d.lds_aux = 0; d.save_gates = 0;
    const int yy = y0 + wave;
    const long rowbase = (b * H + yy) * (long)W;
    const long p0 = rowbase + x0 + 4 * lh;
    const long plimit = yy < H ? rowbase + W : p0;            // nothing below the map
    if ((W & 31) != 0 || (H % FS_TR) != 0) tile_epilogue<4, true>(d, acc, 0, li, p0, plimit);
    else tile_epilogue<4, false>(d, acc, 0, li, p0, 0);
}

}  // namespace

int pf_flow_stem_launch(const PfFlowStemMulti& m, int n, void* stream) {
    if (n < 1 || n > 4 || m.B <= 0 || m.H <= 0 || m.W <= 0) return PF_ERR_BAD_SHAPE;
    const long tiles = (long)m.B * ((m.H + FS_TR - 1) / FS_TR) * ((m.W + 31) / 32);
    if (tiles >= (1L << 31) || (long)m.B * m.H * m.W >= (1L << 31)) return PF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pf_flow_stem_kernel, dim3((unsigned)tiles, (unsigned)n), dim3(256), FS_LDS, (hipStream_t)stream, m);
    return (int)hipGetLastError();
}
