// pf_enc_stem: the encoders' first convolution, 7x7 stride 2 pad 3, 3 -> 64 channels (core/extractor.py:122 conv1, applied
// to the [-1, 1] images of fnet's 4B and cnet's 2B batch, :136-147), straight from the NCHW image (round 4).
//
// Rounds 1-3 ran it as a 4x4 stride-1 convolution over the 2x2 space-to-depth image on the halo conv kernel: 16 taps x one
// 32-channel K chunk each, of which 12 channels are real -- K = 512 for 147 real products per output, behind a 25 MB
// space-to-depth pass (profiles/r4_encoder_alone.txt: 30 + 141 us for fnet's four 512 x 1024 images, 19 + 75 us for cnet's two).
// Here the K axis is the 7x7x3 patch itself: k = ky * 24 + kx * 3 + c (the 21 interleaved floats of one patch row, padded
// to 24) -> K = 176 = 11 MFMA K-steps of 16 (2.9x fewer matrix FLOPs), and the A operand is built in the MFMA gaps from
// an interleaved-RGB copy of the input patch in LDS -- a pixel's patch row is 21 CONTIGUOUS floats there, so a fragment piece
// (8 consecutive k) is four ds_read_b64 + the bf16 hi|lo split (same split arithmetic and pass order as every other
// PF_PREC_BF16X3 kernel: x_lo * w_hi, x_hi * w_lo, x_hi * w_hi, fp32 accumulate).
//
// One workgroup = 4 waves = an 8-row x 32-column tile of output pixels x all 64 channels; wave w owns rows 2w, 2w + 1 (two
// 32x32 accumulators per row).  LDS: the 64 x 176 weights as bf16 hi|lo (45 KB, loaded once: workgroups are persistent over
// tiles) + the 21 x 69 x 3 input patch (17 KB) = 62 KB, so TWO workgroups share a CU and one's patch staging / epilogue
// runs under the other's MFMAs.  The epilogue is the shared tile epilogue (bias, optional ReLU, fp32 rows and / or split
// twin) plus the InstanceNorm statistics of pf_conv_desc.stats_out (fp64 per-tile partials, 8-row tiles: the layout
// pf_channel_stats_final reads), so fnet's and cnet's plans use it exactly like the convolution it replaces.
#include <stdlib.h>
#include "pf_conv_priv.h"

namespace {
using namespace pfconv;

constexpr int ST_K = 176;                 // padded K: 7 patch rows x 24 (21 real) + one zero piece
constexpr int ST_PIECES = ST_K / 8;       // 22 pieces of 8 k
constexpr int ST_STEPS = ST_K / 16;       // 11 MFMA K-steps
constexpr int ST_WROW = ST_PIECES * 32;   // 704 bytes per channel in memory: per piece {hi[8], lo[8]} bf16
constexpr int ST_WLDS = ST_WROW + 16;     // 720-byte LDS row stride: conflict-free ds_read_b128 over 32 channel rows
constexpr int ST_PROWS = 21;              // input rows of a tile's patch: 2 * 8 + 5
constexpr int ST_PW = 208;                // floats per patch row: 69 pixels x 3 channels = 207, padded
// + 16 bytes: the last fragment piece of patch row 20 reads floats 202 .. 209 of the row, two floats past it (x zero weights)
constexpr int ST_LDS = 64 * ST_WLDS + ST_PROWS * ST_PW * 4 + 16;

struct StemArgs {
    const float* img; const char* w; const float* bias;
    float* out; void* out_split; double* stats;
    int relu, Bn, H, W, H2, W2, tiles_x, tiles_y;
    long ntiles;
};

__global__ void __launch_bounds__(256, 2)
pf_enc_stem_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;
    float* const patch = reinterpret_cast<float*>(smem + 64 * ST_WLDS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- weights -> LDS, once (16-byte pieces; 44 per channel row)
    for (int e = tid; e < 64 * (ST_WROW / 16); e += 256) {
        const int row = e / (ST_WROW / 16), q = e % (ST_WROW / 16);
        *reinterpret_cast<f32x4*>(wl + row * ST_WLDS + q * 16) = *reinterpret_cast<const f32x4*>(a.w + (long)row * ST_WROW + q * 16);
    }
    // this lane's fragment pieces: piece 2 s + lh = (patch row ky, floats 8 j .. 8 j + 7 of the 24); the 22nd piece has zero
    // weights and re-reads the 21st
    int a_off[ST_STEPS], b_off[ST_STEPS];
#pragma unroll
    for (int s = 0; s < ST_STEPS; ++s) {
        const int pi = 2 * s + lh, pc = pi < 21 ? pi : 20;
        a_off[s] = (pc / 3) * ST_PW + (pc % 3) * 8 + 6 * li;
        b_off[s] = li * ST_WLDS + pi * 32;
    }
    pf_conv_desc d;
    d.bias = a.bias; d.out = a.out; d.ld_out = 64; d.off_out = 0; d.cout = 64; d.epilogue = a.relu ? PF_EPI_RELU : PF_EPI_LINEAR;
    d.scale = 1.f; d.h = nullptr; d.ld_h = 0; d.z = nullptr; d.ld_z = 0; d.aux_out = nullptr; d.ld_aux = 0;
    d.precision = PF_PREC_BF16X3; d.out_split = a.out_split; d.lds_out = 2; d.aux_split = nullptr; d.lds_aux = 0; d.save_gates = 0;
    const bool ragged = (a.W2 % 32) != 0 || (a.H2 % 8) != 0;
    const long per_img = (long)a.tiles_x * a.tiles_y;

    // Patch staging, software-pipelined: the 17 elements a thread owns (e = tid + 256 u over 21 rows x 207 floats) are LOADED for
    // the next tile before the current tile's MFMAs (all 17 loads in flight at once; a per-element loop with the LDS store behind
    // each load measured 21 us per tile) and written to LDS after them.
    constexpr int ST_PER = (ST_PROWS * 207 + 255) / 256;      // 17
    int p_ry[ST_PER], p_px[ST_PER], p_c[ST_PER], p_dst[ST_PER];
#pragma unroll
    for (int u = 0; u < ST_PER; ++u) {
        const int e = tid + 256 * u;
        const int r = e / 207, t = e % 207;
        p_ry[u] = r; p_px[u] = t / 3; p_c[u] = t % 3;
        p_dst[u] = e < ST_PROWS * 207 ? r * ST_PW + t : -1;
    }
    float pv[ST_PER];
    auto load_patch = [&](long tile) __attribute__((always_inline)) {
        const int tx = (int)(tile % a.tiles_x), ty = (int)((tile / a.tiles_x) % a.tiles_y);
        const long im = tile / per_img;
        const int iy0 = 16 * ty - 3, ix0 = 64 * tx - 3;
        const float* const ib = a.img + im * 3 * (long)a.H * a.W;
#pragma unroll
        for (int u = 0; u < ST_PER; ++u) {
            const int yy = iy0 + p_ry[u], xx = ix0 + p_px[u];
            const bool in = p_dst[u] >= 0 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const float* src = ib + ((long)p_c[u] * a.H + (in ? yy : 0)) * a.W + (in ? xx : 0);
            const float v = *src;                              // always a legal address: a branch-free load
            pv[u] = in ? v : 0.f;
        }
    };
    if ((long)blockIdx.x < a.ntiles) load_patch(blockIdx.x);
    // (Round 5: starting the second workgroup of a CU half a tile late, so that one's stores run under the other's MFMAs, only
    // added the delay -- 72 -> 77 / 79 / 86 us at 2 / 4 / 8 sleep rounds, profiles/r5_encoder_ablation.txt: a workgroup has 4 tiles.)
    for (long tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % a.tiles_x), ty = (int)((tile / a.tiles_x) % a.tiles_y);
        const long im = tile / per_img;
        const int y0 = ty * 8, x0 = tx * 32;
        __syncthreads();                                      // the previous tile's fragment / statistics reads are done
#pragma unroll
        for (int u = 0; u < ST_PER; ++u)
            if (p_dst[u] >= 0) patch[p_dst[u]] = pv[u];
        // the padding float of every row and the four floats behind the last row are read too (x zero weights): they must hold
        // finite values -- LDS keeps what an earlier kernel left there, and NaN x 0 = NaN would end in column x0 + 31 of the tile
        // and in the fused InstanceNorm statistics.  Every tile: the statistics buffer below aliases rows 0 .. 4.
        if (tid < ST_PROWS) patch[tid * ST_PW + 207] = 0.f;
        else if (tid < ST_PROWS + 4) patch[ST_PROWS * ST_PW + tid - ST_PROWS] = 0.f;
        __syncthreads();
#ifndef PF_STEM_ABL_NO_STAGE
        if (tile + gridDim.x < a.ntiles) load_patch(tile + gridDim.x);      // in flight during this tile's MFMAs
#endif
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
#ifndef PF_STEM_ABL_NO_MFMA
        static_for<0, ST_STEPS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            bf16x8 fw[2][2];                                   // [channel tile][hi, lo]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fw[t][0] = *reinterpret_cast<const bf16x8*>(wl + b_off[s] + t * 32 * ST_WLDS);
                fw[t][1] = *reinterpret_cast<const bf16x8*>(wl + b_off[s] + t * 32 * ST_WLDS + 16);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // output row 2 wave + m reads patch rows 2 (2 wave + m) + ky, output column li reads floats from 6 li
                const float* ap = patch + (2 * (2 * wave + m)) * ST_PW + a_off[s];
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
                for (int t = 0; t < 2; ++t) {
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, fw[t][0], acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, fw[t][1], acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, fw[t][0], acc[m][t], 0, 0, 0);
                }
            }
        });
#endif
        // ---- epilogue: acc[m][t][r] = pixel (row y0 + 2 wave + m, column x0 + (r & 3) + 8 (r >> 2) + 4 lh), channel 32 t + li
        const int xlim_raw = a.W2 - x0 - 4 * lh;
#ifdef PF_STEM_ABL_NO_EPI
        if (acc[0][0][0] == 123.456f) a.out[0] = acc[1][1][3];
        continue;
#endif
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int yy = y0 + 2 * wave + m;
            const int xlim = yy < a.H2 ? xlim_raw : 0;
            const long p0 = (im * a.H2 + yy) * (long)a.W2 + x0 + 4 * lh;
            if (ragged) tile_epilogue<2, true>(d, acc[m], 0, li, p0, p0 + (xlim > 0 ? xlim : 0));
            else tile_epilogue<2, false>(d, acc[m], 0, li, p0, 0);
        }
        if (a.stats != nullptr) {
            // InstanceNorm statistics of the stored values, as pf_conv_halo_kernel leaves them: fp64 sum / sum of squares per channel
            // over this tile, partial [tile][64][2]
            __syncthreads();                                  // every wave is done with the patch
            double* red = reinterpret_cast<double*>(patch);   // [4 waves][64 channels][2]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bias = a.bias[32 * t + li];
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int xlim = (y0 + 2 * wave + m < a.H2) ? xlim_raw : 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float f = acc[m][t][r] + bias;
                        if (a.relu) f = fmaxf(f, 0.f);
                        const double v = ((r & 3) + 8 * (r >> 2) < xlim) ? (double)f : 0.0;
                        s1 += v; s2 += v * v;
                    }
                }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lh == 0) {
                    red[(wave * 64 + 32 * t + li) * 2 + 0] = s1;
                    red[(wave * 64 + 32 * t + li) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < 64) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s1 += red[(w * 64 + tid) * 2]; s2 += red[(w * 64 + tid) * 2 + 1]; }
                double* o = a.stats + (tile * 64 + tid) * 2;
                o[0] = s1; o[1] = s2;
            }
        }
    }
}

}  // namespace

namespace {
__global__ void __launch_bounds__(256) pf_dirty_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = pattern;
    __syncthreads();
    if (lds_words[(threadIdx.x * 97) % (160 * 1024 / 4)] != pattern && sink) sink[0] = 1;      // keeps the stores alive
}
}  // namespace

extern "C" int pf_debug_dirty_lds(unsigned pattern, void* stream) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_dirty_lds_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    // one 160 KB workgroup per CU at a time, a few rounds so that every CU gets one
    hipLaunchKernelGGL(pf_dirty_lds_kernel, dim3(1024), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int pf_enc_stem(const float* img, const void* weight, const float* bias, float* out, void* out_split, int relu,
                           double* stats_out, int Bn, int H, int W, void* stream) {
    if (!img || !weight || !bias || (!out && !out_split)) return PF_ERR_BAD_ARG;
    if (Bn <= 0 || H < 16 || W < 16 || (H & 1) || (W & 1)) return PF_ERR_BAD_SHAPE;
    StemArgs a;
    a.img = img; a.w = reinterpret_cast<const char*>(weight); a.bias = bias; a.out = out; a.out_split = out_split; a.stats = stats_out;
    a.relu = relu ? 1 : 0; a.Bn = Bn; a.H = H; a.W = W; a.H2 = H / 2; a.W2 = W / 2;
    a.tiles_x = (a.W2 + 31) / 32; a.tiles_y = (a.H2 + 7) / 8;
    a.ntiles = (long)Bn * a.tiles_x * a.tiles_y;
    if ((long)Bn * a.H2 * a.W2 >= (1L << 31)) return PF_ERR_BAD_SHAPE;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long cap = 2L * cus;                                // two workgroups per CU, persistent over the tiles
    const dim3 grid((unsigned)(a.ntiles < cap ? a.ntiles : cap));
    hipLaunchKernelGGL(pf_enc_stem_kernel, grid, dim3(256), ST_LDS, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
