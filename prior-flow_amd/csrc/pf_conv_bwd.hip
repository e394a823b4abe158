// Convolution backward building blocks (the tape that would use them is not built yet).
//
//   pf_conv2d_wgrad:  dW[o][tap][c] += sum_p dY[p][o] * X[p + off(tap)][c]      (stride 1, zero padding)
//   pf_col_sums:      db[o]          = sum_p dY[p][o]
//
// (dgrad needs no kernel of its own: dX = conv(dY, W') with W'[c][o][ky][kx] = W[o][c][KH-1-ky][KW-1-kx]
//  runs on pf_conv2d -- engine.Conv.dgrad_of.)
//
// wgrad is a GEMM whose REDUCTION runs over pixels: C[o][c] = sum_k A[o][k] B[k][c] with k = pixel, while
// both tensors are channel-last ([pixel][channel]: the channel is contiguous, the pixel is strided).  The
// 32x32x16 MFMA wants, per lane, 8 consecutive k of one row / column -- i.e. the TRANSPOSE of the memory
// layout for both operands.  gfx950's ds_read_b64_tr_b16 does that transpose inside the LDS read: tiles are
// staged row-major with plain coalesced 8-byte writes (bf16 hi | lo planes of 32 channels = 64-byte rows,
// the 3-pass split of pf_conv_mfma.hip) and each lane reads rows 8h .. 8h+7 of its column with two
// transposed reads (addressing validated on hardware in round 1 by a stand-alone lane/element dump).  64-byte rows
// make every transposed read conflict-free: a 32-lane half covers 4 consecutive rows = 256 contiguous bytes.
//
// Work decomposition: workgroup (8 waves) = 128 output channels x 64 input channels x a share of the
// 4 x 32 pixel tiles (split-K); per tile it stages the dY tile and the X HALO once (like the forward halo
// kernel: the taps read shifted rows of the same LDS image); wave (mb, cb) owns one 32 x 32 (output, input)
// channel block and accumulates EVERY tap (<= 9 accumulators).  Partial sums of the splits are added to dW
// with fp32 atomics (dW must be zeroed by the caller).  The kernel is bound by re-staging: every pixel
// tile is read by (Cout/128) x (Cin/64) workgroups; the first version's 64 x 32 tiles doubled that traffic
// (~700 MB of L2 reads per GRU-gate launch, 14-68 TFLOP/s).
#include <type_traits>
#include "pf_common.h"
#include "../../include/priorflow_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32, TPX = TH * TW;     // pixel tile
constexpr int WG_O = 128, WG_C = 64;              // output / input channels per workgroup

struct WgradArgs {
    const float* x0; int ld0, off0, c0;
    const float* x1; int ld1, off1, c1;
    const float* dy; int ld_dy, off_dy, cout;
    float* dw;                    // [Cout_pad128][taps][cin_pad]
    float* db;                    // [Cout] or NULL: db[o] += sum_p dY[p][o], by the workgroups of the first input-channel block
    int B, H, W, kh, kw, cin_pad, nsplit;
};

// rows 8h .. 8h+7 of column (lane & 31) of a row-major bf16 tile with 64-byte rows, starting at row `row0`:
// the MFMA 32x32x16 operand of this lane (both for A = dY^T and for B = X)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const int grp = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
    const __bf16* base = tile + (row0 + 8 * (grp >> 1) + q) * 32 + 16 * (grp & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)base);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(base + 4 * 32));
    bf16x8 f;
    const bf16x4 a = __builtin_bit_cast(bf16x4, lo), b = __builtin_bit_cast(bf16x4, hi);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
    f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    return f;
#else
    (void)tile; (void)row0; (void)lane;
    return bf16x8{};
#endif
}

template <int KH, int KW>
__global__ void __launch_bounds__(512, 2)
pf_wgrad_kernel(const WgradArgs a) {
    constexpr int TAPS = KH * KW, HH = TH + KH - 1, HW = TW + KW - 1, HPX = HH * HW;
    constexpr int ph = KH / 2, pw = KW / 2;
    static_assert(TAPS <= 10, "at most 5 accumulators per wave");
    // TAPS <= 5: the two wave groups take the two 32-channel input blocks (workgroup = 128 x 64 channels);
    // 6..10 taps: they take the even / odd taps of ONE input block (128 x 32 channels, <= 5 accumulators).
    constexpr bool SPLIT_T = TAPS > 5;
    constexpr int NCB = SPLIT_T ? 1 : 2, NACC = SPLIT_T ? (TAPS + 1) / 2 : TAPS;
    // LDS (bf16): dY tile [2 planes][4 blocks of 32 out-channels][128 px][32], X halo [2 planes][2 blocks][XROWS][32]
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    constexpr int XROWS = (HPX + 15) / 16 * 16;
    __bf16* const dyt = lds;                                   // ((plane*4 + mb)*TPX + px)*32 + c
    __bf16* const xt = lds + 2 * 4 * TPX * 32;                 // ((plane*NCB + cb)*XROWS + hp)*32 + c
    // bias gradient (round 5: it was a launch of its own per convolution before): the workgroups of the first
    // input-channel block add the fp32 dY values they stage anyway -- per thread over its 8 pixels of a tile, then into 128 LDS
    // words, and once per workgroup into db
    float* const bsum = reinterpret_cast<float*>(xt + 2 * NCB * XROWS * 32);
    const bool do_db = a.db != nullptr && blockIdx.y == 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave & 3;                                   // this wave's 32 output channels
    const int cb = SPLIT_T ? 0 : wave >> 2, tpar = SPLIT_T ? wave >> 2 : 0;       // input block / tap parity
    const int o0 = blockIdx.x * WG_O, c00 = blockIdx.y * (32 * NCB), split = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = a.B * tiles_img;
    const long N = (long)a.H * a.W;
    const int ctot = a.c0 + a.c1;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (do_db && tid < WG_O) bsum[tid] = 0.f;                 // (the first tile's barrier orders it before the first add)

    // Software pipeline over this split's tiles: the operands of tile i+1 are fetched into registers while
    // tile i is multiplied out of LDS (the first version loaded, waited, converted and computed serially and
    // was latency bound: ~180 us even for a 1x1 conv).
    constexpr int XV = 8 * NCB;                                // float4 per halo pixel
    constexpr int NX = (XROWS * XV + 511) / 512;               // X-halo float4 per thread
    f32x4 ry[8], rx[NX];
    auto load_y = [&](int tile) __attribute__((always_inline)) {
        const int img = tile / tiles_img, tin = tile % tiles_img;
        const int y0 = (tin / tiles_x) * TH, x0 = (tin % tiles_x) * TW;
#pragma unroll
        for (int q = 0; q < 8; ++q) {                          // dY: 128 px x 128 channels
            const int e = tid + 512 * q;                       // float4 index: px = e / 32, channel group = e % 32
            const int px = e >> 5, cg = (e & 31) * 4;
            const int yy = y0 + (px >> 5), xx = x0 + (px & 31), o = o0 + cg;
            ry[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tile < ntiles && yy < a.H && xx < a.W && o < a.cout)      // cout % 4 == 0 is checked by the launcher
                ry[q] = *reinterpret_cast<const f32x4*>(a.dy + ((long)img * N + (long)yy * a.W + xx) * a.ld_dy + a.off_dy + o);
        }
    };
    auto load_x = [&](int tile) __attribute__((always_inline)) {
        const int img = tile / tiles_img, tin = tile % tiles_img;
        const int y0 = (tin / tiles_x) * TH, x0 = (tin % tiles_x) * TW;
#pragma unroll
        for (int q = 0; q < NX; ++q) {                         // X halo: XROWS px x 32*NCB channels
            const int e = tid + 512 * q;
            const int hp = e / XV, cg = (e % XV) * 4;
            const int yy = y0 + hp / HW - ph, xx = x0 + hp % HW - pw, c = c00 + cg;
            rx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tile < ntiles && hp < HPX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && c < ctot) {
                const long p = (long)img * N + (long)yy * a.W + xx;
                rx[q] = c < a.c0 ? *reinterpret_cast<const f32x4*>(a.x0 + p * a.ld0 + a.off0 + c)
                                 : *reinterpret_cast<const f32x4*>(a.x1 + p * a.ld1 + a.off1 + (c - a.c0));
            }
        }
    };
    auto store_y = [&]() __attribute__((always_inline)) {
        if (do_db) {                                           // rows past the map / channels past cout were loaded as zeros
            f32x4 sb = ry[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) sb = sb + ry[q];
            float* bp = bsum + (tid & 31) * 4;
            atomicAdd(bp, sb.x); atomicAdd(bp + 1, sb.y); atomicAdd(bp + 2, sb.z); atomicAdd(bp + 3, sb.w);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 512 * q, px = e >> 5, cg = (e & 31) * 4;
            const bf16x4 hi = __builtin_convertvector(ry[q], bf16x4);
            const bf16x4 lo = __builtin_convertvector(ry[q] - __builtin_convertvector(hi, f32x4), bf16x4);
            const int blk = cg >> 5, c = cg & 31;
            *reinterpret_cast<bf16x4*>(dyt + ((0 * 4 + blk) * TPX + px) * 32 + c) = hi;
            *reinterpret_cast<bf16x4*>(dyt + ((1 * 4 + blk) * TPX + px) * 32 + c) = lo;
        }
    };
    auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int e = tid + 512 * q, hp = e / XV, cg = (e % XV) * 4;
            if (hp < XROWS) {
                const bf16x4 hi = __builtin_convertvector(rx[q], bf16x4);
                const bf16x4 lo = __builtin_convertvector(rx[q] - __builtin_convertvector(hi, f32x4), bf16x4);
                const int blk = cg >> 5, cc = cg & 31;
                *reinterpret_cast<bf16x4*>(xt + ((0 * NCB + blk) * XROWS + hp) * 32 + cc) = hi;
                *reinterpret_cast<bf16x4*>(xt + ((1 * NCB + blk) * XROWS + hp) * 32 + cc) = lo;
            }
        }
    };

    constexpr bool PREFETCH = true;
    if constexpr (PREFETCH) { load_y(split); load_x(split); }
    for (int tile = split; tile < ntiles; tile += a.nsplit) {
        __syncthreads();                                       // previous tile's fragment reads are done
        if constexpr (PREFETCH) {
            store_y(); store_x();
        } else {                                               // one operand after the other: they share registers
            load_y(tile); store_y();
            asm volatile("" ::: "memory");
            load_x(tile); store_x();
        }
        __syncthreads();
        if constexpr (PREFETCH) {
            load_y(tile + a.nsplit); load_x(tile + a.nsplit);  // zeros past the end
            asm volatile("" ::: "memory");                     // keep the loads above the MFMA block
        }
        // ---- 8 K-steps of 16 pixels (tile row y, half row xh); every tap -------------------------------
        for (int ks = 0; ks < 8; ++ks) {
            const int y = ks >> 1, xh = (ks & 1) * 16;
            const bf16x8 a_hi = tr_frag(dyt + (0 * 4 + mb) * TPX * 32, y * 32 + xh, lane);
            const bf16x8 a_lo = tr_frag(dyt + (1 * 4 + mb) * TPX * 32, y * 32 + xh, lane);
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                const int tap = SPLIT_T ? 2 * t + tpar : t;    // wave-uniform
                if (tap < TAPS) {
                    const int ky = tap / KW, kx = tap % KW;
                    const int hrow = (y + ky) * HW + xh + kx;  // halo pixel of this K-step's first pixel under the tap
                    const bf16x8 b_hi = tr_frag(xt + (0 * NCB + cb) * XROWS * 32, hrow, lane);
                    const bf16x8 b_lo = tr_frag(xt + (1 * NCB + cb) * XROWS * 32, hrow, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[t], 0, 0, 0);
                }
            }
        }
    }
    if (do_db) {
        __syncthreads();
        if (tid < WG_O && o0 + tid < a.cout) atomicAdd(a.db + o0 + tid, bsum[tid]);
    }
    // ---- epilogue: D[row = output channel][col = input channel]; lane = column ---------------------------
    const int li = lane & 31, lh = lane >> 5;
    const int c = c00 + 32 * cb + li;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int tap = SPLIT_T ? 2 * t + tpar : t;
        if (tap < TAPS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < a.cout && c < ctot)
                    atomicAdd(a.dw + ((long)o * TAPS + tap) * a.cin_pad + c, acc[t][r]);
            }
        }
    }
}

template <int KH, int KW>
int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    constexpr int HPX = (TH + KH - 1) * (TW + KW - 1), XROWS = (HPX + 15) / 16 * 16;
    constexpr int NCB = KH * KW > 5 ? 1 : 2;
    constexpr size_t lds = (size_t)(2 * 4 * TPX * 32 + 2 * NCB * XROWS * 32) * 2 + WG_O * sizeof(float);      // + the bias sums
    static_assert(lds <= 160 * 1024, "LDS budget");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_wgrad_kernel<KH, KW>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    dim3 grid((unsigned)((a.cout + WG_O - 1) / WG_O), (unsigned)((a.cin_pad + 32 * NCB - 1) / (32 * NCB)), (unsigned)a.nsplit);
    hipLaunchKernelGGL((pf_wgrad_kernel<KH, KW>), grid, dim3(512), lds, stream, a);
    return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// Weight gradient of the SMALL-Cin convolutions (the 7x7 stems: core/extractor.py:122 3->64 stride 2,
// core/update.py:87,173,175 2->128 stride 1; their inputs -- images, detached flows -- need no data gradient):
//   dW[o][c][ky][kx] += sum_p dY[p][o] * X[stride*p + (ky, kx) - pad][c],   db[o] += sum_p dY[p][o]
// K = KH*KW*Cin <= 196 is far too small for the pixel-reduction MFMA kernel above, and the products are exact-fp32
// work anyway (the forward of these layers runs on the exact-fp32 small-Cin kernel).  A workgroup walks over 8 x 8
// output-pixel tiles: the dY tile (64 px x 64 channels) and the input patch live in LDS; thread (o = tid % 64,
// tg = tid / 64) owns output channel o and every fourth tap, all input channels, in registers across ALL its tiles
// (<= 13 taps x 4 channels), and adds them to dW once at the end (fp32 atomics; dW, db are ACCUMULATED: zero first).
// ----------------------------------------------------------------------------------------------
struct WgradSmallArgs {
    const float* x; int nchw, ld_in, off_in, cin;
    const float* dy; int ld_dy, off_dy, cout;
    float* dw; float* db;
    int B, Ho, Wo, kh, kw, stride;
    float* ws;              // optional: per-workgroup partial sums [gridDim.y][gridDim.x][taps*cin + 1][64] (two-stage reduction)
};
constexpr int WS_T = 8, WS_MAXTAPS = 13, WS_MAXC = 4, WS_SLICES = 8;

template <int CIN>
__global__ void __launch_bounds__(256) pf_wgrad_small_kernel(const WgradSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];
    const int PW = (WS_T - 1) * a.stride + a.kw, PH = (WS_T - 1) * a.stride + a.kh;
    float* const dyt = ws_lds;                          // [64 px][64 o]
    float* const patch = ws_lds + 64 * 64;              // [PH][PW][cin]
    const int tid = threadIdx.x, o = tid & 63, tg = tid >> 6;
    const int o0 = blockIdx.y * 64;
    const int taps = a.kh * a.kw, pad_h = a.kh / 2, pad_w = a.kw / 2;
    const int Hi = a.Ho * a.stride, Wi = a.Wo * a.stride;
    const int tiles_x = (a.Wo + WS_T - 1) / WS_T, tiles_y = (a.Ho + WS_T - 1) / WS_T;
    const long ntiles = (long)a.B * tiles_x * tiles_y;
    int toff[WS_MAXTAPS];                               // patch offset of this thread's j-th tap (or -1)
#pragma unroll
    for (int j = 0; j < WS_MAXTAPS; ++j) {
        const int tap = tg + 4 * j;
        toff[j] = tap < taps ? ((tap / a.kw) * PW + tap % a.kw) * CIN : 0;      // a tap past the window re-reads tap 0 (never stored)
    }
    float acc[WS_MAXTAPS][CIN];
#pragma unroll
    for (int j = 0; j < WS_MAXTAPS; ++j)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[j][c] = 0.f;
    float bsum = 0.f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = (int)(tile / (tiles_x * tiles_y)), tin = (int)(tile % (tiles_x * tiles_y));
        const int y0 = (tin / tiles_x) * WS_T, x0 = (tin % tiles_x) * WS_T;
        __syncthreads();                                // the previous tile's reads are done
        for (int e = tid; e < 64 * 16; e += 256) {      // dY tile: 64 px x 16 float4
            const int px = e >> 4, c4 = (e & 15) * 4;
            const int yy = y0 + (px >> 3), xx = x0 + (px & 7);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (yy < a.Ho && xx < a.Wo && o0 + c4 < a.cout)
                v = *reinterpret_cast<const f32x4*>(a.dy + ((long)img * a.Ho * a.Wo + (long)yy * a.Wo + xx) * a.ld_dy + a.off_dy + o0 + c4);
            *reinterpret_cast<f32x4*>(dyt + px * 64 + c4) = v;
        }
        for (int e = tid; e < PH * PW * CIN; e += 256) {
            const int c = e % CIN, pp = e / CIN;
            const int yy = y0 * a.stride + pp / PW - pad_h, xx = x0 * a.stride + pp % PW - pad_w;
            float v = 0.f;
            if (yy >= 0 && yy < Hi && xx >= 0 && xx < Wi)
                v = a.nchw ? a.x[(((long)img * CIN + c) * Hi + yy) * Wi + xx]
                           : a.x[((long)img * Hi * Wi + (long)yy * Wi + xx) * a.ld_in + a.off_in + c];
            patch[e] = v;
        }
        __syncthreads();
        for (int px = 0; px < 64; ++px) {
            const float g = dyt[px * 64 + o];
            if (tg == 0) bsum += g;
            const float* pb = patch + (((px >> 3) * PW + (px & 7)) * a.stride) * CIN;
            // no data-dependent branch in here: the 13 x CIN patch reads of a pixel are independent and issue back to back
            // (with a `break` on the first unused tap they were one exposed LDS round trip each: 136 us per flow stem)
            float xv[WS_MAXTAPS][CIN];
#pragma unroll
            for (int j = 0; j < WS_MAXTAPS; ++j)
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[j][c] = pb[toff[j] + c];
#pragma unroll
            for (int j = 0; j < WS_MAXTAPS; ++j)
#pragma unroll
                for (int c = 0; c < CIN; ++c) acc[j][c] = __builtin_fmaf(g, xv[j][c], acc[j][c]);
        }
    }
    if (a.ws != nullptr) {
        // two-stage: this workgroup's sums go to its own slice of the workspace (64 consecutive floats per (tap, channel):
        // coalesced), pf_wgrad_small_reduce adds the slices -- 8 atomics per weight instead of one per workgroup
        float* const w = a.ws + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (taps * CIN + 1) * 64;
#pragma unroll
        for (int j = 0; j < WS_MAXTAPS; ++j) {
            const int tap = tg + 4 * j;
            if (tap < taps)
#pragma unroll
                for (int c = 0; c < CIN; ++c) w[(tap * CIN + c) * 64 + o] = acc[j][c];
        }
        if (tg == 0) w[taps * CIN * 64 + o] = bsum;
        return;
    }
    if (o0 + o < a.cout) {
#pragma unroll
        for (int j = 0; j < WS_MAXTAPS; ++j) {
            const int tap = tg + 4 * j;
            if (tap < taps)
#pragma unroll
                for (int c = 0; c < CIN; ++c) atomicAdd(a.dw + ((long)(o0 + o) * CIN + c) * taps + tap, acc[j][c]);
        }
        if (tg == 0 && a.db) atomicAdd(a.db + o0 + o, bsum);
    }
}

// Second stage of the workspace form: thread (slot, o) of column block y adds the partial sums of one slice of the workgroups
// (in order) and hands the result to dW / db with one atomic -- WS_SLICES atomics per weight.  The one-stage form ended with one
// atomic per weight PER WORKGROUP on the same 9 408 addresses (294 cache lines): 414 us per encoder stem, most of it that.
__global__ void __launch_bounds__(256) pf_wgrad_small_reduce(const float* __restrict__ ws, int nwg, int ny, int taps, int cin,
                                                             int cout, float* __restrict__ dw, float* __restrict__ db) {
    const int slots = taps * cin + 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)ny * slots * 64) return;
    const int o = (int)(idx & 63);
    const int slot = (int)((idx >> 6) % slots);
    const int y = (int)((idx >> 6) / slots);
    if (y * 64 + o >= cout) return;
    const int per = (nwg + WS_SLICES - 1) / WS_SLICES;
    const int w0 = blockIdx.y * per, w1 = (w0 + per < nwg) ? w0 + per : nwg;
    float sum = 0.f;
    const float* q = ws + (((long)y * nwg + w0) * slots + slot) * 64 + o;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        const float v0 = q[0], v1 = q[(long)slots * 64], v2 = q[(long)slots * 128], v3 = q[(long)slots * 192];
        sum += v0; sum += v1; sum += v2; sum += v3;
        q += (long)slots * 256;
    }
    for (; w < w1; ++w) { sum += q[0]; q += (long)slots * 64; }
    if (w1 <= w0) return;
    if (slot == slots - 1) {
        if (db) atomicAdd(db + y * 64 + o, sum);
    } else {
        const int tap = slot / cin, c = slot % cin;
        atomicAdd(dw + ((long)(y * 64 + o) * cin + c) * taps + tap, sum);
    }
}

}  // namespace

static unsigned wgrad_small_groups(int B, int Hout, int Wout) {
    const long ntiles = (long)B * ((Hout + WS_T - 1) / WS_T) * ((Wout + WS_T - 1) / WS_T);
    return (unsigned)(ntiles < 256 ? ntiles : 256);
}

extern "C" long pf_conv2d_wgrad_small_ws_floats(int cin, int cout, int kh, int kw, int B, int Hout, int Wout) {
    if (cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || B <= 0 || Hout <= 0 || Wout <= 0) return 0;
    return (long)wgrad_small_groups(B, Hout, Wout) * ((cout + 63) / 64) * ((long)kh * kw * cin + 1) * 64;
}

static int wgrad_small_impl(const float* x, int nchw, int ld_in, int off_in, int cin,
                            const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                            int kh, int kw, int stride, int B, int Hout, int Wout, float* ws, void* stream);

extern "C" int pf_conv2d_wgrad_small(const float* x, int nchw, int ld_in, int off_in, int cin,
                                     const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                                     int kh, int kw, int stride, int B, int Hout, int Wout, void* stream) {
    return wgrad_small_impl(x, nchw, ld_in, off_in, cin, dy, ld_dy, off_dy, cout, dw, db, kh, kw, stride, B, Hout, Wout, nullptr, stream);
}

extern "C" int pf_conv2d_wgrad_small_ws(const float* x, int nchw, int ld_in, int off_in, int cin,
                                        const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                                        int kh, int kw, int stride, int B, int Hout, int Wout,
                                        float* workspace, long workspace_floats, void* stream) {
    if (!workspace || workspace_floats < pf_conv2d_wgrad_small_ws_floats(cin, cout, kh, kw, B, Hout, Wout)) return PF_ERR_BAD_ARG;
    return wgrad_small_impl(x, nchw, ld_in, off_in, cin, dy, ld_dy, off_dy, cout, dw, db, kh, kw, stride, B, Hout, Wout, workspace, stream);
}

static int wgrad_small_impl(const float* x, int nchw, int ld_in, int off_in, int cin,
                            const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                            int kh, int kw, int stride, int B, int Hout, int Wout, float* ws, void* stream) {
    if (!x || !dy || !dw) return PF_ERR_BAD_ARG;
    if (B <= 0 || Hout <= 0 || Wout <= 0 || cin <= 0 || cin > WS_MAXC || cout <= 0) return PF_ERR_BAD_SHAPE;
    if (kh < 1 || kw < 1 || kh * kw > 4 * WS_MAXTAPS || (stride != 1 && stride != 2)) return PF_ERR_BAD_SHAPE;
    if ((ld_dy | off_dy | cout) & 3) return PF_ERR_BAD_SHAPE;                       // 16-byte dY loads
    if (off_dy < 0 || off_dy + cout > ld_dy || (!nchw && (off_in < 0 || off_in + cin > ld_in))) return PF_ERR_BAD_ARG;
    WgradSmallArgs a;
    a.x = x; a.nchw = nchw; a.ld_in = ld_in; a.off_in = off_in; a.cin = cin;
    a.dy = dy; a.ld_dy = ld_dy; a.off_dy = off_dy; a.cout = cout; a.dw = dw; a.db = db;
    a.B = B; a.Ho = Hout; a.Wo = Wout; a.kh = kh; a.kw = kw; a.stride = stride; a.ws = ws;
    const int PW = (WS_T - 1) * stride + kw, PH = (WS_T - 1) * stride + kh;
    const size_t lds = (size_t)(64 * 64 + PH * PW * cin) * sizeof(float);
    // every workgroup ends with one atomic per weight on the SAME Cout*Cin*taps addresses: 1 024 workgroups on the encoder stem
    // (9 408 addresses) spent ~0.8 of its 0.9 ms in that contention -- one workgroup per CU walks more tiles instead
    dim3 grid(wgrad_small_groups(B, Hout, Wout), (unsigned)((cout + 63) / 64));
    switch (cin) {
        case 1: hipLaunchKernelGGL(pf_wgrad_small_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(pf_wgrad_small_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(pf_wgrad_small_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(pf_wgrad_small_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, a); break;
    }
    if (ws != nullptr) {
        const long total = (long)grid.y * (kh * kw * cin + 1) * 64;
        hipLaunchKernelGGL(pf_wgrad_small_reduce, dim3((unsigned)((total + 255) / 256), WS_SLICES), dim3(256), 0, (hipStream_t)stream,
                           ws, (int)grid.x, (int)grid.y, kh * kw, cin, cout, dw, db);
    }
    return (int)hipGetLastError();
}

extern "C" int pf_conv2d_wgrad(const float* x0, int ld0, int off0, int c0, const float* x1, int ld1, int off1, int c1,
                               const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                               int kh, int kw, int B, int H8, int W8, void* stream) {
    if (!x0 || !dy || !dw || (c1 > 0 && !x1)) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 0 || W8 <= 0 || c0 <= 0 || c1 < 0 || cout <= 0) return PF_ERR_BAD_SHAPE;
    if ((ld0 | off0 | c0 | c1 | ld_dy | off_dy | cout) & 3) return PF_ERR_BAD_SHAPE;            // 16-byte loads
    if (c1 > 0 && (((ld1 | off1) & 3) || (c0 % 32) != 0)) return PF_ERR_BAD_SHAPE;
    if (off0 < 0 || off0 + c0 > ld0 || (c1 > 0 && (off1 < 0 || off1 + c1 > ld1)) || off_dy < 0 || off_dy + cout > ld_dy)
        return PF_ERR_BAD_ARG;
    WgradArgs a;
    a.x0 = x0; a.ld0 = ld0; a.off0 = off0; a.c0 = c0; a.x1 = x1; a.ld1 = ld1; a.off1 = off1; a.c1 = c1;
    a.dy = dy; a.ld_dy = ld_dy; a.off_dy = off_dy; a.cout = cout; a.dw = dw;
    a.db = db;            // the bias gradient rides in the weight-gradient kernel (round 5: 57 launches per step less)
    a.B = B; a.H = H8; a.W = W8; a.kh = kh; a.kw = kw;
    a.cin_pad = (c0 + c1 + 31) / 32 * 32;
    // split-K: enough workgroups for ~4 per CU, at most one per pixel tile
    const long ntiles = (long)B * ((H8 + TH - 1) / TH) * ((W8 + TW - 1) / TW);
    const int wg_c = kh * kw > 5 ? 32 : WG_C;
    const long wg_base = (long)((cout + WG_O - 1) / WG_O) * ((a.cin_pad + wg_c - 1) / wg_c);
    long ns = (512 + wg_base - 1) / wg_base;          // one 123-135 KB workgroup per CU: about two rounds
    if (ns > ntiles) ns = ntiles;
    if (ns < 1) ns = 1;
    a.nsplit = (int)ns;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (kh == 3 && kw == 3) rc = launch_wgrad<3, 3>(a, s);
    else if (kh == 1 && kw == 5) rc = launch_wgrad<1, 5>(a, s);
    else if (kh == 5 && kw == 1) rc = launch_wgrad<5, 1>(a, s);
    else if (kh == 1 && kw == 1) rc = launch_wgrad<1, 1>(a, s);
    else return PF_ERR_BAD_SHAPE;
    if (rc) return rc;
    return rc;
}
