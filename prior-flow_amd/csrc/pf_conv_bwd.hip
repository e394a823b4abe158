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
// transposed reads (addressing validated on hardware by profiles/scratch/tr_read_check.hip).  64-byte rows
// make every transposed read conflict-free: a 32-lane half covers 4 consecutive rows = 256 contiguous bytes.
//
// Work decomposition: workgroup = (64 output channels) x (one 32-input-channel chunk) x (a share of the
// 4 x 32 pixel tiles, split-K); per tile it stages the dY tile and the X HALO once (like the forward halo
// kernel: the taps read shifted rows of the same LDS image) and accumulates every tap; 4 waves =
// 2 (32-channel halves of the 64 outputs) x 2 (tap parity), up to 5 taps x one 32x32 accumulator per wave.
// Partial sums of the splits are added to dW with fp32 atomics (dW must be zeroed by the caller).
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
constexpr int MAX_TAPS_PER_WAVE = 5;

struct WgradArgs {
    const float* x0; int ld0, off0, c0;
    const float* x1; int ld1, off1, c1;
    const float* dy; int ld_dy, off_dy, cout;
    float* dw;                    // [Cout_pad128][taps][cin_pad]
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
__global__ void __launch_bounds__(256, 2)
pf_wgrad_kernel(const WgradArgs a) {
    constexpr int TAPS = KH * KW, HH = TH + KH - 1, HW = TW + KW - 1, HPX = HH * HW;
    constexpr int ph = KH / 2, pw = KW / 2;
    static_assert(TAPS <= 2 * MAX_TAPS_PER_WAVE, "tap parity groups of at most 5 taps");
    // LDS: dY tile [2 planes][2 halves of 32 channels][128 px][32]  and X halo [2 planes][HPX + pad][32], bf16
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* const dyt = lds;                                   // index ((plane*2 + mb)*TPX + px)*32 + c
    constexpr int XROWS = (HPX + 15) / 16 * 16 + 16;           // transposed reads touch up to 15 rows past a fragment's 8
    __bf16* const xt = lds + 2 * 2 * TPX * 32;                 // index (plane*XROWS + hp)*32 + c

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave & 1, tpar = wave >> 1;                 // 32-channel half of the 64 outputs, tap parity
    const int o0 = blockIdx.x * 64, cchunk = blockIdx.y, split = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = a.B * tiles_img;
    const long N = (long)a.H * a.W;
    const int ctot = a.c0 + a.c1;

    f32x16 acc[MAX_TAPS_PER_WAVE];
#pragma unroll
    for (int t = 0; t < MAX_TAPS_PER_WAVE; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = split; tile < ntiles; tile += a.nsplit) {
        const int img = tile / tiles_img, tin = tile % tiles_img;
        const int y0 = (tin / tiles_x) * TH, x0 = (tin % tiles_x) * TW;
        __syncthreads();                                       // previous tile's fragment reads are done
        // ---- stage dY: 128 px x 64 channels, 8 float4 per thread ---------------------------------------
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;                       // float4 index: px = e / 16, channel group = e % 16
            const int px = e >> 4, cg = (e & 15) * 4;
            const int yy = y0 + (px >> 5), xx = x0 + (px & 31), o = o0 + cg;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (yy < a.H && xx < a.W && o < a.cout)            // cout % 4 == 0 is checked by the launcher
                v = *reinterpret_cast<const f32x4*>(a.dy + ((long)img * N + (long)yy * a.W + xx) * a.ld_dy + a.off_dy + o);
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
            const int half = cg >> 5, c = cg & 31;
            *reinterpret_cast<bf16x4*>(dyt + ((0 * 2 + half) * TPX + px) * 32 + c) = hi;
            *reinterpret_cast<bf16x4*>(dyt + ((1 * 2 + half) * TPX + px) * 32 + c) = lo;
        }
        // ---- stage the X halo: HPX px x 32 channels ------------------------------------------------------
        for (int e = tid; e < XROWS * 8; e += 256) {
            const int hp = e >> 3, cg = (e & 7) * 4;
            const int yy = y0 + hp / HW - ph, xx = x0 + hp % HW - pw, c = cchunk * 32 + cg;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hp < HPX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && c < ctot) {
                const long p = (long)img * N + (long)yy * a.W + xx;
                v = c < a.c0 ? *reinterpret_cast<const f32x4*>(a.x0 + p * a.ld0 + a.off0 + c)
                             : *reinterpret_cast<const f32x4*>(a.x1 + p * a.ld1 + a.off1 + (c - a.c0));
            }
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
            *reinterpret_cast<bf16x4*>(xt + (0 * XROWS + hp) * 32 + cg) = hi;
            *reinterpret_cast<bf16x4*>(xt + (1 * XROWS + hp) * 32 + cg) = lo;
        }
        __syncthreads();
        // ---- 8 K-steps of 16 pixels (tile row y, half row xh); every tap of this wave's parity ----------
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {
            const int y = ks >> 1, xh = (ks & 1) * 16;
            const bf16x8 a_hi = tr_frag(dyt + (0 * 2 + mb) * TPX * 32, y * 32 + xh, lane);
            const bf16x8 a_lo = tr_frag(dyt + (1 * 2 + mb) * TPX * 32, y * 32 + xh, lane);
#pragma unroll
            for (int t = 0; t < MAX_TAPS_PER_WAVE; ++t) {
                const int tap = 2 * t + tpar;                  // compile-time bound below keeps acc[] in registers
                if (tap < TAPS) {
                    const int ky = tap / KW, kx = tap % KW;
                    const int hrow = (y + ky) * HW + xh + kx;  // halo pixel of this K-step's first pixel under the tap
                    const bf16x8 b_hi = tr_frag(xt + 0 * XROWS * 32, hrow, lane);
                    const bf16x8 b_lo = tr_frag(xt + 1 * XROWS * 32, hrow, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: D[row = output channel][col = input channel]; lane = column ---------------------------
    const int li = lane & 31, lh = lane >> 5;
    const int c = cchunk * 32 + li;
#pragma unroll
    for (int t = 0; t < MAX_TAPS_PER_WAVE; ++t) {
        const int tap = 2 * t + tpar;
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
    constexpr int HPX = (TH + KH - 1) * (TW + KW - 1), XROWS = (HPX + 15) / 16 * 16 + 16;
    constexpr size_t lds = (size_t)(2 * 2 * TPX * 32 + 2 * XROWS * 32) * 2;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_wgrad_kernel<KH, KW>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (attr != hipSuccess) return (int)attr;
    dim3 grid((unsigned)((a.cout + 63) / 64), (unsigned)(a.cin_pad / 32), (unsigned)a.nsplit);
    hipLaunchKernelGGL((pf_wgrad_kernel<KH, KW>), grid, dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}

// db[o] = sum over pixels of dY[p][o]; block = 64 channels x 4 pixel lanes, grid.y pixel chunks, fp32 atomics
__global__ void __launch_bounds__(256) pf_col_sum_kernel(const float* __restrict__ dy, int ld, int off, int cout,
                                                        long rows, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const long chunk = (rows + gridDim.y - 1) / gridDim.y;
    const long lo = blockIdx.y * chunk, hi = lo + chunk < rows ? lo + chunk : rows;
    float s = 0.f;
    if (c < cout)
        for (long p = lo + part; p < hi; p += 4) s += dy[p * ld + off + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < cout)
        atomicAdd(db + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

}  // namespace

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
    a.B = B; a.H = H8; a.W = W8; a.kh = kh; a.kw = kw;
    a.cin_pad = (c0 + c1 + 31) / 32 * 32;
    // split-K: enough workgroups for ~4 per CU, at most one per pixel tile
    const long ntiles = (long)B * ((H8 + TH - 1) / TH) * ((W8 + TW - 1) / TW);
    const long wg_base = (long)((cout + 63) / 64) * (a.cin_pad / 32);
    long ns = (1024 + wg_base - 1) / wg_base;
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
    if (db) {
        const long rows = (long)B * H8 * W8;
        int chunks = (int)((rows + 2047) / 2048);
        if (chunks > 256) chunks = 256;
        hipLaunchKernelGGL(pf_col_sum_kernel, dim3((unsigned)((cout + 63) / 64), (unsigned)chunks), dim3(256), 0, s,
                           dy, ld_dy, off_dy, cout, rows, db);
        rc = (int)hipGetLastError();
    }
    return rc;
}
