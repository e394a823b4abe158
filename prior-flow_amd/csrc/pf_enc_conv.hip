// pf_enc_conv64: the encoders' 3x3 convolutions with 64 input and 64 output channels at 1/2 resolution (core/extractor.py:16-17,
// ResidualBlock conv1 / conv2 of layer1; fnet: 4B images of 256 x 512 at 512x1024), PF_PREC_BF16X3 arithmetic, with the WEIGHTS
// STATIONARY (round 5).
//
// The halo kernel (pf_conv_halo_kernel<2,3,3,.,8>) re-stages the layer's 147 KB of split weights for every 256-pixel tile -- more
// bytes than the tile's input halo (87 KB) and output (64 KB) together -- and its staging and matrix phases do not overlap
// (profiles/r5_encoder_ablation.txt: loop 108 us = data path alone 57 + matrix alone 51, of 155 us).  Here a workgroup keeps the
// weights for its whole life:
//   * W_hi, the bf16 high halves, of the wave's 32 output channels x all K = 576 in REGISTERS (36 fragments = 144 VGPRs: the
//     MFMA's B operand);
//   * W_lo, the low halves of all 64 output channels, in LDS (73.7 KB, one ds_read_b128 per K-step);
// and walks DOWN a strip of 32 columns in steps of 4 rows: the input lives in a 10-row ring in LDS (the 6 rows a step reads + the
// 4 new rows of the next step, staged -- affine + ReLU of the previous layer's norm, bf16 hi|lo split -- while the step multiplies),
// so a pixel is fetched once per strip (+ 2 halo columns), not once per tile and chunk.  8 waves = 2 channel tiles x 4 rows, one
// 32 x 32 accumulator each; the two channel tiles ALTERNATE between a matrix phase (a whole step) and a vector phase (epilogue +
// staging), one barrier per phase.  K order, pass order and the statistics' summation order are the halo kernel's:
// outputs and fused InstanceNorm partials are BIT-IDENTICAL to it (tests/test_hip_kernels.py).
// LDS: 73 728 + 10 x 34 x 256 = 160 768 B of the 163 840.
#include <stdlib.h>
#include "pf_conv_priv.h"

namespace {
using namespace pfconv;

constexpr int EC_ROWB = 34 * 256;                    // bytes of a ring row: 34 pixels x 2 chunks x {hi[32], lo[32]} bf16
constexpr int EC_RING = 10;
constexpr int EC_KSTEPS = 36;                        // 2 chunks x 9 taps x 2 K-steps of 16 channels
constexpr int EC_WLO = EC_KSTEPS * 2048;             // [k-step][channel tile][k half][32 channels][16 B]
constexpr int EC_AFF = EC_WLO + EC_RING * EC_ROWB;   // input affine of the image: scale[64], shift[64] fp32
constexpr int EC_LDS = EC_AFF + 512;
constexpr int EC_WROW = 9 * 64 * 4;                  // bytes of one output channel's packed weights: [tap][64 ch as 2 x {hi[32], lo[32]}]
#ifndef PF_EC_AHEAD                                  // K-steps the fragment reads run ahead of the MFMAs (1: 123-126 us, 2: 130-135 us on 4 images)
#define PF_EC_AHEAD 1
#endif
#ifndef PF_EC_ABL                                    // timing-only diagnosis builds: 1 no staging, 2 no epilogue, 4 no K loop
#define PF_EC_ABL 0
#endif

struct EcArgs {
    const float* in; int ld_in;                      // input rows (channel offset folded in)
    const char* w; const float* bias;
    float* out; int ld_out;                          // output rows (channel offset folded in)
    const float* in_scale; const float* in_shift; int in_relu;
    double* stats;                                   // [image][nseg * 4 * strips][64][2] or NULL: one partial per (segment, row phase, strip)
    const float* res; int ld_res;                    // PF_EPI_RELU_RES: out = relu(res + relu(acc + bias)) (relu = 1); else NULL
    float scale; int relu;
    int Bn, H, W, seg, nseg, strips;                 // seg: rows per work item (a multiple of 4); nseg = H / seg; strips = W / 32
};

template <int I, int N, class F>
__device__ __forceinline__ void ec_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ec_for<I + 1, N>(f);
    }
}

__global__ void __launch_bounds__(512)
pf_enc_conv64_kernel(const EcArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int abl = PF_EC_ABL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nt = wave >> 2, mr = wave & 3;         // channel tile, row of the step
    // work item: (image, segment of rows, strip of 32 columns); consecutive workgroups = neighbouring strips of one segment
    const int sx = blockIdx.x % a.strips;
    const int sg = (blockIdx.x / a.strips) % a.nseg;
    const int im = blockIdx.x / (a.strips * a.nseg);
    const int x0 = 32 * sx, yseg = sg * a.seg;
    const int nsteps = a.seg >> 2;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(unsigned long)(lds_char*)lds;
    const unsigned ring_base = lds_base + EC_WLO;

    // K order = the halo kernel's (bit-identical sums): chunk c of 32 channels, tap, K-step h of that chunk; inside an MFMA the
    // lane half kh multiplies channels 16 kh + 8 h .. + 7 of the chunk (the staging kernels' K permutation)
    // ---- weights: W_lo -> LDS, W_hi -> registers -------------------------------------------------------------------
    for (int e = tid; e < EC_KSTEPS * 128; e += 512) {
        const int s = e >> 7, r = e & 127, tnt = r >> 6, kh = (r >> 5) & 1, n = r & 31;
        const int c = s / 18, tap = (s >> 1) % 9, h = s & 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.w + (long)(32 * tnt + n) * EC_WROW + (tap * 64 + c * 32) * 4 + 64 + (kh * 16 + h * 8) * 2);
        *reinterpret_cast<f32x4*>(lds + s * 2048 + tnt * 1024 + kh * 512 + n * 16) = v;
    }
    if (tid < 128) {
        float v = tid < 64 ? 1.f : 0.f;
        if (a.in_scale != nullptr) v = tid < 64 ? a.in_scale[(long)im * 64 + tid] : a.in_shift[(long)im * 64 + tid - 64];
        reinterpret_cast<float*>(lds + EC_AFF)[tid] = v;
    }
    bf16x8 whi[EC_KSTEPS];
    {
        const char* wrow = a.w + (long)(32 * nt + li) * EC_WROW + lh * 32;
#pragma unroll
        for (int s = 0; s < EC_KSTEPS; ++s) {
            const int c = s / 18, tap = (s >> 1) % 9, h = s & 1;
            whi[s] = *reinterpret_cast<const bf16x8*>(wrow + (tap * 64 + c * 32) * 4 + h * 16);
        }
    }
    __syncthreads();                                     // the affine is in LDS

    // ---- staging of input rows.  Thread -> 4 channels f4 of column cg (and, for the first 64 threads, of a halo column):
    // a half of a step's new rows (2 rows) is 3 items per thread, all addresses by shifts
    const int f4 = tid & 15, cg = tid >> 4;              // cg 0..31
    const bool affine = a.in_scale != nullptr, in_relu = a.in_relu != 0;
    const float* const in_img = a.in + (long)im * a.H * a.W * a.ld_in + 4 * f4;
    // LDS byte offset of this thread's hi half inside a pixel: chunk (f4 >> 3), 16-byte piece ((f4 & 7) >> 1), half piece (f4 & 1)
    const unsigned st_piece = (unsigned)((f4 >> 3) * 8 + ((f4 & 7) >> 1)), st_in = (unsigned)((f4 & 1) * 8);
    // A thread's items always sit in the same two columns: cg + 1 (rows 0..3 of a step's new rows) and, for cg < 8, the halo column
    // 33 (cg & 1) of row cg >> 1 -- column validity, source offset and LDS offsets are per-thread constants, a row adds scalars
    const int hcol = (cg & 1) * 33, hrow = cg >> 1;
    const bool has_halo = cg < 8;
    const bool xok_m = x0 + cg < a.W, xok_h = hcol ? (x0 + 32 < a.W) : (x0 > 0);
    const int goff_m = (x0 + cg) * a.ld_in, goff_h = (x0 - 1 + hcol) * a.ld_in;          // floats from the row start (valid when xok)
    auto lds_px = [&](int col) { const unsigned sw = (unsigned)(col & 15);
                                 return (unsigned)(EC_WLO + col * 256) + ((st_piece ^ sw) << 4) + st_in; };
    auto lds_px_lo = [&](int col) { const unsigned sw = (unsigned)(col & 15);
                                    return (unsigned)(EC_WLO + col * 256) + (((st_piece + 4) ^ sw) << 4) + st_in; };
    const unsigned lm_hi = lds_px(cg + 1), lm_lo = lds_px_lo(cg + 1), lh_hi = lds_px(hcol), lh_lo = lds_px_lo(hcol);
    auto convert_store = [&](f32x4 x, bool ok, unsigned off_hi, unsigned off_lo) __attribute__((always_inline)) {
        if (affine) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(lds + EC_AFF + 16 * f4), sh = *reinterpret_cast<const f32x4*>(lds + EC_AFF + 256 + 16 * f4);
            x = x * sc + sh;
            if (in_relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        }
        const f32x4 v = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};              // zero padding applies AFTER the affine
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const f32x4 rest = v - __builtin_convertvector(hi, f32x4);
        const bf16x4 lo = __builtin_convertvector(rest, bf16x4);
        *reinterpret_cast<bf16x4*>(lds + off_hi) = hi;
        *reinterpret_cast<bf16x4*>(lds + off_lo) = lo;
    };
    // rows rr0 .. rr0 + 3 (relative to the segment): item u < 4 = (row rr0 + u, column cg + 1), item 4 = the halo item
    auto rows_load = [&](int rr0, bool live, f32x4 (&st)[5], bool (&ok)[5]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int rr = rr0 + (u < 4 ? u : hrow), y = yseg + rr;
            const bool yok = y >= 0 && y < a.H;                            // (u < 4: wave-uniform)
            ok[u] = live && yok && (u < 4 ? xok_m : (xok_h && has_halo));
            const float* src = in_img + (long)(yok ? y : 0) * a.W * a.ld_in + (u < 4 ? goff_m : goff_h);
            st[u] = ok[u] ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto rows_store = [&](int rr0, bool live, const f32x4 (&st)[5], const bool (&ok)[5]) __attribute__((always_inline)) {
        if (!live) return;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (u == 4 && !has_halo) continue;
            const int rr = rr0 + (u < 4 ? u : hrow);
            const unsigned rowb = (unsigned)(((rr + 1 + EC_RING) % EC_RING) * EC_ROWB);
            convert_store(st[u], ok[u], rowb + (u < 4 ? lm_hi : lh_hi), rowb + (u < 4 ? lm_lo : lh_lo));
        }
    };
    // The same for ONE group of four waves (256 threads) and TWO rows (the ping-pong schedule below: a group stages half of a
    // step's new rows in its vector phase).  Thread -> channels f4 of columns 1 + cgl and 17 + cgl of both rows; threads with
    // cgl < 4 also the halo column 33 (cgl & 1) of row cgl >> 1.
    const int cgl = (tid & 255) >> 4;
    const int hcol2 = (cgl & 1) * 33, hrow2 = cgl >> 1;
    const bool has_halo2 = cgl < 4;
    const bool xok_h2 = hcol2 ? (x0 + 32 < a.W) : (x0 > 0);
    const int goff_g0 = (x0 + cgl) * a.ld_in, goff_g1 = (x0 + cgl + 16) * a.ld_in, goff_gh = (x0 - 1 + hcol2) * a.ld_in;
    const unsigned lg0_hi = lds_px(cgl + 1), lg0_lo = lds_px_lo(cgl + 1), lg1_hi = lds_px(cgl + 17), lg1_lo = lds_px_lo(cgl + 17);
    const unsigned lgh_hi = lds_px(hcol2), lgh_lo = lds_px_lo(hcol2);
    auto grp_load = [&](int rr0, bool live, f32x4 (&st)[5], bool (&ok)[5]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int rr = rr0 + (u < 4 ? (u >> 1) : hrow2), y = yseg + rr;
            const bool yok = y >= 0 && y < a.H;
            ok[u] = live && yok && (u < 4 ? true : (xok_h2 && has_halo2));
            const float* src = in_img + (long)(yok ? y : 0) * a.W * a.ld_in + (u < 4 ? ((u & 1) ? goff_g1 : goff_g0) : goff_gh);
            st[u] = ok[u] ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto grp_store = [&](int rr0, bool live, const f32x4 (&st)[5], const bool (&ok)[5]) __attribute__((always_inline)) {
        if (!live) return;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (u == 4 && !has_halo2) continue;
            const int rr = rr0 + (u < 4 ? (u >> 1) : hrow2);
            const unsigned rowb = (unsigned)(((rr + 1 + EC_RING) % EC_RING) * EC_ROWB);
            convert_store(st[u], ok[u], rowb + (u < 4 ? ((u & 1) ? lg1_hi : lg0_hi) : lgh_hi), rowb + (u < 4 ? ((u & 1) ? lg1_lo : lg0_lo) : lgh_lo));
        }
    };
    // prologue: rows -1 .. 4 of the segment (what step 0 reads); rows 5, 6 are staged again by the loop (harmless)
    {
        f32x4 pst[5]; bool pok[5];
        rows_load(-1, true, pst, pok);
        rows_store(-1, true, pst, pok);
        rows_load(3, true, pst, pok);
        rows_store(3, true, pst, pok);
    }

    // ---- fragment addresses: pixel column (li + kx), 16-byte piece = (8 c + h [+ 4 for lo]) ^ (2 kh) ^ swizzle(column)
    unsigned a_kx[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a_kx[kx] = (unsigned)((li + kx) * 256 + ((((li + kx) & 15) ^ (lh << 1)) << 4));
    const unsigned wlo_lane = lds_base + (unsigned)(nt * 1024 + lh * 512 + li * 16);
    const float bias = a.bias[32 * nt + li];

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // ---- state of the main loop (below): the two waves of a SIMD are wave w and w + 4 -- the same row of a step, the two channel
    // tiles -- and the channel tile is the wave's group
    const int grp = nt;
    f32x16 acc = zero16;
    double s1 = 0.0, s2 = 0.0;          // InstanceNorm statistics of this wave's rows of the segment (one partial per wave and item)
    f32x4 st[5]; bool st_ok[5];
    unsigned ar9[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};        // LDS address of pixel (row ky, column li + kx) of this step, per tap
    constexpr int NB = PF_EC_AHEAD + 1;
    bf16x8 ahi[NB], alo[NB], blo[NB];       // fragments in flight: the reads run PF_EC_AHEAD K-steps ahead of the MFMAs
    auto frag = [&](auto S, int buf) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        constexpr int c = s / 18, tap = (s >> 1) % 9, h = s & 1, ky = tap / 3, kx = tap % 3;
        constexpr unsigned phi = (unsigned)((c * 8 + h) << 4), plo = phi + 64;
        const unsigned ad_hi = ar9[ky * 3 + kx] ^ phi, ad_lo = ar9[ky * 3 + kx] ^ plo;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b128 %0, %1" : "=v"(ahi[buf]) : "v"(ad_hi));
        asm volatile("ds_read_b128 %0, %1" : "=v"(alo[buf]) : "v"(ad_lo));
        if constexpr (s * 2048 < 65536)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(blo[buf]) : "v"(wlo_lane), "n"(s * 2048));
        else {
            const unsigned wl2 = wlo_lane + 32 * 2048;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(blo[buf]) : "v"(wl2), "n"((s - 32) * 2048));
        }
#endif
    };
    // A whole step: 36 K-steps, the operands of the next PF_EC_AHEAD K-steps in flight while one multiplies
    auto kloop = [&]() __attribute__((always_inline)) {
        ec_for<0, PF_EC_AHEAD>([&](auto S) __attribute__((always_inline)) { frag(S, decltype(S)::value % NB); });
        ec_for<0, EC_KSTEPS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value, b = s % NB;
            constexpr int left = EC_KSTEPS - 1 - s;                 // K-steps after this one
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (left >= PF_EC_AHEAD) frag(std::integral_constant<int, s + PF_EC_AHEAD>{}, (s + PF_EC_AHEAD) % NB);
            // (everything but the reads of the K-steps still ahead has arrived)
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(ahi[b]), "+v"(alo[b]), "+v"(blo[b]) : "n"(3 * (left < PF_EC_AHEAD ? left : PF_EC_AHEAD)));
            __builtin_amdgcn_sched_barrier(0);
            // the halo kernel's pass order: x_lo * w_hi, x_hi * w_lo, x_hi * w_hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[b], whi[s], s == 0 ? zero16 : acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[b], blo[b], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[b], whi[s], acc, 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);

    };
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        // this wave's row of step t: 32 pixels x 32 channels; acc[r] = pixel (r & 3) + 8 (r >> 2) + 4 lh, channel 32 nt + li
        if (abl & 2) { asm volatile("" :: "v"(acc)); return; }
        const int y = yseg + 4 * t + mr;
        // values + statistics in the accumulator layout (a lane = one channel, 16 pixels), then a 4 x 4 transpose inside every quad
        // of lanes (DPP: lane j of a quad ends up with pixel j's four channels) so that a lane stores 16 contiguous bytes: 4 store
        // instructions of 1 KB per wave and step instead of 16 of 256 B -- the CU issues one vector-memory instruction per ~46
        // cycles whatever its size, and the narrow stores alone kept that path busy for 40 % of the kernel
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = a.relu ? fmaxf(acc[r] + bias, 0.f) : (acc[r] + bias) * a.scale;
        if (a.stats != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double dv = (double)v[r];
                s1 += dv; s2 += dv * dv;
            }
        }
        const bool b0 = lane & 1, b1 = lane & 2;
        auto xch = [&](float& x0, float& x1, bool odd, auto CTRL) __attribute__((always_inline)) {   // 2 x 2 exchange with the lane CTRL selects
            const float send = odd ? x0 : x1;
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), decltype(CTRL)::value, 0xf, 0xf, true));
            x0 = odd ? recv : x0;
            x1 = odd ? x1 : recv;
        };
        using X1 = std::integral_constant<int, 0xB1>;      // quad_perm [1,0,3,2]: lane ^ 1
        using X2 = std::integral_constant<int, 0x4E>;      // quad_perm [2,3,0,1]: lane ^ 2
        const long opix = ((long)im * a.H + y) * a.W + x0 + 4 * lh + (lane & 3);
        float* const ot = a.out + opix * a.ld_out + 32 * nt + (li & ~3);
        const float* const rt = a.res != nullptr ? a.res + opix * a.ld_res + 32 * nt + (li & ~3) : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xch(v[4 * g], v[4 * g + 1], b0, X1{});
            xch(v[4 * g + 2], v[4 * g + 3], b0, X1{});
            xch(v[4 * g], v[4 * g + 2], b1, X2{});
            xch(v[4 * g + 1], v[4 * g + 3], b1, X2{});
            // lane (quad position j) now holds pixel 8 g + j (+ 4 lh): channels 4 (li >> 2) .. + 3
            f32x4 w4 = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            if (a.res != nullptr) {                           // ResidualBlock tail (core/extractor.py:47), the residual in the stores' layout
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(rt + (long)(8 * g) * a.ld_res);
                w4.x = fmaxf(h4.x + w4.x, 0.f); w4.y = fmaxf(h4.y + w4.y, 0.f); w4.z = fmaxf(h4.z + w4.z, 0.f); w4.w = fmaxf(h4.w + w4.w, 0.f);
            }
            *reinterpret_cast<f32x4*>(ot + (long)(8 * g) * a.ld_out) = w4;
        }
        if (a.stats != nullptr && t == nsteps - 1) {
            // InstanceNorm statistics of the stored values: fp64 sum / sum of squares per channel over this wave's rows of the
            // segment (rows mr, mr + 4, ... x 32 pixels), one partial per (segment, row phase, strip) --
            // [image][nseg * 4 * strips][64][2], reduced by pf_channel_stats_final (pf_conv2d_stats_blocks gives the count).
            // (A partial per row and strip made that reduction 20 us instead of 5 at fnet's layer-1 size.)
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                double* qd = a.stats + (((((long)im * a.nseg + sg) * 4 + mr) * a.strips + sx) * 64 + 32 * nt + li) * 2;
                qd[0] = s1; qd[1] = s2;
            }
        }
    };
    // ---- main loop: the two groups ALTERNATE.  In a half-step one group multiplies a whole step (36 K-steps, 108 MFMAs per wave)
    // while the other does everything that is not matrix work: the epilogue of the step it has just finished (bias / ReLU /
    // residual, transposes, stores, statistics) and the staging of half of the next step's new rows (loads first, their latency
    // under the epilogue, then affine + split + LDS writes).  So on every SIMD one wave's MFMAs run beside its partner's vector
    // and memory instructions all the time -- in the round's first schedule (both waves multiplied half a step per half-step, then
    // both converted) the two waves of a SIMD shared the matrix pipe and then both left it idle: 13.8 K cycles per step for 6.9 K
    // of MFMA, 142 us per launch on four images against 123 with the alternation.  Group 0 multiplies step t in half-step 2 t, group 1 in 2 t + 1.  New rows of step s (4 s + 1 .. 4 s + 4):
    // group 1 stages the first two in half-step 2 s - 2, group 0 the last two in 2 s - 1; their ring slots held rows only step s - 2
    // read, which both groups have left by then, and the slots the multiplying group reads (rows 4 t - 1 .. 4 t + 4) are never
    // written in the same half-step.  One barrier per half-step.
    for (int hs = 0; hs <= 2 * nsteps; ++hs) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (only the LDS traffic has to be complete here)
        if ((hs & 1) == grp) {
            const int t = (hs - grp) >> 1;
            if (t < nsteps && !(abl & 4)) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const unsigned rb = __builtin_amdgcn_readfirstlane(ring_base + (unsigned)(((4 * t + mr + ky) % EC_RING) * EC_ROWB));
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) ar9[ky * 3 + kx] = rb + a_kx[kx];
                }
                kloop();
            }
        } else {
            const int tf = (hs - 1 - grp) >> 1;                    // the step this group multiplied in the previous half-step (-1: none yet)
            const int sn = tf + 1 + grp;                            // the step whose rows it stages now: group 1 runs one step further ahead
            const int rr0 = 4 * sn + 1 + 2 * (1 - grp);             // group 1: rows 4 sn + 1, + 2; group 0: rows 4 sn + 3, + 4
            const bool more = tf >= 0 && sn < nsteps && !(abl & 1); // (step 1's first two rows came with the prologue)
            grp_load(rr0, more, st, st_ok);
            if (tf >= 0 && tf < nsteps && !(abl & 4)) epilogue(tf);
            grp_store(rr0, more, st, st_ok);
        }
    }
}

}  // namespace

// Whether pf_conv2d hands this launch to pf_enc_conv64_kernel (PRIORFLOW_ENC_CONV64=0: never): one group, 3x3 stride 1, 64 -> 64,
// fp32 rows in and out, LINEAR, RELU or RELU_RES epilogue, a map of whole 32-column strips and 4-row steps that fills the chip.
bool pf_enc_conv64_applies(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout) {
    static const int mode = [] { const char* e = getenv("PRIORFLOW_ENC_CONV64"); return e ? atoi(e) : 1; }();     // 2: also on small maps (tests)
    if (mode <= 0 || ngroups != 1) return false;
    const pf_conv_desc& d = grp.d[0];
    return d.precision == PF_PREC_BF16X3 && g.kh == 3 && g.kw == 3 && g.stride == 1 && d.c0 == 64 && d.c1 == 0 && d.cout == 64 &&
           max_cout == 64 && d.in0 != nullptr && d.out != nullptr && d.out_split == nullptr && d.pre == nullptr &&
           (d.epilogue == PF_EPI_LINEAR || d.epilogue == PF_EPI_RELU || (d.epilogue == PF_EPI_RELU_RES && d.h != nullptr && (d.ld_h % 4) == 0)) &&
           (g.W % 32) == 0 && (g.H % 8) == 0 &&
           (d.ld0 % 4) == 0 && (d.off0 % 4) == 0 && (d.ld_out % 4) == 0 && (d.off_out % 4) == 0 &&
           (mode >= 2 || (long)(g.M / g.N) * (g.H / 8) * (g.W / 32) >= 256);
}

// rows per work item: the longest segment (a divisor of H, a multiple of 8: the weights are loaded once per item and a strip's
// rows are fetched once) that still gives every CU an item (4 images of 256 x 512: 64 rows, 256 items -- two rounds of
// 32-row items measured 146-159 us against 139)
static int ec_segment(const pfconv::ConvGeom& g) {
    const int Bn = g.M / g.N;
    for (int sg = g.H; sg > 8; sg -= 8)
        if (g.H % sg == 0 && (long)Bn * (g.H / sg) * (g.W / 32) >= 256) return sg;
    return 8;
}

// fp64 statistics partials per image of a launch with stats_out: one per (segment, row phase of the 4-row step, strip)
int pf_enc_conv64_stats_blocks(const pfconv::ConvGeom& g) { return (g.H / ec_segment(g)) * 4 * (g.W / 32); }

int pf_enc_conv64_launch(const pfconv::ConvGroups& grp, const pfconv::ConvGeom& g, hipStream_t stream) {
    const pf_conv_desc& d = grp.d[0];
    EcArgs a;
    a.in = d.in0 + d.off0; a.ld_in = d.ld0;
    a.w = reinterpret_cast<const char*>(d.weight); a.bias = d.bias;
    a.out = d.out + d.off_out; a.ld_out = d.ld_out;
    a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_relu = d.in_relu;
    a.stats = d.stats_out; a.scale = d.scale; a.relu = d.epilogue == PF_EPI_RELU || d.epilogue == PF_EPI_RELU_RES;
    a.res = d.epilogue == PF_EPI_RELU_RES ? d.h : nullptr; a.ld_res = d.ld_h;
    a.Bn = g.M / g.N; a.H = g.H; a.W = g.W;
    a.seg = ec_segment(g);
    a.nseg = g.H / a.seg; a.strips = g.W / 32;
    const long items = (long)a.Bn * a.nseg * a.strips;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_enc_conv64_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, EC_LDS);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(pf_enc_conv64_kernel, dim3((unsigned)items), dim3(512), EC_LDS, stream, a);
    return (int)hipGetLastError();
}
