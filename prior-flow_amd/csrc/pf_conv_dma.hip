// pf_conv_dma_kernel: the update blocks' multi-tap convolutions (3x3, 1x5, 5x1 at 1/8 resolution; core/update.py:6-14 FlowHead,
// :35-60 SepConvGRU, :81-99 / :162-201 motion encoders) on PRE-SPLIT activations -- both MFMA operands go global -> LDS by
// DMA (global_load_lds_dwordx4), nothing is converted or written to LDS by a wave.
//
// Why.  pf_conv_ws_kernel (pf_conv_mfma.hip) already gave the MFMAs to four waves and all staging to the four others, and
// its s_memtime stamps showed the loader waves bounding the K-step: ~800 cycles for a plain step (four ds_write_b128 + four
// global_load_dwordx4 per wave) and 1 200-1 500 at the two steps of a chunk that split the fp32 halo into bf16 hi|lo on the
// VALU and store it -- against 768 cycles of matrix pipe.  The producers' epilogues now write the
// split form themselves ("split twin", include/priorflow_hip.h: per pixel and 32-channel chunk the 128-byte row
// {hi[32], lo[32]} = the LDS row image), so a halo row and a weight row are both plain 128-byte copies and the loader waves
// issue nothing but 1-KiB DMA pieces (8 rows each): 16 + ~6 per K-step and workgroup instead of 20 KB through registers.
//
// Tile and roles as in pf_conv_ws_kernel: 8 waves; waves 0..3 multiply (64 px x 32*NT channels each, acc[2][NT], the
// fragment reads of step s+1 in the gaps between the MFMAs of step s), waves 4..7 load.  WN = 2: 128 px (4 rows x 32) x
// 64*NT channels; WN = 1: 256 px (8 rows) x 32*NT channels.  One K-step = one tap of one 32-channel chunk, one barrier.
//
// LDS images (unpadded 128-byte rows: a DMA instruction writes 1 KiB lane-linear).  The 16-byte pieces of row r sit at
// piece ^ ((r >> 1) & 7): applied to the per-lane SOURCE address of the DMA and to the fragment reads, conflict-free for
// ds_read_b128 at any row shift (a tap reads rows shifted by ky*HW + kx).
//   halo  [2 buffers][HALO_ROWS][128 B]   chunk c in buffer c & 1; rows outside the image read pf_conv_desc.zeros
//   ring  [4 slots][BN][128 B]            weight tile of step s in slot s & 3
//
// Protocol (L = loader waves, M = MFMA waves; barrier(s) opens step s for both):
//   L, step s:  barrier(s); DMA W(s+3) -> slot (s+3)&3   (that slot held tile s-1: fetched in step s-2, multiplied in s-1)
//               DMA the step's share of halo(c+1) -> buffer (c+1)&1 at taps 0..TAPS-4 (last read by the fetches of chunk c-1)
//               s_waitcnt vmcnt(N), N = DMAs issued after W(s+2) = H(s-1) + W(s+3) + H(s): tile s+2 and every halo piece
//               issued before step s-1 have landed when L reaches barrier(s+1); the last 3 steps issue nothing and wait 0.
//   M, step s:  barrier(s); MFMAs of step s (fragments fetched in step s-1) with the fetch of step s+1 in the gaps:
//               it reads slot (s+1)&3 and the halo of step s+1, both retired by L's wait before barrier(s).
// A staged buffer is read one barrier after the counted wait that retires it, and rewritten two barriers after its last
// read was consumed (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it").
// Per accumulator the MFMA order over (chunk, tap, ks, pass) is pf_conv_halo_kernel's and the operand bits are the ones it
// would have made from fp32: results are bit-identical to the fp32-staged kernels (tests/test_hip_kernels.py).
#include <stdlib.h>
#include "pf_conv_priv.h"

namespace {
using namespace pfconv;

// pieces of the next chunk's halo issued at tap t: [hbeg(t), hbeg(t+1))
template <int HP, int HSTEPS>
constexpr int hbeg_(int t) { return t >= HSTEPS ? HP : HP * t / HSTEPS; }

#ifdef PF_DMA_STAMPS      // diagnostic build only (profiles/microbench_conv_dma.py): s_memtime stamps of workgroup 0, per wave and K-step
__device__ unsigned long long pf_dma_stamp_buf[8 * 64 * 4 + 4];
#define PF_DSTAMP(slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (blockIdx.x == 0 && lane == 0 && stamp_s < 64) pf_dma_stamp_buf[(wave * 64 + stamp_s) * 4 + (slot)] = t_; } while (0)
#else
#define PF_DSTAMP(slot) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wg_barrier() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PF_DMA_ABL_NO_BARRIER)
    asm volatile("s_barrier" ::: "memory");
#endif
}

template <int NT, int KH, int KW, int WN>
__global__ void __launch_bounds__(512, 2)
pf_conv_dma_kernel(const ConvGroups groups, const ConvGeom g) {
    static_assert(WN == 1 || WN == 2, "");
    constexpr int TH = 8 / WN, TW = 32, BN = 32 * NT * WN, TAPS = KH * KW;
    static_assert(TAPS >= 5, "the halo schedule needs TAPS - 3 >= 2 issue steps");
    constexpr int HW = TW + KW - 1, HH = TH + KH - 1;
    constexpr int HP = (HH * HW + 31) / 32;                 // halo DMA pieces (8 rows) per loader wave and chunk
    constexpr int HALO_ROWS = 32 * HP, HALO_BYTES = HALO_ROWS * 128;
    constexpr int WP = BN / 32;                             // weight DMA pieces per loader wave and K-step
    constexpr int SLOT_BYTES = BN * 128;
    constexpr int HSTEPS = TAPS - 3;                        // halo(c+1) is issued at taps 0 .. HSTEPS-1 of chunk c
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int RING = 2 * HALO_BYTES;                    // byte offset of the weight ring

    int grp_i, ntile_i, tile_i;                             // XCD-aware work mapping: see pf_conv_halo_kernel
    {
        const unsigned nwg = gridDim.x, orig = blockIdx.x;
        const unsigned xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
        const unsigned q = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
        ntile_i = (int)(q % (unsigned)g.ntn);
        const unsigned r = q / (unsigned)g.ntn;
        tile_i = (int)(r % (unsigned)g.ntiles);
        grp_i = (int)(r / (unsigned)g.ntiles);
    }
    pf_conv_desc d = groups.d[0];
    if (grp_i == 1) d = groups.d[1];
    else if (grp_i == 2) d = groups.d[2];
    else if (grp_i == 3) d = groups.d[3];
    const int n0 = ntile_i * BN;
    if (n0 >= d.cout) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int x0 = (tile_i % tiles_x) * TW;
    const int y0 = ((tile_i / tiles_x) % tiles_y) * TH;
    const long pix0 = (long)(tile_i / (tiles_x * tiles_y)) * g.N;
    constexpr int ph = KH / 2, pw = KW / 2;
    const int nchunks = g.nchunks;

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int lw = wave - 4;
        const int lrow = lane >> 3, lpc = lane & 7;
        // The loaders' stream is a handful of DMA issues per step, but every one of them gates the whole workgroup at the
        // next barrier, and as the younger wave of its SIMD a loader only gets the issue slots its MFMA partner leaves
        // (stamps: 150 cycles per DMA instruction, 600-1 100 per step).  Static priority for this half, no per-step flips.
        __builtin_amdgcn_s_setprio(3);
        // halo: piece j of this wave covers halo rows (lw*HP + j)*8 + lrow; per-lane source offsets in either segment
        const char* const seg0 = reinterpret_cast<const char*>(d.in0_split) + (long)(d.off0 >> 5) * 128;
        const char* const seg1 = reinterpret_cast<const char*>(d.in1_split) + (long)(d.off1 >> 5) * 128;
        long a_off0[HP], a_off1[HP];
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const int hr = (lw * HP + j) * 8 + lrow;
            const int yy = y0 + hr / HW - ph, xx = x0 + hr % HW - pw;
            const bool in = hr < HH * HW && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            // rows outside the image read the caller's block of zeros (pf_conv_desc.zeros, at least one row of the widest
            // operand): a piece's source is `uniform base + per-lane offset` either way, no select at issue time
            const long pix = pix0 + (long)yy * g.W + xx;
            const long pc = (long)((lpc ^ ((hr >> 1) & 7)) * 16);
            a_off0[j] = (in ? pix * d.lds0 * 128 : reinterpret_cast<const char*>(d.zeros) - seg0) + pc;
            a_off1[j] = (in ? pix * d.lds1 * 128 : reinterpret_cast<const char*>(d.zeros) - seg1) + pc;
        }
        const int c0chunks = d.c1 > 0 ? d.c0 >> 5 : nchunks;          // chunks of segment 0 (the whole K when there is one segment)
        // weights: piece j covers tile rows (lw*WP + j)*8 + lrow
        unsigned b_goff[WP];
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int r = (lw * WP + j) * 8 + lrow;
            b_goff[j] = (unsigned)(((long)(n0 + r) * TAPS * nchunks) * 128) + (unsigned)((lpc ^ ((r >> 1) & 7)) * 16);
        }
        const char* const wbytes = reinterpret_cast<const char*>(d.weight);
        typedef __attribute__((address_space(3))) void lds_void;
        auto dma_W = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
            const char* wp = wbytes + ((long)tap * nchunks + chunk) * 128;            // wave-uniform
#pragma unroll
            for (int j = 0; j < WP; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                lds_void* dst = (lds_void*)(smem + RING + slot * SLOT_BYTES + (lw * WP + j) * 1024);
                __builtin_amdgcn_global_load_lds(wp + b_goff[j], dst, 16, 0, 0);
#else
                (void)wp; (void)slot;
#endif
            }
        };
        auto dma_H = [&](int chunk, int buf, auto J0, auto J1) __attribute__((always_inline)) {
            const bool s0 = chunk < c0chunks;                                        // wave-uniform: a chunk lies in ONE segment
            const char* base = s0 ? seg0 + (long)chunk * 128 : seg1 + (long)(chunk - c0chunks) * 128;
            static_for<decltype(J0)::value, decltype(J1)::value>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const char* src = base + (s0 ? a_off0[j] : a_off1[j]);
#if defined(__HIP_DEVICE_COMPILE__)
                lds_void* dst = (lds_void*)(smem + buf * HALO_BYTES + (lw * HP + j) * 1024);
                __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
                (void)src; (void)buf;
#endif
            });
        };
        using I0 = std::integral_constant<int, 0>;
        // prologue: halo 0 and weight tiles 0, 1, 2; everything landed before the first fetch
        dma_H(0, 0, I0{}, std::integral_constant<int, HP>{});
        dma_W(0, 0, 0);
        dma_W(1 / TAPS, 1 % TAPS, 1);
        dma_W(2 / TAPS, 2 % TAPS, 2);
        wait_vmcnt<0>();
        wg_barrier();                                                                  // barrier(P)
        int slot3 = 3;                                                                 // (s + 3) & 3
#ifdef PF_DMA_STAMPS
        int stamp_s = 0;
        if (blockIdx.x == 0 && tid == 256) {
            unsigned long long t0, r0;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
            pf_dma_stamp_buf[8 * 64 * 4 + 0] = t0; pf_dma_stamp_buf[8 * 64 * 4 + 1] = r0;
        }
#endif
        for (int c = 0; c < nchunks; ++c) {
            const bool ih = c + 1 < nchunks;
            static_for<0, TAPS>([&](auto T) __attribute__((always_inline)) {
                constexpr int tap = decltype(T)::value;
                constexpr int tap3 = (tap + 3) % TAPS, dc3 = (tap + 3) / TAPS;
                constexpr int h0 = hbeg_<HP, HSTEPS>(tap), h1 = hbeg_<HP, HSTEPS>(tap + 1);
                constexpr int hc = h1 - h0, hp = tap > 0 ? h0 - hbeg_<HP, HSTEPS>(tap - 1) : 0;
                PF_DSTAMP(0);
                wg_barrier();                                                          // barrier(s)
                PF_DSTAMP(1);
                const bool iw = c + dc3 < nchunks;
#ifndef PF_DMA_ABL_NO_DMA          // timing-only ablations (profiles/microbench_conv_dma.py); never defined in the product build
                if (iw) dma_W(c + dc3, tap3, slot3);
                if constexpr (hc > 0) {
                    if (ih) dma_H(c + 1, (c + 1) & 1, std::integral_constant<int, h0>{}, std::integral_constant<int, h1>{});
                }
#endif
                PF_DSTAMP(2);
                if (iw) {
                    if (ih) wait_vmcnt<hp + WP + hc>();
                    else wait_vmcnt<WP>();
                } else {
                    wait_vmcnt<0>();
                }
                PF_DSTAMP(3);
#ifdef PF_DMA_STAMPS
                ++stamp_s;
#endif
                slot3 = (slot3 + 1) & 3;
            });
        }
#ifdef PF_DMA_STAMPS
        if (blockIdx.x == 0 && tid == 256) {
            unsigned long long t0, r0;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
            pf_dma_stamp_buf[8 * 64 * 4 + 2] = t0; pf_dma_stamp_buf[8 * 64 * 4 + 3] = r0;
        }
#endif
        return;                                             // the epilogue belongs to the MFMA waves (no barrier in it)
    }

    // ================================== MFMA waves ==================================
    const int wy2 = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;     // pixel rows 2*wy2, 2*wy2 + 1; channel part wn
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    // fragment addressing (byte offsets into smem).  A: halo row of M-tile m at tap (0, 0); B: this lane's 4 pieces of tile 0.
    const unsigned arow0 = (unsigned)((2 * wy2) * HW + li);
    const unsigned P0 = 2u * lh;
    // (the four pieces of a lane differ in the piece bits only: base ^ {0, 16, 64, 80})
    const unsigned b_off = (unsigned)(32 * NT * wn + li) * 128 + ((P0 ^ (((unsigned)(32 * NT * wn + li) >> 1) & 7u)) << 4);
    // fragments, double buffered in registers: [set][...][piece]; pieces 0,1 = hi K-halves, 2,3 = lo K-halves
    bf16x8 fa[2][2][4], fb[2][NT][4];
    unsigned a_addr[2] = {0, 0}, b_addr = 0;
    // A fragment of tap (ky, kx), M-tile m: halo row hr = arow0 + (ky + m) * HW + kx, piece P0 at hr*128 + ((P0 ^ swz(hr)) << 4).
    // These TAPS * 2 offsets are computed once (stamps of the first version: the dependent add / bfe / xor / shift-add chain in
    // front of a fragment read, inside one MFMA gap, stretched a 768-cycle step to 1 050); per step one add (buffer) and
    // three XORs (pieces P0+1, P0+4, P0+5) per M-tile remain, all independent.
    unsigned a_base[TAPS][2];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned hr = arow0 + (unsigned)((tp / KW + m) * HW + tp % KW);
            a_base[tp][m] = (hr << 7) + ((P0 ^ ((hr >> 1) & 7u)) << 4);
        }
    constexpr int NP = 8 + 4 * NT;                          // fragment reads per step
    constexpr int NM = 12 * NT;                             // MFMAs per step
    constexpr int FETCH_GAPS = NM - 2;
    auto fetch_piece = [&](auto SET, auto P, unsigned halo_off, auto TAP, unsigned slot_off) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value, p = decltype(P)::value, tp = decltype(TAP)::value;
        if constexpr (p < 8) {
            constexpr int m = p / 4, k = p % 4;
            if constexpr (k == 0) {
                unsigned a0 = halo_off + a_base[tp][m];
#if defined(__HIP_DEVICE_COMPILE__)       // (the host pass parses kernel bodies too: device-only constraints / builtins are fenced)
                asm volatile("" : "+v"(a0));             // opaque: hipcc would otherwise hoist the XOR variants of every (tap, buffer) too
#endif
                a_addr[m] = a0;
            }
            constexpr unsigned x = (k & 1) * 16u + (k >> 1) * 64u;          // pieces P0+1, P0+4, P0+5: XOR on the piece bits
            fa[set][m][k] = *reinterpret_cast<const bf16x8*>(smem + (a_addr[m] ^ x));
        } else {
            constexpr int t = (p - 8) / 4, k = (p - 8) % 4;
            if constexpr (t == 0 && k == 0) b_addr = (unsigned)RING + slot_off + b_off;
            constexpr unsigned x = (k & 1) * 16u + (k >> 1) * 64u;
            fb[set][t][k] = *reinterpret_cast<const bf16x8*>(smem + (b_addr ^ x) + t * 4096);
        }
    };
    wg_barrier();                                           // barrier(P): the loader waves' prologue is in LDS
    static_for<0, NP>([&](auto P) { fetch_piece(std::integral_constant<int, 0>{}, P, 0u, std::integral_constant<int, 0>{}, 0u); });
    int slot1 = 1;                                          // (s + 1) & 3
#ifdef PF_DMA_STAMPS
    int stamp_s = 0;
#endif
    auto mstep = [&](auto U, int chunk) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        constexpr int tap = u % TAPS, cur = u & 1;
        constexpr int ntap = (tap + 1) % TAPS;
        const int nchunk = (tap == TAPS - 1) ? chunk + 1 : chunk;
        const unsigned halo_off = (unsigned)((nchunk & 1) * HALO_BYTES), slot_off = (unsigned)(slot1 * SLOT_BYTES);
        using NXT = std::integral_constant<int, cur ^ 1>;
        PF_DSTAMP(0);
        wg_barrier();                 // barrier(s): slot (s+1)&3 and the halo of step s+1 are complete
        PF_DSTAMP(1);
        static_for<0, NM>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            constexpr int idx = i % (2 * NT), m = idx / NT, t = idx % NT, j = i / (2 * NT), ks = j / 3, pass = j % 3;
#ifdef PF_DMA_ABL_NO_MFMA
            if constexpr (pass == 0)                      // keep the fragment reads alive, no matrix work
                asm volatile("" :: "v"(fa[cur][m][ks]), "v"(fa[cur][m][2 + ks]), "v"(fb[cur][t][ks]), "v"(fb[cur][t][2 + ks]));
#else
            if constexpr (pass == 0)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][2 + ks], fb[cur][t][ks], acc[m][t], 0, 0, 0);
            else if constexpr (pass == 1)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][ks], fb[cur][t][2 + ks], acc[m][t], 0, 0, 0);
            else
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][ks], fb[cur][t][ks], acc[m][t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef PF_DMA_ABL_NO_READS
            static_for<0, NP>([&](auto P) __attribute__((always_inline)) {
                if constexpr (decltype(P)::value * FETCH_GAPS / NP == i) fetch_piece(NXT{}, P, halo_off, std::integral_constant<int, ntap>{}, slot_off);
            });
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef PF_DMA_STAMPS
        ++stamp_s;
#endif
        slot1 = (slot1 + 1) & 3;
    };
    {
        int c2 = 0;
        for (; c2 + 1 < nchunks; c2 += 2)
            static_for<0, 2 * TAPS>([&](auto U) { mstep(U, c2 + decltype(U)::value / TAPS); });
        if (nchunks & 1)
            static_for<0, TAPS>([&](auto U) { mstep(U, nchunks - 1); });
    }
    PF_DSTAMP(0);                                           // (stamp row nsteps: end of the K loop)
    const bool ragged = (g.W % TW) != 0 || (g.H % TH) != 0;
    long p0[2], plim[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int wy = 2 * wy2 + m;
        const bool row_ok = y0 + wy < g.H;
        const int xlim = row_ok ? g.W - x0 - 4 * lh : 0;
        p0[m] = pix0 + (long)(y0 + wy) * g.W + x0 + 4 * lh;
        plim[m] = p0[m] + (xlim > 0 ? xlim : 0);
    }
    if (ragged) tile_epilogue_pair<NT, true>(d, acc, n0 + 32 * NT * wn, li, p0, plim);
    else tile_epilogue_pair<NT, false>(d, acc, n0 + 32 * NT * wn, li, p0, plim);
#ifdef PF_DMA_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PF_DSTAMP(1);                                           // (stamp row nsteps, slot 1: epilogue stores retired)
#endif
}

#ifdef PF_DMA_STAMPS
}  // namespace
extern "C" int pf_conv_dma_read_stamps(unsigned long long* out) {   // [8 waves][64 steps][4] + {memtime, realtime} x {loop start, loop end}
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_dma_stamp_buf), sizeof(unsigned long long) * (8 * 64 * 4 + 4));
}
namespace {
#endif

template <int NT, int KH, int KW, int WN>
int launch_conv_dma_t(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, hipStream_t stream) {
    constexpr int BN = 32 * NT * WN, TH = 8 / WN;
    constexpr int HALO_ROWS = ((TH + KH - 1) * (32 + KW - 1) + 31) / 32 * 32;
    constexpr size_t lds = (size_t)2 * HALO_ROWS * 128 + (size_t)4 * BN * 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int B = g.M / g.N;
    ConvGeom gg = g;
    gg.ntiles = B * ((g.H + TH - 1) / TH) * ((g.W + 31) / 32);
    gg.ntn = (max_cout + BN - 1) / BN;
    gg.xcd_map = 1;
    const dim3 grid((unsigned)((long)gg.ntiles * gg.ntn * ngroups));
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&pf_conv_dma_kernel<NT, KH, KW, WN>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((pf_conv_dma_kernel<NT, KH, KW, WN>), grid, dim3(512), lds, stream, grp, gg);
    return (int)hipGetLastError();
}

}  // namespace

int pf_conv_dma_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, int nt, int roles,
                       hipStream_t stream) {
    for (int i = 0; i < ngroups; ++i)
        if (!grp.d[i].zeros || grp.d[i].zeros_bytes < 128 * (grp.d[i].lds0 > grp.d[i].lds1 ? grp.d[i].lds0 : grp.d[i].lds1)) return PF_ERR_BAD_ARG;
    if (roles == 2) {
        if (g.kh == 3 && g.kw == 3) return launch_conv_dma_t<2, 3, 3, 1>(grp, ngroups, g, max_cout, stream);
        return PF_ERR_BAD_SHAPE;
    }
    if (nt == 2) {
        if (g.kh == 3 && g.kw == 3) return launch_conv_dma_t<2, 3, 3, 2>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 1 && g.kw == 5) return launch_conv_dma_t<2, 1, 5, 2>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 5 && g.kw == 1) return launch_conv_dma_t<2, 5, 1, 2>(grp, ngroups, g, max_cout, stream);
    } else {
        if (g.kh == 3 && g.kw == 3) return launch_conv_dma_t<1, 3, 3, 2>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 1 && g.kw == 5) return launch_conv_dma_t<1, 1, 5, 2>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 5 && g.kw == 1) return launch_conv_dma_t<1, 5, 1, 2>(grp, ngroups, g, max_cout, stream);
    }
    return PF_ERR_BAD_SHAPE;
}
