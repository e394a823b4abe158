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

// One unit of work: output-channel tile `ntile` of pixel tile `tile` of convolution `grp`.
struct DmaItem { int grp, tile, ntile; };


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

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    constexpr int ph = KH / 2, pw = KW / 2;
    const int nchunks = g.nchunks;

    // ---- the workgroup's items (PERSISTENT: the grid is at most one workgroup per CU) ------------------------------------
    // Work sequence q = (group, pixel tile in raster order, output-channel tile), channel tile fastest, cut into 8 contiguous
    // ranges -- one per XCD, like pf_conv_halo_kernel's XCD-aware grid: workgroups that read the same or overlapping halos share
    // an L2.  The workgroups of an XCD (blockIdx % 8) walk its range with stride = their number.  Items of a group whose Cout
    // ends before the channel tile are skipped by both roles alike.  The K loop runs THROUGH item boundaries: chunk / step /
    // ring-slot / halo-buffer counters simply continue, the loaders prefetch the next item's first halo and weight tiles
    // during the last steps of the current one, and the MFMA waves' epilogue stores drain under the next item's MFMAs.
    const unsigned total = (unsigned)g.ntiles * (unsigned)g.ntn * (unsigned)g.ngroups;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    const unsigned wg_x = (nwg >> 3) + ((nwg & 7) > xcd ? 1u : 0u);                        // workgroups on this XCD
    const unsigned qd = total >> 3, rem = total & 7;
    const unsigned q_begin = xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd, q_count = qd + (xcd < rem ? 1u : 0u);
    // The descriptors through the kernel-argument segment (they are its first bytes): a wave-uniform run-time group index then
    // is a scalar load -- indexing the by-value `groups` argument would put the struct in scratch memory.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) pf_conv_desc* desc_ptr;            // constant address space: s_load
    const desc_ptr gdesc = (desc_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const pf_conv_desc* const gdesc = &groups.d[0];
#endif
    // next non-empty item of this workgroup at or after position j (in units of wg_x); false when exhausted
    auto next_item = [&](unsigned& j, DmaItem& it) -> bool {
        for (; j < q_count; j += wg_x) {
            const unsigned q = q_begin + j;
            const unsigned r = fd_div(q, g.d_ntn);
            it.ntile = (int)(q - r * (unsigned)g.ntn);
            it.grp = (int)fd_div(r, g.d_ntiles);
            it.tile = (int)(r - (unsigned)it.grp * (unsigned)g.ntiles);
            if (it.ntile * BN < gdesc[it.grp].cout) { j += wg_x; return true; }
        }
        return false;
    };
    // pixel-tile origin of an item
    auto tile_origin = [&](const DmaItem& it, int& x0, int& y0, long& pix0) __attribute__((always_inline)) {
        const unsigned ty_ = fd_div((unsigned)it.tile, g.d_tx);                          // tile / tiles_x
        x0 = (int)((unsigned)it.tile - ty_ * (unsigned)tiles_x) * TW;
        const unsigned im = fd_div(ty_, g.d_ty);                                        // image
        y0 = (int)(ty_ - im * (unsigned)tiles_y) * TH;
        pix0 = (long)im * g.N;
    };

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int lw = wave - 4;
        const int lrow = lane >> 3, lpc = lane & 7;
        // The loaders' stream is a handful of DMA issues per step, but every one of them gates the whole workgroup at the
        // next barrier, and as the younger wave of its SIMD a loader only gets the issue slots its MFMA partner leaves.
        // Static priority for this half, no per-step flips.
        __builtin_amdgcn_s_setprio(3);
        // Loader state of the item in flight: source offsets of this lane in every halo piece (either segment), weight row
        // offsets, segment / weight bases.  ONE set, switched to the next item at the two points where the current item's values
        // die: the halo offsets at the start of the item's LAST chunk (its last halo was issued during the chunk before; what is
        // issued during the last chunk is the next item's chunk 0), the weight offsets at the first step whose tile s+3 lies in
        // the next item.  (A second set selected per item ends up in scratch memory -- and a scratch load's vmcnt(0) would drain
        // the DMA queue.)  Item-independent per-piece constants are computed once.
        int h_dyx[HP]; long h_pc[HP];                      // (dy << 16 | dx & 0xffff), or INT_MIN for a row past the halo; piece offset
        static_for<0, HP>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int hr = (lw * HP + j) * 8 + lrow;
            h_dyx[j] = hr < HH * HW ? (int)(((unsigned)(hr / HW - ph) << 16) | ((unsigned)(hr % HW - pw) & 0xffffu)) : (int)0x80000000;
            h_pc[j] = (long)((lpc ^ ((hr >> 1) & 7)) * 16);
        });
        long a0[HP], a1[HP];
        unsigned bg[WP];
        const char* seg0 = nullptr; const char* seg1 = nullptr; const char* wb = nullptr;
        int c0s = 0;
        auto set_halo = [&](const DmaItem& it) __attribute__((always_inline)) {
            const auto& dd = gdesc[it.grp];
            const void* in0s = dd.in0_split; const void* in1s = dd.in1_split; const void* zeros = dd.zeros;
            const int off0 = dd.off0, off1 = dd.off1, c0 = dd.c0, c1 = dd.c1;
            const long rs0 = (long)dd.lds0 * 128, rs1 = (long)dd.lds1 * 128;
            int x0, y0; long pix0;
            tile_origin(it, x0, y0, pix0);
            seg0 = reinterpret_cast<const char*>(in0s) + (long)(off0 >> 5) * 128;
            seg1 = reinterpret_cast<const char*>(in1s) + (long)(off1 >> 5) * 128;
            c0s = c1 > 0 ? c0 >> 5 : nchunks;                     // chunks of segment 0 (the whole K when there is one segment)
            const long z0 = reinterpret_cast<const char*>(zeros) - seg0, z1 = reinterpret_cast<const char*>(zeros) - seg1;
            static_for<0, HP>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int yy = y0 + (h_dyx[j] >> 16), xx = x0 + (int)(short)(h_dyx[j] & 0xffff);
                const bool in = h_dyx[j] != (int)0x80000000 && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                // rows outside the image read the caller's block of zeros (pf_conv_desc.zeros, at least one row of the widest
                // operand): a piece's source is `uniform base + per-lane offset` either way, no select at issue time
                const long pix = pix0 + (long)yy * g.W + xx;
                a0[j] = (in ? pix * rs0 : z0) + h_pc[j];
                a1[j] = (in ? pix * rs1 : z1) + h_pc[j];
            });
        };
        auto set_w = [&](const DmaItem& it) __attribute__((always_inline)) {
            wb = reinterpret_cast<const char*>(gdesc[it.grp].weight);
            const int n0 = it.ntile * BN;
            static_for<0, WP>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int r = (lw * WP + j) * 8 + lrow;
                bg[j] = (unsigned)(((long)(n0 + r) * TAPS * nchunks) * 128) + (unsigned)((lpc ^ ((r >> 1) & 7)) * 16);
            });
        };
        typedef __attribute__((address_space(3))) void lds_void;
        auto dma_W = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
            const char* wp = wb + ((long)tap * nchunks + chunk) * 128;                // wave-uniform
            static_for<0, WP>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PF_DMA_ABL_NO_DMA)      // (timing-only ablation: no DMA issued)
                lds_void* dst = (lds_void*)(smem + RING + slot * SLOT_BYTES + (lw * WP + j) * 1024);
                __builtin_amdgcn_global_load_lds(wp + bg[j], dst, 16, 0, 0);
#else
                (void)wp; (void)slot;
#endif
            });
        };
        auto dma_H = [&](int chunk, int buf, auto J0, auto J1) __attribute__((always_inline)) {
            // a chunk lies in ONE segment (wave-uniform): two straight-line versions
            if (chunk < c0s) {
                const char* base = seg0 + (long)chunk * 128;
                static_for<decltype(J0)::value, decltype(J1)::value>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PF_DMA_ABL_NO_DMA)
                    lds_void* dst = (lds_void*)(smem + buf * HALO_BYTES + (lw * HP + j) * 1024);
                    __builtin_amdgcn_global_load_lds(base + a0[j], dst, 16, 0, 0);
#else
                    (void)base; (void)buf;
#endif
                });
            } else {
                const char* base = seg1 + (long)(chunk - c0s) * 128;
                static_for<decltype(J0)::value, decltype(J1)::value>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PF_DMA_ABL_NO_DMA)
                    lds_void* dst = (lds_void*)(smem + buf * HALO_BYTES + (lw * HP + j) * 1024);
                    __builtin_amdgcn_global_load_lds(base + a1[j], dst, 16, 0, 0);
#else
                    (void)base; (void)buf;
#endif
                });
            }
        };
        using I0 = std::integral_constant<int, 0>;
        unsigned j = wslot;
        DmaItem it, nit;
        if (!next_item(j, it)) return;                                                 // (the MFMA waves leave the same way)
        bool have_next = next_item(j, nit);
        // prologue: weight tiles 0, 1, 2 and halo 0 of the first item (the weights first: their addresses are the cheaper ones);
        // everything landed before the first fetch
        set_w(it);
        dma_W(0, 0, 0);
        dma_W(1 / TAPS, 1 % TAPS, 1);
        dma_W(2 / TAPS, 2 % TAPS, 2);
        set_halo(it);
        dma_H(0, 0, I0{}, std::integral_constant<int, HP>{});
        wait_vmcnt<0>();
        wg_barrier();                                                                  // barrier(P)
        int slot3 = 3;                                                                 // (s + 3) & 3 of the global step counter
        int gc = 0;                                                                    // global chunk counter (halo buffer = gc & 1)
        for (bool have = true; have;) {
            for (int c = 0; c < nchunks; ++c, ++gc) {
                const bool more_c = c + 1 < nchunks;
                const bool ih = more_c || have_next;                                   // a next chunk exists (here or in the next item)
                if (!more_c && have_next) set_halo(nit);                               // the item's own halo offsets are dead from here
                static_for<0, TAPS>([&](auto T) __attribute__((always_inline)) {
                    constexpr int tap = decltype(T)::value;
                    constexpr int tap3 = (tap + 3) % TAPS, dc3 = (tap + 3) / TAPS;
                    constexpr int h0 = hbeg_<HP, HSTEPS>(tap), h1 = hbeg_<HP, HSTEPS>(tap + 1);
                    constexpr int hc = h1 - h0, hp = tap > 0 ? h0 - hbeg_<HP, HSTEPS>(tap - 1) : 0;
                    wg_barrier();                                                      // barrier(s)
                    const bool in_cur = c + dc3 < nchunks;
                    const bool iw = in_cur || have_next;
                    if constexpr (tap == TAPS - 3) {                                   // first step whose tile s + 3 is in the next item
                        if (!more_c && have_next) set_w(nit);
                    }
                    if (iw) dma_W(in_cur ? c + dc3 : c + dc3 - nchunks, tap3, slot3);
                    if constexpr (hc > 0) {
                        if (ih) dma_H(more_c ? c + 1 : 0, (gc + 1) & 1, std::integral_constant<int, h0>{}, std::integral_constant<int, h1>{});
                    }
                    if (iw) {
                        if (ih) wait_vmcnt<hp + WP + hc>();
                        else wait_vmcnt<WP>();
                    } else {
                        wait_vmcnt<0>();
                    }
                    slot3 = (slot3 + 1) & 3;
                });
            }
            have = have_next;
            if (have) have_next = next_item(j, nit);
        }
        return;                                             // the epilogues belong to the MFMA waves (no barrier in them)
    }

    // ================================== MFMA waves ==================================
    const int wy2 = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;     // pixel rows 2*wy2, 2*wy2 + 1; channel part wn
    f32x16 acc[2][NT];
    // fragment addressing (byte offsets into smem).  A: halo row of M-tile m at tap (0, 0); B: this lane's piece P0 of tile 0
    // (the four pieces of a lane differ in the piece bits only: base ^ {0, 16, 64, 80}).  None of it depends on the item.
    const unsigned arow0 = (unsigned)((2 * wy2) * HW + li);
    const unsigned P0 = 2u * lh;
    const unsigned b_off = (unsigned)(32 * NT * wn + li) * 128 + ((P0 ^ (((unsigned)(32 * NT * wn + li) >> 1) & 7u)) << 4);
    // fragments, double buffered in registers: [set][...][piece]; pieces 0,1 = hi K-halves, 2,3 = lo K-halves
    bf16x8 fa[2][2][4], fb[2][NT][4];
#ifdef PF_DMA_ABL_NO_READS
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 4; ++k) {
            for (int m = 0; m < 2; ++m) for (int e = 0; e < 8; ++e) fa[i][m][k][e] = (__bf16)(float)(lane & 7);
            for (int t = 0; t < NT; ++t) for (int e = 0; e < 8; ++e) fb[i][t][k][e] = (__bf16)(float)(lane & 3);
        }
#endif
    unsigned a_addr[2] = {0, 0}, b_addr = 0;
    // A fragment of tap (ky, kx), M-tile m: halo row hr = arow0 + (ky + m) * HW + kx, piece P0 at hr*128 + ((P0 ^ swz(hr)) << 4).
    // These TAPS * 2 offsets are computed once (stamps of the first version: the dependent add / bfe / xor / shift-add chain in
    // front of a fragment read, inside one MFMA gap, stretched a 768-cycle step to 1 050); per step one add (buffer) and
    // three XORs (pieces P0+1, P0+4, P0+5) per M-tile remain, all independent.
    unsigned a_base[TAPS][2];
    auto set_a_base = [&]() __attribute__((always_inline)) {
        // recomputed per item from an opaque copy of the row: 2 * TAPS registers that are NOT live across the epilogue (with the
        // accumulators and the GRU operands it otherwise spills)
        unsigned ar = arow0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ar));
#endif
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned hr = ar + (unsigned)((tp / KW + m) * HW + tp % KW);
                a_base[tp][m] = (hr << 7) + ((P0 ^ ((hr >> 1) & 7u)) << 4);
            }
    };
    set_a_base();
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
#ifndef PF_DMA_ABL_NO_READS                               // (timing-only ablation: fragments stay whatever they are)
            fa[set][m][k] = *reinterpret_cast<const bf16x8*>(smem + (a_addr[m] ^ x));
#endif
        } else {
            constexpr int t = (p - 8) / 4, k = (p - 8) % 4;
            if constexpr (t == 0 && k == 0) b_addr = (unsigned)RING + slot_off + b_off;
            constexpr unsigned x = (k & 1) * 16u + (k >> 1) * 64u;
#ifndef PF_DMA_ABL_NO_READS
            fb[set][t][k] = *reinterpret_cast<const bf16x8*>(smem + (b_addr ^ x) + t * 4096);
#endif
        }
    };
    unsigned jm = wslot;
    DmaItem it;
    if (!next_item(jm, it)) return;
    wg_barrier();                                           // barrier(P): the loader waves' prologue is in LDS
    static_for<0, NP>([&](auto P) __attribute__((always_inline)) { fetch_piece(std::integral_constant<int, 0>{}, P, 0u, std::integral_constant<int, 0>{}, 0u); });
    int slot1 = 1;                                          // (s + 1) & 3 of the global step counter
    int gc = 0;                                             // global chunk counter
    auto mstep = [&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        constexpr int tap = u % TAPS, cur = u & 1;
        constexpr int ntap = (tap + 1) % TAPS;
        // the halo of step s + 1: this chunk's buffer, or the next chunk's (of this item or the next one) after the last tap
        const unsigned halo_off = (unsigned)(((tap == TAPS - 1 ? gc + 1 : gc) & 1) * HALO_BYTES), slot_off = (unsigned)(slot1 * SLOT_BYTES);
        using NXT = std::integral_constant<int, cur ^ 1>;
        wg_barrier();                 // barrier(s): slot (s+1)&3 and the halo of step s+1 are complete
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
            static_for<0, NP>([&](auto P) __attribute__((always_inline)) {
                if constexpr (decltype(P)::value * FETCH_GAPS / NP == i) fetch_piece(NXT{}, P, halo_off, std::integral_constant<int, ntap>{}, slot_off);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        slot1 = (slot1 + 1) & 3;
        if constexpr (tap == TAPS - 1) ++gc;
    };
    const bool ragged = (g.W % TW) != 0 || (g.H % TH) != 0;
    bool have = true;
    while (have) {
        if (const float* pre = gdesc[it.grp].pre) {
            // accumulators start from pf_conv_desc.pre (the iteration-invariant part of a GRU conv) instead of zero: element
            // (m, t, r) is pixel p0[m] + roff(r), channel jb + 32 t + li -- the epilogue's mapping.  The loads of an item's
            // first launch run under the loader waves' prologue; each MFMA waits only for its own accumulator.
            const auto& dp = gdesc[it.grp];
            const int ldp = dp.ld_pre, jb = it.ntile * BN + 32 * NT * wn + li;
            const int cout = dp.cout;
            pre += dp.off_pre;
            int x0, y0; long pix0;
            tile_origin(it, x0, y0, pix0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int wy = 2 * wy2 + m;
                const int xlim = (y0 + wy < g.H) ? g.W - x0 - 4 * lh : 0;
                const long pm = pix0 + (long)(y0 + wy) * g.W + x0 + 4 * lh;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool jok = jb + 32 * t < cout;
                    const float* pp = pre + pm * ldp + jb + 32 * t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        acc[m][t][r] = (jok && ro < xlim) ? pp[(long)ro * ldp] : 0.f;
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
        }
        {
            int c2 = 0;
            for (; c2 + 1 < nchunks; c2 += 2)
                static_for<0, 2 * TAPS>([&](auto U) __attribute__((always_inline)) { mstep(U); });
            if (nchunks & 1)
                static_for<0, TAPS>([&](auto U) __attribute__((always_inline)) { mstep(U); });
        }
        // ---- epilogue of this item (stores are not waited for: they drain under the next item's MFMAs) ----
        const auto& dd = gdesc[it.grp];
        pf_conv_desc d;                      // the fields the epilogue reads (scalar loads)
        d.bias = dd.bias; d.out = dd.out; d.ld_out = dd.ld_out; d.off_out = dd.off_out; d.cout = dd.cout;
        d.epilogue = dd.epilogue; d.scale = dd.scale; d.h = dd.h; d.ld_h = dd.ld_h; d.z = dd.z; d.ld_z = dd.ld_z;
        d.aux_out = dd.aux_out; d.ld_aux = dd.ld_aux; d.precision = dd.precision;
        d.out_split = dd.out_split; d.lds_out = dd.lds_out; d.aux_split = dd.aux_split; d.lds_aux = dd.lds_aux; d.save_gates = dd.save_gates;
        const int n0 = it.ntile * BN;
        int x0, y0; long pix0;
        tile_origin(it, x0, y0, pix0);
        long p0[2], plim[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int wy = 2 * wy2 + m;
            const bool row_ok = y0 + wy < g.H;
            const int xlim = row_ok ? g.W - x0 - 4 * lh : 0;
            p0[m] = pix0 + (long)(y0 + wy) * g.W + x0 + 4 * lh;
            plim[m] = p0[m] + (xlim > 0 ? xlim : 0);
        }
#ifdef PF_DMA_ABL_NO_EPI                                  // (timing-only ablation: the accumulators stay alive, nothing is stored)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[0][NT - 1]), "v"(acc[1][NT - 1]));
#endif
        (void)d; (void)p0; (void)plim;
#else
        if (ragged) tile_epilogue_pair<NT, true>(d, acc, n0 + 32 * NT * wn, li, p0, plim);
        else tile_epilogue_pair<NT, false>(d, acc, n0 + 32 * NT * wn, li, p0, plim);
#endif
        have = next_item(jm, it);
        // The fragments of the next item's first step were fetched during this item's last step, but keeping them (64 VGPRs)
        // alive across the epilogue (accumulators + GRU operands) does not fit the register file: fetch them again -- their
        // halo buffer and ring slot were retired before barrier(s_last) and nothing rewrites them before barrier(s_last + 2).
        if (have) {
            set_a_base();
            const unsigned hoff = (unsigned)((gc & 1) * HALO_BYTES), soff = (unsigned)(((slot1 + 3) & 3) * SLOT_BYTES);
            static_for<0, NP>([&](auto P) __attribute__((always_inline)) { fetch_piece(std::integral_constant<int, 0>{}, P, hoff, std::integral_constant<int, 0>{}, soff); });
        }
    }
}

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
    gg.ngroups = ngroups;
    gg.d_ntn = fd_make((unsigned)gg.ntn); gg.d_ntiles = fd_make((unsigned)gg.ntiles);
    gg.d_tx = fd_make((unsigned)((g.W + 31) / 32)); gg.d_ty = fd_make((unsigned)((g.H + TH - 1) / TH));
    // persistent: at most one workgroup per CU (the kernel's LDS footprint admits no second one), each walking its share of
    // the work sequence.  Per DEVICE: the CU count and the > 64 KB dynamic-LDS attribute belong to the device that is current
    // at launch time (one process may drive several cards).
    constexpr int MAX_DEV = 64;
    static int cus_of[MAX_DEV];              // 0 = not asked yet
    static bool attr_set[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    if (cus_of[dev] == 0) {
        int n = 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus_of[dev] = n;
    }
    int cap = cus_of[dev];
    const long items = (long)gg.ntiles * gg.ntn * ngroups;
    // the kernel cuts the work sequence into 8 ranges (one per XCD) that the workgroups with blockIdx % 8 == range walk: a
    // capped grid needs at least one workgroup per range, or the ranges without one are never computed
    if (cap > 0 && cap < 8 && items > cap) cap = 8;
    const dim3 grid((unsigned)((cap > 0 && items > cap) ? cap : items));
    if (!attr_set[dev]) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_conv_dma_kernel<NT, KH, KW, WN>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return (int)attr;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pf_conv_dma_kernel<NT, KH, KW, WN>), grid, dim3(512), lds, stream, grp, gg);
    return (int)hipGetLastError();
}

}  // namespace

// One translation unit per kernel shape (PF_DMA_PART = 0..6, compiled in parallel by __graft_entry__.build_hip: an unrolled
// K-step body takes about a minute per instantiation) plus the dispatcher (PF_DMA_PART = 7).
//   part: 0 <2,3,3,1>  1 <2,3,3,2>  2 <2,1,5,2>  3 <2,5,1,2>  4 <1,3,3,2>  5 <1,1,5,2>  6 <1,5,1,2>  8 <2,1,5,1>  9 <2,5,1,1>   (<NT, KH, KW, WN>)
#ifndef PF_DMA_PART
#error "compile pf_conv_dma.hip with -DPF_DMA_PART=0..7"
#endif
#define PF_DMA_DEFINE_PART(N, NT, KH, KW, WN)                                                                              \
    int pf_conv_dma_part##N##_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout,   \
                                     hipStream_t stream) {                                                                  \
        return launch_conv_dma_t<NT, KH, KW, WN>(grp, ngroups, g, max_cout, stream);                                        \
    }
#if PF_DMA_PART == 0
PF_DMA_DEFINE_PART(0, 2, 3, 3, 1)
#elif PF_DMA_PART == 1
PF_DMA_DEFINE_PART(1, 2, 3, 3, 2)
#elif PF_DMA_PART == 2
PF_DMA_DEFINE_PART(2, 2, 1, 5, 2)
#elif PF_DMA_PART == 3
PF_DMA_DEFINE_PART(3, 2, 5, 1, 2)
#elif PF_DMA_PART == 4
PF_DMA_DEFINE_PART(4, 1, 3, 3, 2)
#elif PF_DMA_PART == 5
PF_DMA_DEFINE_PART(5, 1, 1, 5, 2)
#elif PF_DMA_PART == 6
PF_DMA_DEFINE_PART(6, 1, 5, 1, 2)
#elif PF_DMA_PART == 8
PF_DMA_DEFINE_PART(8, 2, 1, 5, 1)
#elif PF_DMA_PART == 9
PF_DMA_DEFINE_PART(9, 2, 5, 1, 1)
#else
#define PF_DMA_DECLARE_PART(N) int pf_conv_dma_part##N##_launch(const pfconv::ConvGroups&, int, const pfconv::ConvGeom&, int, hipStream_t);
PF_DMA_DECLARE_PART(0) PF_DMA_DECLARE_PART(1) PF_DMA_DECLARE_PART(2) PF_DMA_DECLARE_PART(3)
PF_DMA_DECLARE_PART(4) PF_DMA_DECLARE_PART(5) PF_DMA_DECLARE_PART(6) PF_DMA_DECLARE_PART(8) PF_DMA_DECLARE_PART(9)

int pf_conv_dma_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, int nt, int roles,
                       hipStream_t stream) {
    for (int i = 0; i < ngroups; ++i)
        if (!grp.d[i].zeros || grp.d[i].zeros_bytes < 128 * (grp.d[i].lds0 > grp.d[i].lds1 ? grp.d[i].lds0 : grp.d[i].lds1)) return PF_ERR_BAD_ARG;
    const bool k33 = g.kh == 3 && g.kw == 3, k15 = g.kh == 1 && g.kw == 5, k51 = g.kh == 5 && g.kw == 1;
    if (roles == 2) {           // 256 px x 64 channels per workgroup: half the weight bytes staged per output, twice the halo
        if (k33) return pf_conv_dma_part0_launch(grp, ngroups, g, max_cout, stream);
        if (k15) return pf_conv_dma_part8_launch(grp, ngroups, g, max_cout, stream);
        if (k51) return pf_conv_dma_part9_launch(grp, ngroups, g, max_cout, stream);
        return PF_ERR_BAD_SHAPE;
    }
    if (nt == 2) {
        if (k33) return pf_conv_dma_part1_launch(grp, ngroups, g, max_cout, stream);
        if (k15) return pf_conv_dma_part2_launch(grp, ngroups, g, max_cout, stream);
        if (k51) return pf_conv_dma_part3_launch(grp, ngroups, g, max_cout, stream);
    } else {
        if (k33) return pf_conv_dma_part4_launch(grp, ngroups, g, max_cout, stream);
        if (k15) return pf_conv_dma_part5_launch(grp, ngroups, g, max_cout, stream);
        if (k51) return pf_conv_dma_part6_launch(grp, ngroups, g, max_cout, stream);
    }
    return PF_ERR_BAD_SHAPE;
}
#endif
