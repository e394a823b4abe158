// pf_conv2d: stride-1 "same" convolution on channel-last activations as an implicit GEMM on the
// gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32), with fused bias / activation /
// SepConvGRU gating epilogues.  Replaces the nn.Conv2d forwards of the reference's update
// blocks (PriOr-RAFT/core/update.py:6-14 FlowHead, :35-60 SepConvGRU, :81-99 and :162-201 motion
// encoders, :124-127/:147-150 mask heads).
//
// GEMM view:  M = B*H8*W8 pixels,  N = Cout,  K = KH*KW*Cin (tap-major, channel-minor).
//   A[m][k]  = input pixel (y+dy, x+dx) channel c   (zero outside the map), K-contiguous rows
//   B[n][k]  = packed weight [Cout_pad][KH*KW][Cin_pad]                     K-contiguous rows
// One K-step = one tap x one 32-channel chunk.  Both tiles are staged through LDS with
// coalesced 16-byte loads (8 lanes per 128-byte row) and read back as MFMA operands with
// ds_read_b128: the K index inside a chunk is PERMUTED so that lane (row i, half h) owns the
// 16 consecutive channels [16h, 16h+16) -- MFMA step s multiplies channel 16h+s of A and B --
// which turns the one-float-per-lane f32 operand into four 16-byte LDS reads per chunk.
// Row stride 36 floats (144 B) makes those reads and the 16-byte staging writes bank-conflict
// free (9*row mod 16 is a bijection over any 16 distinct rows).
//
// Precision modes (pf_conv_desc.precision):
//   PF_PREC_F32    exact fp32: v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains (1/16 of the bf16 rate)
//   PF_PREC_BF16X3 3-pass bf16 split: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
//                  a*b ~= hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
//                  (16 mantissa bits per operand; the dropped lo*lo term is 2^-16 relative).
//                  3/16 of the exact-fp32 MFMA time.  Same tiling and LDS footprint: a 32-channel
//                  LDS row is [hi bf16 x32 | lo bf16 x32] (128 B + 16 B pad).  Weights are split
//                  offline into that row format (the B path stays a pure 16-byte copy);
//                  activations stay fp32 in HBM and are split when staged into LDS.
//
// Software pipeline: global loads of K-step s+1 are issued before the MFMAs of step s and
// written to the other LDS buffer at the top of the next iteration; one __syncthreads per K-step.
#include <stdlib.h>
#include "pf_conv_priv.h"

namespace {
using namespace pfconv;

template <int WM, int WN, int NT, bool SPLIT>
__global__ void __launch_bounds__(256, 2)   // 2 waves/SIMD -> 256-register budget, no spills
pf_conv_mfma_kernel(const ConvGroups groups, const ConvGeom g) {
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int A_V4 = BM * 8 / 256;     // float4 per thread per K-step (A tile)
    constexpr int B_V4 = BN * 8 / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for 256 loader threads");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                  // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;                // [2][BN][LDS_LD]

    // static indices only: a dynamic index into the by-value kernarg struct would be
    // lowered through scratch memory
    pf_conv_desc d = groups.d[0];
    if (blockIdx.z == 1) d = groups.d[1];
    else if (blockIdx.z == 2) d = groups.d[2];
    else if (blockIdx.z == 3) d = groups.d[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar wave id
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    if (n0 >= d.cout) return;                          // groups may have different Cout

    // ---- loader assignment: A rows -----------------------------------------------------------
    int a_y[A_V4], a_x[A_V4];                          // INPUT-space position of the tap centre
    long a_pix[A_V4];                                  // its global input pixel index, or -1
#pragma unroll
    for (int q = 0; q < A_V4; ++q) {
        const int idx = tid + 256 * q;
        const int r = idx >> 3;
        const long p = (long)m0 + r;
        if (p < g.M) {
            const int n = (int)(p % g.N);
            a_y[q] = (n / g.W) * g.stride; a_x[q] = (n % g.W) * g.stride;
            a_pix[q] = (p / g.N) * g.Nin + (long)a_y[q] * g.Win + a_x[q];
        } else {
            a_y[q] = 0; a_x[q] = 0; a_pix[q] = -1;
        }
    }
    const int c4 = (tid & 7) * 4;                      // channel offset inside the chunk
    const int ctot = d.c0 + d.c1;
    const int ph = g.kh / 2, pw = g.kw / 2;
    const long wrow = (long)g.taps * g.cin_pad;        // floats per packed weight row

    f32x4 ra[A_V4], rb[B_V4];
    unsigned a_ok = 0;                                 // bit q: ra[q] is a real (not padded) load

    auto load_step = [&](int step) __attribute__((always_inline)) {
        const int tap = step / g.nchunks;
        const int cbase = (step - tap * g.nchunks) * KC;
        const int dy = tap / g.kw - ph, dx = tap % g.kw - pw;
        const int c = cbase + c4;
        const float* src; int ld, cc;
        if (c < d.c0) { src = d.in0 + d.off0; ld = d.ld0; cc = c; }
        else          { src = d.in1 + d.off1; ld = d.ld1; cc = c - d.c0; }
        const bool cok = c < ctot;
        unsigned okbits = 0;
#pragma unroll
        for (int q = 0; q < A_V4; ++q) {
            const int yy = a_y[q] + dy, xx = a_x[q] + dx;
            const bool ok = cok && a_pix[q] >= 0 && yy >= 0 && yy < g.Hin && xx >= 0 && xx < g.Win;
            // branch-free: masked lanes read a valid dummy address (row 0 of segment 0) and
            // are zeroed afterwards, so the whole K-step stays in one basic block
            const long sp = a_pix[q] + (long)dy * g.Win + dx;
            const float* ptr = ok ? src + sp * ld + cc : d.in0 + d.off0;
            ra[q] = *reinterpret_cast<const f32x4*>(ptr);
            okbits |= ok ? (1u << q) : 0u;
        }
        a_ok = okbits;   // the zeroing select happens at LDS-store time: nothing in the MFMA block
                         // of this iteration waits on these loads
        const float* wp = d.weight + (long)tap * g.cin_pad + cbase + c4;
#pragma unroll
        for (int q = 0; q < B_V4; ++q) {
            const int r = (tid + 256 * q) >> 3;
            rb[q] = *reinterpret_cast<const f32x4*>(wp + (long)(n0 + r) * wrow);
        }
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
        float* as = As + buf * BM * LDS_LD;
        float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int q = 0; q < A_V4; ++q) {
            const int r = (tid + 256 * q) >> 3;
            const f32x4 v = ((a_ok >> q) & 1u) ? ra[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                // hi = bf16(v) (RNE, v_cvt_pk_bf16_f32); lo = bf16(v - hi): the subtraction is exact
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const f32x4 rest = v - __builtin_convertvector(hi, f32x4);
                const bf16x4 lo = __builtin_convertvector(rest, bf16x4);
                char* row = reinterpret_cast<char*>(as + r * LDS_LD);
                *reinterpret_cast<bf16x4*>(row + 2 * c4) = hi;
                *reinterpret_cast<bf16x4*>(row + 64 + 2 * c4) = lo;
            } else {
                *reinterpret_cast<f32x4*>(as + r * LDS_LD + c4) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < B_V4; ++q) {
            const int r = (tid + 256 * q) >> 3;
            *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c4) = rb[q];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = g.taps * g.nchunks;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (32 * wm + li) * LDS_LD + 16 * lh;
    const int b_off = (32 * NT * wn + li) * LDS_LD + 16 * lh;

    // registers hold K-step `step` at the top of each iteration; its LDS buffer was last read
    // two iterations ago (all waves have passed the previous barrier), so ONE barrier suffices
    load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        store_step(buf);
        __syncthreads();
        // unconditional prefetch (the last one re-reads the final K-step; never stored)
        load_step(step + 1 < nsteps ? step + 1 : step);
        // keep the prefetch ABOVE the MFMA block: without this compiler-level barrier hipcc sinks
        // the loads to their first use (top of the next iteration) and waits for them there
        asm volatile("" ::: "memory");

        const float* as = As + buf * BM * LDS_LD + a_off;
        const float* bs = Bs + buf * BN * LDS_LD + b_off;
        if constexpr (SPLIT) {
            // lane (row li, half lh) owns channels [16lh, 16lh+16): bytes [32lh, 32lh+32) of the hi
            // part and the same of the lo part (+64 B); K-step ks uses the ks-th 16 bytes of each
            const char* ap = reinterpret_cast<const char*>(as);     // a_off already holds 16*lh floats = 64lh B
            ap -= 32 * lh;                                          // -> 32lh bytes into the row
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks] = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
                al[ks] = *reinterpret_cast<const bf16x8*>(ap + 64 + 16 * ks);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const char* bp = reinterpret_cast<const char*>(bs + t * 32 * LDS_LD) - 32 * lh;
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bh[ks] = *reinterpret_cast<const bf16x8*>(bp + 16 * ks);
                    bl[ks] = *reinterpret_cast<const bf16x8*>(bp + 64 + 16 * ks);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh[ks], acc[t], 0, 0, 0);
                }
            }
        } else {
            f32x4 af[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(as + 4 * q);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 bf[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bf[q] = *reinterpret_cast<const f32x4*>(bs + t * 32 * LDS_LD + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].w, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: acc[t][r] = D[row (r&3)+8(r>>2)+4h][col lane&31] ---------------------------
    tile_epilogue<NT, true>(d, acc, n0 + 32 * NT * wn, li, (long)m0 + 32 * wm + 4 * lh, g.M);
    if (d.stats_out != nullptr) {
        // InstanceNorm statistics of this tile's stored values (round 4: the stride-2 layers of fnet used to pay a separate
        // pf_stats_partial pass over their output): fp64 sum / sum of squares per channel over the workgroup's BM pixels,
        // partial [m tile][cout][2].  The launcher only passes stats_out when N % BM == 0 (a tile never straddles two images), so
        // m tile = image * (N / BM) + tile in image: the [image][nblk] order pf_channel_stats_final reads.
        __syncthreads();                                   // every wave has left the K loop: the operand LDS is free
        double* red = reinterpret_cast<double*>(smem);     // [4 waves][32 * NT channels][2]
        const long prow = (long)m0 + 32 * wm + 4 * lh;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = n0 + 32 * NT * wn + 32 * t + li;
            const float bias = j < d.cout ? d.bias[j] : 0.f;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double v = (prow + (r & 3) + 8 * (r >> 2) < (long)g.M) ? (double)((acc[t][r] + bias) * d.scale) : 0.0;
                s1 += v; s2 += v * v;
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if ((lane >> 5) == 0) {
                red[((wave * (32 * NT)) + 32 * t + li) * 2 + 0] = s1;
                red[((wave * (32 * NT)) + 32 * t + li) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.cout) {
            const int cw = tid / (32 * NT), cl = tid % (32 * NT);      // channel part (wn) and slot inside it
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int y = 0; y < WM; ++y) {                               // the waves (pixel rows) of this part, in order
                const int wv = y * WN + cw;
                s1 += red[(wv * (32 * NT) + cl) * 2 + 0];
                s2 += red[(wv * (32 * NT) + cl) * 2 + 1];
            }
            double* o = d.stats_out + ((long)blockIdx.x * d.cout + n0 + tid) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Halo-tile kernel (PF_PREC_BF16X3; 3x3, 1x5, 5x1, 4x4 and 1x1 taps; any map size, partial edge tiles).
//
// With 3-pass bf16 MFMAs a K-step carries 5x less matrix time than in exact fp32, so the generic
// kernel above becomes bound by what surrounds the MFMAs: every tap re-loads and re-splits the
// same activations, and one step of prefetch no longer covers the global-load latency.  Here a
// workgroup (8 waves) owns a 4-row x 32-column pixel tile x BN output channels and, per
// 32-channel chunk, stages the (4+KH-1) x (32+KW-1) input HALO once (fp32 -> bf16 hi|lo split done
// once per chunk instead of once per tap; 1.6-2x fewer activation bytes than per-tap tiles).  The
// KH*KW taps of the chunk then read SHIFTED rows of that LDS image as their A operand.  Weight
// tiles (already split offline) stream through a 3-slot LDS ring fed by two register sets (global
// loads issued 4 K-steps ahead); the next chunk's halo is loaded at the chunk's first tap and
// converted/written at its last-but-one tap.  MFMA operand fragments are double buffered in
// registers: the ds_reads of step s+1 are issued before the MFMAs of step s, so the LDS round
// trip is off the critical path (measured: without this the loop ran at ~40 % of the MFMA rate
// even with barriers, LDS writes and global loads removed).  One barrier per K-step.
// Wave w: tile row w>>1 (32 pixels = one MFMA M-block), output channels (w&1)*32*NT + [0, 32*NT).
// ----------------------------------------------------------------------------------------------
#ifdef PF_STAMPS      // diagnostic build only (profiles/stamp_conv.py): per-wave s_memtime stamps inside the K-step
#define PF_STAMP_OFF (140 * 1024 / 4)      // floats: the stamp area sits above the operand buffers of the TH = 4 instantiations
__device__ unsigned long long pf_stamp_buf[8 * 40 * 8];
#endif

template <int NT, int KH, int KW, bool AFFINE, int TH>
__global__ void __launch_bounds__(512, 2)      // 8 waves = 2 per SIMD, 256-register budget
pf_conv_halo_kernel(const ConvGroups groups, const ConvGeom g) {
    // TH = 4: waves 4 (rows) x 2 (channel halves), tile 128 px x 64*NT channels.
    // TH = 8: waves 8 (rows) x 1, tile 256 px x 32*NT channels -- for Cout <= 64 layers (encoder
    //         stem / layer 1) this gives every wave two accumulators per staged K-step instead of one.
    static_assert(TH == 4 || TH == 8, "");
    constexpr int TW = 32, WN = (TH == 4) ? 2 : 1, BN = 32 * NT * WN, TAPS = KH * KW;
    static_assert(BN % 32 == 0, "");
    // weight loader: 64 rows per pass of the 512 threads.  BN = 96 (TH = 8, NT = 3: the encoders' layer 2): the second pass has 32
    // rows left -- both halves of the workgroup load and store rows 64..95 (the same bits to the same place twice) instead of
    // half of the threads branching around the pass, which would cost the K loop its counted waits
    constexpr int B_V4 = (BN + 63) / 64;      // weight float4 per thread per K-step
    constexpr int HW = TW + KW - 1, HH = TH + KH - 1;
    constexpr int A_V4 = (HH * HW + 63) / 64; // halo float4 per thread (64 rows per pass of 512 threads)
    constexpr int HALO_ROWS = 64 * A_V4;      // rows past HH*HW get zeros
    // Weight tiles by LDS-DMA (global_load_lds_dwordx4; multi-tap kernels): no VGPR staging, no ds_write.
    // The DMA writes lane-linear 1-KiB pieces (8 rows x 128 B), so the ring slots are UNPADDED 128-B rows
    // with the 16-byte pieces XOR-swizzled by (row >> 1) & 7 on the source address and on the fragment read
    // (conflict-free ds_read_b128); 4 slots: the tile of step s+3 is issued at step s and retired by a
    // counted s_waitcnt vmcnt(N) in front of the raw barrier of step s+2 (a __syncthreads would add
    // vmcnt(0) while a DMA is in flight); the halo loads are then inline-asm loads with counted waits too.
    // OPT-IN (-DPF_DMA_B): validated by the full GPU suite, but measured a wash -- back-to-back launches of
    // one conv: -3..-6 % (z|r 55.3 vs 57.4 us, heads 58.2 vs 62.1, q 33.5 vs 34.4) at TH = 4, +5 % at TH = 8;
    // inside the forward (same-call traces): NT = 2 kernels -2..-3 %, NT = 1 kernels up to +6 %, end to end
    // 107.7 vs 108.8 pairs/s.  The default stays the register-staged weight ring.
#ifdef PF_DMA_B
    constexpr bool DMA_B = TAPS > 1 && TH == 4;
#else
    constexpr bool DMA_B = false;
#endif
    constexpr int B_SLOTS = DMA_B ? 4 : 3;
    constexpr int B_ROW = DMA_B ? 128 : LDS_LD * 4;     // bytes per weight row in LDS
    constexpr int B_DMA = (BN + 63) / 64;               // 1-KiB DMA pieces per wave per K-step (DMA_B: TH = 4 only, BN % 64 == 0)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ah = smem;                                   // [2][HALO_ROWS][LDS_LD]
    float* Bs = smem + 2 * HALO_ROWS * LDS_LD;          // [B_SLOTS][BN][B_ROW bytes]  (ring)

    // XCD-aware work mapping (1-D grid).  Workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2; every
    // XCD gets a CONTIGUOUS range of the work sequence  q = (group, pixel tile in raster order, output-channel tile) with
    // the channel tile fastest: the workgroups that read the same input halo (the channel tiles of one pixel tile) and
    // overlapping halos (vertically / horizontally adjacent pixel tiles) then run on the same XCD at the same time and
    // find each other's lines in its L2.  With the plain 3-D grid they were spread over all XCDs and every halo came
    // from beyond L2: the PMC pass read 89 MB of fabric-side fetches per 5x1 GRU launch for 25 MB of input.
    int grp_i, ntile_i, tile_i;
    if (g.xcd_map) {
        const unsigned nwg = gridDim.x, orig = blockIdx.x;
        const unsigned xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;     // bijective for any nwg
        const unsigned q = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
        ntile_i = (int)(q % (unsigned)g.ntn);
        const unsigned r = q / (unsigned)g.ntn;
        tile_i = (int)(r % (unsigned)g.ntiles);
        grp_i = (int)(r / (unsigned)g.ntiles);
    } else {
        tile_i = blockIdx.x; ntile_i = blockIdx.y; grp_i = blockIdx.z;
    }
    pf_conv_desc d = groups.d[0];
    if (grp_i == 1) d = groups.d[1];
    else if (grp_i == 2) d = groups.d[2];
    else if (grp_i == 3) d = groups.d[3];
    const int n0 = ntile_i * BN;
    if (n0 >= d.cout) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar wave id
    const int li = lane & 31, lh = lane >> 5;
    const int wy = (TH == 4) ? wave >> 1 : wave, wn = (TH == 4) ? wave & 1 : 0;
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;   // edge tiles may be partial
    const int tile = tile_i;
    const int x0 = (tile % tiles_x) * TW;
    const int y0 = ((tile / tiles_x) % tiles_y) * TH;
    const long pix0 = (long)(tile / (tiles_x * tiles_y)) * g.N;      // batch offset in pixels
    constexpr int ph = KH / 2, pw = KW / 2;

    // ---- halo loader assignment: thread -> 4 (row, 16-byte column) slots ------------------------
    long a_pix[A_V4];         // source pixel (global row) or -1 (outside the image / past the halo)
#pragma unroll
    for (int q = 0; q < A_V4; ++q) {
        const int r = (tid + 512 * q) >> 3;
        const int yy = y0 + r / HW - ph, xx = x0 + r % HW - pw;
        a_pix[q] = (r < HH * HW && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)
                       ? pix0 + (long)yy * g.W + xx : -1;
    }
    const int c4 = (tid & 7) * 4;
    const int ctot = d.c0 + d.c1;
    const long wrow = (long)TAPS * g.cin_pad;
    const int nchunks = g.nchunks;
    const int nsteps = nchunks * TAPS;
    // TAPS == 1 (1x1 convs): a chunk lasts ONE K-step, so the halo follows the weight ring's cadence --
    // two register sets, chunk s+2 stored and chunk s+4 issued at step s (see `step`).
    constexpr int NSET = TAPS == 1 ? 2 : 1;
    static_assert(!(AFFINE && TAPS == 1), "the input affine is loaded once per chunk at tap 0 of a multi-tap conv");
    f32x4 ra[NSET][A_V4];
    unsigned a_loff[A_V4];
#pragma unroll
    for (int q = 0; q < A_V4; ++q) a_loff[q] = (unsigned)((((tid + 512 * q) >> 3) * LDS_LD) * 4 + 2 * c4);
    f32x4 a_sc = {1.f, 1.f, 1.f, 1.f}, a_sh = {0.f, 0.f, 0.f, 0.f};   // input affine of this thread's 4 channels
    const long aff_row = (long)(tile / (tiles_x * tiles_y)) * ctot;     // [image][channel]
    unsigned a_ok[NSET] = {};
    // 16-byte global load.  With the weight DMA in flight hipcc cannot count past it and waits vmcnt(0) at
    // the first use of ANY ordinary load result (draining the DMA queue once per chunk); in that mode the
    // halo loads are issued by inline asm -- invisible to the compiler's wait tracking -- and retired by the
    // counted waits of `wait_A` below.
    auto load16 = [&](f32x4& dst, const float* ptr) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DMA_B) {
            f32x4 t;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
            dst = t;
        } else {
            dst = *reinterpret_cast<const f32x4*>(ptr);
        }
#else
        dst = *reinterpret_cast<const f32x4*>(ptr);
#endif
    };
    // The halo loader/converter works in A_V4 independent slices (q) so that a K-step can spread
    // them between its MFMAs (see `step`).
    auto load_A_affine = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (AFFINE) {     // compile-time: a runtime-conditional load would break the counted waits
            if (chunk >= nchunks) chunk = nchunks - 1;
            const int c = chunk * KC + c4;
            const int ca = c < ctot ? c : 0;
            load16(a_sc, d.in_scale + aff_row + ca);
            load16(a_sh, d.in_shift + aff_row + ca);
        }
    };
    auto load_A_q = [&](int chunk, auto Q, auto SET) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, set = decltype(SET)::value;
        if (chunk >= nchunks) chunk = nchunks - 1;       // tail: harmless re-read, stored to the idle buffer
        const int c = chunk * KC + c4;
        const float* src; int ld, cc;
        if (c < d.c0) { src = d.in0 + d.off0; ld = d.ld0; cc = c; }
        else          { src = d.in1 + d.off1; ld = d.ld1; cc = c - d.c0; }
        const bool ok = c < ctot && a_pix[q] >= 0;
        const float* ptr = ok ? src + a_pix[q] * ld + cc : d.in0 + d.off0;
        load16(ra[set][q], ptr);
        a_ok[set] = (a_ok[set] & ~(1u << q)) | (ok ? (1u << q) : 0u);
    };
    auto store_A_q = [&](int buf, auto Q, auto SET) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value, set = decltype(SET)::value;
        // previous layer's norm (+ReLU) folded into this load; zero padding applies AFTER it
        f32x4 x = ra[set][q];
        if constexpr (AFFINE) {
            x = x * a_sc + a_sh;
            if (d.in_relu) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
            }
        }
        const f32x4 v = ((a_ok[set] >> q) & 1u) ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        // hi = bf16(v) (RNE); lo = bf16(v - hi): the subtraction is exact in fp32
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const f32x4 rest = v - __builtin_convertvector(hi, f32x4);
        const bf16x4 lo = __builtin_convertvector(rest, bf16x4);
        char* row = reinterpret_cast<char*>(Ah) + buf * (HALO_ROWS * LDS_LD * 4) + a_loff[q];
        *reinterpret_cast<bf16x4*>(row) = hi;
        *reinterpret_cast<bf16x4*>(row + 64) = lo;
    };
    using SET0 = std::integral_constant<int, 0>;
    using SET1 = std::integral_constant<int, NSET - 1>;
    auto load_A = [&](int chunk, auto SET) __attribute__((always_inline)) {
        load_A_affine(chunk);
        static_for<0, A_V4>([&](auto Q) { load_A_q(chunk, Q, SET); });
    };
    auto store_A = [&](int buf, auto SET) __attribute__((always_inline)) {
        static_for<0, A_V4>([&](auto Q) { store_A_q(buf, Q, SET); });
    };

    // ---- weight ring: 3 LDS slots, 2 register sets (steps s+2, s+3 staged; s+4 issued) -----------
    // Per-thread address parts are computed ONCE (byte offsets, 32-bit): a K-step's loads are then
    // `wave-uniform base + constant VGPR offset` -- the first version recomputed 64-bit products
    // per step and spent ~60 VALU instructions per wave per step on addressing alone.
    f32x4 rb0[B_V4], rb1[B_V4];
    unsigned b_goff[B_V4];        // global: ((n0 + row) * wrow + c4) * 4 bytes
    unsigned b_loff[B_V4];        // LDS:    (row * LDS_LD + c4) * 4 bytes
    // (Rotating the row order per workgroup, so that the CUs of an XCD do not ask its L2 for the same weight line at the
    // same moment, was measured: no effect.)
#pragma unroll
    for (int q = 0; q < B_V4; ++q) {
        const int r = (BN % 64 != 0 && q == B_V4 - 1) ? 64 * q + ((tid & 255) >> 3) : (tid + 512 * q) >> 3;
        b_goff[q] = (unsigned)(((long)(n0 + r) * wrow + c4) * 4);
        b_loff[q] = (unsigned)((r * LDS_LD + c4) * 4);
    }
    const char* const wbytes = reinterpret_cast<const char*>(d.weight);
    char* const bs_bytes = reinterpret_cast<char*>(Bs);
    auto load_B = [&](int step, f32x4 (&rb)[B_V4]) __attribute__((always_inline)) {
        if (step >= nsteps) step = nsteps - 1;           // tail: harmless re-read
        const int chunk = step / TAPS, tap = step - chunk * TAPS;
        const char* wp = wbytes + ((long)tap * g.cin_pad + chunk * KC) * 4;      // wave-uniform
#pragma unroll
        for (int q = 0; q < B_V4; ++q)
            rb[q] = *reinterpret_cast<const f32x4*>(wp + b_goff[q]);
    };
    auto store_B = [&](int slot, const f32x4 (&rb)[B_V4]) __attribute__((always_inline)) {
        char* bs = bs_bytes + slot * (BN * B_ROW);                                 // wave-uniform
#pragma unroll
        for (int q = 0; q < B_V4; ++q)
            *reinterpret_cast<f32x4*>(bs + b_loff[q]) = rb[q];
    };

    // DMA path: piece j of this wave covers rows (wave*B_DMA + j)*8 + (lane>>3), LDS position lane&7
    unsigned dma_goff[B_DMA];
#pragma unroll
    for (int j = 0; j < B_DMA; ++j) {
        const int r = (wave * B_DMA + j) * 8 + (lane >> 3);
        dma_goff[j] = (unsigned)(((long)(n0 + r) * wrow) * 4 + (((lane & 7) ^ ((r >> 1) & 7)) * 16));
    }
    auto dma_B = [&](int step, int slot) __attribute__((always_inline)) {
        if (step >= nsteps) step = nsteps - 1;           // tail: harmless re-read into a free slot
        const int chunk = step / TAPS, tap = step - chunk * TAPS;
        const char* wp = wbytes + ((long)tap * g.cin_pad + chunk * KC) * 4;      // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)       // (the host pass parses kernel bodies too; these are device-only constructs)
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) {
            typedef __attribute__((address_space(3))) void lds_void;
            lds_void* dst = (lds_void*)(bs_bytes + slot * (BN * B_ROW) + (wave * B_DMA + j) * 1024);
            __builtin_amdgcn_global_load_lds(wp + dma_goff[j], dst, 16, 0, 0);
        }
#else
        (void)wp; (void)slot;
#endif
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const char* a_lane = reinterpret_cast<const char*>(Ah + (wy * HW + li) * LDS_LD) + 32 * lh;
    const char* b_lane = bs_bytes + (32 * NT * wn + li) * B_ROW + (DMA_B ? 0 : 32 * lh);
    // byte offsets of this lane's four 16-byte pieces (hi k0-7, hi k8-15, lo k0-7, lo k8-15) inside its row
    unsigned b_piece[4];
    {
        const unsigned swz = DMA_B ? (unsigned)((li >> 1) & 7) : 0u, p0 = DMA_B ? 2u * lh : 0u;
        b_piece[0] = ((p0 + 0) ^ swz) * 16; b_piece[1] = ((p0 + 1) ^ swz) * 16;
        b_piece[2] = ((p0 + 4) ^ swz) * 16; b_piece[3] = ((p0 + 5) ^ swz) * 16;
    }

    // MFMA operand fragments, double buffered in registers: set (s&1) feeds step s while set
    // ((s+1)&1) is being filled from LDS for step s+1, so the LDS round trip hides behind MFMAs.
    // [0],[1] = hi K-halves, [2],[3] = lo K-halves.
    bf16x8 fa[2][4], fb[2][NT][4];
    auto fetch_A = [&](auto SET, int halo_buf, int ky, int kx) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const char* ap = a_lane + (halo_buf * HALO_ROWS + ky * HW + kx) * (LDS_LD * 4);
        fa[set][0] = *reinterpret_cast<const bf16x8*>(ap);
        fa[set][1] = *reinterpret_cast<const bf16x8*>(ap + 16);
        fa[set][2] = *reinterpret_cast<const bf16x8*>(ap + 64);
        fa[set][3] = *reinterpret_cast<const bf16x8*>(ap + 80);
    };
    auto fetch_B = [&](auto SET, auto T, int slot) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value, t = decltype(T)::value;
        const char* bp = b_lane + (slot * BN + 32 * t) * B_ROW;
        fb[set][t][0] = *reinterpret_cast<const bf16x8*>(bp + b_piece[0]);
        fb[set][t][1] = *reinterpret_cast<const bf16x8*>(bp + b_piece[1]);
        fb[set][t][2] = *reinterpret_cast<const bf16x8*>(bp + b_piece[2]);
        fb[set][t][3] = *reinterpret_cast<const bf16x8*>(bp + b_piece[3]);
    };

    // One 16-byte fragment read (piece P of the 4 A pieces + 4*NT B pieces of a step).  The K loop issues them ONE OR TWO
    // PER MFMA GAP: with whole fragments fetched in the first gaps (4 reads per gap, all 8 waves right after the barrier)
    // the LDS queue filled up, the reads blocked the in-order instruction stream and the MFMAs behind them: s_memtime
    // stamps (profiles/stamp_conv.py) showed ~900 cycles of issue phase for a wave's 12 MFMAs (384 cycles of matrix pipe).
    constexpr int NP = 4 + 4 * NT;
    constexpr int FETCH_GAPS = 6 * NT - (NT == 1 ? 1 : 2);      // the last gap(s) stay free: a read's latency is hidden, not exposed at the barrier
    auto fetch_piece = [&](auto SET, auto P, int halo_buf, int ky, int kx, int slot) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value, p = decltype(P)::value;
        if constexpr (p < 4) {
            const char* ap = a_lane + (halo_buf * HALO_ROWS + ky * HW + kx) * (LDS_LD * 4);
            constexpr int off = (p & 1) * 16 + (p >> 1) * 64;
            fa[set][p] = *reinterpret_cast<const bf16x8*>(ap + off);
        } else {
            constexpr int t = (p - 4) / 4, k = (p - 4) % 4;
            const char* bp = b_lane + (slot * BN + 32 * t) * B_ROW;
            fb[set][t][k] = *reinterpret_cast<const bf16x8*>(bp + b_piece[k]);
        }
    };

    // ---- prologue: halo 0, weight steps 0 and 1 synchronously; steps 2, 3 in flight; frags(0) ------
    // (TAPS == 1: halo chunks 0 and 1 synchronously, chunks 2 and 3 in flight, like the weights)
    // (DMA path: weight steps 0, 1, 2 by DMA, all retired here; steps 3, 4, ... issued from the K loop)
    load_A(0, SET0{});
    if constexpr (TAPS == 1) load_A(1, SET1{});
    if constexpr (DMA_B) {
        dma_B(0, 0); dma_B(1, 1); dma_B(2, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<0, A_V4>([&](auto Q) { f32x4 t = ra[0][decltype(Q)::value]; asm volatile("" : "+v"(t)); ra[0][decltype(Q)::value] = t; });
        if constexpr (AFFINE) { f32x4 t0 = a_sc, t1 = a_sh; asm volatile("" : "+v"(t0), "+v"(t1)); a_sc = t0; a_sh = t1; }
        store_A(0, SET0{});
    } else {
        load_B(0, rb0);
        load_B(1, rb1);
        store_A(0, SET0{});
        if constexpr (TAPS == 1) store_A(1, SET1{});
        store_B(0, rb0);
        store_B(1, rb1);
        if constexpr (TAPS == 1) { load_A(2, SET0{}); load_A(3, SET1{}); }
        load_B(2, rb0);
        load_B(3, rb1);
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    fetch_A(std::integral_constant<int, 0>{}, 0, 0, 0);
    static_for<0, NT>([&](auto T) { fetch_B(std::integral_constant<int, 0>{}, T, 0); });

    // One K-step.  U = position inside a PAIR of chunks (0 .. 2*TAPS-1): tap, register-set parity,
    // ring slot arithmetic (the pair loop carries s mod 3 in `slot3`) and the halo load/store
    // points are compile-time, so the instruction stream of a pair is straight-line and hipcc
    // uses exact counted vmcnt waits (a conditional load inside the loop forces vmcnt(0) at the
    // loop header and drains the whole prefetch queue).
    //
    // Everything a step does besides its MFMAs is independent of them (they read the fragment set
    // fetched during the previous step), and the two waves of a SIMD leave the barrier together.
    // s_memtime stamps of the first version (barrier | ring store | loads | fetch | 6*NT MFMAs)
    // showed both waves spending ~450 cycles on staging before the first MFMA of a step, and
    // 600-950 cycles of pure VALU at the halo-split step, with the matrix pipe idle: 39 % MFMA
    // utilisation (PMC).  So the MFMAs go FIRST and the staging work is cut into pieces issued
    // between them (sched_barrier fences pin the interleave):
    //   M | fetch A(s+1) | M | fetch B(s+1) ... | M | B(s+2) regs -> slot (s+2)%3, issue B(s+4) |
    //   M | halo slice 0 | M | halo slice 1 | ...      (halo(c+1): loads at tap 0, split+store at tap TAPS-2)
    int slot3 = 0;            // s % 3
#ifdef PF_STAMPS
    unsigned long long tM[4] = {0, 0, 0, 0}, tR[2] = {0, 0};
#endif
#ifdef PF_NO_PIN
#define PF_PIN() do {} while (0)
#else
#define PF_PIN() __builtin_amdgcn_sched_barrier(0)
#endif
    auto step = [&](auto U, int chunk) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        constexpr int tap = u % TAPS;
        constexpr int cur = u & 1;                     // parity of s == parity of u (pairs start even)
        const int s = chunk * TAPS + tap;
        constexpr int ntap = (tap + 1) % TAPS;         // tap of step s+1
        constexpr int nky = ntap / KW, nkx = ntap % KW;
        const int nchunk = (tap == TAPS - 1) ? chunk + 1 : chunk;
        const int s1 = slot3 == B_SLOTS - 1 ? 0 : slot3 + 1;   // (s+1) % B_SLOTS
        const int s2 = s1 == B_SLOTS - 1 ? 0 : s1 + 1;         // (s+2) % B_SLOTS
        const int s3 = s2 == B_SLOTS - 1 ? 0 : s2 + 1;         // (s+3) % B_SLOTS   (DMA path)
        using NXT = std::integral_constant<int, cur ^ 1>;
#ifdef PF_STAMPS
        unsigned long long tA, tB;
        asm volatile("s_memtime %0" : "=s"(tA) :: "memory");          // everything of step s-1 issued; its LDS ops may be in flight
#endif
#ifndef PF_ABLATE_NO_BARRIER
        if constexpr (DMA_B) {
            // the weight DMA of step s+1 was issued at step s-2; VMEM operations issued after it, in order:
            // the halo loads (+ affine rows) of step s-2 / s-1 when that was a chunk's first tap, and the DMA
            // of step s+2.  Everything older must have landed before any wave reads slot (s+1).
            constexpr int A_OPS = A_V4 + (AFFINE ? 2 : 0);
            constexpr int YOUNGER = B_DMA + A_OPS * (((tap + TAPS - 1) % TAPS == 0 ? 1 : 0) + ((tap + TAPS - 2) % TAPS == 0 ? 1 : 0));
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(YOUNGER) : "memory");
        } else {
            __syncthreads();  // slot (s+1)%3 and the halo of step s+1 are complete; slot (s+2)%3 is idle
        }
#endif
#ifdef PF_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB), "+s"(tA) :: "memory");
        if (TH == 4 && blockIdx.x == 0 && lane == 0 && s < 40) {
            unsigned long long* st = reinterpret_cast<unsigned long long*>(smem + PF_STAMP_OFF);
            st[(wave * 40 + s) * 8 + 0] = tA;
            st[(wave * 40 + s) * 8 + 1] = tB;
            if (s > 0) {                                                                         // mid-step stamps of step s-1
                for (int k = 0; k < 4; ++k) st[(wave * 40 + s - 1) * 8 + 2 + k] = tM[k];
                st[(wave * 40 + s - 1) * 8 + 6] = tR[0];
                st[(wave * 40 + s - 1) * 8 + 7] = tR[1];
            }
        }
#endif
        constexpr int NM = 6 * NT;                     // MFMAs of the step; accumulators alternate
        auto halo_slice = [&](auto Q) __attribute__((always_inline)) {
            using CUR = std::integral_constant<int, TAPS == 1 ? cur : 0>;
            if constexpr (TAPS == 1) {
                // chunk == step: chunk s+2 goes from register set s&1 into halo buffer s&1 (last read by
                // the fragment fetch of step s-1), then the set is re-issued for chunk s+4
#ifndef PF_ABLATE_NO_LDS_WRITE
                store_A_q(cur, Q, CUR{});
#endif
#ifndef PF_ABLATE_NO_GLOBAL
                load_A_q(chunk + 4, Q, CUR{});
#endif
            } else {
#ifndef PF_ABLATE_NO_GLOBAL
                if constexpr (tap == 0) load_A_q(chunk + 1, Q, CUR{});
#endif
#ifndef PF_ABLATE_NO_LDS_WRITE
                if constexpr (tap == TAPS - 2) {
                    if constexpr (DMA_B) {
                        // slice q was loaded at the chunk's first tap; younger VMEM operations: the later slices'
                        // loads and the weight DMAs of the TAPS-2 steps since (this step's included)
                        constexpr int q = decltype(Q)::value;
                        f32x4 t = ra[0][q];
                        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(t) : "n"(A_V4 - 1 - q + B_DMA * (TAPS - 2)));
                        ra[0][q] = t;
                        if constexpr (AFFINE && q == 0) {
                            f32x4 t0 = a_sc, t1 = a_sh;
                            asm volatile("" : "+v"(t0), "+v"(t1));
                            a_sc = t0; a_sh = t1;
                        }
                    }
                    store_A_q((chunk + 1) & 1, Q, CUR{});
                }
#endif
            }
        };
        static_for<0, NM>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            constexpr int t = i % NT, j = i / NT, ks = j / 3, pass = j % 3;
#ifdef PF_ABLATE_NO_MFMA
            if constexpr (pass == 0)                  // keep the fragment reads alive, no matrix work
                asm volatile("" :: "v"(fa[cur][ks]), "v"(fa[cur][2 + ks]), "v"(fb[cur][t][ks]), "v"(fb[cur][t][2 + ks]));
#else
            if constexpr (pass == 0)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][2 + ks], fb[cur][t][ks], acc[t], 0, 0, 0);
            else if constexpr (pass == 1)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][ks], fb[cur][t][2 + ks], acc[t], 0, 0, 0);
            else
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][ks], fb[cur][t][ks], acc[t], 0, 0, 0);
#endif
            PF_PIN();
#ifndef PF_ABLATE_NO_FETCH
            // the 4 + 4*NT fragment reads of step s+1, spread over the first gaps (see fetch_piece)
            static_for<0, NP>([&](auto P) __attribute__((always_inline)) {
                if constexpr (decltype(P)::value * FETCH_GAPS / NP == i) fetch_piece(NXT{}, P, nchunk & 1, nky, nkx, s1);
            });
#endif
            if constexpr (i == NT + 1 && DMA_B) {
                dma_B(s + 3, s3);
#ifndef PF_ABLATE_NO_GLOBAL
                if constexpr (tap == 0 && TAPS > 1) load_A_affine(chunk + 1);
#endif
            }
            if constexpr (i == NT + 1 && !DMA_B) {
#if defined(PF_ABLATE_NO_B_STORE)           // timing-only: bound on what an LDS-DMA weight path could save
                if constexpr (cur == 0) asm volatile("" :: "v"(rb0[0]), "v"(rb0[B_V4 - 1]));
                else asm volatile("" :: "v"(rb1[0]), "v"(rb1[B_V4 - 1]));
#elif !defined(PF_ABLATE_NO_LDS_WRITE)
#ifdef PF_STAMPS
                if constexpr (NT == 2) asm volatile("s_memtime %0" : "=s"(tR[0]) :: "memory");     // ring piece starts
#endif
                if constexpr (cur == 0) store_B(s2, rb0); else store_B(s2, rb1);
#ifdef PF_STAMPS
                if constexpr (NT == 2) asm volatile("s_memtime %0" : "=s"(tR[1]) :: "memory");     // ... stored; now addresses + loads
#endif
#endif
#ifndef PF_ABLATE_NO_GLOBAL
                if constexpr (cur == 0) load_B(s + 4, rb0); else load_B(s + 4, rb1);
                if constexpr (tap == 0 && TAPS > 1) load_A_affine(chunk + 1);
#endif
            }
#ifdef PF_STAMPS
            if constexpr (NT == 2 && (i == 2 || i == 3 || i == 6 || i == 9))
                asm volatile("s_memtime %0" : "=s"(tM[i == 2 ? 0 : i == 3 ? 1 : i == 6 ? 2 : 3]) :: "memory");
#endif
            if constexpr (NT == 1) {                    // 6 MFMAs: half of the halo slices in each of two gaps
                if constexpr (i == 3) static_for<0, A_V4 / 2>([&](auto Q) { halo_slice(Q); });
                if constexpr (i == 4) static_for<A_V4 / 2, A_V4>([&](auto Q) { halo_slice(Q); });
            } else {
                static_assert(NT == 1 || NT + 2 + A_V4 <= 6 * NT, "halo slices fit between the MFMAs");
                if constexpr (i >= NT + 2 && i < NT + 2 + A_V4) halo_slice(std::integral_constant<int, i - NT - 2>{});
            }
            PF_PIN();
        });
        slot3 = s1;
    };
    int c2 = 0;
#ifdef PF_ABLATE_NO_LOOP                      // prologue + epilogue only: the fixed cost of a launch
    if (g.nchunks > 0) c2 = 1 << 20;
#endif
    for (; c2 + 1 < nchunks; c2 += 2)
        static_for<0, 2 * TAPS>([&](auto U) { step(U, c2 + decltype(U)::value / TAPS); });
    if ((nchunks & 1) && c2 < (1 << 20))
        static_for<0, TAPS>([&](auto U) { step(U, nchunks - 1); });

#ifdef PF_STAMPS
    __syncthreads();
    if (TH == 4 && blockIdx.x == 0) {
        const unsigned long long* st = reinterpret_cast<const unsigned long long*>(smem + PF_STAMP_OFF);
        for (int i = tid; i < 8 * 40 * 8; i += 512) pf_stamp_buf[i] = st[i];
    }
#endif
    // ---- epilogue -----------------------------------------------------------------------------
#ifndef PF_ABLATE_NO_EPILOGUE
    // ragged maps (W % 32 or H % TH != 0): the loader already zero-fills what lies outside the image;
    // here columns >= W and rows >= H of an edge tile are neither stored nor counted
    const bool ragged = (g.W % TW) != 0 || (g.H % TH) != 0;                       // uniform
    const bool row_ok = y0 + wy < g.H;
    const int xlim = row_ok ? g.W - x0 - 4 * lh : 0;                              // pixel r is live iff roff(r) < xlim
    const long p0 = pix0 + (long)(y0 + wy) * g.W + x0 + 4 * lh;
    if (ragged) tile_epilogue<NT, true>(d, acc, n0 + 32 * NT * wn, li, p0, p0 + (xlim > 0 ? xlim : 0));
    else tile_epilogue<NT, false>(d, acc, n0 + 32 * NT * wn, li, p0, 0);
    if (d.stats_out != nullptr) {
        // InstanceNorm statistics of this tile's outputs (the stored values, (acc + bias) * scale): fp64
        // sum / sum of squares per channel over the wave's 32 pixels, the wave's two halves by a
        // lane exchange, the workgroup's pixel rows through the (idle) operand LDS in a fixed order.
        __syncthreads();                            // every wave has left the K loop: LDS is free
        double* red = reinterpret_cast<double*>(smem);          // [8 waves][32*NT channels][2]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = n0 + 32 * NT * wn + 32 * t + li;
            const float bias = j < d.cout ? d.bias[j] : 0.f;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double v = ((r & 3) + 8 * (r >> 2) < xlim) ? (double)((acc[t][r] + bias) * d.scale) : 0.0;
                s1 += v; s2 += v * v;
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                red[((wave * (32 * NT)) + 32 * t + li) * 2 + 0] = s1;
                red[((wave * (32 * NT)) + 32 * t + li) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.cout) {
            const int cw = tid / (32 * NT), cl = tid % (32 * NT);      // channel half (wn) and slot inside it
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int y = 0; y < 8 / WN; ++y) {                           // the waves (pixel rows) of this half, in order
                const int wv = y * WN + cw;
                s1 += red[(wv * (32 * NT) + cl) * 2 + 0];
                s2 += red[(wv * (32 * NT) + cl) * 2 + 1];
            }
            // tile = image * tiles_per_image + tile_in_image: exactly the [image][nblk] order of the partials
            double* o = d.stats_out + ((long)tile * d.cout + n0 + tid) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
#else
    if (acc[0][0] == 123.456f) d.out[0] = acc[NT - 1][3];
#endif
}

template <int NT, int KH, int KW, bool AFFINE, int TH>
int launch_conv_halo_t(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, hipStream_t stream) {
    constexpr int BN = 32 * NT * (TH == 4 ? 2 : 1);
    constexpr int HALO_ROWS = ((TH + KH - 1) * (32 + KW - 1) + 63) / 64 * 64;
#ifdef PF_DMA_B
    constexpr bool DMA_B = KH * KW > 1 && TH == 4;
#else
    constexpr bool DMA_B = false;
#endif
    constexpr size_t lds = (size_t)2 * HALO_ROWS * LDS_LD * sizeof(float) +
                           (DMA_B ? (size_t)4 * BN * 128 : (size_t)3 * BN * LDS_LD * sizeof(float));
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int B = g.M / g.N;
    ConvGeom gg = g;
    gg.ntiles = B * ((g.H + TH - 1) / TH) * ((g.W + 31) / 32);
    gg.ntn = (max_cout + BN - 1) / BN;
    gg.xcd_map = 1;                       // XCD-aware 1-D grid (round 2: fabric-side fetches of a 5x1 GRU launch 89 -> 54 MB)
    const dim3 grid((unsigned)((long)gg.ntiles * gg.ntn * ngroups));
    // up to ~130 KB of dynamic LDS: above the 64 KB default limit
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&pf_conv_halo_kernel<NT, KH, KW, AFFINE, TH>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
#ifdef PF_STAMPS
    static_assert(TH == 8 || lds <= 140 * 1024, "stamp area");
    hipLaunchKernelGGL((pf_conv_halo_kernel<NT, KH, KW, AFFINE, TH>), grid, dim3(512), 160 * 1024, stream, grp, gg);
#else
    hipLaunchKernelGGL((pf_conv_halo_kernel<NT, KH, KW, AFFINE, TH>), grid, dim3(512), lds, stream, grp, gg);
#endif
    return (int)hipGetLastError();
}

template <int NT, int KH, int KW, int WN>
int launch_conv_ws_t(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, hipStream_t stream);

// Which form of the halo kernel a TH = 4 launch takes: 0 the symmetric pf_conv_halo_kernel, 1 pf_conv_ws_kernel with the
// call's own 128-px tile (WN = 2), 2 pf_conv_ws_kernel with the 256 px x 64 channel tile (WN = 1; only where that still
// gives every CU a workgroup).  PRIORFLOW_CONV_WS (A/B knob): 0 / 1 cap the choice, default 2.
int conv_ws_choice(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout) {
    static const int ws = [] { const char* e = getenv("PRIORFLOW_CONV_WS"); return e ? atoi(e) : 2; }();
    if (ws <= 0) return 0;
    const bool shape = (g.kh == 3 && g.kw == 3) || (g.kh == 1 && g.kw == 5) || (g.kh == 5 && g.kw == 1);
    if (!shape || g.stride != 1) return 0;
    for (int i = 0; i < ngroups; ++i)
        if (grp.d[i].stats_out != nullptr || grp.d[i].in_scale != nullptr) return 0;
    const long wgs256 = (long)(g.M / g.N) * ((g.H + 7) / 8) * ((g.W + 31) / 32) * ngroups * ((max_cout + 63) / 64);
    // (measured: the 256-px tile wins for the 3x3 convs, -2 % at B=1; for 1x5 / 5x1 its taller halo costs more than the weights save)
    if (ws >= 3 && g.kh == 3 && wgs256 >= 256) return 2;       // A/B: the 256 px x 64 channel roles kernel for Cout <= 64 as well
    return (ws >= 2 && g.kh == 3 && max_cout > 64 && wgs256 >= 256) ? 2 : 1;
}

template <int NT, int TH>
int launch_conv_halo(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, hipStream_t stream) {
    // every group of a launch must agree on having an input affine (one instantiation per launch)
    bool affine = grp.d[0].in_scale != nullptr;
    for (int i = 1; i < ngroups; ++i)
        if ((grp.d[i].in_scale != nullptr) != affine) return PF_ERR_BAD_ARG;
    if (affine) {                           // only the encoders' 3x3 convs use it
        if (g.kh == 3 && g.kw == 3) return launch_conv_halo_t<NT, 3, 3, true, TH>(grp, ngroups, g, max_cout, stream);
        return PF_ERR_BAD_SHAPE;
    }
    if constexpr (TH == 4) {                // role-specialised waves (pf_conv_ws_kernel); PRIORFLOW_CONV_WS=0: symmetric kernel
        const int ws = conv_ws_choice(grp, ngroups, g, max_cout);
        if (ws == 2) {
            if (g.kh == 3 && g.kw == 3) return pf_conv_ws256_launch(grp, ngroups, g, max_cout, stream);   // (3x3 only: conv_ws_choice)
        }
        if (ws == 1) {
            if (g.kh == 3 && g.kw == 3) return launch_conv_ws_t<NT, 3, 3, 2>(grp, ngroups, g, max_cout, stream);
            if (g.kh == 1 && g.kw == 5) return launch_conv_ws_t<NT, 1, 5, 2>(grp, ngroups, g, max_cout, stream);
            if (g.kh == 5 && g.kw == 1) return launch_conv_ws_t<NT, 5, 1, 2>(grp, ngroups, g, max_cout, stream);
        }
    }
    if (g.kh == 3 && g.kw == 3) return launch_conv_halo_t<NT, 3, 3, false, TH>(grp, ngroups, g, max_cout, stream);
    if (g.kh == 4 && g.kw == 4) return launch_conv_halo_t<NT, 4, 4, false, TH>(grp, ngroups, g, max_cout, stream);
    if constexpr (TH == 4) {                // the GRU's separable convs and the 1x1 convs only exist at 1/8 resolution
        if (g.kh == 1 && g.kw == 1) return launch_conv_halo_t<NT, 1, 1, false, TH>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 1 && g.kw == 5) return launch_conv_halo_t<NT, 1, 5, false, TH>(grp, ngroups, g, max_cout, stream);
        if (g.kh == 5 && g.kw == 1) return launch_conv_halo_t<NT, 5, 1, false, TH>(grp, ngroups, g, max_cout, stream);
    }
    return PF_ERR_BAD_SHAPE;
}

template <int WM, int WN, int NT>
int launch_conv(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, bool split,
                hipStream_t stream) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((max_cout + BN - 1) / BN), (unsigned)ngroups);
    if (split)
        hipLaunchKernelGGL((pf_conv_mfma_kernel<WM, WN, NT, true>), grid, dim3(256), lds, stream, grp, g);
    else
        hipLaunchKernelGGL((pf_conv_mfma_kernel<WM, WN, NT, false>), grid, dim3(256), lds, stream, grp, g);
    return (int)hipGetLastError();
}


// =====================================================================================================================
// Role-specialised form of the halo kernel for the update blocks' multi-tap convs (3x3, 1x5, 5x1; TH = 4, no input
// affine, no fused statistics).  Same tile (128 px x 64*NT channels), same LDS images, same ring protocol and ONE
// barrier per K-step as pf_conv_halo_kernel -- but the 8 waves no longer all do everything:
//   waves 0..3  MFMA waves: 64 px x 32*NT channels each (two pixel rows, acc[2][NT]); their stream is the step's
//               12*NT MFMAs with the 8 + 4*NT fragment reads of the next step in the gaps, nothing else;
//   waves 4..7  loader waves (256 threads): every global load, the halo's hi|lo split, every LDS write.
// Why: s_memtime stamps of the symmetric kernel (profiles/stamp_conv.py) showed that a wave's
// staging instructions (weight registers -> LDS, global loads, halo split) block its own in-order stream for
// 150-350 cycles per K-step and that its SIMD partner, running the same stream a little later, blocks on the same
// shared ports -- the matrix pipe idles for the SUM of both.  Here the partner of every MFMA wave is a loader wave,
// so staging runs beside the matrix pipe by construction (waves w and w + 4 share a SIMD).  The MFMA waves are the
// older half: VALU / LDS issue arbitration favours them.
// Per accumulator the MFMA order over (ks, pass) is the symmetric kernel's, so results are bit-identical.
// WN = 2: tile 128 px (4 rows) x 64*NT channels, MFMA waves 2 (row pairs) x 2 (channel halves);
// WN = 1: tile 256 px (8 rows) x 32*NT channels, MFMA waves 4 (row pairs) x 1 -- half the weight bytes per MFMA: the
//         per-CU stream out of L2 (weights + halo, ~18 B/clk at best) is what bounds the K-step.
template <int NT, int KH, int KW, int WN>
__global__ void __launch_bounds__(512, 2)
pf_conv_ws_kernel(const ConvGroups groups, const ConvGeom g) {
    static_assert(WN == 1 || WN == 2, "");
    constexpr int TH = 8 / WN, TW = 32, BN = 32 * NT * WN, TAPS = KH * KW;
    static_assert(TAPS >= 3, "multi-tap convs only (the 1x1 cadence lives in pf_conv_halo_kernel)");
    constexpr int HW = TW + KW - 1, HH = TH + KH - 1;
    constexpr int LROWS = 32;                               // rows per loader pass: 256 loader threads, 8 x 16 B per row
    constexpr int A_V4 = (HH * HW + LROWS - 1) / LROWS;     // halo float4 per loader thread and chunk
    constexpr int HALO_ROWS = LROWS * A_V4;
    static_assert(BN % LROWS == 0, "");
    constexpr int B_V4 = BN / LROWS;                        // weight float4 per loader thread and K-step
    constexpr int B_ROW = LDS_LD * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ah = smem;                                       // [2][HALO_ROWS][LDS_LD]
    float* Bs = smem + 2 * HALO_ROWS * LDS_LD;              // [3][BN][LDS_LD]  (ring)

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
    const long wrow = (long)TAPS * g.cin_pad;
    char* const ah_bytes = reinterpret_cast<char*>(Ah);
    char* const bs_bytes = reinterpret_cast<char*>(Bs);
    int slot3 = 0;                                          // s % 3, carried by both roles alike

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int ltid = tid - 256;
        const int c4 = (ltid & 7) * 4;
        const int ctot = d.c0 + d.c1;
        // per-slice source rows: byte offsets of the pixel row in either input segment, or "outside"
        long a_off0[A_V4], a_off1[A_V4];
        unsigned a_loff[A_V4], a_in = 0;
#pragma unroll
        for (int q = 0; q < A_V4; ++q) {
            const int r = (ltid + 256 * q) >> 3;
            const int yy = y0 + r / HW - ph, xx = x0 + r % HW - pw;
            const bool in = r < HH * HW && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            const long pix = in ? pix0 + (long)yy * g.W + xx : 0;
            a_off0[q] = pix * d.ld0 * 4;
            a_off1[q] = pix * d.ld1 * 4;
            a_in |= in ? (1u << q) : 0u;
            a_loff[q] = (unsigned)((r * LDS_LD) * 4 + 2 * c4);
        }
        unsigned b_goff[B_V4], b_loff[B_V4];
#pragma unroll
        for (int q = 0; q < B_V4; ++q) {
            const int r = (ltid + 256 * q) >> 3;
            b_goff[q] = (unsigned)(((long)(n0 + r) * wrow + c4) * 4);
            b_loff[q] = (unsigned)((r * LDS_LD + c4) * 4);
        }
        const char* const wbytes = reinterpret_cast<const char*>(d.weight);
        // Weights: FOUR register sets -- tile k lives in set k % 4, is loaded at step k-6 and stored into ring slot k % 3 at
        // step k-2, so four K-steps of weights (64 KB per CU at BN = 128) are in flight: the stamps of the two-set form showed
        // the ring store waiting ~400 cycles for data requested two steps (~2 300 cycles) earlier.
        // Halo: the slices of a chunk form G groups; group k is loaded at tap k and split + stored at tap G + k, so no step
        // carries more than 1/G of the halo work (all of it at tap 0 / tap TAPS-2 made those two steps 1.3x / 1.6x as long).
        constexpr int G = (TAPS - 1) / 2;
        static_assert(2 * G <= TAPS - 1, "the last group is stored before the chunk's last tap");
        f32x4 ra[A_V4], rb[4][B_V4];
        bool a_chunk_ok = false;                            // this thread's 4 channels of the chunk being loaded exist
        const char* a_base = nullptr;
        bool a_seg0 = true;
        auto begin_A = [&](int chunk) __attribute__((always_inline)) {
            if (chunk >= nchunks) chunk = nchunks - 1;      // tail: harmless re-read, stored to the idle buffer
            const int c = chunk * KC + c4;
            a_seg0 = c < d.c0;                              // a chunk lies in ONE segment (c0 % 32 == 0 when c1 > 0)
            a_base = a_seg0 ? reinterpret_cast<const char*>(d.in0 + d.off0 + c)
                            : reinterpret_cast<const char*>(d.in1 + d.off1 + (c - d.c0));
            a_chunk_ok = c < ctot;
            if (!a_chunk_ok) a_base = reinterpret_cast<const char*>(d.in0 + d.off0);
        };
        auto load_A = [&](auto K) __attribute__((always_inline)) {                  // group K of the chunk begun last
            constexpr int k = decltype(K)::value;
            static_for<k * A_V4 / G, (k + 1) * A_V4 / G>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                const long off = a_chunk_ok ? (a_seg0 ? a_off0[q] : a_off1[q]) : 0;
                ra[q] = *reinterpret_cast<const f32x4*>(a_base + off);
            });
        };
        auto store_A = [&](auto K, int buf) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            char* hb = ah_bytes + buf * (HALO_ROWS * LDS_LD * 4);
            static_for<k * A_V4 / G, (k + 1) * A_V4 / G>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                const bool ok = a_chunk_ok && ((a_in >> q) & 1u);
                const f32x4 v = ok ? ra[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);          // hi = bf16(v) (RNE); lo = bf16(v - hi)
                const f32x4 rest = v - __builtin_convertvector(hi, f32x4);
                const bf16x4 lo = __builtin_convertvector(rest, bf16x4);
                *reinterpret_cast<bf16x4*>(hb + a_loff[q]) = hi;
                *reinterpret_cast<bf16x4*>(hb + a_loff[q] + 64) = lo;
            });
        };
        auto load_B = [&](int chunk, int tap, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            if (chunk >= nchunks) chunk = nchunks - 1;      // tail: harmless re-read
            const char* wp = wbytes + ((long)tap * g.cin_pad + chunk * KC) * 4;      // wave-uniform
#pragma unroll
            for (int q = 0; q < B_V4; ++q) rb[set][q] = *reinterpret_cast<const f32x4*>(wp + b_goff[q]);
        };
        auto store_B = [&](int slot, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            char* bs = bs_bytes + slot * (BN * B_ROW);
#pragma unroll
            for (int q = 0; q < B_V4; ++q) *reinterpret_cast<f32x4*>(bs + b_loff[q]) = rb[set][q];
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
        // prologue: halo 0 and weight tiles 0, 1 synchronously; tiles 2..5 in flight
        begin_A(0);
        static_for<0, G>([&](auto K) { load_A(K); });
        load_B(0, 0, S0{});
        load_B(1 / TAPS, 1 % TAPS, S1{});
        static_for<0, G>([&](auto K) { store_A(K, 0); });
        store_B(0, S0{});
        store_B(1, S1{});
        load_B(2 / TAPS, 2 % TAPS, S2{});
        load_B(3 / TAPS, 3 % TAPS, S3{});
        load_B(4 / TAPS, 4 % TAPS, S0{});
        load_B(5 / TAPS, 5 % TAPS, S1{});
        __syncthreads();
#ifdef PF_STAMPS
        unsigned long long tL[2] = {0, 0};
#endif
        // U = position inside a group of FOUR chunks (the set index (s + 2) % 4 must be compile-time: 4 * TAPS steps)
        auto lstep = [&](auto U, int chunk) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            constexpr int tap = u % TAPS;
            using SET = std::integral_constant<int, (u + 2) % 4>;
            const int s1 = slot3 == 2 ? 0 : slot3 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
#ifdef PF_STAMPS
            unsigned long long tA, tB;
            asm volatile("s_memtime %0" : "=s"(tA) :: "memory");
#endif
            __syncthreads();          // the MFMA waves are done with slot (s+2)%3 and with the other halo buffer
#ifdef PF_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB), "+s"(tA) :: "memory");
            { const int s = chunk * TAPS + tap;
              if (blockIdx.x == 0 && lane == 0 && s < 40) {
                unsigned long long* st = reinterpret_cast<unsigned long long*>(smem + PF_STAMP_OFF);
                st[(wave * 40 + s) * 8 + 0] = tA; st[(wave * 40 + s) * 8 + 1] = tB;
                if (s > 0) { st[(wave * 40 + s - 1) * 8 + 2] = tL[0]; st[(wave * 40 + s - 1) * 8 + 3] = tL[1]; } } }
#endif
            store_B(s2, SET{});                                                          // tile s+2 (loaded at step s-4)
#ifdef PF_STAMPS
            asm volatile("s_memtime %0" : "=s"(tL[0]) :: "memory");
#endif
            constexpr int t6 = (tap + 6) % TAPS, c6 = (tap + 6) / TAPS;
            load_B(chunk + c6, t6, SET{});                                               // tile s+6
#ifdef PF_STAMPS
            asm volatile("s_memtime %0" : "=s"(tL[1]) :: "memory");
#endif
            if constexpr (tap == 0) begin_A(chunk + 1);
            if constexpr (tap < G) load_A(std::integral_constant<int, tap>{});
            if constexpr (tap >= G && tap < 2 * G) store_A(std::integral_constant<int, tap - G>{}, (chunk + 1) & 1);
            slot3 = s1;
        };
        int c4i = 0;
        for (; c4i + 3 < nchunks; c4i += 4)
            static_for<0, 4 * TAPS>([&](auto U) { lstep(U, c4i + decltype(U)::value / TAPS); });
        if (c4i < nchunks) static_for<0, TAPS>([&](auto U) { lstep(U, c4i); });
        if (c4i + 1 < nchunks) static_for<TAPS, 2 * TAPS>([&](auto U) { lstep(U, c4i + 1); });
        if (c4i + 2 < nchunks) static_for<2 * TAPS, 3 * TAPS>([&](auto U) { lstep(U, c4i + 2); });
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
    const char* a_lane = ah_bytes + ((2 * wy2) * HW + li) * (LDS_LD * 4) + 32 * lh;
    const char* b_lane = bs_bytes + (32 * NT * wn + li) * B_ROW + 32 * lh;
    // fragments, double buffered in registers: [set][...][piece]; pieces 0,1 = hi K-halves, 2,3 = lo K-halves
    bf16x8 fa[2][2][4], fb[2][NT][4];
    constexpr int NP = 8 + 4 * NT;                          // fragment reads per step
    constexpr int NM = 12 * NT;                             // MFMAs per step
    constexpr int FETCH_GAPS = NM - 2;
    auto fetch_piece = [&](auto SET, auto P, int halo_buf, int ky, int kx, int slot) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value, p = decltype(P)::value;
        if constexpr (p < 8) {
            constexpr int m = p / 4, k = p % 4;
            const char* ap = a_lane + (halo_buf * HALO_ROWS + (ky + m) * HW + kx) * (LDS_LD * 4);
            fa[set][m][k] = *reinterpret_cast<const bf16x8*>(ap + (k & 1) * 16 + (k >> 1) * 64);
        } else {
            constexpr int t = (p - 8) / 4, k = (p - 8) % 4;
            const char* bp = b_lane + (slot * BN + 32 * t) * B_ROW;
            fb[set][t][k] = *reinterpret_cast<const bf16x8*>(bp + (k & 1) * 16 + (k >> 1) * 64);
        }
    };
    __syncthreads();                                        // prologue of the loader waves is in LDS
    static_for<0, NP>([&](auto P) { fetch_piece(std::integral_constant<int, 0>{}, P, 0, 0, 0, 0); });
    auto mstep = [&](auto U, int chunk) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        constexpr int tap = u % TAPS, cur = u & 1;
        constexpr int ntap = (tap + 1) % TAPS, nky = ntap / KW, nkx = ntap % KW;
        const int nchunk = (tap == TAPS - 1) ? chunk + 1 : chunk;
        const int s1 = slot3 == 2 ? 0 : slot3 + 1;
        using NXT = std::integral_constant<int, cur ^ 1>;
#ifdef PF_STAMPS
        unsigned long long tA, tB;
        asm volatile("s_memtime %0" : "=s"(tA) :: "memory");
#endif
        __syncthreads();              // slot (s+1)%3 and the halo of step s+1 are complete; fragment set `cur` has landed
#ifdef PF_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB), "+s"(tA) :: "memory");
        { const int s = chunk * TAPS + tap;
          if (blockIdx.x == 0 && lane == 0 && s < 40) {
            unsigned long long* st = reinterpret_cast<unsigned long long*>(smem + PF_STAMP_OFF);
            st[(wave * 40 + s) * 8 + 0] = tA; st[(wave * 40 + s) * 8 + 1] = tB; } }
#endif
        static_for<0, NM>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            constexpr int idx = i % (2 * NT), m = idx / NT, t = idx % NT, j = i / (2 * NT), ks = j / 3, pass = j % 3;
            if constexpr (pass == 0)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][2 + ks], fb[cur][t][ks], acc[m][t], 0, 0, 0);
            else if constexpr (pass == 1)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][ks], fb[cur][t][2 + ks], acc[m][t], 0, 0, 0);
            else
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m][ks], fb[cur][t][ks], acc[m][t], 0, 0, 0);
            PF_PIN();
            static_for<0, NP>([&](auto P) __attribute__((always_inline)) {
                if constexpr (decltype(P)::value * FETCH_GAPS / NP == i) fetch_piece(NXT{}, P, nchunk & 1, nky, nkx, s1);
            });
            PF_PIN();
        });
        slot3 = s1;
    };
    {
        int c2 = 0;
        for (; c2 + 1 < nchunks; c2 += 2)
            static_for<0, 2 * TAPS>([&](auto U) { mstep(U, c2 + decltype(U)::value / TAPS); });
        if (nchunks & 1)
            static_for<0, TAPS>([&](auto U) { mstep(U, nchunks - 1); });
    }
#ifdef PF_STAMPS
    if (blockIdx.x == 0) {            // (the loader waves wrote their stamps before their last barrier arrival... give them time)
        __builtin_amdgcn_s_sleep(127);
        const unsigned long long* st = reinterpret_cast<const unsigned long long*>(smem + PF_STAMP_OFF);
        for (int i = tid; i < 8 * 40 * 8; i += 256) pf_stamp_buf[i] = st[i];
    }
#endif
    const bool ragged = (g.W % TW) != 0 || (g.H % TH) != 0;
    static_for<0, 2>([&](auto M) __attribute__((always_inline)) {      // (a runtime-indexed acc[m] would put the accumulators in scratch)
        constexpr int m = decltype(M)::value;
        const int wy = 2 * wy2 + m;
        const bool row_ok = y0 + wy < g.H;
        const int xlim = row_ok ? g.W - x0 - 4 * lh : 0;
        const long p0 = pix0 + (long)(y0 + wy) * g.W + x0 + 4 * lh;
        if (ragged) tile_epilogue<NT, true>(d, acc[m], n0 + 32 * NT * wn, li, p0, p0 + (xlim > 0 ? xlim : 0));
        else tile_epilogue<NT, false>(d, acc[m], n0 + 32 * NT * wn, li, p0, 0);
    });
}

template <int NT, int KH, int KW, int WN>
int launch_conv_ws_t(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, hipStream_t stream) {
    constexpr int BN = 32 * NT * WN, TH = 8 / WN;
    constexpr int HALO_ROWS = ((TH + KH - 1) * (32 + KW - 1) + 31) / 32 * 32;
    constexpr size_t lds = (size_t)2 * HALO_ROWS * LDS_LD * sizeof(float) + (size_t)3 * BN * LDS_LD * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int B = g.M / g.N;
    ConvGeom gg = g;
    gg.ntiles = B * ((g.H + TH - 1) / TH) * ((g.W + 31) / 32);
    gg.ntn = (max_cout + BN - 1) / BN;
    gg.xcd_map = 1;
    const dim3 grid((unsigned)((long)gg.ntiles * gg.ntn * ngroups));
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&pf_conv_ws_kernel<NT, KH, KW, WN>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
#ifdef PF_STAMPS
    static_assert(lds <= 140 * 1024, "stamp area");
    hipLaunchKernelGGL((pf_conv_ws_kernel<NT, KH, KW, WN>), grid, dim3(512), 160 * 1024, stream, grp, gg);
#else
    hipLaunchKernelGGL((pf_conv_ws_kernel<NT, KH, KW, WN>), grid, dim3(512), lds, stream, grp, gg);
#endif
    return (int)hipGetLastError();
}

}  // namespace

// The product build compiles this file as five translation units in parallel (-DPF_CONV_PART=0..4, __graft_entry__.py):
// each instantiates only the kernels its launcher names.  Without the macro (diagnostic builds: profiles/stamp_conv.py,
// profiles/ablate_conv.sh) everything is one unit.
#ifndef PF_CONV_PART
#define PF_CONV_PART (-1)
#endif
#define PF_PART(k) (PF_CONV_PART == -1 || PF_CONV_PART == (k))

#if PF_PART(4)      // tile 8: halo kernel 256 px x 96 channels, 3x3 (round 6: the encoders' layer 2)
int pf_conv_part4_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t s) {
    if (g.kh != 3 || g.kw != 3) return PF_ERR_BAD_SHAPE;             // (the 4x4 halo of this tile would not fit the LDS)
    const bool affine = grp.d[0].in_scale != nullptr;
    for (int i = 1; i < ngroups; ++i)
        if ((grp.d[i].in_scale != nullptr) != affine) return PF_ERR_BAD_ARG;
    return affine ? launch_conv_halo_t<3, 3, 3, true, 8>(grp, ngroups, g, max_cout, s)
                  : launch_conv_halo_t<3, 3, 3, false, 8>(grp, ngroups, g, max_cout, s);
}
#endif
#if PF_PART(0)      // generic kernel (stride 2, exact fp32, small problems); the wave-organisation rule
int pf_conv_ws_choice(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout) {
    return conv_ws_choice(grp, ngroups, g, max_cout);
}
int pf_conv_part0_launch(int tile_id, const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout,
                         bool split, hipStream_t s) {
    switch (tile_id) {
        case 0: return launch_conv<4, 1, 1>(grp, ngroups, g, max_cout, split, s);
        case 1: return launch_conv<2, 2, 1>(grp, ngroups, g, max_cout, split, s);
        case 7: return launch_conv<4, 1, 3>(grp, ngroups, g, max_cout, split, s);
        default: return launch_conv<2, 2, 2>(grp, ngroups, g, max_cout, split, s);
    }
}
#endif
#if PF_PART(1)      // tile 3: halo / role-specialised kernels, 128 px x 64 channels
int pf_conv_part1_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t s) {
    return launch_conv_halo<1, 4>(grp, ngroups, g, max_cout, s);
}
#endif
#if PF_PART(2)      // tile 4: 128 px x 128 channels
int pf_conv_part2_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t s) {
    return launch_conv_halo<2, 4>(grp, ngroups, g, max_cout, s);
}
#endif
#if PF_PART(3)      // tile 5: 8-row halo kernel; the role-specialised 256 px x 64 channel tile of the 3x3 convs
int pf_conv_part3_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t s) {
    return launch_conv_halo<2, 8>(grp, ngroups, g, max_cout, s);
}
int pf_conv_ws256_launch(const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout, hipStream_t s) {
    return launch_conv_ws_t<2, 3, 3, 1>(grp, ngroups, g, max_cout, s);
}
#endif

#if PF_CONV_PART == -1
int pf_conv_kernels_launch(int tile_id, const pfconv::ConvGroups& grp, int ngroups, const pfconv::ConvGeom& g, int max_cout,
                           bool split, hipStream_t s) {
    switch (tile_id) {
        case 0: case 1: case 2: case 7: return pf_conv_part0_launch(tile_id, grp, ngroups, g, max_cout, split, s);
        case 3: return pf_conv_part1_launch(grp, ngroups, g, max_cout, s);
        case 4: return pf_conv_part2_launch(grp, ngroups, g, max_cout, s);
        case 8: return pf_conv_part4_launch(grp, ngroups, g, max_cout, s);
        default: return pf_conv_part3_launch(grp, ngroups, g, max_cout, s);
    }
}
#endif

#if defined(PF_STAMPS) && PF_CONV_PART == -1
extern "C" int pf_conv_read_stamps(unsigned long long* out) {   // [8 waves][40 steps][before barrier, after barrier, 4 mid-step stamps, 2 ring-piece stamps]
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_stamp_buf), sizeof(unsigned long long) * 8 * 40 * 8);
}
#endif
