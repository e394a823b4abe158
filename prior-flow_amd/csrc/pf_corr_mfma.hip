// pf_corr_pyramid: all-pairs correlation volume with the 4-level pyramid fused into the GEMM
// epilogue (PriOr-RAFT/core/prior_raft.py:69-75 `corr`, core/corr.py:99-111 `build_pyramid`).
//
//   level0[b][n1][n2] = (1/sqrt(C)) * sum_c f1[b][n1][c] * f2[b][n2][c]
//   level(i+1)        = 2x2 mean of level i over (y2, x2)
//
// The reference writes the volume with a matmul, re-reads and re-writes it for the divide, then
// re-reads it three more times for the avg_pool2d chain.  Here every level is written exactly
// once and nothing is re-read: algorithmic HBM traffic = 4*N^2*(85/64) + 2*4*N*C bytes.
//
// Tiling (fast path, W8 % 32 == 0, H8 % 8 == 0): one workgroup = 4 waves = 128 query pixels n1
// x one 8-row x 32-column block of target pixels n2 (256 columns).  A wave owns 32 n1 rows and
// ALL 256 n2 columns as eight 32x32 MFMA accumulators (one per target row), so
//   * 2x2 pooling along y2 is a register-to-register add between accumulators t and t+1,
//   * pooling along x2 is a lane shuffle (lane = x2 inside the block),
//   * an 8x32 block is closed under three poolings -> levels 1..3 come out of registers,
//   * level-0 stores are 128-byte row segments (32 lanes x 4 B), two rows per instruction.
// Exact fp32: v_mfma_f32_32x32x2_f32 with the same K permutation / LDS staging as
// pf_conv_mfma.hip (16-byte coalesced fills, ds_read_b128 operands, 144-byte padded rows).
//
// Generic path (any H8, W8 >= 16): same GEMM on 256 consecutive n2 columns, level 0 only; levels
// 1..3 then come from a small pooling kernel (odd sizes drop the last row / column like
// F.avg_pool2d(2, stride=2), core/corr.py:108).
#include <stdlib.h>
#include <type_traits>
#include "pf_common.h"
#include "../../include/priorflow_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: plain 16-byte loads, no struct memcpy
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32;
constexpr int LDS_LD = 36;
constexpr int BM = 128;     // n1 per workgroup
constexpr int BN = 256;     // n2 per workgroup

struct CorrArgs {
    const float* f1; const float* f2;
    float* lvl[4];
    int B, H, W, N, C;
    float inv_scale;      // sqrt(C): level0 = acc / inv_scale
    float scale_mul;      // 1/sqrt(C) when that is exact (sqrt(C) a power of two: C = 256), else 0 -> true division
    int tiles_x;          // W/32 (fused path)
    int n2_tiles;         // number of n2 tiles per batch element
    int m_tiles;          // number of 128-row n1 tiles per batch element
};

// value of lane (lane ^ 1), (lane ^ 2) [quad permutes] or (lane + 4) [row shift] without the LDS
// crossbar: the pooling partners of the consuming lanes (x2 even / multiple of 4 / of 8) always
// sit in the same 16-lane DPP row.  The first version used __shfl_down (ds_bpermute + lgkmcnt wait,
// 224 per wave) and was VALU/latency bound in its store phase.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#ifdef PF_ABLATE_NO_POOL_STORE
constexpr bool PF_POOL_STORE = false;
#else
constexpr bool PF_POOL_STORE = true;
#endif
constexpr int DPP_XOR1 = 0xB1;      // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;      // quad_perm [2,3,0,1]
constexpr int DPP_SHL4 = 0x104;     // row_shl:4  (lane i <- lane i+4)

// SPLIT: operands are pre-split bf16 hi|lo rows (pf_split_bf16; same row stride / chunk offsets in
// bytes as the fp32 rows, so staging is the same 16-byte copy) and the products run as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (3/16 of the exact-fp32 MFMA time).
//
// Occupancy: the kernel is a short GEMM (K = C = 256: 8 K-steps) followed by a long store phase
// (128 x 256 outputs + pooled levels = 170 KB per workgroup, ~as long as the GEMM).  With
// double-buffered operand tiles (110 KB of LDS) only one workgroup fitted a CU, so all 256 CUs
// alternated in lock-step between an MFMA phase with HBM idle and a store phase with the matrix
// cores idle (223 us per 373 MB launch = 21 % of the HBM peak).  The tiles are therefore SINGLE
// buffered (55 KB, two barriers per K-step): two workgroups share a CU and one's store phase
// runs under the other's GEMM; the second workgroup also fills the extra barrier's bubble.
template <bool FUSED_POOL, bool SPLIT>
__global__ void __launch_bounds__(256, 2)
pf_corr_kernel(const CorrArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];     // operand tiles; re-used by the epilogue
    float* const As = smem;                       // [BM][LDS_LD]
    float* const Bs = smem + BM * LDS_LD;         // [BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware work mapping.  Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its
    // own 4 MB L2; every XCD gets a contiguous range of the work sequence
    //     q = ((band * m_tiles + m) * tiles_x + tx)        (band = batch x 8-row band of n2 tiles)
    // i.e. ONE band of target tiles (tiles_x * 256 rows of f2 = 1 MB, L2 resident) against all query
    // tiles m, the tiles_x neighbours of a band back to back (they share the f1 tile and complete each
    // other's pooled-level lines).  The first mapping (n2 tile fastest over ALL tiles, then m) made
    // every XCD stream the whole of f2 (8 MB > L2) once per pair of query tiles: the PMC pass read
    // 383 MB of fabric-side fetches per launch against 16.8 MB of inputs.
    int b, m0, tile;
    {
        const unsigned nwg = gridDim.x, orig = blockIdx.x;
        const unsigned xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;     // bijective for any nwg
        const unsigned q = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
        const unsigned tx_n = FUSED_POOL ? (unsigned)a.tiles_x : 1u;    // generic path: one "band" per n2 tile
        const unsigned bands = (unsigned)a.n2_tiles / tx_n;
        const unsigned tx = q % tx_n, r = q / tx_n;
        const unsigned bb = r / (unsigned)a.m_tiles;                    // batch * bands + band
        m0 = (int)(r % (unsigned)a.m_tiles) * BM;
        b = (int)(bb / bands);
        tile = (int)((bb % bands) * tx_n + tx);
    }
    // n2 of local column col (0..255)
    int ty0 = 0, tx0 = 0;
    if (FUSED_POOL) { ty0 = (tile / a.tiles_x) * 8; tx0 = (tile % a.tiles_x) * 32; }
    auto n2_of = [&](int col) -> int {
        if (FUSED_POOL) return (ty0 + (col >> 5)) * a.W + tx0 + (col & 31);
        return tile * BN + col;
    };

    const float* f1b = a.f1 + (long)b * a.N * a.C;
    const float* f2b = a.f2 + (long)b * a.N * a.C;
    const int c4 = (tid & 7) * 4;
    constexpr int A_V4 = BM * 8 / 256;   // 4
    constexpr int B_V4 = BN * 8 / 256;   // 8
    int a_src[A_V4], b_src[B_V4];        // row offsets (floats; N*C < 2^31 is checked by the launcher) or -1
#pragma unroll
    for (int q = 0; q < A_V4; ++q) {
        const int r = (tid + 256 * q) >> 3;
        a_src[q] = (m0 + r < a.N) ? (m0 + r) * a.C : -1;
    }
#pragma unroll
    for (int q = 0; q < B_V4; ++q) {
        const int r = (tid + 256 * q) >> 3;
        const int n2 = n2_of(r);
        b_src[q] = (n2 < a.N) ? n2 * a.C : -1;
    }
    f32x4 ra[A_V4], rb[B_V4];
    auto load_step = [&](int step) __attribute__((always_inline)) {
        const int c = step * KC + c4;
#pragma unroll
        for (int q = 0; q < A_V4; ++q) {     // branch-free: out-of-range rows read row 0 and are zeroed at LDS-store time
            ra[q] = *reinterpret_cast<const f32x4*>(f1b + (a_src[q] >= 0 ? a_src[q] : 0) + c);
        }
#pragma unroll
        for (int q = 0; q < B_V4; ++q) {
            rb[q] = *reinterpret_cast<const f32x4*>(f2b + (b_src[q] >= 0 ? b_src[q] : 0) + c);
        }
    };
    auto store_step = [&](int) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < A_V4; ++q)
            *reinterpret_cast<f32x4*>(&As[((tid + 256 * q) >> 3) * LDS_LD + c4]) =
                (FUSED_POOL || a_src[q] >= 0) ? ra[q] : f32x4{0.f, 0.f, 0.f, 0.f};     // fused path: every row exists
#pragma unroll
        for (int q = 0; q < B_V4; ++q)
            *reinterpret_cast<f32x4*>(&Bs[((tid + 256 * q) >> 3) * LDS_LD + c4]) =
                (FUSED_POOL || b_src[q] >= 0) ? rb[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#ifdef PF_ABLATE_NO_GEMM                // timing-only builds (profiles/): store phase alone / GEMM alone
    const int nsteps = a.C > (1 << 20) ? a.C / KC : 0;
#else
    const int nsteps = a.C / KC;
#endif
    const int a_off = (32 * wave + li) * LDS_LD + 16 * lh;
    const int b_off = li * LDS_LD + 16 * lh;

    load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        constexpr int buf = 0;
        if (step) __syncthreads();       // every wave has read the previous step's fragments
        store_step(buf);
        __syncthreads();
        // unconditional prefetch (the last one re-reads the final K-step; never stored)
        load_step(step + 1 < nsteps ? step + 1 : step);
        asm volatile("" ::: "memory");   // keep the prefetch above the MFMA block (see pf_conv_mfma.hip)
        if constexpr (SPLIT) {
            // lane (row li, half lh) owns channels [16lh,16lh+16): bytes [32lh,32lh+32) of hi and of lo (+64)
            const char* ap = reinterpret_cast<const char*>(&As[a_off]) - 32 * lh;
            bf16x8 fa[4];
            fa[0] = *reinterpret_cast<const bf16x8*>(ap);
            fa[1] = *reinterpret_cast<const bf16x8*>(ap + 16);
            fa[2] = *reinterpret_cast<const bf16x8*>(ap + 64);
            fa[3] = *reinterpret_cast<const bf16x8*>(ap + 80);
            const char* bp0 = reinterpret_cast<const char*>(&Bs[b_off]) - 32 * lh;
            // (no register double buffer for the B fragments: the second workgroup of the CU hides the
            // LDS round trip, and 16 more registers spilled inside this loop)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const char* bp = bp0 + t * 32 * LDS_LD * 4;
                bf16x8 fb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    fb[q] = *reinterpret_cast<const bf16x8*>(bp + (q >> 1) * 64 + (q & 1) * 16);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 + ks], fb[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[2 + ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks], acc[t], 0, 0, 0);
                }
            }
        } else {
            f32x4 af[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(&As[a_off + 4 * q]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x4 bf[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bf[q] = *reinterpret_cast<const f32x4*>(&Bs[b_off + t * 32 * LDS_LD + 4 * q]);
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

    // ---- epilogue -----------------------------------------------------------------------------
    // acc[t][r]: n1 = m0 + 32*wave + (r&3) + 8*(r>>2) + 4*lh ; n2 column = 32*t + li
    const long N = a.N;
    float* l0 = a.lvl[0] + (long)b * N * N;
#ifdef PF_ABLATE_NO_STORE
    if (acc[0][0] != 123.456f) return;
#endif
    if (!FUSED_POOL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n1 = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n1 >= a.N) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int n2 = tile * BN + 32 * t + li;
                if (n2 < a.N) l0[(long)n1 * N + n2] = acc[t][r] / a.inv_scale;
            }
        }
        return;
    }
    // Fused-pool epilogue.  The accumulator layout puts the target column on the lane and the query
    // row in the register, so a direct store is 4 bytes per lane: 240 store instructions per wave,
    // and the store phase was bound by their number (~20 cycles each per CU; timing-only builds:
    // 82 us of stores, 46 us of them for the pooled levels' 24 % of the bytes).  Each wave therefore
    // transposes its results through the (now idle) operand LDS: rows of the staging image are
    // query rows, so a lane reads 16 contiguous bytes and a wave-store writes whole row segments --
    // 32 + 8 + 2 + 1 = 43 store instructions per wave.  LDS traffic of one wave is processed in
    // issue order, so the write -> read -> overwrite sequence on a wave-private region needs no barrier.
    const int W1 = a.W >> 1, W2 = a.W >> 2, W3 = a.W >> 3;
    const long N1 = N >> 2, N2 = N >> 4, N3 = N >> 6;
    __syncthreads();                               // every wave is done with the operand tiles
    float* const stage = smem + wave * ((BM + BN) * LDS_LD / 4);      // 3456 floats per wave
    const long row0 = (long)b * N + m0 + 32 * wave;                   // first query row of this wave (batch folded in)
    auto epilogue = [&](auto MUL) __attribute__((always_inline)) {
        auto scaled = [&](float x) { return decltype(MUL)::value ? x * a.scale_mul : x / a.inv_scale; };
        // ---- level 0: one target row (t) at a time, two alternating 32 x 36 staging images --------
        constexpr int S0 = 36;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float* st = stage + (t & 1) * (32 * S0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[((r & 3) + 8 * (r >> 2) + 4 * lh) * S0 + li] = scaled(acc[t][r]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (lane >> 3) + 8 * k, piece = lane & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * S0 + 4 * piece);
                *reinterpret_cast<f32x4*>(a.lvl[0] + (row0 + row) * N + (long)(ty0 + t) * a.W + tx0 + 4 * piece) = v;
            }
        }
        // ---- pooled levels: DPP pooling in registers, staged per query row -------------------------
        constexpr int S1 = 68, S2 = 20, S3 = 4;                       // padded row strides (floats)
        float* const st1 = stage;                                     // [32][4 rows x 16 cols]
        float* const st2 = stage + 32 * S1;                           // [32][2 x 8]
        float* const st3 = st2 + 32 * S2;                             // [32][4]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = scaled(acc[t][r]);
            // level 1: ((v00 + v01) + v10) + v11, * 0.25  (avg_pool2d order: row-major window sum)
            float p1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float top = v[2 * u], bot = v[2 * u + 1];
                float q = top + dpp_get<DPP_XOR1>(top);
                q = q + bot;
                q = q + dpp_get<DPP_XOR1>(bot);
                p1[u] = q * 0.25f;
                if ((li & 1) == 0) st1[row * S1 + u * 16 + (li >> 1)] = p1[u];
            }
            float p2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float top = p1[2 * u], bot = p1[2 * u + 1];
                float q = top + dpp_get<DPP_XOR2>(top);
                q = q + bot;
                q = q + dpp_get<DPP_XOR2>(bot);
                p2[u] = q * 0.25f;
                if ((li & 3) == 0) st2[row * S2 + u * 8 + (li >> 2)] = p2[u];
            }
            {
                const float top = p2[0], bot = p2[1];
                float q = top + dpp_get<DPP_SHL4>(top);
                q = q + bot;
                q = q + dpp_get<DPP_SHL4>(bot);
                if ((li & 7) == 0) st3[row * S3 + (li >> 3)] = q * 0.25f;
            }
        }
        if (PF_POOL_STORE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {          // level 1: 32 rows x (4 x 64 B)
                const int idx = lane + 64 * k, row = idx >> 4, u = (idx >> 2) & 3, xp = idx & 3;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st1 + row * S1 + u * 16 + 4 * xp);
                *reinterpret_cast<f32x4*>(a.lvl[1] + (row0 + row) * N1 + (long)((ty0 >> 1) + u) * W1 + (tx0 >> 1) + 4 * xp) = v;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {          // level 2: 32 rows x (2 x 32 B)
                const int idx = lane + 64 * k, row = idx >> 2, u = (idx >> 1) & 1, xp = idx & 1;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st2 + row * S2 + u * 8 + 4 * xp);
                *reinterpret_cast<f32x4*>(a.lvl[2] + (row0 + row) * N2 + (long)((ty0 >> 2) + u) * W2 + (tx0 >> 2) + 4 * xp) = v;
            }
        }
        if (lane < 32) {                           // level 3: 32 rows x 16 B
            const f32x4 v = *reinterpret_cast<const f32x4*>(st3 + lane * S3);
            *reinterpret_cast<f32x4*>(a.lvl[3] + (row0 + lane) * N3 + (long)(ty0 >> 3) * W3 + (tx0 >> 3)) = v;
        }
    };
    if (a.scale_mul != 0.f) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

// ----------------------------------------------------------------------------------------------
// Ring kernel (fused pool, bf16x3, C = 256, W8 % 64 == 0): the HBM-bound form of the corr + pyramid build.
//
// What the measurements on MI355X said about the tile kernel above and the first ring kernels (profiles/r2_corr_*):
//   * the kernel is bound by its STORES, and among them by the pooled levels: 24 % of the bytes, written as
//     64- / 32- / 16-byte pieces of cache lines whose other half arrives 10 us later from the neighbouring block,
//     cost as much as the 76 % of level 0 (store-only builds: level 0 alone 61 us, pooled levels alone 49 us,
//     together 120 us; a fill kernel writes the same bytes in 56 us = 6.7 TB/s);
//   * an LDS transpose in front of the stores buys nothing, and the stores of a wave must have left before the
//     first operand tile issued after them can be consumed (vmcnt retires in issue order).
// So this kernel is built around the store stream:
//   * the GEMM is TRANSPOSED: target pixels are the MFMA's A operand (rows), query pixels its B operand (columns).
//     An accumulator register group then holds FOUR CONSECUTIVE TARGET COLUMNS of one query row (x = 8 g + 4 lh + i
//     in register 4 g + i): level 0 is stored straight from registers, 16 bytes per lane, and every 2x2 pooling
//     is a register-to-register add of one lane (same products, same k order, same pooling order as the tile
//     kernel: bit-identical results);
//   * a workgroup's unit of work is a ring tile of 2 map rows x 64 columns (128 target pixels): per query row
//     it closes 2 x 256 contiguous bytes of level 0 and ONE WHOLE 128-byte line of level 1 in one burst; a
//     workgroup owns a region of 8 map rows x 128 columns (8 tiles, ordered so that level-2 lines close after
//     4 tiles: they are collected in a wave-private LDS image and stored as whole lines; level 3 follows from
//     the two level-2 rows);
//   * a wave keeps its 32 query rows x all 256 channels in REGISTERS for its whole life (128 VGPRs of bf16 hi|lo
//     fragments, loaded once); the target operand streams through a 3-slot LDS ring by LDS-DMA
//     (global_load_lds_dwordx4: no VGPR staging, no ds_write), two tiles ahead, retired by counted
//     s_waitcnt vmcnt(N) in front of ONE raw barrier per K-step (a __syncthreads would add vmcnt(0));
//   * 64 accumulator VGPRs per tile leave room for two workgroups per CU: one's stores run under the other's MFMAs.
// DMA writes are lane-linear (8 rows x 128 B per wave instruction), so ring rows are unpadded and the 16-byte
// pieces are XOR-swizzled by (row >> 1) & 7 on the source address and on the fragment reads (conflict-free
// ds_read_b128).
// ----------------------------------------------------------------------------------------------
constexpr int RING_SLOTS = 3;
constexpr int RING_AHEAD = RING_SLOTS - 1;           // tiles in flight
constexpr int RING_TILE = 128 * 128;                 // bytes: 128 target pixels x 32 channels x {hi, lo} bf16
constexpr int RING_IMG2 = 33;                        // row stride (floats) of the level-2 image [32 queries][32 columns]
constexpr int RING_STAGE = 32 * RING_IMG2 + 32 * 16; // floats per wave: level-2 image + level-3 carry [32][16]
constexpr int RING_LDS = RING_SLOTS * RING_TILE + 4 * RING_STAGE * 4;
constexpr int RING_NK = 8;                           // K-steps: C = 256
constexpr int RING_STORES = 24;                      // global stores of a tile's epilogue: 16 + 8 (+ level 2 / 3 lines: lower bound)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void ring_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ring_for<I + 1, N>(f);
    }
}

// NCH: 64-column chunks per region (2: regions of 8 x 128, level-2 lines complete; 1: W8 % 128 != 0, regions of 8 x 64)
template <bool MUL, int NCH>
__global__ void __launch_bounds__(256, 2)
pf_corr_ring_kernel(const CorrArgs a, const int ablate) {
    // ablate (PRIORFLOW_CORR_ABLATE, timing-only diagnosis; results are garbage): 1 no global stores, 2 no DMA, 4 no MFMA,
    // 8 no epilogue, 256 no pooled-level stores, 512 no level-0 stores
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    float* const img2 = reinterpret_cast<float*>(ring + RING_SLOTS * RING_TILE) + wave * RING_STAGE;    // [32][RING_IMG2]
    float* const carry3 = img2 + 32 * RING_IMG2;                                                         // [32][16]
    constexpr int NTILES = 4 * NCH;                  // tiles of a region
    constexpr int RW = 64 * NCH;                     // region width (map columns)

    // ---- work mapping: XCD-contiguous ranges of q = ((batch * regions + region) * m_tiles + m) -------------------
    // (workgroup ids go round-robin over the 8 XCDs: the workgroups of one XCD share target regions, whose operand
    // rows stay in that XCD's L2 while the query tiles stream over them)
    int b, m0, ty0, tx0;
    {
        const unsigned nwg = gridDim.x, orig = blockIdx.x;
        const unsigned xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
        const unsigned q = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
        const unsigned rpb = (unsigned)a.W / RW;                       // regions per band of 8 map rows
        const unsigned regions = (unsigned)(a.H >> 3) * rpb;
        const unsigned bg = q / (unsigned)a.m_tiles;
        m0 = (int)(q % (unsigned)a.m_tiles) * BM;
        b = (int)(bg / regions);
        const unsigned reg = bg % regions;
        ty0 = (int)(reg / rpb) * 8;
        tx0 = (int)(reg % rpb) * RW;
    }
    const long rowbytes = 4L * a.C;                  // split image: per 32-channel chunk {hi[32], lo[32]} bf16
    const char* const f1b = reinterpret_cast<const char*>(a.f1) + (long)b * a.N * rowbytes;
    const char* const f2b = reinterpret_cast<const char*>(a.f2) + (long)b * a.N * rowbytes;

    // ---- query fragments (the MFMA's B operand: column = query li, K-half lh): registers, once -------------------
    bf16x8 fq[RING_NK][4];
    {
        const char* qrow = f1b + (long)(m0 + 32 * wave + li) * rowbytes + 32 * lh;
#pragma unroll
        for (int ks = 0; ks < RING_NK; ++ks) {
            fq[ks][0] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128);
            fq[ks][1] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 16);
            fq[ks][2] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 64);
            fq[ks][3] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 80);
        }
        // retire these loads HERE, before the first DMA is issued: an ordinary load still pending at the loop header
        // makes hipcc wait vmcnt(0) inside the loop, which would drain the DMA queue on every pass
#pragma unroll
        for (int ks = 0; ks < RING_NK; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fq[ks][i]));
    }

    // ---- target ring ---------------------------------------------------------------------------------------------
    // Tile i of the region: map rows 2 rp, 2 rp + 1 with rp = 2 (i / (2 NCH)) + (i & 1), columns 64 ch .. with
    // ch = (i >> 1) % NCH: the two row pairs of a level-2 row come back to back, then the other column chunk.
    // Ring row 32 w + k of a tile: map row (w >> 1) of the pair, column 32 (w & 1) + k.  Wave w loads rows 32 w ..:
    // piece j covers rows 32 w + 8 j + (lane >> 3); the swizzle term (r >> 1) & 7 repeats with period 2 in j, so
    // pieces j and j + 2 differ by 16 columns only, added as a scalar.
    unsigned dma_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        dma_off[j] = (unsigned)((8 * j + (lane >> 3)) * (int)rowbytes + (((lane & 7) ^ ((r >> 1) & 7)) * 16));
    }
    auto tile_rp = [](int i) { return 2 * (i / (2 * NCH)) + (i & 1); };
    auto tile_ch = [](int i) { return (i >> 1) % NCH; };
    int d_tile = 0, d_ks = 0;                        // position of the DMA stream (RING_AHEAD tiles ahead of the compute stream)
    auto issue_dma = [&](int slot) __attribute__((always_inline)) {
        const int row = ty0 + 2 * tile_rp(d_tile) + (wave >> 1), col = tx0 + 64 * tile_ch(d_tile) + 32 * (wave & 1);
        const char* src = f2b + ((long)row * a.W + col) * rowbytes + d_ks * 128;       // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
        if (!(ablate & 2)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef __attribute__((address_space(3))) void lds_void;
                lds_void* dst = (lds_void*)(ring + slot * RING_TILE + (4 * wave + j) * 1024);
                __builtin_amdgcn_global_load_lds(src + (j >> 1) * 16 * rowbytes + dma_off[j & 1], dst, 16, 0, 0);
            }
        }
#else
        (void)src; (void)slot;
#endif
        // advance; past the end the last tile is re-read into a free slot (keeps the vmcnt counts uniform)
        if (++d_ks == RING_NK) {
            if (d_tile + 1 < NTILES) { d_ks = 0; ++d_tile; } else { d_ks = RING_NK - 1; }
        }
    };
    // fragment read offsets: ring row 32 t + li, pieces (hi k0-7, hi k8-15, lo k0-7, lo k8-15) of K-half lh
    unsigned t_piece[4];
    {
        const unsigned swz = (unsigned)((li >> 1) & 7), p0 = 2u * lh;
        t_piece[0] = ((p0 + 0) ^ swz) * 16; t_piece[1] = ((p0 + 1) ^ swz) * 16;
        t_piece[2] = ((p0 + 4) ^ swz) * 16; t_piece[3] = ((p0 + 5) ^ swz) * 16;
    }
    const char* const t_lane = ring + li * 128;

    // acc[2 r + c][4 g + i]: target (map row r of the pair, column 32 c + 8 g + 4 lh + i of the tile), query row li
    f32x16 acc[4];
    const long N = a.N;
    const int W1 = a.W >> 1, W2 = a.W >> 2, W3 = a.W >> 3;
    const long N1 = N >> 2, N2 = N >> 4, N3 = N >> 6;
    const long row0 = (long)b * N + m0 + 32 * wave;      // first query row of this wave
    // store addresses = wave-uniform base (scalars) + ONE 32-bit per-lane byte offset per level; the pooled levels'
    // offsets are recomputed where they are used from a lane id the compiler cannot see through (hoisted out of the loop
    // they cost two registers each -- zero-extended -- and the kernel has none to spare)
    const unsigned off0 = (unsigned)((li * N + 4 * lh) * 4);
    auto opaque_lane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto scaled = [&](float x) { return MUL ? x * a.scale_mul : x / a.inv_scale; };
    auto pool = [](float tl, float tr, float bl, float br) { float q = tl + tr; q = q + bl; q = q + br; return q * 0.25f; };   // avg_pool2d's order
    float hs[2][4];                                      // level-1 pair sums of an even row pair (level 2's top row)
    const bool st0 = !(ablate & (1 | 512)), stp = !(ablate & (1 | 256));

    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        const int rp = tile_rp(tile), ch = tile_ch(tile);
        // ---- level 0: registers -> memory, 16 bytes per lane, 2 x 256 contiguous bytes per query row --------------
        char* const l0 = reinterpret_cast<char*>(a.lvl[0] + row0 * N + (long)(ty0 + 2 * rp) * a.W + tx0 + 64 * ch) + off0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {scaled(acc[t][4 * g]), scaled(acc[t][4 * g + 1]), scaled(acc[t][4 * g + 2]), scaled(acc[t][4 * g + 3])};
                if (st0) *reinterpret_cast<f32x4*>(l0 + ((long)(t >> 1) * a.W + 32 * (t & 1) + 8 * g) * 4) = v;
            }
        // ---- level 1: one whole 128-byte line per query row (register adds; operation order of F.avg_pool2d) -------
        const int ol = opaque_lane(), oli = ol & 31, olh = ol >> 5;
        char* const l1 = reinterpret_cast<char*>(a.lvl[1] + row0 * N1 + (long)((ty0 >> 1) + rp) * W1 + (tx0 >> 1) + 32 * ch)
                         + (unsigned)((oli * N1 + 2 * olh) * 4);
        f32x2 p1[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& top = acc[c];
                const f32x16& bot = acc[2 + c];
                p1[c][g].x = pool(scaled(top[4 * g]), scaled(top[4 * g + 1]), scaled(bot[4 * g]), scaled(bot[4 * g + 1]));
                p1[c][g].y = pool(scaled(top[4 * g + 2]), scaled(top[4 * g + 3]), scaled(bot[4 * g + 2]), scaled(bot[4 * g + 3]));
                if (stp) *reinterpret_cast<f32x2*>(l1 + (16 * c + 4 * g) * 4) = p1[c][g];
            }
        if (!(rp & 1)) {                             // upper row pair of a level-2 row: keep tl + tr (the first add of the pooling)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) hs[c][g] = p1[c][g].x + p1[c][g].y;
            return;
        }
        // ---- level 2: values into the wave-private image [query][column]; whole lines leave once the row is complete ----
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float q = hs[c][g] + p1[c][g].x;
                q = q + p1[c][g].y;
                img2[oli * RING_IMG2 + 16 * ch + 8 * c + 2 * g + olh] = q * 0.25f;
            }
        if (ch != NCH - 1) return;
        constexpr int C2 = 16 * NCH;                 // level-2 columns of the region
        constexpr int LPR = C2 / 4;                  // lanes per query row (16 bytes each)
        constexpr int RPI = 64 / LPR;                // query rows per store instruction
        const int qr = ol / LPR, pc = ol % LPR;
        char* const l2 = reinterpret_cast<char*>(a.lvl[2] + row0 * N2 + (long)((ty0 >> 2) + (rp >> 1)) * W2 + (tx0 >> 2))
                         + (unsigned)((qr * N2 + 4 * pc) * 4);
        char* const l3 = reinterpret_cast<char*>(a.lvl[3] + row0 * N3 + (long)(ty0 >> 3) * W3 + (tx0 >> 3))
                         + (unsigned)((qr * N3 + 2 * pc) * 4);
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
            const float* src = img2 + (qr + RPI * k) * RING_IMG2 + 4 * pc;
            const f32x4 v = {src[0], src[1], src[2], src[3]};
            if (stp) *reinterpret_cast<f32x4*>(l2 + (long)RPI * k * N2 * 4) = v;
            // ---- level 3 from the two level-2 rows of the region ------------------------------------------------------
            float* c3 = carry3 + (qr + RPI * k) * 16 + 2 * pc;
            if (rp == 1) {
                c3[0] = v.x + v.y;
                c3[1] = v.z + v.w;
            } else {
                float q0 = c3[0] + v.x, q1 = c3[1] + v.z;
                q0 = q0 + v.y; q1 = q1 + v.w;
                const f32x2 o = {q0 * 0.25f, q1 * 0.25f};
                if (stp) *reinterpret_cast<f32x2*>(l3 + (long)RPI * k * N3 * 4) = o;
            }
        }
    };

#pragma unroll
    for (int i = 0; i < RING_AHEAD; ++i) issue_dma(i);
    int slot = 0;                                  // g % RING_SLOTS
    for (int tile = 0; tile < NTILES; ++tile) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        ring_for<0, RING_NK>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            // tile g was issued RING_AHEAD steps ago; younger vector-memory operations of this wave, in issue order: the
            // 4 DMA pieces of each of the RING_AHEAD - 1 tiles behind it, and -- when the previous tile's epilogue lies in
            // between -- its stores (a lower bound of their number: waiting for a few more is harmless)
            // (An LDS arrival flag written by one more DMA per tile and polled by the consumer -- no vmcnt wait at all -- was
            // measured too: 171 us against 152 us.  The stores ahead of a DMA block it inside the CU's memory pipeline anyway.)
            if (ks < RING_AHEAD && tile > 0)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(4 * (RING_AHEAD - 1) + RING_STORES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(4 * (RING_AHEAD - 1)) : "memory");
            issue_dma(slot == 0 ? RING_SLOTS - 1 : slot - 1);             // tile g + RING_AHEAD -> slot (g - 1) % RING_SLOTS
            const char* tp = t_lane + slot * RING_TILE;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 ft[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ft[q] = *reinterpret_cast<const bf16x8*>(tp + t * 4096 + t_piece[q]);
                if (ablate & 4) {
                    asm volatile("" :: "v"(ft[0]), "v"(ft[1]), "v"(ft[2]), "v"(ft[3]));
                    continue;
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {       // per element: (query lo * target hi) + (query hi * target lo) + (hi * hi), as the tile kernel
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[k2], fq[ks][2 + k2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[2 + k2], fq[ks][k2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[k2], fq[ks][k2], acc[t], 0, 0, 0);
                }
            }
            slot = slot == RING_SLOTS - 1 ? 0 : slot + 1;
        });
        if (!(ablate & 8)) epilogue(tile);
        else asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    }
    // the tail DMAs (re-reads) must land before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 2x2 mean of one level into the next (generic path only)
__global__ void __launch_bounds__(256)
pf_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int Hs, int Ws) {
    const int Hd = Hs >> 1, Wd = Ws >> 1;
    const long total = rows * Hd * Wd;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; idx < total; idx += stride) {
        const int x = (int)(idx % Wd);
        const int y = (int)((idx / Wd) % Hd);
        const long row = idx / ((long)Wd * Hd);
        const float* s = src + row * Hs * Ws + (long)(2 * y) * Ws + 2 * x;
        float v = s[0] + s[1];
        v = v + s[Ws];
        v = v + s[Ws + 1];
        dst[idx] = v * 0.25f;
    }
}

}  // namespace

static int corr_launch(const float* f1, const float* f2, float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                       int B, int H8, int W8, int C, bool split, void* stream) {
    if (!f1 || !f2 || !lvl0 || !lvl1 || !lvl2 || !lvl3) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 0 || W8 <= 0 || C <= 0 || (C % KC) != 0) return PF_ERR_BAD_SHAPE;
    // any map with a 2x2 (or larger) level 3; odd sizes pool with avg_pool2d's floor semantics on the generic path
    if ((H8 >> 3) < 2 || (W8 >> 3) < 2) return PF_ERR_BAD_SHAPE;
    if ((long)H8 * W8 * C >= (1L << 31)) return PF_ERR_BAD_SHAPE;      // 32-bit row offsets in the kernel
    if ((long)B * ((H8 * W8 + BM - 1) / BM) * ((H8 * W8 + 31) / 32) >= (1L << 31)) return PF_ERR_BAD_SHAPE;   // 1-D grid
    CorrArgs a;
    a.f1 = f1; a.f2 = f2;
    a.lvl[0] = lvl0; a.lvl[1] = lvl1; a.lvl[2] = lvl2; a.lvl[3] = lvl3;
    a.B = B; a.H = H8; a.W = W8; a.N = H8 * W8; a.C = C;
    a.inv_scale = sqrtf((float)C);
    {   // x / 2^k == x * 2^-k bit for bit; any other divisor keeps the reference's division
        int e = 0;
        const float m = frexpf(a.inv_scale, &e);
        a.scale_mul = (m == 0.5f && a.inv_scale * a.inv_scale == (float)C) ? 1.f / a.inv_scale : 0.f;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool fused = (W8 % 32) == 0 && (H8 % 8) == 0 && (a.N % BM) == 0;
    if (fused) {
        a.tiles_x = W8 / 32;
        a.n2_tiles = (H8 / 8) * a.tiles_x;
        a.m_tiles = a.N / BM;
        // Ring kernel: measured against the tile kernel in one process (profiles/ab_corr.py, profiles/r2_corr_ablation.txt):
        // 152 vs 145 us at B = 1 (both sit on the ~110 us their scattered store stream needs), 4-5 % faster from B = 8 on.
        // PRIORFLOW_CORR_RING = 0 / 1 forces one of them (tests, A/B comparisons).
        const char* env = getenv("PRIORFLOW_CORR_RING");
        const bool ring = env ? env[0] == '1' : B >= 4;
        if (split && C == 32 * RING_NK && (W8 % 64) == 0 && ring) {
            const int nch = (W8 % 128) == 0 ? 2 : 1;                 // regions of 8 x 128 (or 8 x 64) target pixels
            dim3 grid((unsigned)((long)B * a.m_tiles * (H8 / 8) * (W8 / (64 * nch))));
            const bool mul = a.scale_mul != 0.f;
            static const hipError_t attr = [] {
                const void* k[4] = {reinterpret_cast<const void*>(&pf_corr_ring_kernel<true, 2>), reinterpret_cast<const void*>(&pf_corr_ring_kernel<true, 1>),
                                    reinterpret_cast<const void*>(&pf_corr_ring_kernel<false, 2>), reinterpret_cast<const void*>(&pf_corr_ring_kernel<false, 1>)};
                for (int i = 0; i < 4; ++i) {
                    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
                    if (e != hipSuccess) return e;
                }
                return hipSuccess;
            }();
            if (attr != hipSuccess) return (int)attr;
            const char* ab = getenv("PRIORFLOW_CORR_ABLATE");        // timing-only diagnosis (profiles/ab_corr.py)
            const int ablate = ab ? atoi(ab) : 0;
            if (mul && nch == 2) hipLaunchKernelGGL((pf_corr_ring_kernel<true, 2>), grid, dim3(256), RING_LDS, s, a, ablate);
            else if (mul) hipLaunchKernelGGL((pf_corr_ring_kernel<true, 1>), grid, dim3(256), RING_LDS, s, a, ablate);
            else if (nch == 2) hipLaunchKernelGGL((pf_corr_ring_kernel<false, 2>), grid, dim3(256), RING_LDS, s, a, ablate);
            else hipLaunchKernelGGL((pf_corr_ring_kernel<false, 1>), grid, dim3(256), RING_LDS, s, a, ablate);
            return (int)hipGetLastError();
        }
        dim3 grid((unsigned)((long)a.m_tiles * a.n2_tiles * B));
        if (split) hipLaunchKernelGGL((pf_corr_kernel<true, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pf_corr_kernel<true, false>), grid, dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
    a.tiles_x = 0;
    a.n2_tiles = (a.N + BN - 1) / BN;
    a.m_tiles = (a.N + BM - 1) / BM;
    dim3 grid((unsigned)((long)a.m_tiles * a.n2_tiles * B));
    if (split) hipLaunchKernelGGL((pf_corr_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pf_corr_kernel<false, false>), grid, dim3(256), 0, s, a);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    float* lv[4] = {lvl0, lvl1, lvl2, lvl3};
    const long rows = (long)B * a.N;
    for (int i = 0; i < 3; ++i) {
        const int Hs = H8 >> i, Ws = W8 >> i;
        const long total = rows * (Hs >> 1) * (Ws >> 1);
        long blocks = (total + 255) / 256;
        if (blocks > 256L * 64) blocks = 256L * 64;
        hipLaunchKernelGGL(pf_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lv[i], lv[i + 1],
                           rows, Hs, Ws);
        rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return PF_OK;
}

extern "C" int pf_corr_pyramid(const float* f1, const float* f2, float* lvl0, float* lvl1,
                               float* lvl2, float* lvl3, int B, int H8, int W8, int C, void* stream) {
    return corr_launch(f1, f2, lvl0, lvl1, lvl2, lvl3, B, H8, W8, C, false, stream);
}

extern "C" int pf_corr_pyramid_bf16x3(const void* f1_split, const void* f2_split, float* lvl0, float* lvl1,
                                      float* lvl2, float* lvl3, int B, int H8, int W8, int C, void* stream) {
    return corr_launch((const float*)f1_split, (const float*)f2_split, lvl0, lvl1, lvl2, lvl3, B, H8, W8, C,
                       true, stream);
}
