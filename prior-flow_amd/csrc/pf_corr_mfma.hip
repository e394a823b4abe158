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
#include <stdio.h>
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
// Global stores of the volume.  Non-temporal by default (round 5): a plain store keeps its line in the XCD's L2, so the
// 356 MB write stream of a launch evicts the operand rows every workgroup of the XCD re-reads (the role-split kernel:
// 181 us with plain stores, 109 us with nt stores, same instruction stream; the tile kernel 149 -> 144 us at B = 1 and
// 1287 -> 1062 us at B = 8).  Only for stores of whole 128-byte lines (16- / 32-byte pieces took 360 us as nt
// stores instead of 155).  -DPF_CORR_PLAIN_STORES restores plain stores (profiles/ab_corr_libs.py).
// -DPF_CORR_STORE_POLICY=n (timing / A-B builds): 1 "sc1", 2 "sc0 sc1", 3 "nt sc1", 4 "nt sc0 sc1", 5 "sc0", 6 "nt sc0" on the store
// instruction instead of the plain "nt" (round 6: none beats it, profiles/r6_corr_store_floor.txt).
template <class V>
__device__ __forceinline__ void vol_store(V* p, const V& v) {
#if defined(PF_CORR_PLAIN_STORES)
    *p = v;
#elif defined(PF_CORR_STORE_POLICY) && defined(__HIP_DEVICE_COMPILE__)
#define PF_ST_STR2(x) #x
#define PF_ST_STR(x) PF_ST_STR2(x)
#if PF_CORR_STORE_POLICY == 1
#define PF_ST_BITS "sc1"
#elif PF_CORR_STORE_POLICY == 2
#define PF_ST_BITS "sc0 sc1"
#elif PF_CORR_STORE_POLICY == 3
#define PF_ST_BITS "nt sc1"
#elif PF_CORR_STORE_POLICY == 4
#define PF_ST_BITS "nt sc0 sc1"
#elif PF_CORR_STORE_POLICY == 5
#define PF_ST_BITS "sc0"
#else
#define PF_ST_BITS "nt sc0"
#endif
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off " PF_ST_BITS :: "v"(p), "v"(v) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off " PF_ST_BITS :: "v"(p), "v"(v) : "memory");
    else __builtin_nontemporal_store(v, p);
#else
    __builtin_nontemporal_store(v, p);
#endif
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
                vol_store(reinterpret_cast<f32x4*>(a.lvl[0] + (row0 + row) * N + (long)(ty0 + t) * a.W + tx0 + 4 * piece), v);
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
                vol_store(reinterpret_cast<f32x4*>(a.lvl[1] + (row0 + row) * N1 + (long)((ty0 >> 1) + u) * W1 + (tx0 >> 1) + 4 * xp), v);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {          // level 2: 32 rows x (2 x 32 B)
                const int idx = lane + 64 * k, row = idx >> 2, u = (idx >> 1) & 1, xp = idx & 1;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st2 + row * S2 + u * 8 + 4 * xp);
                vol_store(reinterpret_cast<f32x4*>(a.lvl[2] + (row0 + row) * N2 + (long)((ty0 >> 2) + u) * W2 + (tx0 >> 2) + 4 * xp), v);
            }
        }
        if (lane < 32) {                           // level 3: 32 rows x 16 B
            const f32x4 v = *reinterpret_cast<const f32x4*>(st3 + lane * S3);
            vol_store(reinterpret_cast<f32x4*>(a.lvl[3] + (row0 + lane) * N3 + (long)(ty0 >> 3) * W3 + (tx0 >> 3)), v);
        }
    };
    if (a.scale_mul != 0.f) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

// ----------------------------------------------------------------------------------------------
// Building blocks of the role-split kernel below (they were the round-2 "ring kernel"'s, which lost to both other forms at
// every size measured -- 165.5 against 101.9 us, profiles/r5_final_ab_corr_kernels.txt -- and was deleted in round 6):
//   * the GEMM is TRANSPOSED: target pixels are the MFMA's A operand (rows), query pixels its B operand (columns).
//     An accumulator register group then holds FOUR CONSECUTIVE TARGET COLUMNS of one query row, so every 2x2 pooling is a
//     register-to-register add of one lane (same products, same k order, same pooling order as the tile kernel);
//   * the unit of work is a tile of 2 map rows x 64 columns (128 target pixels): per query row it closes 2 x 256
//     contiguous bytes of level 0 and ONE WHOLE 128-byte line of level 1; level-2 lines close after 4 tiles and are
//     collected in a wave-private LDS image, level 3 follows from two level-2 rows;
//   * a wave keeps its 32 query rows x all 256 channels in REGISTERS for its whole life (128 VGPRs of bf16 hi|lo
//     fragments, loaded once); the target operand streams through an LDS ring by LDS-DMA (global_load_lds_dwordx4: no VGPR
//     staging, no ds_write), retired by counted s_waitcnt vmcnt(N) in front of ONE raw barrier per K-step.
// DMA writes are lane-linear (8 rows x 128 B per wave instruction), so ring rows are unpadded and the 16-byte
// pieces are XOR-swizzled by (row >> 1) & 7 on the source address and on the fragment reads (conflict-free
// ds_read_b128).
// ----------------------------------------------------------------------------------------------
constexpr int RING_TILE = 128 * 128;                 // bytes: 128 target pixels x 32 channels x {hi, lo} bf16
constexpr int RING_IMG2 = 33;                        // row stride (floats) of the level-2 image [32 queries][32 columns]
constexpr int RING_STAGE = 32 * RING_IMG2 + 32 * 16; // floats per wave: level-2 image + level-3 carry [32][16]
constexpr int RING_NK = 8;                           // K-steps: C = 256
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void ring_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ring_for<I + 1, N>(f);
    }
}

// ----------------------------------------------------------------------------------------------
// Role-split kernel (round 5; fused pool, bf16x3, C = 256, W8 % 64 == 0, H8 % 8 == 0): a transposed GEMM whose
// whole store side runs on waves of their own.
//
// What rounds 2-4 measured on the two kernels above (profiles/r2_corr_ablation.txt): GEMM alone 80-97 us, stores alone
// 110-120 us, together 145-152 us -- the two phases of a wave add, because a wave that stores cannot issue MFMAs, its
// s_waitcnt vmcnt for the next operand tile also waits for its own stores, and the stores leave in bursts (43-240 per
// wave, then nothing while the next tile is multiplied).  Here a workgroup is 8 waves with fixed roles:
//   * waves 0-3 (one per SIMD) are MFMA waves: query fragments in registers, target tiles by
//     LDS-DMA through a 4-slot ring, counted vmcnt in front of one barrier per K-step -- and NOTHING else.  The stream is
//     software-pipelined by hand: the fragments of a K-step's four 32 x 32 blocks ping-pong between two register sets, the
//     barrier of K-step k + 1 stands in front of the LAST block of K-step k, so the LDS round trip of the first fragments
//     and the DMA issue hide behind MFMAs that need nothing from memory.  After a tile's last K-step a wave dumps its 64
//     raw accumulators into a staging image [32 queries][2 map rows x 64 columns] (16 ds_write_b128, spread over the next
//     tile's first K-step) and never issues a global store, so its vmcnt counts DMA only;
//   * waves 4-7 (their SIMD partners) are STORE waves: after the barrier that publishes a staging image they pull it into
//     registers (the image is free again within one K-step), and over the eight K-steps of the NEXT tile they scale,
//     pool and store it: per K-step 4 query rows = two level-0 stores of 4 x 256 contiguous bytes and one level-1 store
//     of 4 whole lines.  The store stream of a CU is therefore even in time (3 stores per wave per K-step) instead of a
//     burst per tile, every store instruction writes whole 128-byte lines, and no load ever waits behind a store's
//     completion.  Level 2 goes through a wave-private image and leaves as whole lines, level 3 follows
//     from two level-2 rows.
// A work item is 128 query pixels x (RB map rows x 64 NCH columns) of targets, RB = 16 when H8 % 16 == 0 (512x1024:
// 256 items of 16 tiles, one per CU), else 8.  Same products, same k order, same pooling order as the two kernels above:
// bit-identical results (tests/test_hip_kernels.py).
//
// PF_RS_ABL (compile-time, timing-only diagnosis builds of profiles/; results are garbage): 1 no global stores, 2 no DMA,
// 4 no MFMA and no fragment reads, 8 no store-wave work, 16 no staging dump, 32 no barriers, 64 MFMA waves leave at once,
// 256 no pooled-level stores, 512 no level-0 stores, 1024 fragment reads without MFMAs, 2048 store addresses of a workgroup's tiles
// rotated by a per-workgroup amount (do the workgroups, walking their items in step, pile onto the same memory channels? -- no)
// ----------------------------------------------------------------------------------------------
#ifndef PF_RS_ABL
#define PF_RS_ABL 0
#endif
#ifndef PF_RS_AHEAD
#define PF_RS_AHEAD 2
#endif
#ifndef PF_RS_PRIO
#define PF_RS_PRIO 0
#endif
#ifndef PF_RS_MFMA16             // 1: v_mfma_f32_16x16x32_bf16 on the MFMA waves (round 6); 0: v_mfma_f32_32x32x16_bf16 (round 5, bit-identical to the tile kernel)
#define PF_RS_MFMA16 1
#endif
#ifdef PF_RS_STAMP               // diagnosis build: cycles per wave spent in barriers / behind waits / issuing stores
__device__ unsigned long long pf_rs_dbg[2048 * 8 * 4];
#define RS_T() __builtin_amdgcn_s_memtime()
#endif
constexpr int RS_SLOTS = 4;                          // divides the 8 K-steps of a tile: the slot of a K-step is a compile-time constant
constexpr int RS_AHEAD = PF_RS_AHEAD;                // K-steps in flight (2 or 3)
constexpr int RS_PITCH = 132;                        // floats per staging row: 2 x 64 targets + 16 bytes (conflict-free ds_write_b128)
constexpr int RS_STAGE = 32 * RS_PITCH;              // floats per MFMA wave
constexpr int RS_LDS = RS_SLOTS * RING_TILE + 4 * RS_STAGE * 4 + 4 * RING_STAGE * 4;
#define RS_SB() __builtin_amdgcn_sched_barrier(0)

template <bool MUL, int NCH>
__global__ void __launch_bounds__(512)
pf_corr_rs_kernel(const CorrArgs a, const int RB) {
    constexpr int ablate = PF_RS_ABL;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int RW = 64 * NCH;                     // region width (map columns)
    const int NT = (RB >> 1) * NCH;                  // tiles of a work item

    // ---- work mapping: XCD-contiguous ranges of q = ((batch * regions + region) * m_tiles + m) ----------------------
    int b, m0, ty0, tx0;
    {
        const unsigned nwg = gridDim.x, orig = blockIdx.x;
        const unsigned xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
        const unsigned q = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
        const unsigned rpb = (unsigned)a.W / RW;                       // regions per band
        const unsigned regions = (unsigned)(a.H / RB) * rpb;
        const unsigned bg = q / (unsigned)a.m_tiles;
        m0 = (int)(q % (unsigned)a.m_tiles) * BM;
        b = (int)(bg / regions);
        const unsigned reg = bg % regions;
        ty0 = (int)(reg / rpb) * RB;
        tx0 = (int)(reg % rpb) * RW;
    }
    auto tile_rp = [](int i) { return 2 * (i / (2 * NCH)) + (i & 1); };
    auto tile_ch = [](int i) { return (i >> 1) % NCH; };
    float* const stag_all = reinterpret_cast<float*>(ring + RS_SLOTS * RING_TILE);
    // Synchronisation: one s_barrier per K-step for all eight waves.  (Round 5 also built LDS arrival / full / empty counters in its
    // place -- the store waves then never meet a barrier and may lag a whole tile: bit-identical, 111 against 97 us per launch,
    // because the atomic + poll per K-step cost the four MFMA waves more than the barrier waits they remove,
    // profiles/r5_corr_rs_flagsync.txt; the code was removed in round 6.)
    if (wave < 4) {
        // ================================ MFMA waves ================================================================
        if (ablate & 64) return;
#if PF_RS_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        const int li = lane & 31, lh = lane >> 5;
        const long rowbytes = 4L * a.C;
        const char* const f1b = reinterpret_cast<const char*>(a.f1) + (long)b * a.N * rowbytes;
        const char* const f2b = reinterpret_cast<const char*>(a.f2) + (long)b * a.N * rowbytes;
#if PF_RS_MFMA16
        // v_mfma_f32_16x16x32_bf16: one instruction covers a whole 32-channel chunk.  Lane l holds, of operand row / column l & 15,
        // the 8 channels 8 (l >> 4) ..: the 16-byte piece (l >> 4) of the row image's hi half, piece 4 + (l >> 4) of its lo half.
        // A wave's 32 queries are two column tiles qt; fq[ks][2 qt + {0: hi, 1: lo}].
        const int l15 = lane & 15, l4 = lane >> 4;
#endif
        bf16x8 fq[RING_NK][4];                           // loaded behind the first DMA pieces (below)
        // ring row 32 w + k of a tile: map row (w >> 1) of the pair, column 32 (w & 1) + k; wave w loads rows 32 w ..: piece j
        // covers rows 32 w + 8 j + (lane >> 3), the swizzle term (r >> 1) & 7 repeats with period 2 in j
        unsigned dma_off[4];                             // per-lane byte offsets of the 4 pieces from the wave's tile base
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 32 * wave + 8 * (j & 1) + (lane >> 3);
            dma_off[j] = (unsigned)((16 * (j >> 1) + 8 * (j & 1) + (lane >> 3)) * (int)rowbytes + (((lane & 7) ^ ((r >> 1) & 7)) * 16));
        }
        auto tile_src = [&](int i) -> const char* {       // wave-uniform: first target row of this wave's quarter of tile i
            const int row = ty0 + 2 * tile_rp(i) + (wave >> 1), col = tx0 + 64 * tile_ch(i) + 32 * (wave & 1);
            return f2b + ((long)row * a.W + col) * rowbytes;
        };
        const char* cur_src = tile_src(0);
        const char* nxt_src = cur_src;
        typedef __attribute__((address_space(3))) char lds_char;
        const unsigned lds_base = (unsigned)(unsigned long)(lds_char*)ring;
        const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds_base + 4096u * (unsigned)wave);
        // One K-step of this wave's quarter tile: 4 pieces of 8 rows x 128 B, source = scalar tile base + per-lane offset +
        // K-step offset (folded into the scalar base: the instruction's immediate offset would move the LDS address as well).
        // Inline assembly: the builtin wants a 64-bit per-lane pointer (8 VGPRs for the four pieces and a
        // 64-bit vector add per piece; the kernel has no registers to spare), the instruction takes a scalar base.
        auto dma_piece = [&](const char* src, auto KOFF, int slot, int j) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (!(ablate & 2)) {
                unsigned keep;
                const unsigned dst = wave_dst + (unsigned)(slot * RING_TILE + j * 1024);
                const unsigned off = dma_off[j];
                const char* const ksrc = src + decltype(KOFF)::value;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(ksrc), "s"(dst) : "memory");
            }
#else
            (void)src; (void)slot; (void)j;
#endif
        };
        // fragment read addresses (LDS byte offsets): ring row 32 t + li, pieces (hi k0-7, hi k8-15, lo k0-7, lo k8-15) of K-half lh.
        // The reads and their waits are inline assembly: hipcc's own wait insertion drains lgkmcnt to 0 in front of every MFMA
        // group, which exposes the round trip of the reads issued just before it (2 x ~120 cycles per K-step).
        unsigned t_addr[4];
#if PF_RS_MFMA16
        {
            // block t = ring rows 32 t .. 32 t + 31 = two 16-row target tiles j; t_addr[2 j + {0: hi, 1: lo}] of ring row 16 j + l15
            // (the swizzle term (row >> 1) & 7 does not depend on j or t); 16 lanes of a K group read 16 rows x one piece each:
            // 2 x 8 swizzled positions x 4 banks = all 64 banks, conflict-free
            const unsigned swz = (unsigned)((l15 >> 1) & 7);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                t_addr[q] = lds_base + (unsigned)(16 * (q >> 1) + l15) * 128u + ((((unsigned)l4 + 4u * (q & 1)) ^ swz) * 16u);
        }
        // D[target 4 (l >> 4) + i][query l & 15]: a lane's 4 accumulator registers are 4 consecutive targets of one query
        float* const stag = stag_all + wave * RS_STAGE + l15 * RS_PITCH + 4 * l4;
#else
        {
            const unsigned base = lds_base + (unsigned)li * 128u;
            const unsigned swz = (unsigned)((li >> 1) & 7), p0 = 2u * lh;
            t_addr[0] = base + ((p0 + 0) ^ swz) * 16; t_addr[1] = base + ((p0 + 1) ^ swz) * 16;
            t_addr[2] = base + ((p0 + 4) ^ swz) * 16; t_addr[3] = base + ((p0 + 5) ^ swz) * 16;
        }
        float* const stag = stag_all + wave * RS_STAGE + li * RS_PITCH + 4 * lh;

#endif
#if PF_RS_MFMA16
        // acc[t][2 j + qt][i]: target = ring row 32 t + 16 j + 4 (l >> 4) + i of the tile, query 16 qt + (l & 15)
        f32x4 acc[4][4];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[t][g] = zero4;
#else
        // acc[2 r + c][4 g + i]: target (map row r of the pair, column 32 c + 8 g + 4 lh + i of the tile), query row li
        f32x16 acc[4];
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = zero16;
#endif
        bf16x8 fa[4], fb[4];                             // the two fragment register sets
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[q] = bf16x8{}; fb[q] = bf16x8{}; }
        auto reads = [&](bf16x8 (&ft)[4], auto OFF) __attribute__((always_inline)) {
            if (ablate & 4) return;
            constexpr int off = decltype(OFF)::value;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned ad = t_addr[q];
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ft[q]) : "v"(ad), "n"(off));
            }
#endif
        };
        // all but the N youngest LDS operations of this wave have returned: the fragments of `ft` may be used
        auto wait_for = [&](bf16x8 (&ft)[4], auto CNT) __attribute__((always_inline)) {
            if (ablate & 4) return;
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ft[0]), "+v"(ft[1]), "+v"(ft[2]), "+v"(ft[3]) : "n"(decltype(CNT)::value));
        };
#if PF_RS_MFMA16
        // One block of a K-step (32 targets x 32 queries = 2 x 2 tiles of 16 x 16): 12 MFMAs of 16 cycles, pass-major -- (query lo *
        // target hi), (query hi * target lo), (hi * hi), each over the 4 tiles, so the same accumulator comes back every fourth
        // instruction -- with gap(i) behind every second MFMA: the same six gaps of ~32 cycles per block as the 32x32x16 form.
        // Per element the three passes now add over all 32 channels of the chunk at once (the 32x32x16 form: 16 + 16), so the
        // sums differ from the tile kernel's in the last bits; the oracle is the pin (tests/test_hip_kernels.py).
        auto block = [&](bf16x8 (&ft)[4], int t, int ks, bool fresh, auto&& gap) __attribute__((always_inline)) {
            ring_for<0, 12>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value, pass = i / 4, g = i % 4, j = g >> 1, qt = g & 1;
                if (!(ablate & (4 | 1024))) {
                    if (pass == 0) acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ft[2 * j], fq[ks][2 * qt + 1], fresh ? zero4 : acc[t][g], 0, 0, 0);
                    else if (pass == 1) acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ft[2 * j + 1], fq[ks][2 * qt], acc[t][g], 0, 0, 0);
                    else acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ft[2 * j], fq[ks][2 * qt], acc[t][g], 0, 0, 0);
                } else if ((ablate & 1024) && i == 0) asm volatile("" :: "v"(ft[0]), "v"(ft[1]), "v"(ft[2]), "v"(ft[3]));
                if constexpr (i & 1) {
                    RS_SB();
                    gap(std::integral_constant<int, i / 2>{});
                    RS_SB();
                }
            });
        };
#else
        // One 32 x 32 block of a K-step: 6 MFMAs -- per element (query lo * target hi) + (query hi * target lo) + (hi * hi) for
        // both 16-channel halves, the tile kernel's order -- with gap(i) issued behind MFMA i.  What goes into the gaps is
        // everything else this wave does: one DMA piece, or one ds_write_b128 of the staging dump (the LDS takes wide
        // stores at ~80 B/clk per CU: the four waves' 64 KB of accumulators in one burst held the matrix pipe for ~800
        // cycles per tile; one store per gap hides behind the 32 cycles of the MFMA in front of it).
        auto block = [&](bf16x8 (&ft)[4], int t, int ks, bool fresh, auto&& gap) __attribute__((always_inline)) {
            ring_for<0, 6>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value, k2 = i / 3, m = i % 3;
                if (!(ablate & (4 | 1024))) {
                    if (m == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[k2], fq[ks][2 + k2], (fresh && i == 0) ? zero16 : acc[t], 0, 0, 0);
                    else if (m == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[2 + k2], fq[ks][k2], acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[k2], fq[ks][k2], acc[t], 0, 0, 0);
                } else if ((ablate & 1024) && i == 0) asm volatile("" :: "v"(ft[0]), "v"(ft[1]), "v"(ft[2]), "v"(ft[3]));
                RS_SB();
                gap(I);
                RS_SB();
            });
        };
#endif
        auto no_gap = [](auto) {};
#if PF_RS_MFMA16
        auto dump_piece = [&](int t, int g) __attribute__((always_inline)) {   // staging[query 16 qt + l15][32 t + 16 j + 4 l4 ..+3], g = 2 j + qt
            if (ablate & 16) { asm volatile("" :: "v"(acc[t][g])); return; }
            *reinterpret_cast<f32x4*>(stag + (g & 1) * 16 * RS_PITCH + 32 * t + 16 * (g >> 1)) = acc[t][g];
        };
#else
        auto dump_piece = [&](int t, int g) __attribute__((always_inline)) {   // staging[query li][64 r + 32 c + 8 g + 4 lh ..+3], t = 2 r + c
            if (ablate & 16) { asm volatile("" :: "v"(acc[t])); return; }
            const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            *reinterpret_cast<f32x4*>(stag + 64 * (t >> 1) + 32 * (t & 1) + 8 * g) = v;
        };
#endif
        using C0 = std::integral_constant<int, 0>;
        using C4 = std::integral_constant<int, 4>;

        ring_for<0, RS_AHEAD>([&](auto I) __attribute__((always_inline)) {
            ring_for<0, 4>([&](auto J) __attribute__((always_inline)) {
                dma_piece(cur_src, std::integral_constant<int, decltype(I)::value * 128>{}, decltype(I)::value, decltype(J)::value);
            });
        });
        // Query fragments: once, into registers.  Issued BEHIND the first K-steps' DMA pieces (round 6: their latencies overlap; the
        // loads in front cost ~1.5 us of a 100 us launch) and retired HERE: an ordinary load still pending at the loop header makes
        // hipcc wait vmcnt(0) inside the loop (its counter does not see the DMA pieces, which are inline assembly; being older than
        // the loads they only make its counted waits longer, never shorter).
        {
#if PF_RS_MFMA16
            const char* qrow = f1b + (long)(m0 + 32 * wave + l15) * rowbytes + 16 * l4;
#pragma unroll
            for (int ks = 0; ks < RING_NK; ++ks) {
                fq[ks][0] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128);
                fq[ks][1] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 64);
                fq[ks][2] = *reinterpret_cast<const bf16x8*>(qrow + 16 * rowbytes + ks * 128);
                fq[ks][3] = *reinterpret_cast<const bf16x8*>(qrow + 16 * rowbytes + ks * 128 + 64);
            }
#else
            const char* qrow = f1b + (long)(m0 + 32 * wave + li) * rowbytes + 32 * lh;
#pragma unroll
            for (int ks = 0; ks < RING_NK; ++ks) {
                fq[ks][0] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128);
                fq[ks][1] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 16);
                fq[ks][2] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 64);
                fq[ks][3] = *reinterpret_cast<const bf16x8*>(qrow + ks * 128 + 80);
            }
#endif
#pragma unroll
            for (int ks = 0; ks < RING_NK; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fq[ks][i]));
        }
#ifdef PF_RS_STAMP
        unsigned long long st_wait = 0, st_bar = 0;
        const unsigned long long st_begin = RS_T();
#endif
        for (int tile = 0; tile < NT; ++tile) {
            nxt_src = tile_src(tile + 1 < NT ? tile + 1 : tile);      // past the end the last tile is re-read (uniform vmcnt counts)
            ring_for<0, RING_NK>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                constexpr int sbase = (ks % RS_SLOTS) * RING_TILE;
                constexpr int kd = ks + RS_AHEAD;                      // the K-step whose DMA is issued here
                // K-step g's pieces were issued RS_AHEAD steps ago; the only younger vector-memory operations of this wave are
                // the 4 pieces of each of the RS_AHEAD - 1 steps behind it (this wave never stores).  lgkmcnt(0): the last
                // block's fragments of the previous K-step are in registers (its slot may be overwritten from here on) and a
                // staging dump has landed before the barrier that hands it to the store wave.
#ifdef PF_RS_STAMP
                {
                    const unsigned long long t0 = RS_T();
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(4 * (RS_AHEAD - 1)) : "memory");
                    const unsigned long long t1 = RS_T();
                    asm volatile("s_barrier" ::: "memory");
                    const unsigned long long t2 = RS_T();
                    st_wait += t1 - t0; st_bar += t2 - t1;
                }
#else
                if (ablate & 32) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(4 * (RS_AHEAD - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(4 * (RS_AHEAD - 1)) : "memory");
#endif
                RS_SB();
                reads(fa, std::integral_constant<int, sbase>{});
                RS_SB();
                // last block of the previous K-step (operands already in registers) with this step's DMA pieces in its gaps
                const char* const dsrc = kd >= RING_NK ? nxt_src : cur_src;
                using DK = std::integral_constant<int, (kd % RING_NK) * 128>;
                auto dma_gap = [&](auto I) __attribute__((always_inline)) {
                    if constexpr (decltype(I)::value < 4) dma_piece(dsrc, DK{}, kd % RS_SLOTS, decltype(I)::value);
                };
                if (ks > 0 || tile > 0) {
                    wait_for(fb, C4{});
                    block(fb, 3, (ks + RING_NK - 1) % RING_NK, ks == 1, dma_gap);
                } else {
                    ring_for<0, 4>(dma_gap);
                }
                // The staging dump of a tile rides in the gaps of blocks that do not touch the accumulator being dumped:
                // acc[0] / acc[1] (final after blocks 0 / 1 of the tile's last K-step) behind blocks 1 / 2 of that K-step,
                // acc[2] / acc[3] (final after block 2 / the block above) behind blocks 0 / 1 of the next tile's first K-step.
                auto dump_gap = [&](int t, bool on) {
                    return [&, t, on](auto I) __attribute__((always_inline)) {
                        if constexpr (decltype(I)::value < 4) { if (on) dump_piece(t, decltype(I)::value); }
                    };
                };
                RS_SB();
                reads(fb, std::integral_constant<int, sbase + 4096>{});
                RS_SB();
                wait_for(fa, C4{});
                if (ks == 0) block(fa, 0, ks, true, dump_gap(2, tile > 0));
                else block(fa, 0, ks, false, no_gap);
                RS_SB();
                reads(fa, std::integral_constant<int, sbase + 2 * 4096>{});
                RS_SB();
                wait_for(fb, C4{});
                if (ks == 0) block(fb, 1, ks, true, dump_gap(3, tile > 0));
                else if (ks == RING_NK - 1) block(fb, 1, ks, false, dump_gap(0, true));
                else block(fb, 1, ks, false, no_gap);
                RS_SB();
                reads(fb, std::integral_constant<int, sbase + 3 * 4096>{});
                RS_SB();
                wait_for(fa, C4{});
                if (ks == RING_NK - 1) block(fa, 2, ks, false, dump_gap(1, true));
                else block(fa, 2, ks, ks == 0, no_gap);
                RS_SB();
            });
            cur_src = nxt_src;
        }
        wait_for(fb, C0{});
        block(fb, 3, RING_NK - 1, false, no_gap);
#pragma unroll
        for (int g = 0; g < 4; ++g) dump_piece(2, g);
#pragma unroll
        for (int g = 0; g < 4; ++g) dump_piece(3, g);
        // the last dump is handed over; the tail DMAs (re-reads) must land before the workgroup's LDS is released
        if (ablate & 32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PF_RS_STAMP
        if (lane == 0 && blockIdx.x < 2048) {
            unsigned long long* d = pf_rs_dbg + (blockIdx.x * 8 + wave) * 4;
            d[0] = RS_T() - st_begin; d[1] = st_wait; d[2] = st_bar; d[3] = 0;
        }
#endif
        return;
    }

    // ==================================== store waves ===================================================================
    const int sw = wave - 4;
#if PF_RS_PRIO == 1
    __builtin_amdgcn_s_setprio(3);
#endif
    const int qq = lane >> 4, j = lane & 15;         // pass p: query row 4 p + qq of the wave's 32, target columns 4 j .. 4 j + 3 of both map rows
    const float* const stag = stag_all + sw * RS_STAGE + qq * RS_PITCH + 4 * j;
    float* const img2 = stag_all + 4 * RS_STAGE + sw * RING_STAGE;                                       // [32][RING_IMG2]
    float* const carry3 = img2 + 32 * RING_IMG2;                                                         // [32][16]
    const long N = a.N;
    const int W1 = a.W >> 1, W2 = a.W >> 2, W3 = a.W >> 3;
    const long N1 = N >> 2, N2 = N >> 4, N3 = N >> 6;
    const long row0 = (long)b * N + m0 + 32 * sw;        // first query row of this wave
    const unsigned off0 = (unsigned)((qq * N + 4 * j) * 4);
    const unsigned off1 = (unsigned)((qq * N1 + 2 * j) * 4);
    auto scaled = [&](float x) { return MUL ? x * a.scale_mul : x / a.inv_scale; };
    auto pool = [](float tl, float tr, float bl, float br) { float q = tl + tr; q = q + bl; q = q + br; return q * 0.25f; };   // avg_pool2d's order
    constexpr bool st0 = !(ablate & (1 | 512)), stp = !(ablate & (1 | 256));
    float hs[8];                                         // level-1 pair sums of an even row pair (level 2's top row), per pass
    f32x4 raw[8][2];

    auto pass = [&](auto P, const int dt) __attribute__((always_inline)) {       // pass p of data tile dt
        constexpr int p = decltype(P)::value;
        // (PF_RS_ABL & 2048, timing only: every workgroup addresses the tiles of its item in an order of its own -- do the workgroups,
        // which all walk their items in step, pile their stores onto the same address bits 8..12 = the same memory channels?)
        const int dta = (ablate & 2048) ? ((dt + (int)((blockIdx.x * 5u) & 15u)) % NT) : dt;
        const int rp = tile_rp(dta), ch = tile_ch(dta);
        f32x4 r0 = raw[p][0], r1 = raw[p][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { r0[i] = scaled(r0[i]); r1[i] = scaled(r1[i]); }
        // ---- level 0: 4 query rows x 256 contiguous bytes per store, both map rows ----------------------------------
        char* const l0 = reinterpret_cast<char*>(a.lvl[0] + (row0 + 4 * p) * N + (long)(ty0 + 2 * rp) * a.W + tx0 + 64 * ch) + off0;
        // (Round 6 timed a wrong-on-purpose variant in which a store instruction writes ONE query row x 1 KB contiguous instead of
        // 4 query rows x 256 B -- every byte of level 0 still written once: store side alone 86.9 against 86.3 us, whole kernel 97.5
        // against 93.3 us, profiles/r6_corr_rs_m16.txt.  The layout of the store stream is not what bounds it.)
        if (st0) {
            vol_store(reinterpret_cast<f32x4*>(l0), r0);
            vol_store(reinterpret_cast<f32x4*>(l0 + 4L * a.W), r1);
        }
        // ---- level 1: 4 whole lines per store (operation order of F.avg_pool2d) ----------------------------------------
        f32x2 p1;
        p1.x = pool(r0[0], r0[1], r1[0], r1[1]);
        p1.y = pool(r0[2], r0[3], r1[2], r1[3]);
        char* const l1 = reinterpret_cast<char*>(a.lvl[1] + (row0 + 4 * p) * N1 + (long)((ty0 >> 1) + rp) * W1 + (tx0 >> 1) + 32 * ch) + off1;
        if (stp) vol_store(reinterpret_cast<f32x2*>(l1), p1);
        if (!(rp & 1)) {
            hs[p] = p1.x + p1.y;                     // tl + tr: the first add of the level-2 pooling
        } else {
            float q = hs[p] + p1.x;
            q = q + p1.y;
            img2[(4 * p + qq) * RING_IMG2 + 16 * ch + j] = q * 0.25f;
        }
    };
    auto finish = [&](const int dt) __attribute__((always_inline)) {
        // level 2: whole lines out of the wave-private image once both row pairs of all column chunks are in; level 3 from
        // the two level-2 rows of an 8-row band 
        const int rp = tile_rp(dt), ch = tile_ch(dt);
        if (!(rp & 1) || ch != NCH - 1) return;
        constexpr int C2 = 16 * NCH;                 // level-2 columns of the region
        constexpr int LPR = C2 / 4;                  // lanes per query row (16 bytes each)
        constexpr int RPI = 64 / LPR;                // query rows per store instruction
        const int l2r = rp >> 1;                     // level-2 row inside the item
        const int qr = lane / LPR, pc = lane % LPR;
        char* const l2 = reinterpret_cast<char*>(a.lvl[2] + row0 * N2 + (long)((ty0 >> 2) + l2r) * W2 + (tx0 >> 2))
                         + (unsigned)((qr * N2 + 4 * pc) * 4);
        char* const l3 = reinterpret_cast<char*>(a.lvl[3] + row0 * N3 + (long)((ty0 >> 3) + (l2r >> 1)) * W3 + (tx0 >> 3))
                         + (unsigned)((qr * N3 + 2 * pc) * 4);
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
            const float* src = img2 + (qr + RPI * k) * RING_IMG2 + 4 * pc;
            const f32x4 v = {src[0], src[1], src[2], src[3]};
            if (stp) vol_store(reinterpret_cast<f32x4*>(l2 + (long)RPI * k * N2 * 4), v);
            float* c3 = carry3 + (qr + RPI * k) * 16 + 2 * pc;
            if (!(l2r & 1)) {
                c3[0] = v.x + v.y;
                c3[1] = v.z + v.w;
            } else {
                float q0 = c3[0] + v.x, q1 = c3[1] + v.z;
                q0 = q0 + v.y; q1 = q1 + v.w;
                const f32x2 o = {q0 * 0.25f, q1 * 0.25f};
                if (stp) vol_store(reinterpret_cast<f32x2*>(l3 + (long)RPI * k * N3 * 4), o);
            }
        }
    };
    auto pull = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            raw[p][0] = *reinterpret_cast<const f32x4*>(stag + 4 * p * RS_PITCH);
            raw[p][1] = *reinterpret_cast<const f32x4*>(stag + 4 * p * RS_PITCH + 64);
        }
    };

#ifdef PF_RS_STAMP
    unsigned long long st_bar = 0, st_work = 0, st_last = 0;
    const unsigned long long st_begin = RS_T();
#endif
    // The staging image of tile i is complete behind the barrier of K-step (i + 1, 1).  Interval (tile, ks): pass (ks - 1) & 7 of
    // data tile (ks >= 1 ? tile - 1 : tile - 2).
    for (int tile = 0; tile < NT; ++tile) {
        ring_for<0, RING_NK>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            // lgkmcnt(0): the staging reads of this wave have returned before the barrier behind which the image is rewritten
#ifdef PF_RS_STAMP
            {
                const unsigned long long t0 = RS_T();
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned long long t1 = RS_T();
                st_bar += t1 - t0;
                if (st_last) st_work += t0 - st_last;
                st_last = t1;
            }
#else
            if (ablate & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            if (ablate & 8) return;
            const int dt = ks >= 1 ? tile - 1 : tile - 2;
            if (dt < 0) return;
            if (ks == 1) pull();
            pass(std::integral_constant<int, (ks + 7) & 7>{}, dt);
            if (ks == 0) finish(dt);
        });
    }
#ifdef PF_RS_STAMP
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long* d = pf_rs_dbg + (blockIdx.x * 8 + wave) * 4;
        d[0] = RS_T() - st_begin; d[1] = st_work; d[2] = st_bar; d[3] = 1;
    }
#endif
    if (!(ablate & 8) && NT >= 2) {
        pass(std::integral_constant<int, 7>{}, NT - 2);
        finish(NT - 2);
    }
    if (ablate & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ablate & 8) return;
    pull();
    ring_for<0, 8>([&](auto P) __attribute__((always_inline)) { pass(P, NT - 1); });
    finish(NT - 1);
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
        // PRIORFLOW_CORR_RS=0 keeps the tile kernel on every map (the bitwise reference of the role-split kernel: tests, A/Bs)
        const char* const rs_env = getenv("PRIORFLOW_CORR_RS");          // read per launch: the tests switch inside one process
        const bool rs_on = !(rs_env && rs_env[0] == '0');
        if (split && C == 32 * RING_NK && (W8 % 64) == 0 && rs_on) {
            const int nch = (W8 % 128) == 0 ? 2 : 1;                 // work items of RB x 128 (or RB x 64) target pixels
            const int RB = (H8 % 16) == 0 ? 16 : 8;
            dim3 grid((unsigned)((long)B * a.m_tiles * (H8 / RB) * (W8 / (64 * nch))));
            const bool mul = a.scale_mul != 0.f;
            static const hipError_t attr = [] {
                const void* k[4] = {reinterpret_cast<const void*>(&pf_corr_rs_kernel<true, 2>), reinterpret_cast<const void*>(&pf_corr_rs_kernel<true, 1>),
                                    reinterpret_cast<const void*>(&pf_corr_rs_kernel<false, 2>), reinterpret_cast<const void*>(&pf_corr_rs_kernel<false, 1>)};
                for (int i = 0; i < 4; ++i) {
                    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
                    if (e != hipSuccess) return e;
                }
                return hipSuccess;
            }();
            if (attr != hipSuccess) return (int)attr;
            if (mul && nch == 2) hipLaunchKernelGGL((pf_corr_rs_kernel<true, 2>), grid, dim3(512), RS_LDS, s, a, RB);
            else if (mul) hipLaunchKernelGGL((pf_corr_rs_kernel<true, 1>), grid, dim3(512), RS_LDS, s, a, RB);
            else if (nch == 2) hipLaunchKernelGGL((pf_corr_rs_kernel<false, 2>), grid, dim3(512), RS_LDS, s, a, RB);
            else hipLaunchKernelGGL((pf_corr_rs_kernel<false, 1>), grid, dim3(512), RS_LDS, s, a, RB);
#ifdef PF_RS_STAMP
            if (getenv("PRIORFLOW_CORR_STAMP")) {
                hipDeviceSynchronize();
                static unsigned long long h[2048 * 8 * 4];
                hipMemcpyFromSymbol(h, HIP_SYMBOL(pf_rs_dbg), sizeof(h));
                const int nb = grid.x < 2048 ? (int)grid.x : 2048;
                double m[3] = {0, 0, 0}, st[3] = {0, 0, 0};
                for (int bI = 0; bI < nb; ++bI)
                    for (int w = 0; w < 8; ++w)
                        for (int k = 0; k < 3; ++k) (w < 4 ? m : st)[k] += (double)h[(bI * 8 + w) * 4 + k] / (4.0 * nb);
                fprintf(stderr, "[rs stamp] MFMA waves: loop %.0f cyc, vmcnt/lgkm wait %.0f, in barrier %.0f | store waves: loop %.0f, between barriers (work) %.0f, in barrier %.0f   (s_memtime ticks, mean over %d workgroups)\n",
                        m[0], m[1], m[2], st[0], st[1], st[2], nb);
            }
#endif
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
