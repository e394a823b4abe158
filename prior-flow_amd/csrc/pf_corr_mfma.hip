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
