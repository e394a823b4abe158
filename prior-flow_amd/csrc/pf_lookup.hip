// pf_lookup_win_kernel: the DCCL lookups (own-view 9x9x4 window + raw cross-view samples, PriOr-RAFT/core/corr.py:113-137) with
// wave-cooperative windows.  Replaces the one-thread-per-3-taps kernel (pf_lookup_elem, csrc/pf_elem.h) on the device; that
// function stays the scalar statement of the arithmetic (host emulation, PRIORFLOW_LOOKUP_WIN=0) and the bitwise reference.
//
// The 81 taps of one (pixel, level) sample ONE 10-row window of the pixel's own volume row and ONE 10-row window of the
// level-0 grid: the x geometry of a tap depends only on its slow index a, the y geometry only on b.  pf_lookup_elem evaluates
// both per thread and issues 14 scattered loads per 3 taps (timing-only ablations: own-window loads 7 us + grid loads 9 us of a
// 32 us launch); here ONE WAVE owns a pixel and per level
//   P0  36 lanes evaluate the 9 + 9 x parts and 9 + 9 y parts (own volume at level size, grid at level-0 size) once -> LDS;
//   P1  the wave loads the windows cooperatively -- (row slot, a) per lane, so the 9 lanes of a row hit adjacent addresses and
//       a wave instruction touches ~10 cache lines instead of 64: own pairs [10][9], grid quads [10][9] -> LDS;
//   P2  a lane per tap (two levels = 162 taps = 3 rounds of 64): weights from the two parts, own / grid values from the LDS
//       windows, then the cross-view sample -- the only gather left scattered (its position comes out of the grid sample).
// Arithmetic: every expression is the one pf_taps0v / pf_apply_pairs / pf_lookup_elem evaluates, in the same order, on the same
// operands (a pair (row, xb) loaded once instead of per tap is the same pair): outputs are bit-identical (tests).
// Row slots: slot b = row y0 of tap row b, slot 9 = row y1 of tap row 8; row y1 of tap row b < 8 is slot b+1 whenever the two
// row offsets agree (they do unless the fp32 round trip of cy lands an ulp below an integer) -- otherwise that tap loads its
// lower pair straight from memory.
#include <stdlib.h>
#include "pf_common.h"
#include "pf_elem.h"

namespace {

constexpr int LW_WAVES = 4;                       // pixels per workgroup (one wave each)
constexpr int LW_GEO = 4 * 4 * 9;                 // per wave: [level][part: xo, xg, yo, yg][9] x 16 bytes
constexpr int LW_OWN = 2 * 10 * 9;                // per wave: [level of the pass][slot][a] float2
constexpr int LW_GRD = 2 * 10 * 9;                // per wave: [level of the pass][slot][a] float4
constexpr int LW_BYTES = LW_GEO * 16 + LW_OWN * 8 + LW_GRD * 16;      // 6 624 bytes per wave

struct GeoX { float e, w; int xb, flags; };       // flags: 1 xin0, 2 xin1, 4 s0hi, 8 s1lo
struct GeoY { float e, w; int r0, r1f; };         // r0 = y0 * W; r1f = y1 * W | yin0 << 30 | yin1 << 31

__device__ __forceinline__ GeoX geo_x(float x, int W) {                   // the x half of pf_taps0v
    const float ix = pf_roundtrip(x, W);
    const float fx = floorf(ix);
    GeoX g;
    g.w = ix - fx; g.e = 1.f - g.w;
    const bool xin0 = (fx >= 0.f) && (fx <= (float)(W - 1));
    const bool xin1 = (fx >= -1.f) && (fx <= (float)(W - 2));
    g.xb = (fx >= 0.f) ? ((fx <= (float)(W - 2)) ? (int)fx : W - 2) : 0;
    g.flags = (xin0 ? 1 : 0) | (xin1 ? 2 : 0) | (fx == (float)(W - 1) ? 4 : 0) | (fx == -1.f ? 8 : 0);
    return g;
}
__device__ __forceinline__ GeoY geo_y(float y, int H, int W) {            // the y half of pf_taps0v
    const float iy = pf_roundtrip(y, H);
    const float fy = floorf(iy);
    GeoY g;
    g.w = iy - fy; g.e = 1.f - g.w;
    const bool yin0 = (fy >= 0.f) && (fy <= (float)(H - 1));
    const bool yin1 = (fy >= -1.f) && (fy <= (float)(H - 2));
    const int y0 = yin0 ? (int)fy : 0, y1 = yin1 ? (int)fy + 1 : 0;
    g.r0 = y0 * W;
    g.r1f = (y1 * W) | (yin0 ? (1 << 30) : 0) | (yin1 ? (int)0x80000000 : 0);
    return g;
}
// weights of pf_taps0v from the two halves
__device__ __forceinline__ void tap_weights(const GeoX& gx, const GeoY& gy, float (&w)[4]) {
    const bool xin0 = gx.flags & 1, xin1 = gx.flags & 2, yin0 = (gy.r1f >> 30) & 1, yin1 = gy.r1f < 0;
    w[0] = (xin0 && yin0) ? gy.e * gx.e : 0.f;
    w[1] = (xin1 && yin0) ? gy.e * gx.w : 0.f;
    w[2] = (xin0 && yin1) ? gy.w * gx.e : 0.f;
    w[3] = (xin1 && yin1) ? gy.w * gx.w : 0.f;
}
// pf_apply_pairs with the tap described by (flags, weights)
__device__ __forceinline__ float apply_pairs_w(int xflags, const float (&w)[4], const PfPair p0, const PfPair p1) {
    const bool s0hi = xflags & 4, s1lo = xflags & 8;
    float acc = (s0hi ? p0.b : p0.a) * w[0];
    acc = acc + (s1lo ? p0.a : p0.b) * w[1];
    acc = acc + (s0hi ? p1.b : p1.a) * w[2];
    acc = acc + (s1lo ? p1.a : p1.b) * w[3];
    return acc;
}

__global__ void __launch_bounds__(64 * LW_WAVES) pf_lookup_win_kernel(const PfLookupArgs a, const long rows) {
    extern __shared__ __attribute__((aligned(16))) char lw_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * LW_WAVES + wave;            // b * N + n: this wave's pixel
    if (row >= rows) return;                                          // (wave-uniform; no workgroup barrier in this kernel)
    char* const base = lw_smem + wave * LW_BYTES;
    GeoX* const geo = reinterpret_cast<GeoX*>(base);                  // [level][part][9]; y parts are GeoY (same size)
    float2* const ownw = reinterpret_cast<float2*>(base + LW_GEO * 16);
    float4* const grdw = reinterpret_cast<float4*>(base + LW_GEO * 16 + LW_OWN * 8);
    const long N = (long)a.H * a.W;
    const long b = row / N, n = row % N;
    const float c1x = a.coords[(b * 2 + 0) * N + n], c1y = a.coords[(b * 2 + 1) * N + n];

    // ---- P0: geometry of all four levels (144 items) --------------------------------------------------------------------
    for (int id = lane; id < LW_GEO; id += 64) {
        const int lvl = id / 36, part = (id % 36) / 9, i = id % 9;
        const int Hl = a.H >> lvl, Wl = a.W >> lvl;
        const float inv = 1.f / (float)(1 << lvl);                    // coords / 2**i : exact
        if (part < 2) {
            const float cx = c1x * inv + (float)(i - PF_CORR_RADIUS);
            // own view: x wrapped mod W_i; cross view: level-i coordinates index the LEVEL-0 grid (core/corr.py:132-133)
            geo[id] = part == 0 ? geo_x(pf_pymod(cx, (float)Wl), Wl) : geo_x(pf_pymod(cx, (float)a.W), a.W);
        } else {
            const float cy = c1y * inv + (float)(i - PF_CORR_RADIUS);
            const GeoY gy = part == 2 ? geo_y(cy, Hl, Wl) : geo_y(cy, a.H, a.W);
            reinterpret_cast<GeoY*>(geo)[id] = gy;
        }
    }
    const GeoY* const geoy = reinterpret_cast<const GeoY*>(geo);
    // The phases hand data from lane to lane through wave-private LDS.  A wave's LDS operations are processed in issue order, but
    // the compiler must not move a phase's reads above the previous phase's writes of OTHER lanes: a wave-level fence + barrier at
    // every hand-over states the dependency instead of relying on the emitted order.
    auto wave_sync = []() __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    };
    wave_sync();                                                          // P0 -> P1

    for (int pass = 0; pass < 2; ++pass) {
        if (pass) wave_sync();                                            // P2 of the previous pass has read the windows: P1 may rewrite them
        // ---- P1: windows of levels 2*pass, 2*pass + 1 (180 own pairs + 180 grid quads) ----------------------------------
        for (int id = lane; id < LW_OWN; id += 64) {
            const int ll = id / 90, slot = (id % 90) / 9, i = id % 9;
            const int lvl = 2 * pass + ll;
            const long lsz = (long)(a.H >> lvl) * (a.W >> lvl);
            const GeoY yo = geoy[lvl * 36 + 18 + (slot < 9 ? slot : 8)], yg = geoy[lvl * 36 + 27 + (slot < 9 ? slot : 8)];
            const int ro = slot < 9 ? yo.r0 : (yo.r1f & 0x3fffffff), rg = slot < 9 ? yg.r0 : (yg.r1f & 0x3fffffff);
            const PfPair p = pf_load2(a.own[lvl] + row * lsz + ro + geo[lvl * 36 + i].xb);
            ownw[id] = float2{p.a, p.b};
            const long r = (long)rg + geo[lvl * 36 + 9 + i].xb;
            float4 q;
            if (a.g_il) {
                const PfQuad v = pf_load4(a.g_il + 2 * r);
                q = float4{v.a, v.b, v.c, v.d};
            } else {
                const PfPair vx = pf_load2(a.g_w2c + r), vy = pf_load2(a.g_w2c + N + r);
                q = float4{vx.a, vy.a, vx.b, vy.b};
            }
            grdw[id] = q;
        }
        wave_sync();                                                      // P1 -> P2
        // ---- P2: one lane per tap (162 taps) ----------------------------------------------------------------------------
        for (int id = lane; id < 2 * PF_TAPS; id += 64) {
            const int ll = id / PF_TAPS, k = id % PF_TAPS, ta = k / 9, tb = k % 9;
            const int lvl = 2 * pass + ll;
            const int Hl = a.H >> lvl, Wl = a.W >> lvl;
            const long lsz = (long)Hl * Wl;
            const GeoX xo = geo[lvl * 36 + ta], xg = geo[lvl * 36 + 9 + ta];
            const GeoY yo = geoy[lvl * 36 + 18 + tb], yg = geoy[lvl * 36 + 27 + tb];
            float w[4];
            // own view
            tap_weights(xo, yo, w);
            const float2* ow = ownw + ll * 90 + ta;
            const float2 o0 = ow[tb * 9];
            float2 o1;
            {
                const int r1 = yo.r1f & 0x3fffffff;
                const int rn = tb < 8 ? geoy[lvl * 36 + 18 + tb + 1].r0 : r1;       // the row slot tb + 1 holds (slot 9: y1 of row 8)
                if (rn == r1) o1 = ow[(tb + 1) * 9];
                else { const PfPair p = pf_load2(a.own[lvl] + row * lsz + r1 + xo.xb); o1 = float2{p.a, p.b}; }
            }
            const float vo = apply_pairs_w(xo.flags, w, PfPair{o0.x, o0.y}, PfPair{o1.x, o1.y});
            // cross view: grid sample (both components share the taps), then the other volume at the sampled position
            tap_weights(xg, yg, w);
            const float4* gw = grdw + ll * 90 + ta;
            const float4 g0 = gw[tb * 9];
            float4 g1;
            {
                const int r1 = yg.r1f & 0x3fffffff;
                const int rn = tb < 8 ? geoy[lvl * 36 + 27 + tb + 1].r0 : r1;
                if (rn == r1) g1 = gw[(tb + 1) * 9];
                else {
                    const long r = (long)r1 + xg.xb;
                    if (a.g_il) { const PfQuad v = pf_load4(a.g_il + 2 * r); g1 = float4{v.a, v.b, v.c, v.d}; }
                    else { const PfPair vx = pf_load2(a.g_w2c + r), vy = pf_load2(a.g_w2c + N + r); g1 = float4{vx.a, vy.a, vx.b, vy.b}; }
                }
            }
            const float gx = apply_pairs_w(xg.flags, w, PfPair{g0.x, g0.z}, PfPair{g1.x, g1.z});
            const float gy = apply_pairs_w(xg.flags, w, PfPair{g0.y, g0.w}, PfPair{g1.y, g1.w});
            const PfTaps2 t = pf_taps0v(pf_pymod(gx, (float)Wl), gy, Hl, Wl);
            const float raw = pf_apply_v(t, a.other[lvl] + row * lsz);
            const long o = row * a.ld + lvl * PF_TAPS + k;
            a.own_out[o] = vo;
            a.raw_out[o] = raw;
        }
    }
}

}  // namespace

// Launcher used by pf_dccl_lookup_il (pf_elem_kernels.hip); returns -100 when this kernel does not take the launch.
// OPT-IN (PRIORFLOW_LOOKUP_WIN=1, read once).  Measured on MI355X (512x1024, one branch, same process A/B): 45 us per launch
// against 28 us for the per-thread kernel, 122.4 vs 124.3 pairs/s end to end.  The lookup is bound by its VALU work, not by
// its gathers: the cross-view sample's tap evaluation (two fp32 round trips with IEEE divisions + a python-style modulo per
// tap, ~90 instructions) cannot be shared between taps, the shared x / y halves were already amortised over the three taps of a
// thread, and a wave per pixel serialises geometry -> window loads -> taps with two global round trips per pass where the
// per-thread form has thousands of independent threads in flight.  Kept because it is the design the round's brief names
// (LDS-staged windows, one wave per pixel) and the bitwise test pins both forms to each other.
int pf_lookup_win_launch(const PfLookupArgs& a, void* stream) {
    static const bool on = [] { const char* e = getenv("PRIORFLOW_LOOKUP_WIN"); return e && e[0] == '1'; }();
    if (!on) return -100;
    const long rows = (long)a.B * a.H * a.W;
    if ((long)a.H * a.W >= (1L << 29)) return -100;                 // row offsets are packed into 30 bits
    const long blocks = (rows + LW_WAVES - 1) / LW_WAVES;
    if (blocks <= 0 || blocks >= (1L << 31)) return PF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pf_lookup_win_kernel, dim3((unsigned)blocks), dim3(64 * LW_WAVES), LW_WAVES * LW_BYTES, (hipStream_t)stream, a, rows);
    return (int)hipGetLastError();
}
