// Per-element device functions of the non-MFMA kernels (samplers, ERP geometry,
// DCCL lookups, convex upsampling, small direct convolutions).
//
// Each `*_elem(idx, args)` computes ONE output element (or one pixel) and is wrapped by
// a __global__ kernel in pf_elem_kernels.hip.  The same functions compile for the host
// (tests/emu) so that index / wrap / padding logic can be checked against the oracle
// without a GPU; the product only ever runs the device build.
//
// Reference citations are relative to /root/reference/PriOr-RAFT.
#pragma once
#include "pf_common.h"

// ----------------------------------------------------------------------------------------------
// accumulation into gradient buffers: device atomics / omp atomic in the host emulation
#if defined(__HIP_DEVICE_COMPILE__)
#define PF_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
#else
#define PF_ATOMIC_ADD(ptr, v) do { float* pf_p_ = (ptr); const float pf_v_ = (v); _Pragma("omp atomic") *pf_p_ += pf_v_; } while (0)
#endif
// scalar helpers
// ----------------------------------------------------------------------------------------------
// bf16 hi|lo split of one value (the operand format of the PF_PREC_BF16X3 GEMMs): hi = bf16_rne(x), lo = bf16_rne(x - hi).
PF_HD unsigned short pf_bf16_rne(float f) {
    union { float f; unsigned u; } v; v.f = f;
    const unsigned r = v.u + 0x7FFFu + ((v.u >> 16) & 1u);     // round to nearest even (finite inputs)
    return (unsigned short)(r >> 16);
}
PF_HD float pf_bf16_to_f32(unsigned short h) {
    union { float f; unsigned u; } v; v.u = (unsigned)h << 16;
    return v.f;
}
// element (row, ch) of a split twin [rows][lds chunks]{hi[32], lo[32]} (include/priorflow_hip.h, pf_conv_desc)
PF_HD void pf_split_put(void* base, long row, int lds, int ch, float x) {
    unsigned short* o = reinterpret_cast<unsigned short*>(base) + (row * lds + (ch >> 5)) * 64 + (ch & 31);
    const unsigned short hi = pf_bf16_rne(x);
    o[0] = hi;
    o[32] = pf_bf16_rne(x - pf_bf16_to_f32(hi));
}

// Python-style float remainder for b > 0 (`xgrid % W`, core/utils/utils.py:83; ATen: fmod, then +b when
// the signs differ), bit for bit, without the device's (long, loop-based) fmodf:
//   0 <= a < b            -> a                       (fmod is the identity)
//   -b < a < 0            -> a + b, ROUNDED          (ATen adds b to fmod = a; a tiny negative gives exactly b)
//   otherwise             -> a - b*floor(a/b): q from a*(1/b) can be off by one, r = fma(-q, b, a) is exact
//                            (for |a| >= b the true remainder lies on a's own ulp grid), one correction step.
// NaN / +-inf propagate to NaN like fmod.  b is an image width (an integer <= 2^20) here.
PF_HD float pf_pymod(float a, float b) {
    if (a >= 0.f && a < b) return a;
    if (a < 0.f && a > -b) return a + b;
    const float q = floorf(a * (1.f / b));
    float r = fmaf(-q, b, a);
    if (r < 0.f) r += b;
    else if (r >= b) r -= b;
    return r == 0.f ? copysignf(0.f, a) : r;       // fmod keeps the dividend's sign on an exact multiple
}

// pixel -> [-1,1] -> pixel round trip of `2x/(W-1)-1` (core/utils/utils.py:85-86) followed
// by grid_sample(align_corners=True)'s unnormalise, each step rounded to fp32.
PF_HD float pf_roundtrip(float p, int size) {
    const float s = (float)(size - 1);
    float pn = (2.f * p) / s;
    pn = pn - 1.f;
    return (pn + 1.f) * (s * 0.5f);
}

struct PfTaps {          // 4 bilinear taps of a zero-padded sample
    int idx[4];          // y*W+x (clamped into range; weight is 0 when out of bounds)
    float w[4];          // nw, ne, sw, se
};

// Zero-padded bilinear taps at pixel coords (x,y) of an H x W map
// (F.grid_sample bilinear / zeros / align_corners=True after the callers' normalisation).
PF_HD PfTaps pf_taps0(float x, float y, int H, int W) {
    PfTaps t;
    const float ix = pf_roundtrip(x, W), iy = pf_roundtrip(y, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ix - fx, wy = iy - fy;
    const float ex = 1.f - wx, ey = 1.f - wy;
    // NaN / huge coordinates: every comparison below is false -> all taps out of bounds
    const bool xin0 = (fx >= 0.f) && (fx <= (float)(W - 1));
    const bool xin1 = (fx >= -1.f) && (fx <= (float)(W - 2));
    const bool yin0 = (fy >= 0.f) && (fy <= (float)(H - 1));
    const bool yin1 = (fy >= -1.f) && (fy <= (float)(H - 2));
    const int x0 = xin0 ? (int)fx : 0, x1 = xin1 ? (int)fx + 1 : 0;
    const int y0 = yin0 ? (int)fy : 0, y1 = yin1 ? (int)fy + 1 : 0;
    t.idx[0] = y0 * W + x0; t.w[0] = (xin0 && yin0) ? ey * ex : 0.f;
    t.idx[1] = y0 * W + x1; t.w[1] = (xin1 && yin0) ? ey * wx : 0.f;
    t.idx[2] = y1 * W + x0; t.w[2] = (xin0 && yin1) ? wy * ex : 0.f;
    t.idx[3] = y1 * W + x1; t.w[3] = (xin1 && yin1) ? wy * wx : 0.f;
    return t;
}

PF_HD float pf_apply(const PfTaps& t, const float* img) {
    float acc = img[t.idx[0]] * t.w[0];
    acc = acc + img[t.idx[1]] * t.w[1];
    acc = acc + img[t.idx[2]] * t.w[2];
    acc = acc + img[t.idx[3]] * t.w[3];
    return acc;
}
// same, strided (channel-last maps: element (pixel, c) at img[pixel*ld + c])
PF_HD float pf_apply_ld(const PfTaps& t, const float* img, long ld) {
    float acc = img[(long)t.idx[0] * ld] * t.w[0];
    acc = acc + img[(long)t.idx[1] * ld] * t.w[1];
    acc = acc + img[(long)t.idx[2] * ld] * t.w[2];
    acc = acc + img[(long)t.idx[3] * ld] * t.w[3];
    return acc;
}

// Paired form of the zero-padded taps for PLANAR maps: the two x-corners of a bilinear sample are
// adjacent floats, so a sample is two 8-byte loads (rows y0, y1) instead of four 4-byte gathers.
// The pair starts at column xb = clamp(floor(x), 0, W-2), always inside the row; the corner values
// are picked out of it (x0 = W-1 sits in the pair's second slot, x1 = 0 in its first).  Weights,
// rows and the order of the four products are those of pf_taps0 / pf_apply, so results are identical.
// The DCCL lookup is bound by the number of gather instructions (16 per tap before, 8 now).
struct PfPair { float a, b; };
struct PfQuad { float a, b, c, d; };
PF_HD PfQuad pf_load4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float pf_f4u __attribute__((ext_vector_type(4), aligned(4)));     // dword-aligned 16-byte load
    const pf_f4u v = *reinterpret_cast<const pf_f4u*>(p);
    return PfQuad{v.x, v.y, v.z, v.w};
#else
    return PfQuad{p[0], p[1], p[2], p[3]};
#endif
}
PF_HD PfPair pf_load2(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float pf_f2u __attribute__((ext_vector_type(2), aligned(4)));     // dword-aligned 8-byte load
    const pf_f2u v = *reinterpret_cast<const pf_f2u*>(p);
    return PfPair{v.x, v.y};
#else
    return PfPair{p[0], p[1]};
#endif
}
struct PfTaps2 {
    int r0, r1;          // y0*W + xb, y1*W + xb
    bool s0hi, s1lo;     // x0 value = second slot of the pair / x1 value = first slot
    float w[4];          // nw, ne, sw, se (0 when out of bounds)
};
PF_HD PfTaps2 pf_taps0v(float x, float y, int H, int W) {      // W >= 2
    PfTaps2 t;
    const float ix = pf_roundtrip(x, W), iy = pf_roundtrip(y, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ix - fx, wy = iy - fy;
    const float ex = 1.f - wx, ey = 1.f - wy;
    const bool xin0 = (fx >= 0.f) && (fx <= (float)(W - 1));
    const bool xin1 = (fx >= -1.f) && (fx <= (float)(W - 2));
    const bool yin0 = (fy >= 0.f) && (fy <= (float)(H - 1));
    const bool yin1 = (fy >= -1.f) && (fy <= (float)(H - 2));
    // NaN / huge coordinates: every comparison is false -> xb = 0, all weights 0
    const int xb = (fx >= 0.f) ? ((fx <= (float)(W - 2)) ? (int)fx : W - 2) : 0;
    const int y0 = yin0 ? (int)fy : 0, y1 = yin1 ? (int)fy + 1 : 0;
    t.r0 = y0 * W + xb; t.r1 = y1 * W + xb;
    t.s0hi = fx == (float)(W - 1);
    t.s1lo = fx == -1.f;
    t.w[0] = (xin0 && yin0) ? ey * ex : 0.f;
    t.w[1] = (xin1 && yin0) ? ey * wx : 0.f;
    t.w[2] = (xin0 && yin1) ? wy * ex : 0.f;
    t.w[3] = (xin1 && yin1) ? wy * wx : 0.f;
    return t;
}
PF_HD float pf_apply_pairs(const PfTaps2& t, const PfPair p0, const PfPair p1) {
    float acc = (t.s0hi ? p0.b : p0.a) * t.w[0];
    acc = acc + (t.s1lo ? p0.a : p0.b) * t.w[1];
    acc = acc + (t.s0hi ? p1.b : p1.a) * t.w[2];
    acc = acc + (t.s1lo ? p1.a : p1.b) * t.w[3];
    return acc;
}
PF_HD float pf_apply_v(const PfTaps2& t, const float* img) {
    const PfPair p0 = pf_load2(img + t.r0), p1 = pf_load2(img + t.r1);
    float acc = (t.s0hi ? p0.b : p0.a) * t.w[0];
    acc = acc + (t.s1lo ? p0.a : p0.b) * t.w[1];
    acc = acc + (t.s0hi ? p1.b : p1.a) * t.w[2];
    acc = acc + (t.s1lo ? p1.a : p1.b) * t.w[3];
    return acc;
}

// True-wrap (x) / clamp (y) bilinear taps (core/utils/my_cycle_sample.py:31-60).
struct PfWrapTaps { int ia, ib, ic, id; float wa, wb, wc, wd; };
PF_HD PfWrapTaps pf_wraptaps(float gx, float gy, int H, int W) {
    PfWrapTaps t;
    gx = pf_pymod(gx, (float)W);
    float fx = floorf(gx), fy = floorf(gy);
    const float xw = gx - fx, yw = gy - fy;
    // keep the integer conversion defined for NaN / huge values
    if (!(fx >= 0.f && fx <= (float)W)) fx = 0.f;
    if (!(fy >= -1.0e6f)) fy = -1.0e6f;
    if (!(fy <= 1.0e6f)) fy = 1.0e6f;
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 % W; x1 = x1 % W;                      // x0 >= 0 here
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    t.ia = y0 * W + x0; t.ib = y1 * W + x0; t.ic = y0 * W + x1; t.id = y1 * W + x1;
    t.wa = (1.f - xw) * (1.f - yw);
    t.wb = (1.f - xw) * yw;
    t.wc = xw * (1.f - yw);
    t.wd = xw * yw;
    return t;
}
PF_HD float pf_wrapmix(const PfWrapTaps& t, float a, float b, float c, float d) {
    float acc = t.wa * a + t.wb * b;
    acc = acc + t.wc * c;
    acc = acc + t.wd * d;
    return acc;
}
// seam un-wrapping of the m-channel (core/utils/my_cycle_sample.py:82-97)
PF_HD float pf_unwrap_m(float anchor, float v, float W) {
    float t = (v - anchor) + W * 0.5f;
    t = pf_pymod(t, W);
    return (anchor + t) - W * 0.5f;
}

// ----------------------------------------------------------------------------------------------
// K8: sample grid  (core/utils/projection_prim_ortho.py:432-443 and helpers)
// ----------------------------------------------------------------------------------------------
struct PfGridArgs { float* grid; int H, W; float R[9]; };
PF_HD float pf_nudge(float t) {              // diverge_zero, :69-74
    const float eps = 1e-6f;
    const float sgn = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
    return (fabsf(t) < eps) ? t + sgn * eps : t;
}
PF_HD void pf_sample_grid_elem(long idx, const PfGridArgs& a) {
    const float PI = 3.14159274101257324f;       // float32(np.pi)
    const float TWO_PI = 6.28318548202514648f;   // float32(2*np.pi)
    const int m = (int)(idx % a.W), n = (int)(idx / a.W);
    float u = ((float)m + 0.5f) / (float)a.W;
    float theta = ((u - 0.5f) * 2.f) * PI;
    float v = ((float)n + 0.5f) / (float)a.H;
    float phi = (0.5f - v) * PI;
    const float cp = cosf(phi);
    const float x = cp * cosf(theta), y = cp * sinf(theta), z = sinf(phi);
    const float* R = a.R;
    float xr = R[0] * x + R[1] * y; xr = xr + R[2] * z;
    float yr = R[3] * x + R[4] * y; yr = yr + R[5] * z;
    float zr = R[6] * x + R[7] * y; zr = zr + R[8] * z;
    const float phi2 = asinf(zr);
    const float theta2 = atan2f(pf_nudge(yr), pf_nudge(xr));
    float m2 = theta2 / TWO_PI + 0.5f;
    m2 = m2 * (float)a.W - 0.5f;
    float n2 = 0.5f - phi2 / PI;
    n2 = n2 * (float)a.H - 0.5f;
    const long hw = (long)a.H * a.W;
    a.grid[idx] = m2;
    a.grid[hw + idx] = n2;
}

// ----------------------------------------------------------------------------------------------
// K7: img_rotate, NCHW (core/utils/projection_prim_ortho.py:507-514, :119-135)
// ----------------------------------------------------------------------------------------------
struct PfImgRotArgs { const float* img; const float* grid; float* out; int B, C, H, W; };
PF_HD void pf_img_rotate_elem(long idx, const PfImgRotArgs& a) {
    const long hw = (long)a.H * a.W;
    const long pix = idx % hw;
    const long bc = idx / hw;
    const float gx = pf_pymod(a.grid[pix], (float)a.W);
    const float gy = a.grid[hw + pix];
    const PfTaps t = pf_taps0(gx, gy, a.H, a.W);
    a.out[idx] = pf_apply(t, a.img + bc * hw);
}

// ----------------------------------------------------------------------------------------------
// input normalisation 2 * (image / 255.0) - 1.0 of both images (core/prior_raft.py:121-122), written straight into the
// encoders' batches: image1 -> f1 and c1 (feature and context batch), image2 -> f2
// ----------------------------------------------------------------------------------------------
struct PfNormImgArgs { const float* image1; const float* image2; float* f1; float* f2; float* c1; };
PF_HD float pf_norm255(float v) { return 2.f * (v / 255.0f) - 1.0f; }
PF_HD void pf_normalise_images_elem(long idx, const PfNormImgArgs& a) {
    const float v1 = pf_norm255(a.image1[idx]);
    a.f1[idx] = v1;
    if (a.c1) a.c1[idx] = v1;
    a.f2[idx] = pf_norm255(a.image2[idx]);
}

// ----------------------------------------------------------------------------------------------
// The whole input stage in one pass over the pixels (round 6; core/prior_raft.py:121-127): normalise both images and resample
// them into view B, written straight into the encoders' batches
//     f = [im1 | im2 | im1_B | im2_B]   ([4B,3,H,W])        c = [im1 | im1_B]   ([2B,3,H,W], optional)
// A rotated value is sum_q w_q * norm(raw[tap q]): the values pf_img_rotate reads from the normalised image, the same products
// in the same order -- bit-identical to pf_normalise_images + pf_img_rotate (+ the copy of im1_B into c) --, one launch instead of
// three at the head of every forward, where nothing else runs.
// ----------------------------------------------------------------------------------------------
struct PfPrepImgArgs { const float* image1; const float* image2; const float* grid; float* f; float* c; int B, H, W; };
PF_HD void pf_prepare_images_elem(long idx, const PfPrepImgArgs& a) {   // idx over B*3*H*W: one plane position of both images
    const long hw = (long)a.H * a.W;
    const long plane = (idx / hw) * hw;
    long pix = idx % hw;
    // 64 consecutive indices = an 8 x 8 tile of pixels, not 64 pixels of a row: view B is view A turned by 90 degrees, a row
    // segment of B is a column-like curve in A -- up to 64 cache lines per gather instruction, the whole launch bound by the
    // texture cache's tag rate (31 us) --, while an 8 x 8 tile samples an ~8 x 8 region (which thread takes which pixel
    // changes nothing in the results)
    // (8 x 8: 24 us; 4 x 16 and 2 x 32: 27 us; 16 x 4: 34 us; rows: 31 us -- profiles/r6_ab_head_of_forward.txt)
    if (!((a.H | a.W) & 7)) {
        const long tile = pix >> 6;
        const int in = (int)(pix & 63), tpr = a.W >> 3;
        pix = ((tile / tpr) * 8 + (in >> 3)) * a.W + (tile % tpr) * 8 + (in & 7);
    }
    const float gx = pf_pymod(a.grid[pix], (float)a.W);
    const float gy = a.grid[hw + pix];
    const PfTaps t = pf_taps0(gx, gy, a.H, a.W);
    const long img = (long)a.B * 3 * hw;            // elements of one batch of B three-channel images
    const float* p1 = a.image1 + plane;
    const float* p2 = a.image2 + plane;
    // the ten loads first, then the arithmetic, then the six stores (a load behind a possibly aliasing store would wait for it)
    const float o1 = p1[pix], o2 = p2[pix];
    float t1[4], t2[4];
    for (int q = 0; q < 4; ++q) { t1[q] = p1[t.idx[q]]; t2[q] = p2[t.idx[q]]; }
    float r1 = pf_norm255(t1[0]) * t.w[0], r2 = pf_norm255(t2[0]) * t.w[0];      // pf_apply's products and order
    for (int q = 1; q < 4; ++q) { r1 = r1 + pf_norm255(t1[q]) * t.w[q]; r2 = r2 + pf_norm255(t2[q]) * t.w[q]; }
    const float v1 = pf_norm255(o1), v2 = pf_norm255(o2);
    const long o = plane + pix;
    a.f[o] = v1;
    a.f[img + o] = v2;
    a.f[2 * img + o] = r1;
    a.f[3 * img + o] = r2;
    if (a.c) { a.c[o] = v1; a.c[img + o] = r1; }
}

// ----------------------------------------------------------------------------------------------
// flow = coords1 - coords0, scattered to planar + up to two channel-last destinations
// (core/prior_raft.py:172,177; coords_grid core/utils/utils.py:98-101)
// ----------------------------------------------------------------------------------------------
struct PfDst { float* ptr; int ld; int c_off; };     // channel-last [B*N][ld] destination slice
struct PfFlowPrepArgs { const float* coords1; float* flow; PfDst d0, d1; int B, H, W; };
PF_HD void pf_store_dst2(const PfDst& d, long row, float u, float v) {
    if (d.ptr) { d.ptr[row * d.ld + d.c_off] = u; d.ptr[row * d.ld + d.c_off + 1] = v; }
}
PF_HD void pf_flow_prep_elem(long idx, const PfFlowPrepArgs& a) {   // idx over B*N
    const long N = (long)a.H * a.W;
    const long b = idx / N, n = idx % N;
    const float x = (float)(n % a.W), y = (float)(n / a.W);
    const float u = a.coords1[(b * 2 + 0) * N + n] - x;
    const float v = a.coords1[(b * 2 + 1) * N + n] - y;
    if (a.flow) { a.flow[(b * 2 + 0) * N + n] = u; a.flow[(b * 2 + 1) * N + n] = v; }
    pf_store_dst2(a.d0, idx, u, v);
    pf_store_dst2(a.d1, idx, u, v);
}

// ----------------------------------------------------------------------------------------------
// K6: flo_rotate (core/utils/projection_prim_ortho.py:531-546, :200-218, :234-244;
//     core/utils/my_cycle_sample.py:6-97)
// ----------------------------------------------------------------------------------------------
struct PfFloRotArgs {
    const float* flow;       // planar [B,2,N]
    const float* g_w2c;      // [2,N]
    const float* g_c2w;      // [2,N]
    float* out;              // planar [B,2,N] (may be null)
    PfDst d0, d1;
    int B, H, W;
};
// camera-frame flow F at pixel p of batch b
PF_HD void pf_flow_c_at(const PfFloRotArgs& a, long b, int p, float& f0, float& f1) {
    const long N = (long)a.H * a.W;
    const float Wf = (float)a.W;
    const float px = (float)(p % a.W), py = (float)(p / a.W);
    float ex = (px + a.flow[(b * 2 + 0) * N + p]) + 0.5f;
    ex = pf_pymod(ex, Wf) - 0.5f;
    float ey = py + a.flow[(b * 2 + 1) * N + p];
    ey = fminf(fmaxf(ey, -0.5f), (float)a.H - 0.5f);
    const PfWrapTaps t = pf_wraptaps(ex, ey, a.H, a.W);
    const float* g0 = a.g_w2c;
    const float* g1 = a.g_w2c + N;
    const float a0 = g0[t.ia];
    const float e0 = pf_wrapmix(t, a0, pf_unwrap_m(a0, g0[t.ib], Wf), pf_unwrap_m(a0, g0[t.ic], Wf),
                                pf_unwrap_m(a0, g0[t.id], Wf));
    const float e1 = pf_wrapmix(t, g1[t.ia], g1[t.ib], g1[t.ic], g1[t.id]);
    f0 = e0 - g0[p];
    f0 = pf_pymod(f0 + Wf * 0.5f, Wf) - Wf * 0.5f;     // u_clip
    f1 = e1 - g1[p];
}
PF_HD void pf_flo_rotate_elem(long idx, const PfFloRotArgs& a) {  // idx over B*N
    const long N = (long)a.H * a.W;
    const long b = idx / N, n = idx % N;
    const PfWrapTaps t = pf_wraptaps(a.g_c2w[n], a.g_c2w[N + n], a.H, a.W);
    float a0, a1, b0, b1, c0, c1, d0, d1;
    pf_flow_c_at(a, b, t.ia, a0, a1);
    pf_flow_c_at(a, b, t.ib, b0, b1);
    pf_flow_c_at(a, b, t.ic, c0, c1);
    pf_flow_c_at(a, b, t.id, d0, d1);
    const float u = pf_wrapmix(t, a0, b0, c0, d0);
    const float v = pf_wrapmix(t, a1, b1, c1, d1);
    if (a.out) { a.out[(b * 2 + 0) * N + n] = u; a.out[(b * 2 + 1) * N + n] = v; }
    pf_store_dst2(a.d0, idx, u, v);
    pf_store_dst2(a.d1, idx, u, v);
}

// ----------------------------------------------------------------------------------------------
// K3 + K4(a,b): own-view and raw cross-view lookups (core/corr.py:113-137)
// ----------------------------------------------------------------------------------------------
struct PfLookupArgs {
    const float* coords;            // planar [B,2,N]
    const float* own[PF_CORR_LEVELS];    // level i: [B*N][H_i*W_i]
    const float* other[PF_CORR_LEVELS];
    const float* g_w2c;             // [2,N]
    const float* g_il;              // optional: the same grid interleaved [N][2] (x, y per pixel), or null
    float* own_out;                 // channel-last [B*N][ld]
    float* raw_out;                 // channel-last [B*N][ld]
    int B, H, W, ld;
};
// One call = PF_LOOKUP_TPT consecutive taps (same level, same slow index a): their gather chains
// (coords -> grid taps -> other-volume taps) are independent, so the loads of all of them are in
// flight together.  Timing-only ablations (round 1; profiles/microbench_lookup.py is the harness) showed the kernel bound by the
// NUMBER of gather instructions, cache-resident ones included (no own-window loads -7 us, no grid loads -9 us,
// no other-volume loads -4 us, no stores -3 us of 32 us).  Consecutive taps of a column share a row (the lower
// row of tap j is the upper row of tap j+1), so a row pair is loaded once and reused when its ADDRESS matches --
// same address, same value: results are bit-identical to per-tap loads.  Own window and the two grid components:
// 4 loads per 3 taps instead of 6 each.  Measured in one call (round-1 A/B): 32.8 us without
// reuse, 27.4 us with it at 3 taps per call, 43 us at 9 taps per call (a whole column: too few, too fat threads).
#ifndef PF_LOOKUP_TPT
#define PF_LOOKUP_TPT 3
#endif
PF_HD void pf_lookup_elem(long idx, const PfLookupArgs& a) {  // idx over B*N*(324/TPT)
    const long N = (long)a.H * a.W;
    const int per_row = PF_CORR_CH / PF_LOOKUP_TPT;
    const int k0 = (int)(idx % per_row) * PF_LOOKUP_TPT;
    const long row = idx / per_row;                    // b*N + n
    const long b = row / N, n = row % N;
    const int lvl = k0 / PF_TAPS, tap0 = k0 % PF_TAPS;
    const int ta = tap0 / 9, tb0 = tap0 % 9;           // slow axis a offsets x (core/corr.py:120-126)
    const int Hl = a.H >> lvl, Wl = a.W >> lvl;
    const float inv = 1.f / (float)(1 << lvl);         // coords / 2**i : exact
    const float cx = a.coords[(b * 2 + 0) * N + n] * inv + (float)(ta - PF_CORR_RADIUS);
    const float cy0 = a.coords[(b * 2 + 1) * N + n] * inv;
    const long lsz = (long)Hl * Wl;
    const float* own = a.own[lvl] + row * lsz;
    const float* oth = a.other[lvl] + row * lsz;
    // own view: x wrapped mod W_i, zero padded.  Cross view: level-i coordinates index the LEVEL-0
    // grid (core/corr.py:132-133), and the result indexes row n of the OTHER branch's volume (:135-136)
    const float xo = pf_pymod(cx, (float)Wl), xg = pf_pymod(cx, (float)a.W);
    float gx[PF_LOOKUP_TPT], gy[PF_LOOKUP_TPT], vo[PF_LOOKUP_TPT];
    int ro = -1, rg = -1;                              // addresses of the row pairs carried from the previous tap
    PfPair po = {0.f, 0.f}, pgx = {0.f, 0.f}, pgy = {0.f, 0.f};
    for (int j = 0; j < PF_LOOKUP_TPT; ++j) {
        const float cy = cy0 + (float)(tb0 + j - PF_CORR_RADIUS);
        const PfTaps2 t = pf_taps0v(xo, cy, Hl, Wl);
#ifdef PF_ABL_NO_OWN          // timing-only ablations; never defined in the product build
        vo[j] = t.w[0] + cy;
#else
        {
            const PfPair p0 = (t.r0 == ro) ? po : pf_load2(own + t.r0);
            const PfPair p1 = pf_load2(own + t.r1);
            vo[j] = pf_apply_pairs(t, p0, p1);
            ro = t.r1; po = p1;
        }
#endif
        const PfTaps2 tg = pf_taps0v(xg, cy, a.H, a.W);
#ifdef PF_ABL_NO_GRID
        gx[j] = xg + tg.w[0]; gy[j] = cy + tg.w[1];
#else
        {
            const bool hit = tg.r0 == rg;
            PfPair x0 = pgx, y0 = pgy, x1, y1;
            if (a.g_il) {
                // interleaved grid: the row pair of BOTH components is one 16-byte load {x(xb), y(xb), x(xb+1), y(xb+1)}
                if (!hit) { const PfQuad q = pf_load4(a.g_il + 2 * (long)tg.r0); x0 = PfPair{q.a, q.c}; y0 = PfPair{q.b, q.d}; }
                const PfQuad q = pf_load4(a.g_il + 2 * (long)tg.r1);
                x1 = PfPair{q.a, q.c}; y1 = PfPair{q.b, q.d};
            } else {
                if (!hit) { x0 = pf_load2(a.g_w2c + tg.r0); y0 = pf_load2(a.g_w2c + N + tg.r0); }
                x1 = pf_load2(a.g_w2c + tg.r1); y1 = pf_load2(a.g_w2c + N + tg.r1);
            }
            gx[j] = pf_apply_pairs(tg, x0, x1);
            gy[j] = pf_apply_pairs(tg, y0, y1);
            rg = tg.r1; pgx = x1; pgy = y1;
        }
#endif
    }
    for (int j = 0; j < PF_LOOKUP_TPT; ++j) {
        const PfTaps2 t = pf_taps0v(pf_pymod(gx[j], (float)Wl), gy[j], Hl, Wl);
#ifndef PF_ABL_NO_STORE
        a.own_out[row * a.ld + k0 + j] = vo[j];
#endif
#ifdef PF_ABL_NO_OTH
        a.raw_out[row * a.ld + k0 + j] = t.w[0] + t.w[3] + vo[j];
#elif defined(PF_ABL_NO_STORE)
        if (pf_apply_v(t, oth) + vo[j] == 123.456f) a.raw_out[row * a.ld + k0 + j] = 1.f;
#else
        a.raw_out[row * a.ld + k0 + j] = pf_apply_v(t, oth);
#endif
    }
}

// K4(c): rotate the raw cross-view lookup back and add the own-view lookup
// (core/corr.py:138; core/prior_raft.py:187-188).
struct PfCombineArgs {
    const float* own; const float* raw;   // channel-last [B*N][ld]
    const float* g_back;                  // [2,N]
    float* out;                           // channel-last [B*N][ld_out]
    int B, H, W, ld, ld_out;
};
PF_HD void pf_combine_elem(long idx, const PfCombineArgs& a) {  // idx over B*N*324
    const long N = (long)a.H * a.W;
    const int k = (int)(idx % PF_CORR_CH);
    const long row = idx / PF_CORR_CH;
    const long b = row / N, n = row % N;
    const PfTaps t = pf_taps0(pf_pymod(a.g_back[n], (float)a.W), a.g_back[N + n], a.H, a.W);
    const float cross = pf_apply_ld(t, a.raw + b * N * a.ld + k, a.ld);
    a.out[row * a.ld_out + k] = a.own[row * a.ld + k] + cross;
}

// ----------------------------------------------------------------------------------------------
// Backward of K3 + K4 (SURVEY.md 8f-3 groundwork).  coords are detached every iteration
// (core/prior_raft.py:171,176) and the sample grids are constants, so the only gradients are those of the
// sampled VALUES: the bilinear weights of the forward, scattered.
//   combine:  corr = own + rotate_back(raw)  =>  d_own = d_corr (no kernel), d_raw[idx_k(p)] += w_k(p) d_corr[p]
//   lookup:   own[n][k] = sum_j w_j pyr_own[lvl][n][idx_j]   =>  g_own[lvl][n][idx_j]   += w_j d_own[n][k]
//             raw[n][k] = sum_j w_j pyr_oth[lvl][n][idx_j'] =>  g_other[lvl][n][idx_j'] += w_j d_raw[n][k]
// Gradients are ACCUMULATED (fp32 atomics; callers zero the buffers once per step).
// ----------------------------------------------------------------------------------------------
struct PfCombineBwdArgs {
    const float* d_corr;                  // channel-last [B*N][ld_in]
    const float* g_back;                  // [2,N]
    float* d_raw;                         // channel-last [B*N][ld]  (accumulated)
    int B, H, W, ld_in, ld;
};
PF_HD void pf_combine_bwd_elem(long idx, const PfCombineBwdArgs& a) {   // idx over B*N*324
    const long N = (long)a.H * a.W;
    const int k = (int)(idx % PF_CORR_CH);
    const long row = idx / PF_CORR_CH;
    const long b = row / N, n = row % N;
    const PfTaps t = pf_taps0(pf_pymod(a.g_back[n], (float)a.W), a.g_back[N + n], a.H, a.W);
    const float g = a.d_corr[row * a.ld_in + k];
    float* base = a.d_raw + b * N * a.ld + k;
    for (int j = 0; j < 4; ++j)
        if (t.w[j] != 0.f) PF_ATOMIC_ADD(base + (long)t.idx[j] * a.ld, g * t.w[j]);
}
// ----------------------------------------------------------------------------------------------
// SepConvGRU gate backward (core/update.py:46-60; one half-step):
//   z = sigmoid(az), r = sigmoid(ar), q = tanh(aq(cat[r*h, x])), h' = (1 - z) * h + z * q
// stage Q  (before the dgrad of convq):  dq_pre = dh' * z * (1 - q^2),  dz = dh' * q - dh' * h,  dh = dh' * (1 - z)
// stage ZR (after it, d_rh = gradient of r*h): dz_pre = dz * (1 - z) * z,  dr_pre = (d_rh * h) * (1 - r) * r,
//                                              dh += d_rh * r
// (the products are written in the order of torch's sigmoid_backward / tanh_backward / mul backward).
// dzr_pre is the [z | r] gradient of the fused z|r convolution (columns 0..C-1 and C..2C-1).
// ----------------------------------------------------------------------------------------------
struct PfGruQBwdArgs {
    const float* dhn; const float* z; const float* q; const float* h;
    float* dq_pre; float* dz; float* dh;
    int ld_dhn, ld_z, ld_q, ld_h, ld_dq, ld_dz, ld_dh, C;
};
PF_HD void pf_gru_q_bwd_elem(long idx, const PfGruQBwdArgs& a) {          // idx over rows*C
    const long row = idx / a.C; const int c = (int)(idx % a.C);
    const float g = a.dhn[row * a.ld_dhn + c], z = a.z[row * a.ld_z + c], q = a.q[row * a.ld_q + c], h = a.h[row * a.ld_h + c];
    a.dq_pre[row * a.ld_dq + c] = (g * z) * (1.f - q * q);
    a.dz[row * a.ld_dz + c] = g * q - g * h;
    a.dh[row * a.ld_dh + c] = g * (1.f - z);
}
struct PfGruZrBwdArgs {
    const float* dz; const float* d_rh; const float* z; const float* r; const float* h;
    float* dzr_pre; float* dh;
    int ld_dz, ld_drh, ld_z, ld_r, ld_h, ld_dzr, ld_dh, C;
};
PF_HD void pf_gru_zr_bwd_elem(long idx, const PfGruZrBwdArgs& a) {        // idx over rows*C
    const long row = idx / a.C; const int c = (int)(idx % a.C);
    const float dz = a.dz[row * a.ld_dz + c], drh = a.d_rh[row * a.ld_drh + c];
    const float z = a.z[row * a.ld_z + c], r = a.r[row * a.ld_r + c], h = a.h[row * a.ld_h + c];
    a.dzr_pre[row * a.ld_dzr + c] = (dz * (1.f - z)) * z;
    a.dzr_pre[row * a.ld_dzr + a.C + c] = ((drh * h) * (1.f - r)) * r;
    a.dh[row * a.ld_dh + c] = a.dh[row * a.ld_dh + c] + drh * r;
}
// ----------------------------------------------------------------------------------------------
// Gradient of a SepConvGRU's input x = [inp (C) | out (wout) | flows] (core/update.py:155,133) after both half-steps
// (training loop node, prior-flow_amd/train_loop.py): dx = f1 + f2 (the two half-steps' data gradients, columns of wider
// scratch rows); d_inp += dx[:, :C] (inp feeds every iteration); d_out = dx[:, C:C+wout] where out > 0 else 0 (out is the
// ReLU output of the motion encoder's last convolution, stored in x itself).  One element = one (row, column < C + wout).
struct PfGruDxArgs {
    const float* f1; const float* f2; const float* x; float* d_inp; float* d_out;
    int ld_f1, ld_f2, ld_x, ld_dinp, ld_dout, C, wout;
};
PF_HD void pf_gru_dx_finish_elem(long idx, const PfGruDxArgs& a) {
    const int w = a.C + a.wout;
    const long row = idx / w; const int c = (int)(idx % w);
    const float dx = a.f1[row * a.ld_f1 + c] + a.f2[row * a.ld_f2 + c];
    if (c < a.C) a.d_inp[row * a.ld_dinp + c] = a.d_inp[row * a.ld_dinp + c] + dx;
    else a.d_out[row * a.ld_dout + (c - a.C)] = a.x[row * a.ld_x + c] > 0.f ? dx : 0.f;
}
// ----------------------------------------------------------------------------------------------
// Backward of  y = act(x * s[b,c] + t[b,c])  over channel-last rows [B*Np][C] (core/extractor.py:112-147):
// s, t are the scale / shift of pf_channel_stats (InstanceNorm: s = rstd, t = -mean * rstd) or the folded
// BatchNorm(eval) affine.  With xh = x*s + t and g = dy masked by the ReLU (xh > 0):
//   InstanceNorm:     dx = s * (g - mean_p(g) - xh * mean_p(g * xh))      (mean over the Np pixels of (b, c))
//   BatchNorm (eval): dx = s * g
// Three elementwise passes for InstanceNorm: partial sums over pixel chunks (fp64, fixed order), their
// reduction to the two means, the apply pass.
// ----------------------------------------------------------------------------------------------
struct PfNormBwdArgs {
    const float* dy; const float* x; const float* s; const float* t;
    double* part;          // [B][nblk][C][2]
    float* coef;           // [B][C][2]: mean(g), mean(g*xh)
    float* dx;
    int B, Np, C, nblk, relu, instance;
};
PF_HD void pf_norm_bwd_partial_elem(long idx, const PfNormBwdArgs& a) {     // idx over B*nblk*C
    const int c = (int)(idx % a.C);
    const long r = idx / a.C;
    const int blk = (int)(r % a.nblk); const long b = r / a.nblk;
    const int chunk = (a.Np + a.nblk - 1) / a.nblk;
    const int p0 = blk * chunk, p1 = (p0 + chunk < a.Np) ? p0 + chunk : a.Np;
    const float s = a.s[b * a.C + c], t = a.t[b * a.C + c];
    double s1 = 0.0, s2 = 0.0;
    int p = p0;
    for (; p + 4 <= p1; p += 4) {          // four pixels' loads in flight together; the sums keep their pixel order
        float xv[4], gv[4];
        for (int u = 0; u < 4; ++u) {
            const long e = (b * a.Np + p + u) * a.C + c;
            xv[u] = a.x[e]; gv[u] = a.dy[e];
        }
        for (int u = 0; u < 4; ++u) {
            const float xh = xv[u] * s + t;
            const float g = (a.relu && !(xh > 0.f)) ? 0.f : gv[u];
            s1 += (double)g; s2 += (double)g * (double)xh;
        }
    }
    for (; p < p1; ++p) {
        const long e = (b * a.Np + p) * a.C + c;
        const float xh = a.x[e] * s + t;
        const float g = (a.relu && !(xh > 0.f)) ? 0.f : a.dy[e];
        s1 += (double)g; s2 += (double)g * (double)xh;
    }
    a.part[idx * 2] = s1; a.part[idx * 2 + 1] = s2;
}
PF_HD void pf_norm_bwd_final_elem(long idx, const PfNormBwdArgs& a) {       // idx over B*C
    const int c = (int)(idx % a.C); const long b = idx / a.C;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < a.nblk; ++k) {
        const double* q = a.part + ((b * a.nblk + k) * a.C + c) * 2;
        s1 += q[0]; s2 += q[1];
    }
    a.coef[idx * 2] = (float)(s1 / (double)a.Np); a.coef[idx * 2 + 1] = (float)(s2 / (double)a.Np);
}
PF_HD void pf_norm_bwd_apply_elem(long idx, const PfNormBwdArgs& a) {       // idx over B*Np*C
    const int c = (int)(idx % a.C);
    const long b = idx / ((long)a.Np * a.C);
    const float s = a.s[b * a.C + c], t = a.t[b * a.C + c];
    const float xh = a.x[idx] * s + t;
    const float g = (a.relu && !(xh > 0.f)) ? 0.f : a.dy[idx];
    if (a.instance) {
        const float m1 = a.coef[(b * a.C + c) * 2], m2 = a.coef[(b * a.C + c) * 2 + 1];
        a.dx[idx] = s * ((g - m1) - xh * m2);
    } else {
        a.dx[idx] = s * g;
    }
}
// ResidualBlock tail on the training tape (core/extractor.py:47): out = relu(x + y), and its backward dx = dy = (out > 0) ? g : 0
// (one launch each; 4 consecutive floats per call).
struct PfAddReluArgs { const float* x; const float* y; float* out; long n; };
PF_HD void pf_add_relu_elem(long idx, const PfAddReluArgs& a) {        // idx over ceil(n / 4)
    for (long i = idx * 4; i < idx * 4 + 4 && i < a.n; ++i) {
        const float v = a.x[i] + a.y[i];
        a.out[i] = v > 0.f ? v : 0.f;
    }
}
PF_HD void pf_relu_mask_elem(long idx, const PfAddReluArgs& a) {       // x = g, y = the forward output, out = masked g
    for (long i = idx * 4; i < idx * 4 + 4 && i < a.n; ++i) a.out[i] = a.y[i] > 0.f ? a.x[i] : 0.f;
}

// ----------------------------------------------------------------------------------------------
// BatchNorm2d with frozen statistics (freeze_bn, train_flow.py:107-108; the context encoder's norm, core/extractor.py:114-115),
// optionally with the ReLU behind it, on channel-last rows [rows][C]:
//   forward   out = [relu]( x*s + t ),  s = gamma * rsqrt(var + eps),  t = beta - mean * s
//   backward  g_m = relu ? (x*s + t > 0 ? g : 0) : g;   dx = s * g_m;   d gamma = sum g_m * xhat,  d beta = sum g_m,
//             xhat = (x - mean) * rsqrt(var + eps);  the sums: fp64 partials per (row block, channel), fixed order.
// ----------------------------------------------------------------------------------------------
struct PfBnArgs {
    const float* x; const float* dy; const float* gamma; const float* beta; const float* mean; const float* var;
    float* out;            // forward: out; backward: dx
    double* part;          // [nblk][C][2]
    float* dgamma; float* dbeta;
    float eps; long rows; int C, nblk, relu, accumulate;
};
PF_HD void pf_bn_fwd_elem(long idx, const PfBnArgs& a) {              // idx over rows*C
    const int c = (int)(idx % a.C);
    const float s = a.gamma[c] * (1.f / sqrtf(a.var[c] + a.eps));
    const float y = a.x[idx] * s + (a.beta[c] - a.mean[c] * s);
    a.out[idx] = (a.relu && !(y > 0.f)) ? 0.f : y;
}
PF_HD void pf_bn_bwd_partial_elem(long idx, const PfBnArgs& a) {      // idx over nblk*C
    const int c = (int)(idx % a.C);
    const long blk = idx / a.C;
    const long chunk = (a.rows + a.nblk - 1) / a.nblk;
    const long p0 = blk * chunk, p1 = (p0 + chunk < a.rows) ? p0 + chunk : a.rows;
    const float rstd = 1.f / sqrtf(a.var[c] + a.eps), mean = a.mean[c];
    const float s = a.gamma[c] * rstd, t = a.beta[c] - mean * s;
    double s1 = 0.0, s2 = 0.0;
    long p = p0;
    for (; p + 8 <= p1; p += 8) {          // eight rows' loads in flight together; the sums keep their row order
        float xv[8], gv[8];
        for (int u = 0; u < 8; ++u) { xv[u] = a.x[(p + u) * a.C + c]; gv[u] = a.dy[(p + u) * a.C + c]; }
        for (int u = 0; u < 8; ++u) {
            const float g = (a.relu && !(xv[u] * s + t > 0.f)) ? 0.f : gv[u];
            s1 += (double)g; s2 += (double)g * (double)((xv[u] - mean) * rstd);
        }
    }
    for (; p < p1; ++p) {
        const float xv = a.x[p * a.C + c];
        const float g = (a.relu && !(xv * s + t > 0.f)) ? 0.f : a.dy[p * a.C + c];
        s1 += (double)g; s2 += (double)g * (double)((xv - mean) * rstd);
    }
    a.part[idx * 2] = s1; a.part[idx * 2 + 1] = s2;
}
PF_HD void pf_bn_bwd_final_elem(long idx, const PfBnArgs& a) {        // idx over C
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < a.nblk; ++k) {
        const double* q = a.part + ((long)k * a.C + idx) * 2;
        s1 += q[0]; s2 += q[1];
    }
    if (a.accumulate) { a.dbeta[idx] += (float)s1; a.dgamma[idx] += (float)s2; }
    else { a.dbeta[idx] = (float)s1; a.dgamma[idx] = (float)s2; }
}
PF_HD void pf_bn_bwd_apply_elem(long idx, const PfBnArgs& a) {        // idx over rows*C
    const int c = (int)(idx % a.C);
    const float s = a.gamma[c] * (1.f / sqrtf(a.var[c] + a.eps));
    const float y = a.x[idx] * s + (a.beta[c] - a.mean[c] * s);
    a.out[idx] = (a.relu && !(y > 0.f)) ? 0.f : s * a.dy[idx];
}

struct PfLookupBwdArgs {
    const float* coords;                  // planar [B,2,N]
    const float* g_w2c;                   // [2,N]
    const float* d_own; const float* d_raw;      // channel-last [B*N][ld]
    float* g_own[PF_CORR_LEVELS];         // level i: [B*N][H_i*W_i]  (accumulated)
    float* g_other[PF_CORR_LEVELS];
    int B, H, W, ld;
    float* clear_raw;                     // NULL, or d_raw again: every value read is replaced by 0 (the next pf_dccl_combine_bwd scatters into it)
};
PF_HD void pf_lookup_bwd_elem(long idx, const PfLookupBwdArgs& a) {     // idx over B*N*324
    const long N = (long)a.H * a.W;
    const int k = (int)(idx % PF_CORR_CH);
    const long row = idx / PF_CORR_CH;
    const long b = row / N, n = row % N;
    const int lvl = k / PF_TAPS, tap = k % PF_TAPS;
    // Consecutive lanes take taps that are neighbours in X (channel = 81 lvl + 9 ta + tb with the slow axis a offsetting x,
    // core/corr.py:120-126): their four corners then fall into the same one or two 64-byte lines of a map row, and a wave's atomic
    // instruction is a few line requests instead of one per lane (channel order, tb fastest, walks down a column: every lane its
    // own line)
#ifdef PF_LKB_CHANNEL_ORDER
    const int ta = tap / 9, tb = tap % 9;
#else
    const int ta = tap % 9, tb = tap / 9;
#endif
    const int kk = lvl * PF_TAPS + ta * 9 + tb;       // the channel this thread scatters
    const int Hl = a.H >> lvl, Wl = a.W >> lvl;
    const float inv = 1.f / (float)(1 << lvl);
    const float cx = a.coords[(b * 2 + 0) * N + n] * inv + (float)(ta - PF_CORR_RADIUS);
    const float cy = a.coords[(b * 2 + 1) * N + n] * inv + (float)(tb - PF_CORR_RADIUS);
    const long lsz = (long)Hl * Wl;
    const float go = a.d_own[row * a.ld + kk], gr = a.d_raw[row * a.ld + kk];
    if (a.clear_raw) a.clear_raw[row * a.ld + kk] = 0.f;
    const PfTaps t = pf_taps0(pf_pymod(cx, (float)Wl), cy, Hl, Wl);
    float* own = a.g_own[lvl] + row * lsz;
    for (int j = 0; j < 4; ++j)
        if (t.w[j] != 0.f) PF_ATOMIC_ADD(own + t.idx[j], go * t.w[j]);
    // cross view: the level-i coordinates index the LEVEL-0 grid (core/corr.py:132-133)
    const PfTaps tg = pf_taps0(pf_pymod(cx, (float)a.W), cy, a.H, a.W);
    const float gx = pf_apply(tg, a.g_w2c), gy = pf_apply(tg, a.g_w2c + N);
    const PfTaps to = pf_taps0(pf_pymod(gx, (float)Wl), gy, Hl, Wl);
    float* oth = a.g_other[lvl] + row * lsz;
    for (int j = 0; j < 4; ++j)
        if (to.w[j] != 0.f) PF_ATOMIC_ADD(oth + to.idx[j], gr * to.w[j]);
}

// Backward of build_pyramid (core/corr.py:99-111: three F.avg_pool2d(2, stride 2), odd sizes floor): the dense
// volume gradient  dV[row][y][x] = g0 + g1[y/2][x/2]/4 + g2[y/4][x/4]/16 + g3[y/8][x/8]/64, a parent
// contributing only where it exists (y>>i < H>>i and the whole chain of windows was complete).  Written in
// place into the level-0 gradient; one pass.  (The feature gradients are then two plain GEMMs, dV f2 and dV^T f1.)
struct PfPyramidBwdArgs { float* g0; const float* g1; const float* g2; const float* g3; long rows; int H, W; };
PF_HD void pf_pyramid_bwd_elem(long idx, const PfPyramidBwdArgs& a) {    // idx over rows*H*W
    const int N = a.H * a.W;
    const long row = idx / N;
    const int n = (int)(idx % N), y = n / a.W, x = n % a.W;
    float v = a.g0[idx];
    int hy = a.H, hx = a.W, py = y, px = x;
    const float* lv[3] = {a.g1, a.g2, a.g3};
    float scale = 1.f;
    for (int i = 0; i < 3; ++i) {
        hy >>= 1; hx >>= 1; py >>= 1; px >>= 1;
        if (py >= hy || px >= hx) break;           // odd size: the last row / column has no parent (nor grand-parents)
        scale *= 0.25f;
        v = v + lv[i][row * ((long)hy * hx) + (long)py * hx + px] * scale;
    }
    a.g0[idx] = v;
}

// ----------------------------------------------------------------------------------------------
// K12: convex 8x upsampling (core/prior_raft.py:58-67); flow = coords1 - coords0
// ----------------------------------------------------------------------------------------------
struct PfUpsampleArgs {
    const float* coords1;   // planar [B,2,N]
    const float* mask;      // channel-last [B*N][ld], channel = 64k + 8i + j (already x0.25)
    float* out;             // NCHW [B,2,8H,8W]
    int B, H, W, ld;
};
PF_HD void pf_upsample_elem(long idx, const PfUpsampleArgs& a) {  // idx over B*8H*8W
    const int W8 = 8 * a.W, H8 = 8 * a.H;
    const long N = (long)a.H * a.W;
    const int X = (int)(idx % W8);
    const int Y = (int)((idx / W8) % H8);
    const long b = idx / ((long)W8 * H8);
    const int x = X >> 3, j = X & 7, y = Y >> 3, i = Y & 7;
    const float* mrow = a.mask + (b * N + (long)y * a.W + x) * a.ld + 8 * i + j;
    float lg[9];
    float mx = -INFINITY;
    for (int k = 0; k < 9; ++k) { lg[k] = mrow[64 * k]; mx = fmaxf(mx, lg[k]); }
    float den = 0.f;
    for (int k = 0; k < 9; ++k) { lg[k] = expf(lg[k] - mx); den = den + lg[k]; }
    float su = 0.f, sv = 0.f;
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        float fu = 0.f, fv = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {      // F.unfold zero padding
            const long p = (long)yy * a.W + xx;
            fu = 8.f * (a.coords1[(b * 2 + 0) * N + p] - (float)xx);
            fv = 8.f * (a.coords1[(b * 2 + 1) * N + p] - (float)yy);
        }
        const float wk = lg[k] / den;
        su = su + wk * fu;
        sv = sv + wk * fv;
    }
    const long plane = (long)W8 * H8;
    a.out[(b * 2 + 0) * plane + (long)Y * W8 + X] = su;
    a.out[(b * 2 + 1) * plane + (long)Y * W8 + X] = sv;
}

// Backward of K12 (autograd through upsample_flow, core/prior_raft.py:58-67), one fine pixel per call:
// with w = softmax_k(mask[64k + 8i + j]) and nb_k = 8 * flow at the k-th 3x3 neighbour (zero outside),
//   d_mask[64k + 8i + j] = w_k (s_k - sum_k' w_k' s_k'),  s_k = <g, nb_k>        (written; the caller's mask is
//                                                                                  0.25 * conv: scale upstream)
//   d_flow[neighbour k]  += 8 w_k g                                              (fp32 atomics; = d coords1)
struct PfUpsampleBwdArgs {
    const float* coords1;   // planar [B,2,N]
    const float* mask;      // channel-last [B*N][ld]
    const float* g;         // NCHW [B,2,8H,8W]: gradient of the upsampled flow
    float* d_mask;          // channel-last [B*N][ld_d]  (every one of the 576 columns is written)
    float* d_flow;          // planar [B,2,N]  (accumulated)
    int B, H, W, ld, ld_d;
};
PF_HD void pf_upsample_bwd_elem(long idx, const PfUpsampleBwdArgs& a) {  // idx over B*8H*8W
    const int W8 = 8 * a.W, H8 = 8 * a.H;
    const long N = (long)a.H * a.W;
    const int X = (int)(idx % W8);
    const int Y = (int)((idx / W8) % H8);
    const long b = idx / ((long)W8 * H8);
    const int x = X >> 3, j = X & 7, y = Y >> 3, i = Y & 7;
    const long prow = b * N + (long)y * a.W + x;
    const float* mrow = a.mask + prow * a.ld + 8 * i + j;
    const long plane = (long)W8 * H8;
    const float gu = a.g[(b * 2 + 0) * plane + (long)Y * W8 + X], gv = a.g[(b * 2 + 1) * plane + (long)Y * W8 + X];
    float w[9], sk[9];
    float mx = -INFINITY;
    for (int k = 0; k < 9; ++k) { w[k] = mrow[64 * k]; mx = fmaxf(mx, w[k]); }
    float den = 0.f;
    for (int k = 0; k < 9; ++k) { w[k] = expf(w[k] - mx); den = den + w[k]; }
    float dot = 0.f;
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        w[k] = w[k] / den;
        sk[k] = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
            const long p = (long)yy * a.W + xx;
            const float fu = 8.f * (a.coords1[(b * 2 + 0) * N + p] - (float)xx);
            const float fv = 8.f * (a.coords1[(b * 2 + 1) * N + p] - (float)yy);
            sk[k] = gu * fu + gv * fv;
            PF_ATOMIC_ADD(a.d_flow + (b * 2 + 0) * N + p, 8.f * w[k] * gu);
            PF_ATOMIC_ADD(a.d_flow + (b * 2 + 1) * N + p, 8.f * w[k] * gv);
        }
        dot = dot + w[k] * sk[k];
    }
    float* drow = a.d_mask + prow * a.ld_d + 8 * i + j;
    for (int k = 0; k < 9; ++k) drow[64 * k] = w[k] * (sk[k] - dot);
}

// coords1 += delta  (core/prior_raft.py:193,196); delta is channel-last [B*N][ld]
struct PfCoordsAddArgs { float* coords1; const float* delta; int B, N, ld; const float* src; };   // src: coords1 = src + delta (NULL: in place)
PF_HD void pf_coords_add_elem(long idx, const PfCoordsAddArgs& a) {   // idx over B*N
    const long b = idx / a.N, n = idx % a.N;
    const float* from = a.src ? a.src : a.coords1;
    a.coords1[(b * 2 + 0) * a.N + n] = from[(b * 2 + 0) * a.N + n] + a.delta[idx * a.ld + 0];
    a.coords1[(b * 2 + 1) * a.N + n] = from[(b * 2 + 1) * a.N + n] + a.delta[idx * a.ld + 1];
}

// ----------------------------------------------------------------------------------------------
// small direct convolution, channel-last, stride 1, "same" zero padding, optional ReLU.
// Used for the tiny-Cin layers (7x7 2->128, 3x3 8->32, 3x3 32->16; core/update.py:171-178,
// :87).  Weights packed [KH*KW][Cin][Cout].
// ----------------------------------------------------------------------------------------------
struct PfDirectConvArgs {
    const float* in; int ld_in, c_in_off, Cin;
    const float* w; const float* bias;
    float* out; int ld_out, c_out_off, Cout;
    int B, H, W, KH, KW, relu;      // H, W: OUTPUT map
    int stride, nchw, Hin, Win;     // input map = stride x output; nchw: input is [B,Cin,Hin,Win] planes
    void* out_split; int lds_out;   // optional split twin of `out` (same channel offset); `out` may then be null
};
PF_HD void pf_direct_conv_elem(long idx, const PfDirectConvArgs& a) {  // idx over B*N*Cout
    const long N = (long)a.H * a.W, Nin = (long)a.Hin * a.Win;
    const int co = (int)(idx % a.Cout);
    const long row = idx / a.Cout;
    const long b = row / N, n = row % N;
    const int y = (int)(n / a.W) * a.stride, x = (int)(n % a.W) * a.stride;
    const int ph = a.KH / 2, pw = a.KW / 2;
    float acc = 0.f;
    for (int kh = 0; kh < a.KH; ++kh) {
        const int yy = y + kh - ph;
        if (yy < 0 || yy >= a.Hin) continue;
        for (int kw = 0; kw < a.KW; ++kw) {
            const int xx = x + kw - pw;
            if (xx < 0 || xx >= a.Win) continue;
            const float* wp = a.w + ((long)(kh * a.KW + kw) * a.Cin) * a.Cout + co;
            for (int c = 0; c < a.Cin; ++c) {
                const float v = a.nchw ? a.in[(b * a.Cin + c) * Nin + (long)yy * a.Win + xx]
                                       : a.in[(b * Nin + (long)yy * a.Win + xx) * a.ld_in + a.c_in_off + c];
                acc = acc + v * wp[(long)c * a.Cout];
            }
        }
    }
    acc = acc + a.bias[co];
    if (a.relu) acc = fmaxf(acc, 0.f);
    if (a.out) a.out[row * a.ld_out + a.c_out_off + co] = acc;
    if (a.out_split) pf_split_put(a.out_split, row, a.lds_out, a.c_out_off + co, acc);
}

// ----------------------------------------------------------------------------------------------
// FlowHead.conv2 (3x3, C->2, core/update.py:10,13-14) fused with the coordinate update
// coords1 += delta_flow (core/prior_raft.py:193,196).  w packed [2][9][C]; x channel-last.
// Scalar statement (host/emu); the device kernel spreads C over a wavefront.
// ----------------------------------------------------------------------------------------------
struct PfFlowOutArgs {
    const float* x; int ld, C;
    const float* w; const float* bias;
    float* coords1;            // planar [B,2,N], updated in place
    float* delta;              // optional channel-last [B*N][ld_delta] copy of delta_flow
    int ld_delta;
    int B, H, W;
};
PF_HD void pf_flow_out_elem(long idx, const PfFlowOutArgs& a) {   // idx over B*N*2
    const long N = (long)a.H * a.W;
    const int o = (int)(idx & 1);
    const long row = idx >> 1;
    const long b = row / N, n = row % N;
    const int y = (int)(n / a.W), x = (int)(n % a.W);
    float acc = 0.f;
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
        const float* xp = a.x + (b * N + (long)yy * a.W + xx) * a.ld;
        const float* wp = a.w + ((long)o * 9 + t) * a.C;
        for (int c = 0; c < a.C; ++c) acc = acc + xp[c] * wp[c];
    }
    acc = acc + a.bias[o];
    if (a.delta) a.delta[row * a.ld_delta + o] = acc;
    a.coords1[(b * 2 + o) * N + n] += acc;
}

// ----------------------------------------------------------------------------------------------
// fp32 -> bf16 hi|lo split rows (operand format of the PF_PREC_BF16X3 GEMMs):
// out row = [C/32 chunks] x { bf16 hi[32], bf16 lo[32] }, hi = bf16_rne(x), lo = bf16_rne(x - hi).
// ----------------------------------------------------------------------------------------------
struct PfSplitArgs { const float* in; unsigned short* out; long rows; int C; };
PF_HD void pf_split_bf16_elem(long idx, const PfSplitArgs& a) {   // idx over rows*C/4
    const int c4n = a.C / 4;
    const int c = (int)(idx % c4n) * 4;
    const long row = idx / c4n;
    unsigned short* o = a.out + row * 2 * a.C + (c / 32) * 64 + (c % 32);
    for (int i = 0; i < 4; ++i) {
        const float x = a.in[row * a.C + c + i];
        const unsigned short hi = pf_bf16_rne(x);
        o[i] = hi;
        o[32 + i] = pf_bf16_rne(x - pf_bf16_to_f32(hi));
    }
}

// ----------------------------------------------------------------------------------------------
// Weight packing on the device (round 4).  nn.Conv2d weights [Cout][Cin][KH][KW] (one tensor, or two concatenated on Cout:
// the fused z|r convolution of a SepConvGRU half, core/update.py:48-49) -> the PF_PREC_BF16X3 operand of pf_conv2d:
// [Cout_pad][KH*KW][Cin_pad/32] x {bf16 hi[32], bf16 lo[32]}, zero padded, plus the padded bias vector.
//   mode 0: the forward convolution.
//   mode 1: the DATA-GRADIENT convolution dX = conv(dY, W'), W'[c][o][ky][kx] = W[o][(c + cin_rot) % Cin][KH-1-ky][KW-1-kx]:
//           its output channels are the forward input channels (rotated by cin_rot), its input channels the forward output
//           channels; no bias.
// A training step re-packs every convolution of the model after each optimizer step; with torch ops that was ~9 small
// kernels per pack (two fills, a permute copy, two bf16 casts, a subtract, a cat ...), ~800 launches per step.
// One element = one (packed output channel, tap, packed input channel).
// ----------------------------------------------------------------------------------------------
struct PfPackWArgs {
    const float* w0; const float* w1; const float* b0; const float* b1;
    unsigned short* dst_w; float* dst_b;
    int cout0, cout1, cin, kh, kw, mode, cin_rot, cout_pad, cin_pad;
};
PF_HD void pf_pack_conv_weights_elem(long idx, const PfPackWArgs& a) {
    const int taps = a.kh * a.kw;
    const int c = (int)(idx % a.cin_pad);
    const int tap = (int)((idx / a.cin_pad) % taps);
    const int o = (int)(idx / ((long)a.cin_pad * taps));
    const int cout = a.cout0 + a.cout1;
    float v = 0.f;
    if (a.mode == 0) {
        if (o < cout && c < a.cin) {
            const float* w = o < a.cout0 ? a.w0 + (long)o * a.cin * taps : a.w1 + (long)(o - a.cout0) * a.cin * taps;
            v = w[(long)c * taps + tap];
        }
        if (a.dst_b != nullptr && c == 0 && tap == 0)
            a.dst_b[o] = o < a.cout0 ? (a.b0 ? a.b0[o] : 0.f) : (o < cout ? (a.b1 ? a.b1[o - a.cout0] : 0.f) : 0.f);
    } else {
        if (o < a.cin && c < cout) {
            const int ci = (o + a.cin_rot) % a.cin;
            const float* w = c < a.cout0 ? a.w0 + (long)c * a.cin * taps : a.w1 + (long)(c - a.cout0) * a.cin * taps;
            v = w[(long)ci * taps + (taps - 1 - tap)];           // (KH-1-ky) * KW + (KW-1-kx) = taps - 1 - tap
        }
        if (a.dst_b != nullptr && c == 0 && tap == 0) a.dst_b[o] = 0.f;
    }
    unsigned short* d = a.dst_w + ((long)(o * taps + tap) * (a.cin_pad / 32) + c / 32) * 64 + (c % 32);
    const unsigned short hi = pf_bf16_rne(v);
    d[0] = hi;
    d[32] = pf_bf16_rne(v - pf_bf16_to_f32(hi));
}

// Several pf_pack_conv_weights problems in one launch (a training step re-packs ~115 operands after every optimizer step):
// job j owns the indices [start[j], start[j+1]).
#define PF_PACK_MAX_JOBS 16
struct PfPackBatchArgs {
    PfPackWArgs job[PF_PACK_MAX_JOBS];
    long start[PF_PACK_MAX_JOBS + 1];
    int n;
};
PF_HD void pf_pack_conv_weights_batch_elem(long idx, const PfPackBatchArgs& a) {
    int j = 0;
    while (j + 1 < a.n && idx >= a.start[j + 1]) ++j;
    pf_pack_conv_weights_elem(idx - a.start[j], a.job[j]);
}

// ----------------------------------------------------------------------------------------------
// Training: packed weight / bias gradients -> the parameters' own gradient tensors, several convolutions per launch.
//   gw[o][c][tap] += scale * dw[o_off + o][tap][c]      gb[o] += scale * db[o_off + o]
// dw: pf_conv2d_wgrad's layout [Cout_pad128][taps][cin_pad]; gw: [cout][cin][KH][KW] (the .grad of an nn.Conv2d weight, a view
// of the optimiser's flat gradient buffer).  o_off selects the rows of one module inside a fused convolution (convz|convr).
// One call = one element of one job; job j owns the indices [start[j], start[j+1]): cout*cin*taps weight elements, then cout
// bias elements.
// ----------------------------------------------------------------------------------------------
#define PF_UNPACK_MAX_JOBS 16
struct PfUnpackJob {
    const float* dw; const float* db; float* gw; float* gb;
    int cout, cin, taps, cin_pad, o_off; float scale;
};
struct PfUnpackArgs {
    PfUnpackJob job[PF_UNPACK_MAX_JOBS];
    long start[PF_UNPACK_MAX_JOBS + 1];
    int n;
};
PF_HD void pf_unpack_wgrads_elem(long idx, const PfUnpackArgs& a) {
    int j = 0;
    while (j + 1 < a.n && idx >= a.start[j + 1]) ++j;
    const PfUnpackJob& q = a.job[j];
    long i = idx - a.start[j];
    const long nw = (long)q.cout * q.cin * q.taps;
    if (i < nw) {
        const int tap = (int)(i % q.taps);
        const int c = (int)((i / q.taps) % q.cin);
        const int o = (int)(i / ((long)q.taps * q.cin));
        q.gw[i] += q.scale * q.dw[((long)(q.o_off + o) * q.taps + tap) * q.cin_pad + c];
    } else if (q.gb != nullptr) {
        i -= nw;
        q.gb[i] += q.scale * q.db[q.o_off + i];
    }
}

// ----------------------------------------------------------------------------------------------
// Encoder glue (core/extractor.py:41-47, :144-150): out = relu( res' + relu(y*s + t) ) with
// res' = res (identity shortcut) | res*rs + rt (normalised 1x1/2 shortcut) | absent.
// y, res, out: channel-last [B*Np][C]; s,t,rs,rt: [B][C].  One call = 4 consecutive channels.
// ----------------------------------------------------------------------------------------------
struct PfNormActArgs {
    const float* y; const float* s; const float* t;
    const float* res; const float* rs; const float* rt;
    float* out; int B, Np, C; int res_relu;
};
PF_HD void pf_norm_act_elem(long idx, const PfNormActArgs& a) {   // idx over B*Np*C/4
    const int c4n = a.C / 4;
    const int c = (int)(idx % c4n) * 4;
    const long row = idx / c4n;
    const long b = row / a.Np;
    for (int i = 0; i < 4; ++i) {
        const long e = row * a.C + c + i, pc = b * a.C + c + i;
        float v = fmaxf(a.y[e] * a.s[pc] + a.t[pc], 0.f);
        if (a.res) {
            float r = a.res[e];
            if (a.rs) r = r * a.rs[pc] + a.rt[pc];
            if (a.res_relu) r = fmaxf(r, 0.f);
            v = fmaxf(r + v, 0.f);
        }
        a.out[e] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// NCHW -> channel-last slice copy with optional activation (0 none, 1 relu, 2 tanh)
// (core/prior_raft.py:136-142: net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:]))
// ----------------------------------------------------------------------------------------------
struct PfToNhwcArgs { const float* in; float* out; int B, C_total, c_begin, C, N, ld_out, c_out_off, act; };
PF_HD float pf_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return tanhf(v);
    return v;
}
PF_HD void pf_to_nhwc_elem(long idx, const PfToNhwcArgs& a) {   // idx over B*N*C
    const int c = (int)(idx % a.C);
    const long row = idx / a.C;
    const long b = row / a.N, n = row % a.N;
    const float v = a.in[(b * a.C_total + a.c_begin + c) * (long)a.N + n];
    a.out[row * a.ld_out + a.c_out_off + c] = pf_act(v, a.act);
}

// 2x2 space-to-depth of an NCHW image into a channel-last map:
//   out[b][Y][X][(py*2+px)*C + c] = in[b][c][2Y+py][2X+px]
// The encoders' 7x7 stride-2 stem (core/extractor.py:122, Cin = 3) is, on this image, a 4x4
// stride-1 convolution over 12 channels with window rows Y-2..Y+1 (tap KY <-> ky = 2 KY + py - 1),
// which the implicit-GEMM conv kernels run directly (see engine.EncoderPlan).
struct PfS2dArgs { const float* in; float* out; int B, C, H, W, ld_out; };   // H, W: INPUT size (even)
PF_HD void pf_s2d_elem(long idx, const PfS2dArgs& a) {           // idx over B*(H/2)*(W/2)*4*C
    const int C4 = 4 * a.C, Wo = a.W / 2, Ho = a.H / 2;
    const int cc = (int)(idx % C4);
    const long pix = idx / C4;
    const int X = (int)(pix % Wo), Y = (int)((pix / Wo) % Ho);
    const long b = pix / ((long)Wo * Ho);
    const int q = cc / a.C, c = cc % a.C;
    const int py = q >> 1, px = q & 1;
    a.out[pix * a.ld_out + cc] = a.in[((b * a.C + c) * a.H + 2 * Y + py) * (long)a.W + 2 * X + px];
}

// ----------------------------------------------------------------------------------------------
// Evaluation counterpart: per-pixel EPE and SEPE of a predicted flow against the
// ground truth.  EPE = |flow - gt| (evaluate.py:265); SEPE = haversine great-circle distance on the
// unit sphere between the two END POINTS (core/utils/spherical.py:20-53, method 'Haversine'):
// end point x wraps, y clamps (core/utils/projection_prim_ortho.py:200-218), pixel -> (theta, phi)
// by ERP.plane2spherical (:397-411).
// ----------------------------------------------------------------------------------------------
struct PfFlowMetricsArgs { const float* pred; const float* gt; float* epe; float* sd; int B, H, W; int cosine; };
PF_HD void pf_endpoint_sph(float x, float y, float u, float v, int H, int W, float& theta, float& phi) {
    const float pi = 3.14159265358979323846f;
    const float e0 = pf_pymod(x + u + 0.5f, (float)W) - 0.5f;
    float e1 = y + v;
    e1 = e1 < -0.5f ? -0.5f : (e1 > (float)H - 0.5f ? (float)H - 0.5f : e1);
    theta = (((e0 + 0.5f) / (float)W - 0.5f) * 2.f) * pi;
    phi = (0.5f - (e1 + 0.5f) / (float)H) * pi;
}
PF_HD float pf_haversine(float x) { const float s = sinf(x / 2.f); return s * s; }
PF_HD void pf_flow_metrics_elem(long idx, const PfFlowMetricsArgs& a) {   // idx over B*H*W
    const long N = (long)a.H * a.W;
    const long b = idx / N, n = idx % N;
    const float x = (float)(n % a.W), y = (float)(n / a.W);
    const float pu = a.pred[(b * 2 + 0) * N + n], pv = a.pred[(b * 2 + 1) * N + n];
    const float gu = a.gt[(b * 2 + 0) * N + n], gv = a.gt[(b * 2 + 1) * N + n];
    const float du = pu - gu, dv = pv - gv;
    if (a.epe) a.epe[idx] = sqrtf(du * du + dv * dv);
    if (a.sd) {
        float tp, pp, tg, pg;
        pf_endpoint_sph(x, y, pu, pv, a.H, a.W, tp, pp);
        pf_endpoint_sph(x, y, gu, gv, a.H, a.W, tg, pg);
        if (a.cosine) {         // method='Cosine' (core/utils/spherical.py:40-46): spherical law of cosines, as written there
            const float ca = sinf(pp) * sinf(pg) + (cosf(pp) * cosf(pg)) * cosf(tg - tp);
            a.sd[idx] = acosf(ca);
        } else {
            const float hv = pf_haversine(pg - pp) + (cosf(pp) * cosf(pg)) * pf_haversine(tg - tp);
            a.sd[idx] = 2.f * asinf(sqrtf(hv));
        }
    }
}

// Region sums (evaluate.py:246-275): pixel n belongs to region r when bit r of bits[n] is set.
// partials[((b*nblk + k)*R + r)*3 + {0,1,2}] = sum epe, sum sd, sum sd*weight over block k's pixels.
struct PfRegionSumArgs {
    const float* epe; const float* sd; const float* weight; const unsigned char* bits;
    double* partials; int B, N, R, nblk;
};

// ----------------------------------------------------------------------------------------------
// Training-step counterpart, the parts that do not need the network's backward.
// Sequence loss of ONE prediction (train_flow.py:62-71): m = (valid >= 0.5 && |gt| < max_flow) * w[n],
// loss_i = sum m * (|du| + |dv|); its gradient seed d loss / d pred = i_weight * m * sign(pred - gt);
// and the metrics of train_flow.py:73-79 (epe and <1/<3/<5 px fractions over the valid pixels).
// sums[6] = { sum m*(|du|+|dv|), sum_valid epe, n_valid, n(epe<1), n(epe<3), n(epe<5) }
// ----------------------------------------------------------------------------------------------
struct PfSeqLossArgs {
    const float* pred; const float* gt; const float* valid; const float* w;   // [B,2,N] x2, [B,N], [N]
    float* grad;                 // [B,2,N] or null
    double* partials;            // [B][nblk][6]
    int B, N, nblk; float i_weight, max_flow;
};
constexpr int PF_SEQ_LOSS_MAX = 32;
struct PfSeqLossBatch {           // n terms sharing gt / valid / weight: term i = PfSeqLossArgs{pred[i], ..., grad[i], partials + i * B * nblk * 6, i_weight[i]}
    const float* pred[PF_SEQ_LOSS_MAX]; float* grad[PF_SEQ_LOSS_MAX]; float i_weight[PF_SEQ_LOSS_MAX];
    PfSeqLossArgs common;         // pred / grad / i_weight unused
    int n;
};
PF_HD PfSeqLossArgs pf_seq_loss_term(const PfSeqLossBatch& t, int i) {
    PfSeqLossArgs a = t.common;
    a.pred = t.pred[i]; a.grad = t.grad[i]; a.i_weight = t.i_weight[i];
    a.partials = t.common.partials + (long)i * t.common.B * t.common.nblk * 6;
    return a;
}
PF_HD void pf_seq_loss_pixel(const PfSeqLossArgs& a, long b, int n, double (&sums)[6]) {
    const long N = a.N;
    const float gu = a.gt[(b * 2 + 0) * N + n], gv = a.gt[(b * 2 + 1) * N + n];
    const float du = a.pred[(b * 2 + 0) * N + n] - gu, dv = a.pred[(b * 2 + 1) * N + n] - gv;
    const float mag = sqrtf(gu * gu + gv * gv);
    const bool ok = (a.valid[b * N + n] >= 0.5f) && (mag < a.max_flow);
    const float m = ok ? a.w[n] : 0.f;
    if (a.grad) {
        const float gs = a.i_weight * m;
        a.grad[(b * 2 + 0) * N + n] = du > 0.f ? gs : (du < 0.f ? -gs : 0.f);
        a.grad[(b * 2 + 1) * N + n] = dv > 0.f ? gs : (dv < 0.f ? -gs : 0.f);
    }
    sums[0] += (double)(m * (fabsf(du) + fabsf(dv)));
    if (ok) {
        const float e = sqrtf(du * du + dv * dv);
        sums[1] += (double)e; sums[2] += 1.0;
        sums[3] += e < 1.f ? 1.0 : 0.0; sums[4] += e < 3.f ? 1.0 : 0.0; sums[5] += e < 5.f ? 1.0 : 0.0;
    }
}

// sum of squares (gradient norm for clip_grad_norm_, train_flow.py:137): partials[k] = chunk sums
struct PfSumSqArgs { const float* x; double* partials; long n; int nblk; };

// AdamW (torch.optim.AdamW as train_flow.py:86-88 builds it; amsgrad off), one element:
//   g' = g * gscale (the clip coefficient); p *= 1 - lr*wd; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),  bc_k = 1 - beta_k^step
struct PfAdamWArgs { float* p; const float* g; float* m; float* v; long n;
                     float decay, b1, b2, eps, step_size, sqrt_bc2, gscale; };   // scalars rounded from double like torch's
PF_HD void pf_adamw_elem(long i, const PfAdamWArgs& a) {
    const float g = a.g[i] * a.gscale;
    float p = a.p[i];
    p = p * a.decay;                                                    // param.mul_(1 - lr * weight_decay)
    const float m = a.m[i] + (g - a.m[i]) * (1.f - a.b1);               // exp_avg.lerp_(grad, 1 - beta1)
    const float v = a.v[i] * a.b2 + (g * g) * (1.f - a.b2);             // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
    p = p + (m / denom) * (-a.step_size);                               // addcdiv_(exp_avg, denom, value=-step_size)
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
}

// The same update with its step-dependent scalars read from DEVICE memory (hyper = {decay, step_size, sqrt_bc2, gscale}): the
// launch arguments are then the same at every step, which is what a captured HIP graph of the training step needs -- the host
// (or, for the clip coefficient, a kernel of the graph) rewrites the four floats between replays.
struct PfAdamWDevArgs { float* p; const float* g; float* m; float* v; long n; float b1, b2, eps; const float* hyper; };
PF_HD void pf_adamw_dev_elem(long i, const PfAdamWDevArgs& a) {
    PfAdamWArgs b; b.p = a.p; b.g = a.g; b.m = a.m; b.v = a.v; b.n = a.n; b.b1 = a.b1; b.b2 = a.b2; b.eps = a.eps;
    b.decay = a.hyper[0]; b.step_size = a.hyper[1]; b.sqrt_bc2 = a.hyper[2]; b.gscale = a.hyper[3];
    pf_adamw_elem(i, b);
}

// channel-last -> NCHW (debug / boundary export)
struct PfToNchwArgs { const float* in; float* out; int B, C, N, ld_in, c_in_off; };
PF_HD void pf_to_nchw_elem(long idx, const PfToNchwArgs& a) {   // idx over B*C*N
    const long n = idx % a.N;
    const long bc = idx / a.N;
    const long b = bc / a.C, c = bc % a.C;
    a.out[idx] = a.in[(b * a.N + n) * a.ld_in + a.c_in_off + c];
}

// ----------------------------------------------------------------------------------------------
// K5: warp + groupwise correlation, per pixel (core/prior_raft.py:173-174, :77-83).
// One output pixel = 4 group means; the device kernel spreads the 256 channels over one
// wavefront (see pf_elem_kernels.hip); this scalar form is the host/emu statement.
// ----------------------------------------------------------------------------------------------
struct PfWarpGcorrArgs {
    const float* f1; const float* f2;   // channel-last [B*N][C]
    const float* coords;                // planar [B,2,N]: absolute coords, or a flow when add_grid
    PfDst dst;                          // 4 channels written at dst.c_off
    int B, H, W, C, add_grid;
};
PF_HD PfTaps pf_warp_taps(const PfWarpGcorrArgs& a, long b, long n) {
    const long N = (long)a.H * a.W;
    float x = a.coords[(b * 2 + 0) * N + n], y = a.coords[(b * 2 + 1) * N + n];
    if (a.add_grid) { x = (float)(n % a.W) + x; y = (float)(n / a.W) + y; }   // coords0 + flow
    return pf_taps0(pf_pymod(x, (float)a.W), y, a.H, a.W);
}
PF_HD void pf_warp_gcorr_elem(long idx, const PfWarpGcorrArgs& a) {   // idx over B*N*4
    const long N = (long)a.H * a.W;
    const int g = (int)(idx % 4);
    const long row = idx / 4;
    const long b = row / N, n = row % N;
    const PfTaps t = pf_warp_taps(a, b, n);
    const int cg = a.C / 4;
    const float* f2b = a.f2 + b * N * a.C;
    float acc = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c)
        acc = acc + a.f1[row * a.C + c] * pf_apply_ld(t, f2b + c, a.C);
    a.dst.ptr[row * a.dst.ld + a.dst.c_off + g] = acc / (float)cg;
}

// ----------------------------------------------------------------------------------------------
// Per-iteration motion inputs of one pixel, fused (core/prior_raft.py:171-182): the flows of both branches
// (coords1 - coords0), flo_rotate(flow_B) into view A, and the two feature warps + groupwise correlations whose
// sample points are coords1_A and coords0 + flow_B_A.  The scalar part below is the arithmetic of pf_flow_prep_elem
// and pf_flo_rotate_elem, statement for statement (the flow at a corner pixel is formed as coords1 - coords0 first,
// exactly what the separate flow_prep launch stored), so the fused launch is bit-identical to the five it replaces.
// ----------------------------------------------------------------------------------------------
struct PfMotionPrepArgs {
    const float* c1a; const float* c1b;      // planar coords1 [B,2,N] of branch A / branch B
    const float* g_w2c; const float* g_c2w;  // flo_rotate(flow_B): W2C = grid(R_A2B), C2W = grid(R_B2A)   [2,N] each
    const float* f1; const float* f2;        // channel-last features of view A [B*N][256]
    float* flow4_a;                          // [B*N][4]  flow_A | flow_B_A      (input of the 7x7 flow stems)
    float* flow2_b;                          // [B*N][2]  flow_B
    PfDst xa, xb;                            // GRU input tails: 4 columns (flow_A | flow_B_A) / 2 columns (flow_B)
    void* xa_split; int xa_lds;              // optional split twins of the GRU input buffers (same channel offsets as xa / xb)
    void* xb_split; int xb_lds;
    float* conf; int conf_ld;                // [B*N][conf_ld]: columns 0..3 flaw_A, 4..7 flaw_B_A
    int B, H, W;
};
struct PfMotionFlows { float ua, va, ub, vb, uba, vba; };
// camera-frame flow of branch B at pixel p (pf_flow_c_at with flow_B = coords1_B - coords0 formed in place)
PF_HD void pf_flow_c_at_coords(const PfMotionPrepArgs& a, long b, int p, float& f0, float& f1) {
    const long N = (long)a.H * a.W;
    const float Wf = (float)a.W;
    const float px = (float)(p % a.W), py = (float)(p / a.W);
    const float fu = a.c1b[(b * 2 + 0) * N + p] - px, fv = a.c1b[(b * 2 + 1) * N + p] - py;
    float ex = (px + fu) + 0.5f;
    ex = pf_pymod(ex, Wf) - 0.5f;
    float ey = py + fv;
    ey = fminf(fmaxf(ey, -0.5f), (float)a.H - 0.5f);
    const PfWrapTaps t = pf_wraptaps(ex, ey, a.H, a.W);
    const float* g0 = a.g_w2c;
    const float* g1 = a.g_w2c + N;
    const float a0 = g0[t.ia];
    const float e0 = pf_wrapmix(t, a0, pf_unwrap_m(a0, g0[t.ib], Wf), pf_unwrap_m(a0, g0[t.ic], Wf),
                                pf_unwrap_m(a0, g0[t.id], Wf));
    const float e1 = pf_wrapmix(t, g1[t.ia], g1[t.ib], g1[t.ic], g1[t.id]);
    f0 = e0 - g0[p];
    f0 = pf_pymod(f0 + Wf * 0.5f, Wf) - Wf * 0.5f;     // u_clip
    f1 = e1 - g1[p];
}
PF_HD PfMotionFlows pf_motion_flows(long row, const PfMotionPrepArgs& a) {      // row = b*N + n
    const long N = (long)a.H * a.W;
    const long b = row / N, n = row % N;
    const float x = (float)(n % a.W), y = (float)(n / a.W);
    PfMotionFlows f;
    f.ua = a.c1a[(b * 2 + 0) * N + n] - x; f.va = a.c1a[(b * 2 + 1) * N + n] - y;
    f.ub = a.c1b[(b * 2 + 0) * N + n] - x; f.vb = a.c1b[(b * 2 + 1) * N + n] - y;
    const PfWrapTaps t = pf_wraptaps(a.g_c2w[n], a.g_c2w[N + n], a.H, a.W);
    float a0, a1, b0, b1, c0, c1, d0, d1;
    pf_flow_c_at_coords(a, b, t.ia, a0, a1);
    pf_flow_c_at_coords(a, b, t.ib, b0, b1);
    pf_flow_c_at_coords(a, b, t.ic, c0, c1);
    pf_flow_c_at_coords(a, b, t.id, d0, d1);
    f.uba = pf_wrapmix(t, a0, b0, c0, d0);
    f.vba = pf_wrapmix(t, a1, b1, c1, d1);
    return f;
}

// Backward of K5 (autograd through cycle_bilinear_sampler + groupwise_corr, core/prior_raft.py:173-174,
// :77-83; coords detached): with gs = d_flaw[group(c)] / (C/4),
//   d_f1[p][c]        += gs * warped_f2[p][c]
//   d_f2[corner_k][c] += gs * f1[p][c] * w_k          (fp32 atomics: several pixels sample the same corner)
// One call = one (pixel, channel); both outputs are ACCUMULATED.
struct PfWarpGcorrBwdArgs {
    const float* f1; const float* f2;   // channel-last [B*N][C]
    const float* coords;                // planar [B,2,N]
    const float* d_flaw; int ld_d, off_d;   // channel-last gradient of the 4 group means
    float* d_f1; float* d_f2;           // channel-last [B*N][C]
    int B, H, W, C, add_grid;
};
PF_HD void pf_warp_gcorr_bwd_elem(long idx, const PfWarpGcorrBwdArgs& a) {   // idx over B*N*C
    const long N = (long)a.H * a.W;
    const int c = (int)(idx % a.C);
    const long row = idx / a.C;
    const long b = row / N, n = row % N;
    PfWarpGcorrArgs fa; fa.coords = a.coords; fa.H = a.H; fa.W = a.W; fa.add_grid = a.add_grid;
    const PfTaps t = pf_warp_taps(fa, b, n);
    const int cg = a.C / 4;
    const float gs = a.d_flaw[row * a.ld_d + a.off_d + c / cg] / (float)cg;
    const float* f2b = a.f2 + b * N * a.C + c;
    PF_ATOMIC_ADD(a.d_f1 + row * a.C + c, gs * pf_apply_ld(t, f2b, a.C));
    const float v = gs * a.f1[row * a.C + c];
    float* d2 = a.d_f2 + b * N * a.C + c;
    for (int j = 0; j < 4; ++j)
        if (t.w[j] != 0.f) PF_ATOMIC_ADD(d2 + (long)t.idx[j] * a.C, v * t.w[j]);
}

