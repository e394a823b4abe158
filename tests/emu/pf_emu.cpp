// Host emulation build of the per-element kernels -- TEST INFRASTRUCTURE ONLY.
// Compiles prior-flow_amd/csrc/pf_elem.h + pf_api_elem.inc for the CPU so that the sampler /
// geometry / lookup index logic can be checked against the oracle in the GPU-less build
// container (tests/test_emu_kernels.py).  The product never loads this library; the MFMA
// kernels (pf_conv2d, pf_corr_pyramid) have no emulation and are tested on the GPU only.
#include "pf_elem.h"

template <class Args, void (*F)(long, const Args&)>
static int pf_loop(const Args& a, long total) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < total; ++i) F(i, a);
    return PF_OK;
}
#define PF_LAUNCH(name, args, total, stream) pf_loop<decltype(args), pf_##name##_elem>(args, total)
#define PF_DIRECT_CONV_LAUNCH(a, total, stream) pf_loop<PfDirectConvArgs, pf_direct_conv_elem>(a, total)
static int pf_direct_group_loop(const PfDirectConvArgs* ds, int n, long total) {
    for (int i = 0; i < n; ++i) pf_loop<PfDirectConvArgs, pf_direct_conv_elem>(ds[i], total);
    return 0;
}
#define PF_DIRECT_CONV_GROUP_LAUNCH(ds, n, total, stream) pf_direct_group_loop(ds, n, total)

// host statements of pf_seq_loss / pf_sum_squares (same chunk partition, sequential sums inside a chunk)
static int emu_seq_loss(const PfSeqLossArgs& a, void*) {
    const int chunk = (a.N + a.nblk - 1) / a.nblk;
    for (int b = 0; b < a.B; ++b)
        for (int k = 0; k < a.nblk; ++k) {
            double sums[6] = {0, 0, 0, 0, 0, 0};
            for (int n = k * chunk; n < (k + 1) * chunk && n < a.N; ++n) pf_seq_loss_pixel(a, b, n, sums);
            for (int j = 0; j < 6; ++j) a.partials[((long)b * a.nblk + k) * 6 + j] = sums[j];
        }
    return PF_OK;
}
static int emu_sumsq(const PfSumSqArgs& a, void*) {
    const long chunk = (a.n + a.nblk - 1) / a.nblk;
    for (int k = 0; k < a.nblk; ++k) {
        double s = 0;
        for (long i = k * chunk; i < (k + 1) * chunk && i < a.n; ++i) s += (double)a.x[i] * (double)a.x[i];
        a.partials[k] = s;
    }
    return PF_OK;
}
#define PF_SEQ_LOSS_LAUNCH(a, stream) emu_seq_loss(a, stream)
static int emu_seq_loss_batch(const PfSeqLossBatch& t, void* s) {
    for (int i = 0; i < t.n; ++i) {
        const int rc = emu_seq_loss(pf_seq_loss_term(t, i), s);
        if (rc != PF_OK) return rc;
    }
    return PF_OK;
}
#define PF_SEQ_LOSS_BATCH_LAUNCH(t, stream) emu_seq_loss_batch(t, stream)
#define PF_SUMSQ_LAUNCH(a, stream) emu_sumsq(a, stream)

// host statement of pf_channel_stats_final
static int emu_stats_final(const double* part, int B, int Np, int C, int nblk, float eps, float* scale, float* shift, void*) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double s = 0, ss = 0;
            for (int k = 0; k < nblk; ++k) {
                s += part[(((long)b * nblk + k) * C + c) * 2];
                ss += part[(((long)b * nblk + k) * C + c) * 2 + 1];
            }
            const double mean = s / Np;
            double var = ss / Np - mean * mean;
            if (var < 0) var = 0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            scale[(long)b * C + c] = (float)rstd;
            shift[(long)b * C + c] = (float)(-mean * rstd);
        }
    return PF_OK;
}
#define PF_STATS_FINAL_LAUNCH emu_stats_final

// host statement of pf_region_sums (same block partition, sequential sums inside a block)
static int emu_region_sums(const PfRegionSumArgs& a, void*) {
    const int chunk = (a.N + a.nblk - 1) / a.nblk;
    for (int b = 0; b < a.B; ++b)
        for (int k = 0; k < a.nblk; ++k)
            for (int r = 0; r < a.R; ++r) {
                double s0 = 0, s1 = 0, s2 = 0;
                for (int n = k * chunk; n < (k + 1) * chunk && n < a.N; ++n)
                    if ((a.bits[n] >> r) & 1u) {
                        const double e = a.epe[(long)b * a.N + n], s = a.sd[(long)b * a.N + n];
                        s0 += e; s1 += s; s2 += a.weight ? s * (double)a.weight[n] : 0.0;
                    }
                double* o = a.partials + ((long)(b * a.nblk + k) * a.R + r) * 3;
                o[0] = s0; o[1] = s1; o[2] = s2;
            }
    return PF_OK;
}
#define PF_REGION_SUM_LAUNCH(a, stream) emu_region_sums(a, stream)

// host statement of pf_channel_stats (same fp64 two-stage sums as the device kernels)
static int emu_stats(const float* y, int B, int Np, int C, float eps, float* scale, float* shift, double* part,
                     int nblk, void*) {
    const int chunk = (Np + nblk - 1) / nblk;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double s = 0, ss = 0;
            for (int k = 0; k < nblk; ++k) {
                double ps = 0, pss = 0;
                for (int p = k * chunk; p < (k + 1) * chunk && p < Np; ++p) {
                    const double v = y[((long)b * Np + p) * C + c];
                    ps += v; pss += v * v;
                }
                part[(((long)b * nblk + k) * C + c) * 2] = ps;
                part[(((long)b * nblk + k) * C + c) * 2 + 1] = pss;
                s += ps; ss += pss;
            }
            const double mean = s / Np;
            double var = ss / Np - mean * mean;
            if (var < 0) var = 0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            scale[(long)b * C + c] = (float)rstd;
            shift[(long)b * C + c] = (float)(-mean * rstd);
        }
    return PF_OK;
}
#define PF_STATS_LAUNCH emu_stats
#define PF_FLOW_OUT_LAUNCH(a, total, stream) pf_loop<PfFlowOutArgs, pf_flow_out_elem>(a, total)
#define PF_LOOKUP_LAUNCH(a, total, stream) pf_loop<PfLookupArgs, pf_lookup_elem>(a, total)
#define PF_COMBINE_LAUNCH(a, total, stream) pf_loop<PfCombineArgs, pf_combine_elem>(a, total)
#define PF_LOOKUP_BWD_LAUNCH(a, total, stream) pf_loop<PfLookupBwdArgs, pf_lookup_bwd_elem>(a, total)
#define PF_UPSAMPLE_BWD_LAUNCH(a, total, stream) pf_loop<PfUpsampleBwdArgs, pf_upsample_bwd_elem>(a, total)

#include "pf_api_elem.inc"

extern "C" int pf_warp_gcorr(const float* f1, const float* f2, const float* coords, int add_grid,
                             float* dst, int dst_ld, int dst_off, int B, int H8, int W8, int C,
                             void* stream) {
    PfWarpGcorrArgs a;
    const int rc = pf_warp_gcorr_fill(a, f1, f2, coords, add_grid, dst, dst_ld, dst_off, B, H8, W8, C);
    if (rc != PF_OK) return rc;
    return PF_LAUNCH(warp_gcorr, a, (long)B * H8 * W8 * 4, stream);
}
// host statement of the fused confidence stem: two direct convolutions through a temporary
extern "C" int pf_conf_stem(const float* in, int ld_in, int off_in, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* out, int ld_out, int off_out,
                            void* out_split, int lds_out, int B, int H8, int W8, void*) {
    if (!in || !w1 || !b1 || !w2 || !b2 || (!out && !out_split)) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 0 || W8 <= 0) return PF_ERR_BAD_SHAPE;
    const long N = (long)H8 * W8;
    float* mid = new float[(size_t)B * N * 32];
    for (int layer = 0; layer < 2; ++layer) {
        const int cin = layer ? 32 : 8, cout = layer ? 16 : 32;
        const float* src = layer ? mid : in; const int lds = layer ? 32 : ld_in, offs = layer ? 0 : off_in;
        float* dst = layer ? out : mid; const int ldd = layer ? ld_out : 32, offd = layer ? off_out : 0;
        const float* w = layer ? w2 : w1; const float* bias = layer ? b2 : b1;
#pragma omp parallel for
        for (long row = 0; row < B * N; ++row) {
            const long b = row / N, n = row % N;
            const int y = (int)(n / W8), x = (int)(n % W8);
            for (int co = 0; co < cout; ++co) {
                float acc = bias[co];
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy < 0 || yy >= H8 || xx < 0 || xx >= W8) continue;
                    const float* s = src + (b * N + (long)yy * W8 + xx) * lds + offs;
                    for (int ci = 0; ci < cin; ++ci) acc = fmaf(s[ci], w[(t * cin + ci) * cout + co], acc);
                }
                const float r = acc > 0.f ? acc : 0.f;
                if (!layer || out) dst[row * ldd + offd + co] = r;
                if (layer && out_split) pf_split_put(out_split, row, lds_out, off_out + co, r);
            }
        }
    }
    delete[] mid;
    return PF_OK;
}
// scalar statement of the fused motion-prep launch (the device kernel spreads the channels of the two warps over a wave)
extern "C" int pf_motion_prep(const float* c1a, const float* c1b, const float* g_w2c, const float* g_c2w,
                              const float* f1a, const float* f2a, float* flow4_a, float* flow2_b,
                              float* xa, int xa_ld, int xa_off, float* xb, int xb_ld, int xb_off,
                              void* xa_split, int xa_lds, void* xb_split, int xb_lds,
                              float* conf, int conf_ld, int B, int H8, int W8, int C, void*) {
    if (!c1a || !c1b || !g_w2c || !g_c2w || !f1a || !f2a || !flow4_a || !flow2_b || !conf) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 1 || W8 <= 1 || C != 256 || conf_ld < 8) return PF_ERR_BAD_SHAPE;
    PfMotionPrepArgs a;
    a.c1a = c1a; a.c1b = c1b; a.g_w2c = g_w2c; a.g_c2w = g_c2w; a.f1 = f1a; a.f2 = f2a;
    a.flow4_a = flow4_a; a.flow2_b = flow2_b;
    a.xa = pf_dst(xa, xa_ld, xa_off); a.xb = pf_dst(xb, xb_ld, xb_off);
    a.conf = conf; a.conf_ld = conf_ld; a.B = B; a.H = H8; a.W = W8;
    const long N = (long)H8 * W8;
#pragma omp parallel for
    for (long row = 0; row < B * N; ++row) {
        const long b = row / N, n = row % N;
        const PfMotionFlows f = pf_motion_flows(row, a);
        float* o = flow4_a + row * 4;
        o[0] = f.ua; o[1] = f.va; o[2] = f.uba; o[3] = f.vba;
        flow2_b[row * 2] = f.ub; flow2_b[row * 2 + 1] = f.vb;
        if (a.xa.ptr) { float* d = a.xa.ptr + row * a.xa.ld + a.xa.c_off; d[0] = f.ua; d[1] = f.va; d[2] = f.uba; d[3] = f.vba; }
        pf_store_dst2(a.xb, row, f.ub, f.vb);
        if (xa_split) {
            const float v4[4] = {f.ua, f.va, f.uba, f.vba};
            for (int i = 0; i < 4; ++i) pf_split_put(xa_split, row, xa_lds, xa_off + i, v4[i]);
        }
        if (xb_split) { pf_split_put(xb_split, row, xb_lds, xb_off, f.ub); pf_split_put(xb_split, row, xb_lds, xb_off + 1, f.vb); }
        for (int wsel = 0; wsel < 2; ++wsel) {
            const float x = wsel == 0 ? c1a[(b * 2 + 0) * N + n] : (float)(n % W8) + f.uba;
            const float y = wsel == 0 ? c1a[(b * 2 + 1) * N + n] : (float)(n / W8) + f.vba;
            const PfTaps t = pf_taps0(pf_pymod(x, (float)W8), y, H8, W8);
            for (int g = 0; g < 4; ++g) {
                float acc = 0.f;
                for (int c = g * 64; c < (g + 1) * 64; ++c)
                    acc = acc + f1a[row * 256 + c] * pf_apply_ld(t, f2a + b * N * 256 + c, 256);
                conf[row * conf_ld + 4 * wsel + g] = acc / 64.f;
            }
        }
    }
    return PF_OK;
}
extern "C" const char* pf_version(void) { return "priorflow host emulation (tests only)"; }
