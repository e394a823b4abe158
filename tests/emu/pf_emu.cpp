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
extern "C" const char* pf_version(void) { return "priorflow host emulation (tests only)"; }
