// __global__ wrappers + C-ABI launchers of the per-element kernels (gfx950).
// The math lives in pf_elem.h; see include/priorflow_hip.h for the ABI contract.
#include "pf_flow_stem.h"
#include "pf_elem.h"
#include "pf_split.h"

namespace {

constexpr int kBlock = 256;
constexpr long kMaxBlocks = 256L * 64;   // 256 CUs x 64 blocks, grid-stride beyond that

template <class Args, void (*F)(long, const Args&)>
__global__ void __launch_bounds__(kBlock) pf_elem_kernel(const Args a, const long total) {
    long idx = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; idx < total; idx += stride) F(idx, a);
}

template <class Args, void (*F)(long, const Args&)>
int pf_launch_elem(const Args& a, long total, void* stream) {
    if (total <= 0) return PF_OK;
    long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL((pf_elem_kernel<Args, F>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, a, total);
    return (int)hipGetLastError();
}

// Experiment knob (not defined in the product build): the DCCL lookup as its own __global__ with a forced occupancy
// (PF_LOOKUP_WAVES = waves per SIMD the register allocator must make room for).  The generic wrapper uses 90 VGPRs =
// 5 waves, and that is the optimum: 3 / 4 / 5 / 6 / 8 waves -> 26.7 / 24.3 / 22.9 / 25.1 / 43.4 us per launch
// (interleaved grid; round-1 A/B).
#ifdef PF_LOOKUP_WAVES
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PF_LOOKUP_WAVES, PF_LOOKUP_WAVES)))
pf_lookup_kernel(const PfLookupArgs a, const long total) {
    long idx = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; idx < total; idx += stride) pf_lookup_elem(idx, a);
}
int launch_lookup(const PfLookupArgs& a, long total, void* stream) {
    if (total <= 0) return PF_OK;
    long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(pf_lookup_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, total);
    return (int)hipGetLastError();
}
#endif

// K5 on a wavefront: one pixel per wave, 4 channels per lane (C == 256), the four 64-channel
// group sums reduced with 16-lane butterfly shuffles.
__global__ void __launch_bounds__(kBlock) pf_warp_gcorr_wave(const PfWarpGcorrArgs a, const long rows) {
    const int lane = threadIdx.x & 63;
    const long N = (long)a.H * a.W;
    long row = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (kBlock / 64);
    for (; row < rows; row += stride) {
        const long b = row / N, n = row % N;
        const PfTaps t = pf_warp_taps(a, b, n);
        const float4 v1 = reinterpret_cast<const float4*>(a.f1 + row * 256)[lane];
        const float* f2b = a.f2 + b * N * 256;
        float4 s;
        {
            const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[0] * 256)[lane];
            s.x = v.x * t.w[0]; s.y = v.y * t.w[0]; s.z = v.z * t.w[0]; s.w = v.w * t.w[0];
        }
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[q] * 256)[lane];
            s.x = s.x + v.x * t.w[q]; s.y = s.y + v.y * t.w[q];
            s.z = s.z + v.z * t.w[q]; s.w = s.w + v.w * t.w[q];
        }
        float p = v1.x * s.x;
        p = p + v1.y * s.y;
        p = p + v1.z * s.z;
        p = p + v1.w * s.w;
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);
        p += __shfl_xor(p, 8);
        if ((lane & 15) == 0) a.dst.ptr[row * a.dst.ld + a.dst.c_off + (lane >> 4)] = p / 64.f;
    }
}

// Motion inputs of an iteration in ONE launch (replaces flow_prep x2, flo_rotate, warp_gcorr x2: five dependent
// latency-bound launches of 8 192 pixels each -- 8.7 + 33 + 18.8 us in the replay of round 1).
// A workgroup of 256 threads owns 32 consecutive pixels.
//   stage 1 (thread-per-task): flo_rotate's four corner evaluations of a pixel are independent -- thread (pixel p,
//            corner c) = tid & 31, tid >> 5 for the first 128 threads -- and meet in LDS; 32 threads then blend them and
//            write the flows.  (A first version ran this scalar part redundantly in all 64 lanes of a wave-per-pixel
//            kernel: 66 us per launch -- it is ~1 500 instructions of pymod / floor arithmetic.)
//   stage 2 (16 lanes per pixel, 4 pixels per wave at a time): lane j loads float4 number 16 i + j of a feature row for
//            i = 0..3, i.e. the four channels that lane 16 i + j of pf_warp_gcorr_wave holds, and the sums run through the
//            same xor-butterfly over 16 lanes: the results are bit-identical to that kernel's, the loads are 256-byte
//            contiguous per pixel, and the row of f1 is loaded once for both warps.
#ifndef PF_MP_PIX
#define PF_MP_PIX 32            // (16 -- two workgroups per CU at B = 1 -- measured the same: 11.3 vs 11.7 us alone, 131.1 vs 131.6 pairs/s; its 35 us in the replay is sharing the chip with both lookups)
#endif
constexpr int MP_PIX = PF_MP_PIX;
__global__ void __launch_bounds__(256) pf_motion_prep_kernel(const PfMotionPrepArgs a, const long rows) {
    __shared__ float corner[4][MP_PIX][2];
    __shared__ float flows[MP_PIX][6];
    __shared__ float blendw[MP_PIX][4];
    const int tid = threadIdx.x;
    const long N = (long)a.H * a.W;
    const long row0 = (long)blockIdx.x * MP_PIX;
    // ---- stage 1 ---------------------------------------------------------------------------------------------
    if (tid < 4 * MP_PIX) {
        const int p = tid & (MP_PIX - 1), c = tid / MP_PIX;
        const long row = row0 + p;
        if (row < rows) {
            const long b = row / N, n = row % N;
            const PfWrapTaps t = pf_wraptaps(a.g_c2w[n], a.g_c2w[N + n], a.H, a.W);
            const int ci = c == 0 ? t.ia : (c == 1 ? t.ib : (c == 2 ? t.ic : t.id));
            float f0, f1;
            pf_flow_c_at_coords(a, b, ci, f0, f1);
            corner[c][p][0] = f0; corner[c][p][1] = f1;
            if (c == 0) { blendw[p][0] = t.wa; blendw[p][1] = t.wb; blendw[p][2] = t.wc; blendw[p][3] = t.wd; }
        }
    }
    __syncthreads();
    if (tid < MP_PIX && row0 + tid < rows) {
        const long row = row0 + tid;
        const long b = row / N, n = row % N;
        const float x = (float)(n % a.W), y = (float)(n / a.W);
        PfMotionFlows f;
        f.ua = a.c1a[(b * 2 + 0) * N + n] - x; f.va = a.c1a[(b * 2 + 1) * N + n] - y;
        f.ub = a.c1b[(b * 2 + 0) * N + n] - x; f.vb = a.c1b[(b * 2 + 1) * N + n] - y;
        PfWrapTaps t;                               // only the weights are used by pf_wrapmix
        t.wa = blendw[tid][0]; t.wb = blendw[tid][1]; t.wc = blendw[tid][2]; t.wd = blendw[tid][3];
        f.uba = pf_wrapmix(t, corner[0][tid][0], corner[1][tid][0], corner[2][tid][0], corner[3][tid][0]);
        f.vba = pf_wrapmix(t, corner[0][tid][1], corner[1][tid][1], corner[2][tid][1], corner[3][tid][1]);
        reinterpret_cast<float4*>(a.flow4_a)[row] = float4{f.ua, f.va, f.uba, f.vba};
        reinterpret_cast<float2*>(a.flow2_b)[row] = float2{f.ub, f.vb};
        if (a.xa.ptr) {
            float* d = a.xa.ptr + row * a.xa.ld + a.xa.c_off;
            d[0] = f.ua; d[1] = f.va; d[2] = f.uba; d[3] = f.vba;
        }
        pf_store_dst2(a.xb, row, f.ub, f.vb);
        if (a.xa_split) {
            const float v4[4] = {f.ua, f.va, f.uba, f.vba};
#pragma unroll
            for (int i = 0; i < 4; ++i) pf_split_store(pf_split_ptr(a.xa_split, row, a.xa_lds, a.xa.c_off + i), v4[i]);
        }
        if (a.xb_split) {
            pf_split_store(pf_split_ptr(a.xb_split, row, a.xb_lds, a.xb.c_off), f.ub);
            pf_split_store(pf_split_ptr(a.xb_split, row, a.xb_lds, a.xb.c_off + 1), f.vb);
        }
        // sample points of the two warps: coords1_A (:173) and coords0 + flow_B_A (:180)
        flows[tid][0] = a.c1a[(b * 2 + 0) * N + n]; flows[tid][1] = a.c1a[(b * 2 + 1) * N + n];
        flows[tid][2] = x + f.uba; flows[tid][3] = y + f.vba;
    }
    __syncthreads();
    // ---- stage 2 ---------------------------------------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6, j = lane & 15, sub = lane >> 4;
#pragma unroll
    for (int round = 0; round < MP_PIX / 16; ++round) {
        const int p = round * 16 + wave * 4 + sub;
        const long row = row0 + p;
        if (row >= rows) continue;                  // (whole 16-lane groups drop out together)
        const long b = row / N;
        const float4* f1r = reinterpret_cast<const float4*>(a.f1 + row * 256);
        const float* f2b = a.f2 + b * N * 256;
        float4 v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v1[i] = f1r[16 * i + j];
#pragma unroll
        for (int wsel = 0; wsel < 2; ++wsel) {
            const PfTaps t = pf_taps0(pf_pymod(flows[p][2 * wsel], (float)a.W), flows[p][2 * wsel + 1], a.H, a.W);
            float4 s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[0] * 256)[16 * i + j];
                s[i].x = v.x * t.w[0]; s[i].y = v.y * t.w[0]; s[i].z = v.z * t.w[0]; s[i].w = v.w * t.w[0];
            }
#pragma unroll
            for (int q = 1; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = reinterpret_cast<const float4*>(f2b + (long)t.idx[q] * 256)[16 * i + j];
                    s[i].x = s[i].x + v.x * t.w[q]; s[i].y = s[i].y + v.y * t.w[q];
                    s[i].z = s[i].z + v.z * t.w[q]; s[i].w = s[i].w + v.w * t.w[q];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float pr = v1[i].x * s[i].x;
                pr = pr + v1[i].y * s[i].y;
                pr = pr + v1[i].z * s[i].z;
                pr = pr + v1[i].w * s[i].w;
                pr += __shfl_xor(pr, 1);
                pr += __shfl_xor(pr, 2);
                pr += __shfl_xor(pr, 4);
                pr += __shfl_xor(pr, 8);
                if (j == 0) a.conf[row * a.conf_ld + 4 * wsel + i] = pr / 64.f;
            }
        }
    }
}

// Small-Cin convolution on the exact-fp32 matrix cores (7x7 2->128, 3x3 8->32, 3x3 32->16 of the
// motion encoders, core/update.py:171-178,87; and the encoders' 7x7/2 3->64 stem).
//
// GEMM view per 32-pixel row segment: M = 32 output pixels, N = Cout, K = KH*KW*Cin (tap-major,
// channel-minor: exactly the [KH*KW][Cin][Cout] packing of pf_conv2d_direct).  K is too small
// and too ragged for the 32-channel-chunk kernels, so the A operand is GATHERED: the input patch
// (KH x ((32-1)*stride+KW) pixels x Cin, zero padded) is staged in LDS once per segment and lane
// (pixel i, half h) of v_mfma_f32_32x32x2_f32 step s reads element k = 2s+h of its pixel's
// im2col row through a small k -> patch-offset table.  Pixel stride inside the patch is made odd
// (Cin|1) so the 32 lanes of a read spread over the banks.
struct PfSmallConvArgs {
    const float* in; int ld_in, c_in_off, Cin; int nchw;      // channel-last rows, or NCHW planes
    const float* w; const float* bias;                        // [KH*KW*Cin][Cout]
    float* out; int ld_out, c_out_off, Cout;                  // channel-last
    void* out_split; int lds_out;                             // optional split twin of `out` (pf_stem7x7c2_valu only)
    int B, H, W;                                              // INPUT spatial size
    int KH, KW, stride, relu;
    int Ho, Wo;                                               // output size (H/stride, W/stride)
};

// One workgroup = one (32-pixel segment, 32-output-channel group) work item; its 4 waves split
// the K dimension (each <= MAXS MFMA steps) and reduce their 32x32 partial accumulators through
// LDS.  These layers are LATENCY chains with little parallelism at B=1 (a 64x128 map has 256
// segments), so the kernel is built to have one global round trip per phase: the patch is staged
// by all 256 threads with 4 loads in flight each, and every wave prefetches ALL its weights
// (one per lane per step, 128-byte coalesced rows) into registers before the first MFMA.
// Up to 4 same-shape problems per launch (blockIdx.y): the motion encoders' three independent 7x7 flow stems.
struct PfSmallConvMulti { PfSmallConvArgs p[4]; };
template <int MAXS, int PMAX>
__global__ void __launch_bounds__(256) pf_small_conv_mfma(const PfSmallConvMulti multi) {
    const PfSmallConvArgs& a = multi.p[blockIdx.y];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = a.KH * a.KW * a.Cin;
    const int KS = (K + 1) >> 1;                    // MFMA steps
    const int CinP = a.Cin | 1;                     // odd pixel stride (bank spread)
    const int PW = 31 * a.stride + a.KW;            // patch width in pixels
    const int patch_elems = a.KH * PW * CinP;
    float* patch = sm;                              // [KH*PW*CinP]
    float* red = sm + ((patch_elems + 3) & ~3);     // [3][16][64] partial accumulators of waves 1..3

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ngrp = (a.Cout + 31) / 32;
    const int ph = a.KH / 2, pw = a.KW / 2;
    const int segs_x = (a.Wo + 31) / 32;            // the last 32-pixel segment of a row may be partial
    const long nitems = (long)a.B * a.Ho * segs_x * ngrp;
    const long Nin = (long)a.H * a.W, Nout = (long)a.Ho * a.Wo;
    const int a_lane = li * a.stride * CinP;
    const int total = a.KH * PW * a.Cin;
    const int spw = (KS + 3) >> 2;                  // steps per wave
    const int s_begin = wave * spw;
    const int s_end = (s_begin + spw < KS) ? s_begin + spw : KS;
    // patch elements owned by this thread: e = tid + 256*u (same for every item)
    int pe[PMAX], pd[PMAX];
#pragma unroll
    for (int u = 0; u < PMAX; ++u) {
        const int e = tid + 256 * u;
        if (e < total) {
            const int c = e % a.Cin, px = (e / a.Cin) % PW, ky = e / (a.Cin * PW);
            pe[u] = (ky << 20) | (px << 8) | c;
            pd[u] = (ky * PW + px) * CinP + c;
        } else {
            pe[u] = -1; pd[u] = 0;
        }
    }

    // the launcher makes gridDim.x a multiple of ngrp, so a block's channel group never changes:
    // its weights are fetched ONCE (one global round trip) and stay in registers for every item
    const int grp = (int)(blockIdx.x % ngrp);
    const int j = 32 * grp + li;                    // this lane's output channel
    const bool jok = j < a.Cout;
    float bv[MAXS];
#pragma unroll
    for (int q = 0; q < MAXS; ++q) {
        const int k = 2 * (s_begin + q) + lh;
        bv[q] = (jok && s_begin + q < s_end && k < K) ? a.w[(long)k * a.Cout + j] : 0.f;
    }
    const float bias = jok ? a.bias[j] : 0.f;

    for (long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const long seg = item / ngrp;
        const int sx = (int)(seg % segs_x);
        const int yo = (int)((seg / segs_x) % a.Ho);
        const long b = seg / ((long)segs_x * a.Ho);
        const int xi0 = sx * 32 * a.stride - pw, yi0 = yo * a.stride - ph;

        // ---- input patch: up to PMAX loads in flight per thread; the element -> (ky,px,c)
        // decomposition is item-independent and was hoisted out of the item loop -----------------
        __syncthreads();                            // previous item's LDS reads are done
#pragma unroll
        for (int u0 = 0; u0 < PMAX; u0 += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pk = pe[u0 + u];          // packed (ky << 20 | px << 8 | c), or -1
                const int c = pk & 255, px = (pk >> 8) & 4095, ky = pk >> 20;
                const int yy = yi0 + ky, xx = xi0 + px;
                const bool ok = pk >= 0 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                const long src = a.nchw ? (b * a.Cin + c) * Nin + (long)yy * a.W + xx
                                        : (b * Nin + (long)yy * a.W + xx) * a.ld_in + a.c_in_off + c;
                v[u] = ok ? a.in[ok ? src : 0] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pe[u0 + u] >= 0) patch[pd[u0 + u]] = v[u];
        }
        __syncthreads();

        // ---- this wave's MFMA steps; im2col offset advanced incrementally (k = 2s + lh) -------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int k0 = 2 * s_begin + lh;
        if (k0 >= K) k0 = 0;                        // padded K: weight is 0, any valid offset
        int c = k0 % a.Cin, tap = k0 / a.Cin;
        int kx = tap % a.KW, ky = tap / a.KW;
#pragma unroll
        for (int q = 0; q < MAXS; ++q) {
            if (s_begin + q < s_end) {              // wave-uniform
                const float av = patch[(ky * PW + kx) * CinP + c + a_lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[q], acc, 0, 0, 0);
                c += 2;
                while (c >= a.Cin) { c -= a.Cin; if (++kx == a.KW) { kx = 0; ++ky; } }
                if (ky >= a.KH) { ky = 0; kx = 0; c = 0; }   // ran past K (odd K tail): stay in range
            }
        }
        // ---- reduce the 4 partial accumulators ----------------------------------------------------
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave == 0 && jok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[r] + red[r * 64 + lane];
                v = v + red[(16 + r) * 64 + lane];
                v = v + red[(32 + r) * 64 + lane];
                v = v + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (sx * 32 + px < a.Wo)
                    a.out[(b * Nout + (long)yo * a.Wo + sx * 32 + px) * a.ld_out + a.c_out_off + j] = v;
            }
        }
    }
}

// 7x7 convolution over TWO input channels (the motion encoders' flow stems, core/update.py:87,173,175: 2 -> 128, K = 98) on
// the vector ALUs.  The MFMA formulation above spends its time in per-workgroup round trips (weights, patch, LDS reduction of a
// K split: 39 us for the three stems of an iteration, 0.6 GFLOP); here a lane owns one output pixel and CPW output channels,
// the weights of a tap are wave-uniform (scalar loads feeding v_fmac_f32 directly), the 7 x 70 x 2 input patch of a 64-pixel
// row segment sits in LDS as one 8-byte read per (row, tap column), and nothing is reduced across lanes or waves.
// One workgroup = 2 waves = 64 pixels x 2*CPW channels; blockIdx.y = problem (up to 4 same-shape stems per launch).
// Still bound by the latency of the scalar weight loads, not by the 7.8 us of FMA work; in the forward the stems sit on a
// side stream, so halving them moved the wall time by only +0.3 % (same-box A/B).
template <int CPW>
__global__ void __launch_bounds__(128) pf_stem7x7c2_valu(const PfSmallConvMulti multi) {
    const PfSmallConvArgs& a = multi.p[blockIdx.y];
    constexpr int PWD = 64 + 6;
    __shared__ __attribute__((aligned(16))) float patch[7 * PWD * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int segs = (a.W + 63) / 64;
    const int cgroups = a.Cout / (2 * CPW);
    long item = blockIdx.x;
    const int cg = (int)(item % cgroups); item /= cgroups;
    const int sx = (int)(item % segs); item /= segs;
    const int y = (int)(item % a.H);
    const long b = item / a.H;
    const int x0 = sx * 64 - 3;
    for (int e = tid; e < 7 * PWD * 2; e += 128) {
        const int c = e & 1, px = (e >> 1) % PWD, ky = (e >> 1) / PWD;
        const int yy = y + ky - 3, xx = x0 + px;
        float v = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
            v = a.in[((b * a.H + yy) * (long)a.W + xx) * a.ld_in + a.c_in_off + c];
        patch[e] = v;
    }
    __syncthreads();
    const int co0 = cg * 2 * CPW + wave * CPW;
    float acc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[i] = a.bias[co0 + i];
    const float* wbase = a.w + co0;
    for (int ky = 0; ky < 7; ++ky) {
        float2 in[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) in[kx] = *reinterpret_cast<const float2*>(&patch[(ky * PWD + lane + kx) * 2]);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float* w0 = wbase + (long)((ky * 7 + kx) * 2) * a.Cout;
            const float* w1 = w0 + a.Cout;
#pragma unroll
            for (int i = 0; i < CPW; ++i) acc[i] = __builtin_fmaf(in[kx].x, w0[i], acc[i]);
#pragma unroll
            for (int i = 0; i < CPW; ++i) acc[i] = __builtin_fmaf(in[kx].y, w1[i], acc[i]);
        }
    }
    const int x = sx * 64 + lane;
    if (x < a.W) {
        const long row = (b * a.H + y) * (long)a.W + x;
        if (a.relu) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) acc[i] = fmaxf(acc[i], 0.f);
        }
        if (a.out != nullptr) {
            float* o = a.out + row * a.ld_out + a.c_out_off + co0;
#pragma unroll
            for (int i = 0; i < CPW; i += 4) {
                float4 v;
                v.x = acc[i]; v.y = acc[i + 1]; v.z = acc[i + 2]; v.w = acc[i + 3];
                *reinterpret_cast<float4*>(o + i) = v;
            }
        }
        if (a.out_split != nullptr) {               // 8 consecutive channels: one 16-byte store per half
            static_assert(CPW % 8 == 0, "");
#pragma unroll
            for (int i = 0; i < CPW; i += 8) {
                const float v8[8] = {acc[i], acc[i + 1], acc[i + 2], acc[i + 3], acc[i + 4], acc[i + 5], acc[i + 6], acc[i + 7]};
                pf_split_store_n<8>(pf_split_ptr(a.out_split, row, a.lds_out, a.c_out_off + co0 + i), v8);
            }
        }
    }
}

bool stem7x7c2_ok(const PfSmallConvArgs& a) {
    return a.KH == 7 && a.KW == 7 && a.Cin == 2 && a.stride == 1 && !a.nchw && a.Cout % 64 == 0 && a.Ho == a.H && a.Wo == a.W &&
           (!a.out || (a.ld_out % 4 == 0 && a.c_out_off % 4 == 0 && ((uintptr_t)a.out) % 16 == 0)) &&
           (!a.out_split || (a.c_out_off % 8 == 0 && ((uintptr_t)a.out_split) % 16 == 0));
}

int launch_stem7x7c2(const PfSmallConvMulti& m, int n, void* stream) {
    const PfSmallConvArgs& a = m.p[0];
    const long blocks = (long)a.B * a.H * ((a.W + 63) / 64) * (a.Cout / 64);
    if (blocks <= 0 || blocks >= (1L << 31)) return PF_ERR_BAD_SHAPE;
    // CPW = 8: 6 waves per SIMD hide the scalar weight loads best (three stems of 64x128: 22.2 / 19.9 / 17.5 us for
    // CPW = 32 / 16 / 8 against 36.7 us on the MFMA kernel; round-1 micro-benchmark)
    hipLaunchKernelGGL((pf_stem7x7c2_valu<8>), dim3((unsigned)(blocks * 4), (unsigned)n), dim3(128), 0, (hipStream_t)stream, m);
    return (int)hipGetLastError();
}

int launch_small_conv(const PfSmallConvMulti& m, int n, void* stream) {
    const PfSmallConvArgs& a = m.p[0];
    const int K = a.KH * a.KW * a.Cin, KS = (K + 1) / 2;
    const int PW = 31 * a.stride + a.KW;
    const int patch_elems = a.KH * PW * (a.Cin | 1);
    const size_t lds = ((size_t)((patch_elems + 3) & ~3) + 3 * 16 * 64) * 4;
    const long nitems = (long)a.B * a.Ho * ((a.Wo + 31) / 32) * ((a.Cout + 31) / 32);
    const int ngrp = (a.Cout + 31) / 32;
    const long cap = (256L * 8 / ngrp) * ngrp;      // ~8 workgroups per CU, a multiple of ngrp (see kernel)
    const long blocks = nitems < cap ? nitems : cap;
    const int spw = ((KS + 3) / 4);
    const int pmax = (a.KH * PW * a.Cin + 255) / 256;     // patch elements per thread
    dim3 grid((unsigned)blocks, (unsigned)n), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (spw <= 16 && pmax <= 4)
        hipLaunchKernelGGL((pf_small_conv_mfma<16, 4>), grid, blk, lds, st, m);
    else if (spw <= 20 && pmax <= 8)
        hipLaunchKernelGGL((pf_small_conv_mfma<20, 8>), grid, blk, lds, st, m);
    else if (spw <= 40 && pmax <= 8)
        hipLaunchKernelGGL((pf_small_conv_mfma<40, 8>), grid, blk, lds, st, m);
    else if (spw <= 40 && pmax <= 16)
        hipLaunchKernelGGL((pf_small_conv_mfma<40, 16>), grid, blk, lds, st, m);
    else
        return PF_ERR_BAD_SHAPE;
    return (int)hipGetLastError();
}

// Per-(image, channel) statistics of a channel-last map -> the affine of Instance/BatchNorm-style
// normalisation: scale = 1/sqrt(var+eps), shift = -mean*scale (biased variance, eps 1e-5:
// nn.InstanceNorm2d, core/extractor.py:112-113).  Two deterministic stages with fp64 accumulation:
// nblk partial sums per image, then a fixed-order final reduction.
__global__ void __launch_bounds__(256) pf_stats_partial(const float* __restrict__ y, double* __restrict__ part,
                                                         int Np, int C, int nblk) {
    // thread = (pixel group, channel quad): 16-byte loads, 256/(C/4) pixels in flight per block
    __shared__ double sh[8][256];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int cq = C / 4;                      // channel quads (C % 4 == 0)
    const int g = 256 / cq;
    const int q = tid % cq, grp = tid / cq;
    const int chunk = (Np + nblk - 1) / nblk;
    const int p0 = blk * chunk, p1 = (p0 + chunk < Np) ? p0 + chunk : Np;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (grp < g)
        for (int p = p0 + grp; p < p1; p += g) {
            const float4 v = *reinterpret_cast<const float4*>(y + ((long)b * Np + p) * C + 4 * q);
            const double v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
            s[0] += v0; ss[0] += v0 * v0; s[1] += v1; ss[1] += v1 * v1;
            s[2] += v2; ss[2] += v2 * v2; s[3] += v3; ss[3] += v3 * v3;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh[i][tid] = s[i]; sh[4 + i][tid] = ss[i]; }
    __syncthreads();
    if (tid < cq) {                            // fixed-order sum over the pixel groups
        for (int k = 1; k < g; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[i] += sh[i][tid + k * cq]; ss[i] += sh[4 + i][tid + k * cq]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* o = part + (((long)b * nblk + blk) * C + 4 * tid + i) * 2;
            o[0] = s[i]; o[1] = ss[i];
        }
    }
}
__global__ void __launch_bounds__(256) pf_stats_final(const double* __restrict__ part, float* __restrict__ scale,
                                                       float* __restrict__ shift, int C, int CB, int nblk, int Np, float eps) {
    // block per (image, CB channels); thread (grp, c) sums partials k = grp, grp+g, ... (fixed order), then the
    // g group sums are added in order: deterministic, and 256/CB loads in flight per channel.  (One block per image
    // -- CB = C -- left a 64-channel layer's 512 partials to 4 groups: 16 dependent L2 round trips, 10 us per call.)
    __shared__ double sh[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int g = 256 / CB;
    const int c = blockIdx.y * CB + tid % CB, grp = tid / CB;
    double s = 0.0, ss = 0.0;
    if (grp < g)
        for (int k0 = grp; k0 < nblk; k0 += 8 * g) {
            // 8 partials per trip, all loads issued before the first add (the one-load-per-iteration
            // form was a chain of 32 dependent L2 round trips: 15 us for a few KB)
            double v[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * g;
                const double* o = part + (((long)b * nblk + (k < nblk ? k : k0)) * C + c) * 2;
                v[j][0] = o[0]; v[j][1] = o[1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j * g < nblk) { s += v[j][0]; ss += v[j][1]; }
        }
    sh[0][tid] = s; sh[1][tid] = ss;
    __syncthreads();
    if (tid < CB) {
        for (int k = 1; k < g; ++k) { s += sh[0][tid + k * CB]; ss += sh[1][tid + k * CB]; }
        const double mean = s / Np;
        double var = ss / Np - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        scale[(long)b * C + c] = (float)rstd;
        shift[(long)b * C + c] = (float)(-mean * rstd);
    }
}
int launch_stats(const float* y, int B, int Np, int C, float eps, float* scale, float* shift, double* part,
                 int nblk, void* stream) {
    hipLaunchKernelGGL(pf_stats_partial, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, y, part, Np, C, nblk);
    const int CB = (C % 16 == 0) ? 16 : C;          // channels per block
    hipLaunchKernelGGL(pf_stats_final, dim3(B, C / CB), dim3(256), 0, (hipStream_t)stream, part, scale, shift, C, CB, nblk, Np, eps);
    return (int)hipGetLastError();
}

int launch_stats_final(const double* part, int B, int Np, int C, int nblk, float eps, float* scale, float* shift,
                       void* stream) {
    const int CB = (C % 16 == 0) ? 16 : C;          // channels per block
    hipLaunchKernelGGL(pf_stats_final, dim3(B, C / CB), dim3(256), 0, (hipStream_t)stream, part, scale, shift, C, CB, nblk, Np, eps);
    return (int)hipGetLastError();
}

// Final stage of the two-stage fp64 reductions of the norm backward passes (pf_norm_bwd, pf_bn_frozen_bwd): the nblk partial
// (sum g, sum g*xh) pairs of one (group, channel) summed by ONE WAVE -- lane l takes the partials l, l+64, ... in order, then a
// fixed xor butterfly -- instead of one thread looping over all of them (pf_norm_bwd_final_elem: 37 us per call at nblk ~ 220;
// 305 us at nblk = 2048).  Same result for every run (fixed order); the elem form stays as the CPU emulation's path.
//   mode 0: coef[(g*C + c)*2 + {0,1}] = sum / Np            (InstanceNorm backward: means)
//   mode 1: out0[c] (+)= sum g (d beta), out1[c] (+)= sum g*xhat (d gamma); groups = 1
__global__ void __launch_bounds__(256) pf_pair_final_wave(const double* __restrict__ part, int groups, int nblk, int C, int mode,
                                                          double inv_np, float* __restrict__ out0, float* __restrict__ out1,
                                                          int accumulate) {
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (group, channel)
    if (pair >= (long)groups * C) return;
    const int c = (int)(pair % C);
    const long g = pair / C;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        const double* q = part + ((g * nblk + k) * C + c) * 2;
        s1 += q[0]; s2 += q[1];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane != 0) return;
    if (mode == 0) {
        out0[pair * 2] = (float)(s1 * inv_np);
        out0[pair * 2 + 1] = (float)(s2 * inv_np);
    } else if (accumulate) {
        out0[c] += (float)s1; out1[c] += (float)s2;
    } else {
        out0[c] = (float)s1; out1[c] = (float)s2;
    }
}
int launch_pair_final(const double* part, int groups, int nblk, int C, int mode, int Np, float* out0, float* out1, int accumulate,
                      void* stream) {
    const long pairs = (long)groups * C;
    hipLaunchKernelGGL(pf_pair_final_wave, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, part, groups, nblk, C,
                       mode, 1.0 / (double)(Np > 0 ? Np : 1), out0, out1, accumulate);
    return (int)hipGetLastError();
}

// ResidualBlock tail, 16-byte accesses (same arithmetic as pf_norm_act_elem)
__global__ void __launch_bounds__(256) pf_norm_act_vec(const PfNormActArgs a, const long total) {
    const int c4n = a.C / 4;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; idx < total; idx += stride) {
        const int c = (int)(idx % c4n) * 4;
        const long row = idx / c4n;
        const long b = row / a.Np;
        const long e = row * a.C + c, pc = b * a.C + c;
        const float4 y = *reinterpret_cast<const float4*>(a.y + e);
        const float4 s = *reinterpret_cast<const float4*>(a.s + pc);
        const float4 t = *reinterpret_cast<const float4*>(a.t + pc);
        float4 v;
        v.x = fmaxf(y.x * s.x + t.x, 0.f); v.y = fmaxf(y.y * s.y + t.y, 0.f);
        v.z = fmaxf(y.z * s.z + t.z, 0.f); v.w = fmaxf(y.w * s.w + t.w, 0.f);
        if (a.res) {
            float4 r = *reinterpret_cast<const float4*>(a.res + e);
            if (a.rs) {
                const float4 rs = *reinterpret_cast<const float4*>(a.rs + pc);
                const float4 rt = *reinterpret_cast<const float4*>(a.rt + pc);
                r.x = r.x * rs.x + rt.x; r.y = r.y * rs.y + rt.y; r.z = r.z * rs.z + rt.z; r.w = r.w * rs.w + rt.w;
            }
            if (a.res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            v.x = fmaxf(r.x + v.x, 0.f); v.y = fmaxf(r.y + v.y, 0.f);
            v.z = fmaxf(r.z + v.z, 0.f); v.w = fmaxf(r.w + v.w, 0.f);
        }
        *reinterpret_cast<float4*>(a.out + e) = v;
    }
}
int launch_norm_act(const PfNormActArgs& a, long total, void* stream) {
    long blocks = (total + 255) / 256;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(pf_norm_act_vec, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, total);
    return (int)hipGetLastError();
}

// FlowHead.conv2 (3x3, 256 -> 2) + coords1 += delta_flow: one wave per strip of 4 consecutive pixels of a row,
// 4 channels per lane per 256-channel slab.  The 3 x 6 input neighbourhood and the 18 weight vectors are loaded
// once per strip (36 16-byte loads; the first version, one wave per pixel, issued 108 and was twice as slow), and
// the eight partial sums (4 pixels x 2 outputs) are reduced with a halving exchange: 10 cross-lane steps.
// History: this kernel is how the packed-fp32 / MFMA erratum of DESIGN.md section 8 was found -- compiled with
// the SLP vectoriser it was not reproducible beside the mask head's conv.
// NP problems of one shape in one launch (branch A and branch B of an iteration: one launch, no cross-queue hop in front
// of branch B's strip); the problem is a wave-uniform choice between the kernel arguments.
struct PfFlowOutN { PfFlowOutArgs p[2]; };
template <int NP>
__global__ void __launch_bounds__(256) pf_flow_out_strip(const PfFlowOutN pn, const long strips, const int spr) {
    const int lane = threadIdx.x & 63;
    long strip_all = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; strip_all < strips * NP; strip_all += stride) {
        const bool second = NP == 2 && strip_all >= strips;
        const PfFlowOutArgs& a = second ? pn.p[1] : pn.p[0];
        const long strip = second ? strip_all - strips : strip_all;
        const long N = (long)a.H * a.W;
        const long b = strip / ((long)a.H * spr);
        const int rem = (int)(strip % ((long)a.H * spr));
        const int y = rem / spr, x0 = (rem % spr) * 4;
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
        for (int c = lane * 4; c < a.C; c += 256) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= a.H) continue;                           // wave-uniform
                const float* xrow = a.x + (b * N + (long)yy * a.W) * a.ld + c;
                float4 v[6], w0[3], w1[3];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int xx = x0 + i - 1;
                    v[i] = (xx >= 0 && xx < a.W) ? *reinterpret_cast<const float4*>(xrow + (long)xx * a.ld)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    w0[kx] = *reinterpret_cast<const float4*>(a.w + (long)(ky * 3 + kx) * a.C + c);
                    w1[kx] = *reinterpret_cast<const float4*>(a.w + (long)(9 + ky * 3 + kx) * a.C + c);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 q = v[p + kx];
                        s[2 * p] += q.x * w0[kx].x + q.y * w0[kx].y + q.z * w0[kx].z + q.w * w0[kx].w;
                        s[2 * p + 1] += q.x * w1[kx].x + q.y * w1[kx].y + q.z * w1[kx].z + q.w * w1[kx].w;
                    }
            }
        }
        // halving exchange: after the m = 32, 16, 8 steps a lane holds sum j = lane >> 3 over 8 lanes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool up = lane & 32;
            const float keep = up ? s[i + 4] : s[i], send = up ? s[i] : s[i + 4];
            s[i] = keep + __shfl_xor(send, 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool up = lane & 16;
            const float keep = up ? s[i + 2] : s[i], send = up ? s[i] : s[i + 2];
            s[i] = keep + __shfl_xor(send, 16);
        }
        {
            const bool up = lane & 8;
            const float keep = up ? s[1] : s[0], send = up ? s[0] : s[1];
            s[0] = keep + __shfl_xor(send, 8);
        }
        float r = s[0];
        r += __shfl_xor(r, 4); r += __shfl_xor(r, 2); r += __shfl_xor(r, 1);
        const int j = lane >> 3, x = x0 + (j >> 1), o = j & 1;
        if ((lane & 7) == 0 && x < a.W) {
            const long n = (long)y * a.W + x;
            const float acc = r + a.bias[o];
            if (a.delta) a.delta[(b * N + n) * a.ld_delta + o] = acc;
            a.coords1[(b * 2 + o) * N + n] += acc;
        }
    }
}
// The same convolution for big batches (round 6): one wave per TILE of 4 x 4 pixels.  The strip form reads a 3 x 6 neighbourhood
// per 4 pixels -- 4.5 KB of rows per pixel, fine while the map sits in the L2 / memory-side cache (8.4 MB at B = 1: 8 us per launch)
// but 454 us per launch at batch 32, where 268 MB per branch stream from HBM.  A 4 x 4 tile reads 6 x 6 rows per 16 pixels: 2.25 KB
// per pixel.  Per output pixel the products, the order of the additions inside a lane (channel slab, then ky, then kx) and the
// cross-lane tree (partners at distance 32, 16, 8, 4, 2, 1) are the strip kernel's, so the results are bit-identical
// (tests/test_hip_kernels.py); the strip form stays for small batches, where 4x fewer waves would not fill the chip.
__global__ void __launch_bounds__(256) pf_flow_out_tile(const PfFlowOutArgs a, const long tiles, const int tpr, const int tpc) {
    const int lane = threadIdx.x & 63;
    long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    const long N = (long)a.H * a.W;
    for (; tile < tiles; tile += stride) {
        const long b = tile / ((long)tpc * tpr);
        const int rem = (int)(tile % ((long)tpc * tpr));
        const int y0 = (rem / tpr) * 4, x0 = (rem % tpr) * 4;
        float s[32];                                   // s[2 (4 row + col) + output]
#pragma unroll
        for (int j = 0; j < 32; ++j) s[j] = 0.f;
        for (int c = lane * 4; c < a.C; c += 256) {
            float4 w0[9], w1[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                w0[k] = *reinterpret_cast<const float4*>(a.w + (long)k * a.C + c);
                w1[k] = *reinterpret_cast<const float4*>(a.w + (long)(9 + k) * a.C + c);
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {              // input row y0 + r - 1: feeds output row r - ky (ky ascending with r: the strip kernel's order)
                const int yy = y0 + r - 1;
                if (yy < 0 || yy >= a.H) continue;                           // wave-uniform
                const float* xrow = a.x + (b * N + (long)yy * a.W) * a.ld + c;
                float4 v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int xx = x0 + i - 1;
                    v[i] = (xx >= 0 && xx < a.W) ? *reinterpret_cast<const float4*>(xrow + (long)xx * a.ld)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int orow = r - ky;                                 // compile-time after unrolling
                    if (orow < 0 || orow > 3) continue;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 q = v[p + kx];
                            const float4 u0 = w0[ky * 3 + kx], u1 = w1[ky * 3 + kx];
                            s[2 * (4 * orow + p)] += q.x * u0.x + q.y * u0.y + q.z * u0.z + q.w * u0.w;
                            s[2 * (4 * orow + p) + 1] += q.x * u1.x + q.y * u1.y + q.z * u1.z + q.w * u1.w;
                        }
                }
            }
        }
        // halving exchange over the partner distances 32, 16, 8, 4, 2: afterwards a lane holds sum j = (lane >> 1) & 31 over
        // 32 lanes; the last step (distance 1) completes it
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool up = lane & 32;
            const float keep = up ? s[i + 16] : s[i], send = up ? s[i] : s[i + 16];
            s[i] = keep + __shfl_xor(send, 32);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool up = lane & 16;
            const float keep = up ? s[i + 8] : s[i], send = up ? s[i] : s[i + 8];
            s[i] = keep + __shfl_xor(send, 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool up = lane & 8;
            const float keep = up ? s[i + 4] : s[i], send = up ? s[i] : s[i + 4];
            s[i] = keep + __shfl_xor(send, 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool up = lane & 4;
            const float keep = up ? s[i + 2] : s[i], send = up ? s[i] : s[i + 2];
            s[i] = keep + __shfl_xor(send, 4);
        }
        {
            const bool up = lane & 2;
            const float keep = up ? s[1] : s[0], send = up ? s[0] : s[1];
            s[0] = keep + __shfl_xor(send, 2);
        }
        float r = s[0];
        r += __shfl_xor(r, 1);
        const int j = (lane >> 1) & 31, pix = j >> 1, o = j & 1;
        const int y = y0 + (pix >> 2), x = x0 + (pix & 3);
        if ((lane & 1) == 0 && x < a.W && y < a.H) {
            const long n = (long)y * a.W + x;
            const float acc = r + a.bias[o];
            if (a.delta) a.delta[(b * N + n) * a.ld_delta + o] = acc;
            a.coords1[(b * 2 + o) * N + n] += acc;
        }
    }
}
int launch_flow_out(const PfFlowOutArgs& a, long total, void* stream) {
#ifndef PF_FLOW_OUT_TILE_MIN
#define PF_FLOW_OUT_TILE_MIN 65536
#endif
    if (a.C % 4 == 0 && a.ld % 4 == 0 && (long)a.B * a.H * a.W >= PF_FLOW_OUT_TILE_MIN) {      // >= 8 pairs of 512x1024: the tile form
        const int tpr = (a.W + 3) / 4, tpc = (a.H + 3) / 4;
        const long tiles = (long)a.B * tpc * tpr;
        long blocks = (tiles + 3) / 4;
        if (blocks > kMaxBlocks) blocks = kMaxBlocks;
        hipLaunchKernelGGL(pf_flow_out_tile, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, tiles, tpr, tpc);
        return (int)hipGetLastError();
    }
    if (a.C % 4 == 0 && a.ld % 4 == 0) {
        const int spr = (a.W + 3) / 4;
        const long strips = (long)a.B * a.H * spr;
        long blocks = (strips + 3) / 4;
        if (blocks > kMaxBlocks) blocks = kMaxBlocks;
        PfFlowOutN pn; pn.p[0] = a; pn.p[1] = a;
        hipLaunchKernelGGL(pf_flow_out_strip<1>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, pn, strips, spr);
        return (int)hipGetLastError();
    }
    return pf_launch_elem<PfFlowOutArgs, pf_flow_out_elem>(a, total, stream);
}
// Region sums: one block per (pixel chunk k, image b); per-thread fp64 accumulators for up to 8
// regions, wave shuffle reduction, then LDS across the 4 waves.  Deterministic (no atomics): the
// host adds the nblk partials.
__global__ void __launch_bounds__(256) pf_region_sum_kernel(const PfRegionSumArgs a) {
    __shared__ double red[4][8 * 3];
    const int k = blockIdx.x, b = blockIdx.y;
    const int chunk = (a.N + a.nblk - 1) / a.nblk;
    const int lo = k * chunk, hi = (lo + chunk < a.N) ? lo + chunk : a.N;
    double acc[8][3];
#pragma unroll
    for (int r = 0; r < 8; ++r) { acc[r][0] = 0; acc[r][1] = 0; acc[r][2] = 0; }
    for (int n = lo + threadIdx.x; n < hi; n += 256) {
        const double e = a.epe[(long)b * a.N + n], s = a.sd[(long)b * a.N + n];
        const double sw = a.weight ? s * (double)a.weight[n] : 0.0;
        const unsigned bits = a.bits[n];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if ((bits >> r) & 1u) { acc[r][0] += e; acc[r][1] += s; acc[r][2] += sw; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double v = acc[r][j];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) red[wave][r * 3 + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < a.R * 3) {
        const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        a.partials[((long)(b * a.nblk + k) * a.R) * 3 + threadIdx.x] = v;
    }
}

// block (k, b): pixels [k*chunk, (k+1)*chunk) of image b; deterministic two-stage sums as above
__device__ __forceinline__ void pf_seq_loss_block(const PfSeqLossArgs& a) {
    __shared__ double red[4][6];
    const int k = blockIdx.x, b = blockIdx.y;
    const int chunk = (a.N + a.nblk - 1) / a.nblk;
    const int lo = k * chunk, hi = (lo + chunk < a.N) ? lo + chunk : a.N;
    double sums[6] = {0, 0, 0, 0, 0, 0};
    for (int n = lo + threadIdx.x; n < hi; n += 256) pf_seq_loss_pixel(a, b, n, sums);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v = sums[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
        a.partials[((long)b * a.nblk + k) * 6 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) pf_seq_loss_kernel(const PfSeqLossArgs a) { pf_seq_loss_block(a); }
// blockIdx.z = the term (round 6: all predictions of a branch in one launch)
__global__ void __launch_bounds__(256) pf_seq_loss_batch_kernel(const PfSeqLossBatch t) {
    const PfSeqLossArgs a = pf_seq_loss_term(t, blockIdx.z);
    pf_seq_loss_block(a);
}
int launch_seq_loss(const PfSeqLossArgs& a, void* stream) {
    hipLaunchKernelGGL(pf_seq_loss_kernel, dim3((unsigned)a.nblk, (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
int launch_seq_loss_batch(const PfSeqLossBatch& t, void* stream) {
    hipLaunchKernelGGL(pf_seq_loss_batch_kernel, dim3((unsigned)t.common.nblk, (unsigned)t.common.B, (unsigned)t.n), dim3(256), 0,
                       (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) pf_sumsq_kernel(const PfSumSqArgs a) {
    __shared__ double red[4];
    const long chunk = (a.n + a.nblk - 1) / a.nblk;
    const long lo = blockIdx.x * chunk, hi = (lo + chunk < a.n) ? lo + chunk : a.n;
    double s = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) { const double v = a.x[i]; s += v * v; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
int launch_sumsq(const PfSumSqArgs& a, void* stream) {
    hipLaunchKernelGGL(pf_sumsq_kernel, dim3((unsigned)a.nblk), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int launch_region_sums(const PfRegionSumArgs& a, void* stream) {
    hipLaunchKernelGGL(pf_region_sum_kernel, dim3((unsigned)a.nblk, (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// Backward of the convex upsampling on a wavefront: one coarse pixel per wave, lane = sub-pixel 8i + j.
// The 9 softmax logits of a lane are 9 coalesced 256-byte rows of the mask (the per-element version read them
// at a 2.3 KB stride) and the 64 contributions to each of the 9 neighbours' coarse-flow gradients are summed
// with a butterfly before ONE atomic per (neighbour, component) instead of 64 colliding ones.
// Same arithmetic per lane as pf_upsample_bwd_elem (pf_elem.h), which stays the emulated / checked statement.
__global__ void __launch_bounds__(kBlock) pf_upsample_bwd_wave(const PfUpsampleBwdArgs a, const long rows) {
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, j = lane & 7;
    const long N = (long)a.H * a.W;
    const int W8 = 8 * a.W;
    const long plane = (long)W8 * 8 * a.H;
    long prow = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (kBlock / 64);
    for (; prow < rows; prow += stride) {
        const long b = prow / N;
        const int n = (int)(prow % N);
        const int y = n / a.W, x = n % a.W;
        const long fine = (long)(8 * y + i) * W8 + 8 * x + j;
        const float gu = a.g[(b * 2 + 0) * plane + fine], gv = a.g[(b * 2 + 1) * plane + fine];
        const float* mrow = a.mask + prow * a.ld + lane;
        float w[9], sk[9];
        // every load of the pixel up front and unconditional (a neighbour outside the map reads the pixel itself and is not
        // used): inside the per-neighbour branch each pair of coordinate loads was its own round trip to memory, nine in a row
        float cu[9], cv[9];
        bool ok[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            ok[k] = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;           // wave-uniform
            const long p = ok[k] ? (long)yy * a.W + xx : (long)n;
            cu[k] = a.coords1[(b * 2 + 0) * N + p];
            cv[k] = a.coords1[(b * 2 + 1) * N + p];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { w[k] = mrow[64 * k]; mx = fmaxf(mx, w[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { w[k] = expf(w[k] - mx); den = den + w[k]; }
        float su[9], sv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            w[k] = w[k] / den;
            const float fu = 8.f * (cu[k] - (float)xx), fv = 8.f * (cv[k] - (float)yy);
            sk[k] = ok[k] ? gu * fu + gv * fv : 0.f;
            su[k] = 8.f * w[k] * gu; sv[k] = 8.f * w[k] * gv;
        }
        // the 18 butterflies are independent chains: interleaved, not one after the other
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) { su[k] += __shfl_xor(su[k], m); sv[k] += __shfl_xor(sv[k], m); }
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (ok[k] && lane == 0) {
                const long p = (long)(y + k / 3 - 1) * a.W + (x + k % 3 - 1);
                atomicAdd(a.d_flow + (b * 2 + 0) * N + p, su[k]);
                atomicAdd(a.d_flow + (b * 2 + 1) * N + p, sv[k]);
            }
            dot = dot + w[k] * sk[k];
        }
        float* drow = a.d_mask + prow * a.ld_d + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k) drow[64 * k] = w[k] * (sk[k] - dot);
    }
}

// Backward of the DCCL lookup, one query per wave (round 5).  pf_lookup_bwd_elem (pf_elem.h) issues eight global atomics per
// (query, channel): 2 592 per query, and on the coarse levels most of them collide -- the 81 taps of level 3 land on a 3 x 3
// patch of the other view's map, 36 serialised atomics per address.  Here a wave owns a query: its 324 taps (taps that are
// neighbours in x on consecutive lanes) add into wave-private LDS windows, one per level and view -- 12 x 12 cells around the
// own view's 10 x 10 footprint, 16 x 16 around the warped centre tap for the other view, x relative to the window modulo the
// map width (the panorama wraps) -- and the window cells that were touched go to memory as ONE atomic each (~600 per query,
// no two of a launch on the same address except where a narrow map folds a window onto itself).  A corner outside its window
// (a warp that tears near a pole) falls back to the direct global atomic.  Per tap the arithmetic (pf_pymod / pf_taps0 /
// pf_apply) is pf_lookup_bwd_elem's; the sums differ from it only in the order of the additions.  77 -> 42 us per launch at the
// training crop (48 x 64 queries; profiles/microbench_train_elem.py).  A workgroup per query (one tap per thread, two barriers)
// measured 48 us.
constexpr int LKB_OWN = 12, LKB_OTH = 16;
constexpr int LKB_OWN_CELLS = PF_CORR_LEVELS * LKB_OWN * LKB_OWN, LKB_CELLS = LKB_OWN_CELLS + PF_CORR_LEVELS * LKB_OTH * LKB_OTH;   // 1 600
struct LkbXY { int x[2], y[2]; float w[4]; };           // pf_taps0 with the corners as (x, y): w = nw, ne, sw, se
__device__ __forceinline__ LkbXY lkb_taps(float x, float y, int H, int W) {
    LkbXY t;
    const float ix = pf_roundtrip(x, W), iy = pf_roundtrip(y, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ix - fx, wy = iy - fy;
    const float ex = 1.f - wx, ey = 1.f - wy;
    const bool xin0 = (fx >= 0.f) && (fx <= (float)(W - 1));
    const bool xin1 = (fx >= -1.f) && (fx <= (float)(W - 2));
    const bool yin0 = (fy >= 0.f) && (fy <= (float)(H - 1));
    const bool yin1 = (fy >= -1.f) && (fy <= (float)(H - 2));
    t.x[0] = xin0 ? (int)fx : 0; t.x[1] = xin1 ? (int)fx + 1 : 0;
    t.y[0] = yin0 ? (int)fy : 0; t.y[1] = yin1 ? (int)fy + 1 : 0;
    t.w[0] = (xin0 && yin0) ? ey * ex : 0.f;
    t.w[1] = (xin1 && yin0) ? ey * wx : 0.f;
    t.w[2] = (xin0 && yin1) ? wy * ex : 0.f;
    t.w[3] = (xin1 && yin1) ? wy * wx : 0.f;
    return t;
}
// window origin from a coordinate's floor (finite and near the map, else 0: such a tap has no weight anyway)
__device__ __forceinline__ int lkb_origin(float p, int size, int back) {
    const float f = floorf(pf_roundtrip(p, size));
    return (f > -64.f && f < (float)(size + 64)) ? (int)f - back : 0;
}
__global__ void __launch_bounds__(kBlock) pf_lookup_bwd_rows(const PfLookupBwdArgs a, const long rows) {
    __shared__ float win[(kBlock / 64) * LKB_CELLS];
    __shared__ int origins[(kBlock / 64) * 4 * PF_CORR_LEVELS];            // per wave and level: own x, own y, other x, other y
    const int lane = threadIdx.x & 63;
    float* const mine = win + (threadIdx.x >> 6) * LKB_CELLS;
    int* const org = origins + (threadIdx.x >> 6) * 4 * PF_CORR_LEVELS;
    for (int c = lane; c < LKB_CELLS; c += 64) mine[c] = 0.f;
    const long N = (long)a.H * a.W;
    long row = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (kBlock / 64);
    // (no workgroup barrier anywhere: the four waves of a block walk their own queries.  A wave's LDS operations complete in
    // issue order, so its reads below see its adds above; the fences only keep hipcc from reordering them)
    for (; row < rows; row += stride) {
        const long b = row / N, n = row % N;
        const float c0x = a.coords[(b * 2 + 0) * N + n], c0y = a.coords[(b * 2 + 1) * N + n];
        if (lane < PF_CORR_LEVELS) {
            // window origins of level `lane`: own = one cell before tap (0, 0)'s floor; other = seven cells before the warped
            // centre tap.  In LDS, read with the level as index (as register arrays hipcc spills them to scratch memory for that)
            const int l = lane;
            const int Hl = a.H >> l, Wl = a.W >> l;
            const float inv = 1.f / (float)(1 << l);
            const float cx = c0x * inv, cy = c0y * inv;
            org[4 * l] = lkb_origin(pf_pymod(cx - (float)PF_CORR_RADIUS, (float)Wl), Wl, 1);
            org[4 * l + 1] = lkb_origin(cy - (float)PF_CORR_RADIUS, Hl, 1);
            const PfTaps tg = pf_taps0(pf_pymod(cx, (float)a.W), cy, a.H, a.W);
            const float gx = pf_apply(tg, a.g_w2c), gy = pf_apply(tg, a.g_w2c + N);
            org[4 * l + 2] = lkb_origin(pf_pymod(gx, (float)Wl), Wl, LKB_OTH / 2 - 1);
            org[4 * l + 3] = lkb_origin(gy, Hl, LKB_OTH / 2 - 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = lane; k < PF_CORR_CH; k += 64) {
            const int lvl = k / PF_TAPS, tap = k % PF_TAPS;
            const int ta = tap % 9, tb = tap / 9;                 // lanes walk x first (channel = 81 lvl + 9 ta + tb, ta offsets x)
            const int kk = lvl * PF_TAPS + ta * 9 + tb;
            const int Hl = a.H >> lvl, Wl = a.W >> lvl;
            const float inv = 1.f / (float)(1 << lvl);
            const float cx = c0x * inv + (float)(ta - PF_CORR_RADIUS);
            const float cy = c0y * inv + (float)(tb - PF_CORR_RADIUS);
            const long lsz = (long)Hl * Wl;
            const float go = a.d_own[row * a.ld + kk], gr = a.d_raw[row * a.ld + kk];
            if (a.clear_raw) a.clear_raw[row * a.ld + kk] = 0.f;
            const int ox = org[4 * lvl], oy = org[4 * lvl + 1], qx = org[4 * lvl + 2], qy = org[4 * lvl + 3];
            float* own = a.g_own[0];
            float* oth = a.g_other[0];
#pragma unroll
            for (int l = 1; l < PF_CORR_LEVELS; ++l)          // (selects: indexing the by-value argument with lvl would put it in scratch)
                if (lvl == l) { own = a.g_own[l]; oth = a.g_other[l]; }
            own += row * lsz; oth += row * lsz;
            float* const wo = mine + lvl * (LKB_OWN * LKB_OWN);
            float* const wt = mine + LKB_OWN_CELLS + lvl * (LKB_OTH * LKB_OTH);
            const LkbXY t = lkb_taps(pf_pymod(cx, (float)Wl), cy, Hl, Wl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t.w[j] == 0.f) continue;
                const int x = t.x[j & 1], y = t.y[j >> 1];
                int xr = x - ox; xr = xr < 0 ? xr + Wl : (xr >= Wl ? xr - Wl : xr);
                const int yr = y - oy;
                if ((unsigned)xr < (unsigned)LKB_OWN && (unsigned)yr < (unsigned)LKB_OWN) atomicAdd(wo + yr * LKB_OWN + xr, go * t.w[j]);
                else atomicAdd(own + (long)y * Wl + x, go * t.w[j]);
            }
            // cross view: the level-i coordinates index the LEVEL-0 grid (core/corr.py:132-133)
            const PfTaps tg = pf_taps0(pf_pymod(cx, (float)a.W), cy, a.H, a.W);
            const float gx = pf_apply(tg, a.g_w2c), gy = pf_apply(tg, a.g_w2c + N);
            const LkbXY to = lkb_taps(pf_pymod(gx, (float)Wl), gy, Hl, Wl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (to.w[j] == 0.f) continue;
                const int x = to.x[j & 1], y = to.y[j >> 1];
                int xr = x - qx; xr = xr < 0 ? xr + Wl : (xr >= Wl ? xr - Wl : xr);
                const int yr = y - qy;
                if ((unsigned)xr < (unsigned)LKB_OTH && (unsigned)yr < (unsigned)LKB_OTH) atomicAdd(wt + yr * LKB_OTH + xr, gr * to.w[j]);
                else atomicAdd(oth + (long)y * Wl + x, gr * to.w[j]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int c = lane; c < LKB_CELLS; c += 64) {
            const float v = mine[c];
            if (v == 0.f) continue;
            mine[c] = 0.f;                                    // the window is clean again for the wave's next query
            const bool c_own = c < LKB_OWN_CELLS;
            const int cc = c_own ? c : c - LKB_OWN_CELLS;
            const int cl = c_own ? cc / (LKB_OWN * LKB_OWN) : cc / (LKB_OTH * LKB_OTH);
            const int cell = c_own ? cc % (LKB_OWN * LKB_OWN) : cc % (LKB_OTH * LKB_OTH);
            const int yr = c_own ? cell / LKB_OWN : cell / LKB_OTH, xr = c_own ? cell % LKB_OWN : cell % LKB_OTH;
            const int Wc = a.W >> cl;
            const int bx = org[4 * cl + (c_own ? 0 : 2)], by = org[4 * cl + (c_own ? 1 : 3)];
            float* dst = c_own ? a.g_own[0] : a.g_other[0];
#pragma unroll
            for (int l = 1; l < PF_CORR_LEVELS; ++l)
                if (cl == l) dst = c_own ? a.g_own[l] : a.g_other[l];
            // a filled cell came from an in-range (x, y) with x = origin + xr modulo the width (a narrow map folds the window)
            int x = (bx + xr) % Wc; x = x < 0 ? x + Wc : x;
            atomicAdd(dst + row * ((long)(a.H >> cl) * Wc) + (long)(by + yr) * Wc + x, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// K4(c) four channels per thread: the taps (g_back) are per pixel, so a thread's four corner reads and its store are
// 16-byte accesses of one channel-last row.  Same per-element arithmetic and order as pf_combine_elem (pf_elem.h).
__global__ void __launch_bounds__(kBlock) pf_combine_vec4(const PfCombineArgs a, const long total) {   // total = rows * 81
    const long N = (long)a.H * a.W;
    long idx = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; idx < total; idx += stride) {
        const int k = (int)(idx % (PF_CORR_CH / 4)) * 4;
        const long row = idx / (PF_CORR_CH / 4);
        const long b = row / N, n = row % N;
        const PfTaps t = pf_taps0(pf_pymod(a.g_back[n], (float)a.W), a.g_back[N + n], a.H, a.W);
        const float* base = a.raw + b * N * a.ld + k;
        const float4 r0 = *reinterpret_cast<const float4*>(base + (long)t.idx[0] * a.ld);
        const float4 r1 = *reinterpret_cast<const float4*>(base + (long)t.idx[1] * a.ld);
        const float4 r2 = *reinterpret_cast<const float4*>(base + (long)t.idx[2] * a.ld);
        const float4 r3 = *reinterpret_cast<const float4*>(base + (long)t.idx[3] * a.ld);
        const float4 o = *reinterpret_cast<const float4*>(a.own + row * a.ld + k);
        float4 c;
        c.x = r0.x * t.w[0]; c.x = c.x + r1.x * t.w[1]; c.x = c.x + r2.x * t.w[2]; c.x = c.x + r3.x * t.w[3];
        c.y = r0.y * t.w[0]; c.y = c.y + r1.y * t.w[1]; c.y = c.y + r2.y * t.w[2]; c.y = c.y + r3.y * t.w[3];
        c.z = r0.z * t.w[0]; c.z = c.z + r1.z * t.w[1]; c.z = c.z + r2.z * t.w[2]; c.z = c.z + r3.z * t.w[3];
        c.w = r0.w * t.w[0]; c.w = c.w + r1.w * t.w[1]; c.w = c.w + r2.w * t.w[2]; c.w = c.w + r3.w * t.w[3];
        float4 out;
        out.x = o.x + c.x; out.y = o.y + c.y; out.z = o.z + c.z; out.w = o.w + c.w;
        *reinterpret_cast<float4*>(a.out + row * a.ld_out + k) = out;
    }
}

int launch_combine(const PfCombineArgs& a, long total, void* stream) {
    const bool vec = (a.ld % 4 == 0) && (a.ld_out % 4 == 0) &&
                     ((((uintptr_t)a.own) | ((uintptr_t)a.raw) | ((uintptr_t)a.out)) % 16 == 0);
    if (!vec) return pf_launch_elem<PfCombineArgs, pf_combine_elem>(a, total, stream);
    const long n4 = total / 4;
    if (n4 <= 0) return PF_OK;
    long blocks = (n4 + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(pf_combine_vec4, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, n4);
    return (int)hipGetLastError();
}

int launch_upsample_bwd(const PfUpsampleBwdArgs& a, long total, void* stream) {
    const long rows = total / 64;
    if (rows <= 0) return PF_OK;
    long blocks = (rows + 3) / 4;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(pf_upsample_bwd_wave, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, rows);
    return (int)hipGetLastError();
}

int launch_lookup_bwd(const PfLookupBwdArgs& a, long total, void* stream) {
    // PRIORFLOW_LOOKUP_BWD=elem: the per-(query, channel) scatter (A/B and the statement the emulation runs)
    static const bool per_elem = [] { const char* e = getenv("PRIORFLOW_LOOKUP_BWD"); return e && e[0] == 'e'; }();
    if (per_elem) return pf_launch_elem<PfLookupBwdArgs, pf_lookup_bwd_elem>(a, total, stream);
    const long rows = total / PF_CORR_CH;
    if (rows <= 0) return PF_OK;
    long blocks = (rows + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(pf_lookup_bwd_rows, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, rows);
    return (int)hipGetLastError();
}

}  // namespace

#define PF_LOOKUP_BWD_LAUNCH(a, total, stream) launch_lookup_bwd(a, total, stream)
#define PF_UPSAMPLE_BWD_LAUNCH(a, total, stream) launch_upsample_bwd(a, total, stream)
#define PF_COMBINE_LAUNCH(a, total, stream) launch_combine(a, total, stream)
#ifdef PF_LOOKUP_WAVES
#define PF_LOOKUP_LAUNCH(a, total, stream) launch_lookup(a, total, stream)
#else
// the wave-cooperative window kernel (pf_lookup.hip) is OPT-IN (PRIORFLOW_LOOKUP_WIN=1; pf_lookup_win_launch answers -100
// otherwise: it measured 45 us against 28 us per launch, profiles/r3_final_ab_lookup_win.txt); pf_lookup_elem, the per-thread
// kernel, is the default, the scalar statement the window kernel is bit-identical to, and what the host emulation runs
int pf_lookup_win_launch(const PfLookupArgs& a, void* stream);
static int pf_lookup_dispatch(const PfLookupArgs& a, long total, void* stream) {
    const int rc = pf_lookup_win_launch(a, stream);
    return rc == -100 ? pf_launch_elem<PfLookupArgs, pf_lookup_elem>(a, total, stream) : rc;
}
#define PF_LOOKUP_LAUNCH(a, total, stream) pf_lookup_dispatch(a, total, stream)
#endif
#define PF_REGION_SUM_LAUNCH(a, stream) launch_region_sums(a, stream)
#define PF_SEQ_LOSS_LAUNCH(a, stream) launch_seq_loss(a, stream)
#define PF_SEQ_LOSS_BATCH_LAUNCH(t, stream) launch_seq_loss_batch(t, stream)
#define PF_SUMSQ_LAUNCH(a, stream) launch_sumsq(a, stream)
#define PF_FLOW_OUT_LAUNCH(a, total, stream) launch_flow_out(a, total, stream)
#define PF_NORM_ACT_LAUNCH(a, total, stream) launch_norm_act(a, total, stream)
#define PF_STATS_LAUNCH launch_stats
#define PF_STATS_FINAL_LAUNCH launch_stats_final
#define PF_PAIR_FINAL_LAUNCH launch_pair_final
#define PF_LAUNCH(name, args, total, stream) \
    pf_launch_elem<decltype(args), pf_##name##_elem>(args, total, stream)

// device build: route pf_conv2d_direct[_group] to the MFMA small-Cin kernel when the shape allows
// (the n problems of a group share every shape parameter; checked by the entry point)
static int pf_direct_conv_dispatch_n(const PfDirectConvArgs* ds, int n, long total, void* stream) {
    const PfDirectConvArgs& d = ds[0];
    const int K = d.KH * d.KW * d.Cin;
    const size_t lds = ((size_t)d.KH * (31 * d.stride + d.KW) * (d.Cin | 1) + 4 + 3 * 16 * 64) * 4;
    if (lds <= 60 * 1024 && (K + 1) / 2 <= 160 &&
        d.KH * (31 * d.stride + d.KW) * d.Cin <= 16 * 256 && d.Cin <= 255) {
        PfSmallConvMulti m;
        for (int i = 0; i < n; ++i) {
            const PfDirectConvArgs& e = ds[i];
            PfSmallConvArgs& a = m.p[i];
            a.in = e.in; a.ld_in = e.ld_in; a.c_in_off = e.c_in_off; a.Cin = e.Cin; a.nchw = e.nchw;
            a.w = e.w; a.bias = e.bias; a.out = e.out; a.ld_out = e.ld_out; a.c_out_off = e.c_out_off; a.Cout = e.Cout;
            a.B = e.B; a.H = e.Hin; a.W = e.Win; a.KH = e.KH; a.KW = e.KW; a.stride = e.stride; a.relu = e.relu;
            a.Ho = e.H; a.Wo = e.W;
            a.out_split = e.out_split; a.lds_out = e.lds_out;
        }
        for (int i = n; i < 4; ++i) m.p[i] = m.p[0];
        bool valu = true;
        for (int i = 0; i < n; ++i) valu = valu && stem7x7c2_ok(m.p[i]);
        if (valu) {
            // the 2 -> 128 flow stems of the motion encoders: MFMA form (pf_flow_stem.hip, round 4) unless switched off
            bool mf = true;
            for (int i = 0; i < n; ++i) mf = mf && m.p[i].Cout == 128 && m.p[i].B == m.p[0].B && m.p[i].H == m.p[0].H && m.p[i].W == m.p[0].W;
            if (mf) {
                PfFlowStemMulti fm;
                for (int i = 0; i < 4; ++i) {
                    const PfSmallConvArgs& a = m.p[i < n ? i : 0];
                    PfFlowStemProblem& q = fm.p[i];
                    q.in = a.in; q.ld_in = a.ld_in; q.c_in_off = a.c_in_off; q.w = a.w; q.bias = a.bias;
                    q.out = a.out; q.ld_out = a.ld_out; q.c_out_off = a.c_out_off; q.out_split = a.out_split; q.lds_out = a.lds_out;
                    q.relu = a.relu;
                }
                fm.B = m.p[0].B; fm.H = m.p[0].H; fm.W = m.p[0].W;
                return pf_flow_stem_launch(fm, n, stream);
            }
            return launch_stem7x7c2(m, n, stream);
        }
        for (int i = 0; i < n; ++i)
            if (m.p[i].out_split || !m.p[i].out) return PF_ERR_BAD_SHAPE;       // the MFMA form writes fp32 rows only
        return launch_small_conv(m, n, stream);
    }
    for (int i = 0; i < n; ++i) {
        const int rc = pf_launch_elem<PfDirectConvArgs, pf_direct_conv_elem>(ds[i], total, stream);
        if (rc != PF_OK) return rc;
    }
    return PF_OK;
}
static int pf_direct_conv_dispatch(const PfDirectConvArgs& d, long total, void* stream) {
    return pf_direct_conv_dispatch_n(&d, 1, total, stream);
}
#define PF_DIRECT_CONV_GROUP_LAUNCH(ds, n, total, stream) pf_direct_conv_dispatch_n(ds, n, total, stream)
#define PF_DIRECT_CONV_LAUNCH(a, total, stream) pf_direct_conv_dispatch(a, total, stream)

#include "pf_api_elem.inc"

extern "C" int pf_warp_gcorr(const float* f1, const float* f2, const float* coords, int add_grid,
                             float* dst, int dst_ld, int dst_off, int B, int H8, int W8, int C,
                             void* stream) {
    PfWarpGcorrArgs a;
    const int rc = pf_warp_gcorr_fill(a, f1, f2, coords, add_grid, dst, dst_ld, dst_off, B, H8, W8, C);
    if (rc != PF_OK) return rc;
    const long rows = (long)B * H8 * W8;
    if (C == 256) {
        long blocks = (rows + 3) / 4;
        if (blocks > kMaxBlocks) blocks = kMaxBlocks;
        hipLaunchKernelGGL(pf_warp_gcorr_wave, dim3((unsigned)blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, a, rows);
        return (int)hipGetLastError();
    }
    return PF_LAUNCH(warp_gcorr, a, rows * 4, stream);
}

// Confidence stem of the ODDC motion encoder in ONE launch (core/update.py:177-178,193-194):
//   out = relu(conv3x3_{32->16}(relu(conv3x3_{8->32}(x))))      (zero padding 1, stride 1, exact fp32 FMA chains)
// 113 MFLOP per 64x128 map: a latency job (round 1 ran it as two launches of the small-Cin MFMA kernel, 34.7 + 20.1 us
// in the replay).  A workgroup owns a 4 x 16 pixel tile: the 8 x 20 input patch and the 6 x 18 intermediate map live
// in LDS, so the intermediate activation never goes to memory.  Stage 1: thread = (channel pair, 1 of 16 pixel
// lanes), its 2 x 72 weights in registers, 72 patch values per pixel from LDS.  Stage 2: thread = (pixel, group of 4
// output channels), weights [288][16] read from LDS (the same row for every pixel: a broadcast), k ascending.
constexpr int CS_TH = 4, CS_TW = 16, CS_CIN = 8, CS_MID = 32, CS_OUT = 16;
constexpr int CS_PW = CS_TW + 4, CS_PH = CS_TH + 4, CS_MW = CS_TW + 2, CS_MH = CS_TH + 2, CS_MLD = CS_MID + 4;
struct PfConfStemArgs {
    const float* in; int ld_in, off_in;
    const float* w1; const float* b1;       // [9*8][32], [32]
    const float* w2; const float* b2;       // [9*32][16], [16]
    float* out; int ld_out, off_out;
    void* out_split; int lds_out;
    int B, H, W;
};
__global__ void __launch_bounds__(256) pf_conf_stem_kernel(const PfConfStemArgs a) {
    __shared__ __attribute__((aligned(16))) float patch[CS_PH * CS_PW * CS_CIN];     //  5.0 KB
    __shared__ __attribute__((aligned(16))) float mid[CS_MH * CS_MW * CS_MLD];       // 15.2 KB
    __shared__ __attribute__((aligned(16))) float w2s[9 * CS_MID * CS_OUT];          // 18.0 KB
    const int tid = threadIdx.x;
    const int tiles_x = (a.W + CS_TW - 1) / CS_TW, tiles_y = (a.H + CS_TH - 1) / CS_TH;
    const int tile = blockIdx.x;
    const int x0 = (tile % tiles_x) * CS_TW, y0 = ((tile / tiles_x) % tiles_y) * CS_TH;
    const long pix0 = (long)(tile / (tiles_x * tiles_y)) * a.H * a.W;
    // ---- stage 0: input patch (zero outside the map), second-layer weights; first-layer weights to registers ----
    for (int i = tid; i < CS_PH * CS_PW * 2; i += 256) {
        const int p = i >> 1, half = i & 1;
        const int y = y0 + p / CS_PW - 2, x = x0 + p % CS_PW - 2;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < a.H && x >= 0 && x < a.W)
            v = *reinterpret_cast<const float4*>(a.in + (pix0 + (long)y * a.W + x) * a.ld_in + a.off_in + 4 * half);
        *reinterpret_cast<float4*>(patch + p * CS_CIN + 4 * half) = v;
    }
    for (int i = tid; i < 9 * CS_MID * CS_OUT / 4; i += 256)
        reinterpret_cast<float4*>(w2s)[i] = reinterpret_cast<const float4*>(a.w2)[i];
    const int cp = tid & 15, pl = tid >> 4;
    float2 w1r[9 * CS_CIN];
#pragma unroll
    for (int k = 0; k < 9 * CS_CIN; ++k) w1r[k] = *reinterpret_cast<const float2*>(a.w1 + k * CS_MID + 2 * cp);
    const float2 bias1 = *reinterpret_cast<const float2*>(a.b1 + 2 * cp);
    __syncthreads();
    // ---- stage 1: intermediate map on the tile + 1-pixel ring; positions outside the image are conv2's zero padding ----
    for (int p = pl; p < CS_MH * CS_MW; p += 16) {
        const int my = p / CS_MW, mx = p % CS_MW;
        const int y = y0 + my - 1, x = x0 + mx - 1;
        float2 acc = bias1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* src = patch + ((my + t / 3) * CS_PW + mx + t % 3) * CS_CIN;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc.x = __builtin_fmaf(vv[c], w1r[t * 8 + c].x, acc.x);
                acc.y = __builtin_fmaf(vv[c], w1r[t * 8 + c].y, acc.y);
            }
        }
        const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
        float2 r = {inside ? fmaxf(acc.x, 0.f) : 0.f, inside ? fmaxf(acc.y, 0.f) : 0.f};
        *reinterpret_cast<float2*>(mid + p * CS_MLD + 2 * cp) = r;
    }
    __syncthreads();
    // ---- stage 2 -----------------------------------------------------------------------------------------
    const int px = tid >> 2, cg = tid & 3;
    const int oy = px / CS_TW, ox = px % CS_TW;
    float4 acc = *reinterpret_cast<const float4*>(a.b2 + 4 * cg);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float* m = mid + ((oy + t / 3) * CS_MW + ox + t % 3) * CS_MLD;
        const float* w = w2s + t * CS_MID * CS_OUT + 4 * cg;
#pragma unroll
        for (int c4 = 0; c4 < CS_MID / 4; ++c4) {
            const float4 mv = *reinterpret_cast<const float4*>(m + 4 * c4);
            const float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 wv = *reinterpret_cast<const float4*>(w + (4 * c4 + i) * CS_OUT);
                acc.x = __builtin_fmaf(mm[i], wv.x, acc.x); acc.y = __builtin_fmaf(mm[i], wv.y, acc.y);
                acc.z = __builtin_fmaf(mm[i], wv.z, acc.z); acc.w = __builtin_fmaf(mm[i], wv.w, acc.w);
            }
        }
    }
    const int y = y0 + oy, x = x0 + ox;
    if (y < a.H && x < a.W) {
        const float4 r = {fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
        const long row = pix0 + (long)y * a.W + x;
        if (a.out != nullptr) *reinterpret_cast<float4*>(a.out + row * a.ld_out + a.off_out + 4 * cg) = r;
        if (a.out_split != nullptr) {
            const float v4[4] = {r.x, r.y, r.z, r.w};
            pf_split_store_n<4>(pf_split_ptr(a.out_split, row, a.lds_out, a.off_out + 4 * cg), v4);
        }
    }
}

extern "C" int pf_conf_stem(const float* in, int ld_in, int off_in, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* out, int ld_out, int off_out,
                            void* out_split, int lds_out, int B, int H8, int W8, void* stream) {
    if (!in || !w1 || !b1 || !w2 || !b2 || (!out && !out_split)) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 0 || W8 <= 0) return PF_ERR_BAD_SHAPE;
    if (off_in < 0 || off_in + CS_CIN > ld_in || off_out < 0 || (out && off_out + CS_OUT > ld_out)) return PF_ERR_BAD_ARG;
    if (out_split && off_out + CS_OUT > lds_out * 32) return PF_ERR_BAD_ARG;
    if ((ld_in | off_in | off_out) & 3 || (out && (ld_out & 3))) return PF_ERR_BAD_SHAPE;          // 16-byte rows
    PfConfStemArgs a;
    a.out_split = out_split; a.lds_out = lds_out;
    a.in = in; a.ld_in = ld_in; a.off_in = off_in; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.out = out; a.ld_out = ld_out; a.off_out = off_out; a.B = B; a.H = H8; a.W = W8;
    const long tiles = (long)B * ((H8 + CS_TH - 1) / CS_TH) * ((W8 + CS_TW - 1) / CS_TW);
    hipLaunchKernelGGL(pf_conf_stem_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int pf_motion_prep(const float* c1a, const float* c1b, const float* g_w2c, const float* g_c2w,
                              const float* f1a, const float* f2a, float* flow4_a, float* flow2_b,
                              float* xa, int xa_ld, int xa_off, float* xb, int xb_ld, int xb_off,
                              void* xa_split, int xa_lds, void* xb_split, int xb_lds,
                              float* conf, int conf_ld, int B, int H8, int W8, int C, void* stream) {
    if (!c1a || !c1b || !g_w2c || !g_c2w || !f1a || !f2a || !flow4_a || !flow2_b || !conf) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 1 || W8 <= 1 || C != 256 || conf_ld < 8) return PF_ERR_BAD_SHAPE;
    if (xa && (xa_off < 0 || xa_off + 4 > xa_ld)) return PF_ERR_BAD_ARG;
    if (xb && (xb_off < 0 || xb_off + 2 > xb_ld)) return PF_ERR_BAD_ARG;
    if (xa_split && (xa_off < 0 || xa_off + 4 > xa_lds * 32)) return PF_ERR_BAD_ARG;
    if (xb_split && (xb_off < 0 || xb_off + 2 > xb_lds * 32)) return PF_ERR_BAD_ARG;
    PfMotionPrepArgs a;
    a.xa_split = xa_split; a.xa_lds = xa_lds; a.xb_split = xb_split; a.xb_lds = xb_lds;
    a.c1a = c1a; a.c1b = c1b; a.g_w2c = g_w2c; a.g_c2w = g_c2w; a.f1 = f1a; a.f2 = f2a;
    a.flow4_a = flow4_a; a.flow2_b = flow2_b;
    a.xa = pf_dst(xa, xa_ld, xa_off); a.xb = pf_dst(xb, xb_ld, xb_off);
    a.conf = conf; a.conf_ld = conf_ld; a.B = B; a.H = H8; a.W = W8;
    const long rows = (long)B * H8 * W8;
    hipLaunchKernelGGL(pf_motion_prep_kernel, dim3((unsigned)((rows + MP_PIX - 1) / MP_PIX)), dim3(256), 0,
                       (hipStream_t)stream, a, rows);
    return (int)hipGetLastError();
}

extern "C" const char* pf_version(void) {
    return "priorflow-hip r6 gfx950 (bf16x3 / exact-fp32 MFMA implicit-GEMM convs, all-DMA convs on pre-split activations, weights-stationary encoder convs, MFMA encoder stem, role-split corr+pyramid, fused combine+1x1, HIP encoders, HIP training backward as one loop node, capturable step)";
}
