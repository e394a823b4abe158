// pf_conv2d: argument validation, tile choice and dispatch (host side).  The kernels live in pf_conv_mfma.hip (generic,
// halo and role-specialised kernels on fp32 activations) and pf_conv_dma.hip (all-DMA kernel on pre-split activations).
#include <stdlib.h>
#include "pf_conv_priv.h"

using namespace pfconv;

// validation + geometry shared by pf_conv2d and pf_conv2d_tile
static int conv_prepare(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8,
                        ConvGroups& grp, ConvGeom& g, int& max_cout) {   // H8, W8: OUTPUT map size
    if (!descs || ngroups < 1 || ngroups > MAX_GROUPS) return PF_ERR_BAD_ARG;
    if (B <= 0 || H8 <= 0 || W8 <= 0) return PF_ERR_BAD_SHAPE;
    max_cout = 0;
    const pf_conv_desc& f = descs[0];
    for (int i = 0; i < ngroups; ++i) {
        const pf_conv_desc& d = descs[i];
        if ((!d.in0 && !d.in0_split) || !d.weight || !d.bias || (!d.out && !d.out_split)) return PF_ERR_BAD_ARG;
        if (d.c0 <= 0 || d.c1 < 0 || (d.c1 > 0 && !d.in1 && !d.in1_split)) return PF_ERR_BAD_ARG;
        // split twins (include/priorflow_hip.h): bf16x3 arithmetic, chunk-aligned slices that fit their rows
        if ((d.in0_split || d.in1_split || d.out_split || d.aux_split) && d.precision != PF_PREC_BF16X3) return PF_ERR_BAD_ARG;
        if (d.in0_split && ((d.off0 & 31) || d.lds0 * 32 < d.off0 + d.c0)) return PF_ERR_BAD_SHAPE;
        if (d.in1_split && d.c1 > 0 && ((d.off1 & 31) || (d.c0 & 31) || d.lds1 * 32 < d.off1 + d.c1)) return PF_ERR_BAD_SHAPE;
        // The all-DMA kernel copies whole 32-channel chunks: a slice that ends inside a chunk must end at the END OF THE ROW, where
        // the twin's columns past the logical width are zero by contract (anywhere else it would multiply a neighbour's live
        // columns -- possibly Inf / NaN -- by the zero-padded weights)
        if (d.in0_split && d.c1 == 0 && (d.c0 & 31) && d.off0 + ((d.c0 + 31) & ~31) != d.lds0 * 32) return PF_ERR_BAD_SHAPE;
        if (d.in1_split && d.c1 > 0 && (d.c1 & 31) && d.off1 + ((d.c1 + 31) & ~31) != d.lds1 * 32) return PF_ERR_BAD_SHAPE;
        if (d.out_split && ((d.off_out & 31) || d.lds_out <= 0)) return PF_ERR_BAD_SHAPE;
        if (d.aux_split && d.lds_aux * 32 < 128) return PF_ERR_BAD_SHAPE;
        if (d.pre && (d.off_pre < 0 || d.off_pre + d.cout > d.ld_pre)) return PF_ERR_BAD_SHAPE;
        if ((d.in0_split != nullptr) != (f.in0_split != nullptr) || (d.c1 > 0 && (d.in1_split != nullptr) != (d.in0_split != nullptr)))
            return PF_ERR_BAD_ARG;              // every group and both segments agree on the operand form
        // same geometry in every group: one kernel, one K loop
        if (d.kh != f.kh || d.kw != f.kw || d.c0 + d.c1 != f.c0 + f.c1) return PF_ERR_BAD_SHAPE;
        // odd k: window [-k/2, k/2]; even k: [-k/2, k/2 - 1] (the space-to-depth form of the 7x7/2 stem)
        if (d.kh < 1 || d.kw < 1 || d.kh > 7 || d.kw > 7 || d.cout <= 0) return PF_ERR_BAD_SHAPE;
        // 16-byte loads: every channel offset / stride must be a multiple of 4 floats
        if ((d.off0 | d.c0 | d.c1) & 3) return PF_ERR_BAD_SHAPE;
        if (d.in0 && (d.ld0 & 3)) return PF_ERR_BAD_SHAPE;
        if (d.c1 > 0 && ((d.off1 & 3) || (d.in1 && (d.ld1 & 3)) || (d.c0 % KC) != 0)) return PF_ERR_BAD_SHAPE;
        if (d.off0 < 0 || (d.in0 && d.off0 + d.c0 > d.ld0) || (d.c1 > 0 && (d.off1 < 0 || (d.in1 && d.off1 + d.c1 > d.ld1))))
            return PF_ERR_BAD_ARG;
        if (d.epilogue < PF_EPI_LINEAR || d.epilogue > PF_EPI_ADD) return PF_ERR_BAD_ARG;
        if ((d.epilogue == PF_EPI_RELU_RES || d.epilogue == PF_EPI_MASK || d.epilogue == PF_EPI_ADD) && (!d.h || d.ld_h < d.cout)) return PF_ERR_BAD_ARG;
        if (d.save_gates && d.aux_out && ((d.epilogue == PF_EPI_GRU_ZR && d.ld_aux < 256) || (d.epilogue == PF_EPI_GRU_Q && d.ld_aux < 128)))
            return PF_ERR_BAD_ARG;
        if (d.stride != f.stride || (d.stride != 1 && d.stride != 2)) return PF_ERR_BAD_SHAPE;
        if ((d.in_scale == nullptr) != (d.in_shift == nullptr)) return PF_ERR_BAD_ARG;
        if (d.epilogue == PF_EPI_TANH_RELU && (d.cout != 256 || (!d.aux_out && !d.aux_split) || (d.aux_out && d.ld_aux < 128))) return PF_ERR_BAD_ARG;
        if (d.precision != f.precision || (d.precision != PF_PREC_F32 && d.precision != PF_PREC_BF16X3))
            return PF_ERR_BAD_ARG;
        const int out_w = (d.epilogue == PF_EPI_GRU_ZR || d.epilogue == PF_EPI_TANH_RELU) ? 128 : d.cout;
        if (d.off_out < 0 || (d.out && d.off_out + out_w > d.ld_out) || (d.out_split && d.off_out + out_w > d.lds_out * 32))
            return PF_ERR_BAD_ARG;
        if (d.epilogue == PF_EPI_GRU_ZR && (d.cout != 256 || !d.h || (!d.aux_out && !d.aux_split) || (d.aux_out && d.ld_aux < 128) || d.ld_h < 128))
            return PF_ERR_BAD_ARG;
        if (d.epilogue == PF_EPI_GRU_Q && (d.cout != 128 || !d.h || !d.z || d.ld_z < 128 || d.ld_h < 128))
            return PF_ERR_BAD_ARG;
        grp.d[i] = d;
        if (d.cout > max_cout) max_cout = d.cout;
    }
    for (int i = ngroups; i < MAX_GROUPS; ++i) grp.d[i] = descs[0];
    g.H = H8; g.W = W8; g.N = H8 * W8; g.M = B * H8 * W8;
    g.kh = f.kh; g.kw = f.kw; g.taps = f.kh * f.kw;
    g.cin_pad = (f.c0 + f.c1 + KC - 1) / KC * KC;
    g.nchunks = g.cin_pad / KC;
    g.stride = f.stride; g.Hin = H8 * f.stride; g.Win = W8 * f.stride; g.Nin = g.Hin * g.Win;
    if (f.co_groups < 0 || f.co_groups > MAX_GROUPS) return PF_ERR_BAD_ARG;
    return PF_OK;
}

// Tile choice: the packed weights are zero-padded to a multiple of 128 output channels, so any
// BN in {32,64,128} is legal.  Small problems (one 512x1024 pair = 8192 pixels per branch) need
// the smaller tile to put >= 1 workgroup on each of the 256 CUs.
// 0: 128x32 (WM4 WN1 NT1)   1: 64x64 (WM2 WN2 NT1)   2: 64x128 (WM2 WN2 NT2)   7: 128x96 (WM4 WN1 NT3)
// 3: halo kernel 128x64     4: halo kernel 128x128   (bf16x3; any map size: edge tiles may be partial)
// 5: halo kernel 256x64 (8-row tile, Cout <= 64, 3x3 / 4x4, enough pixels to fill the chip)   8: halo kernel 256x96 (8-row tile, 3x3)
// `ngroups` here and in conv_dma_choice is the number of groups the chip sees at once: the launch's own plus pf_conv_desc.co_groups.
static int conv_tile(const ConvGeom& g, int ngroups, int max_cout, int precision) {
    const bool halo_shape = (g.kh == 3 && g.kw == 3) || (g.kh == 1 && g.kw == 5) || (g.kh == 5 && g.kw == 1) ||
                            (g.kh == 4 && g.kw == 4) || (g.kh == 1 && g.kw == 1);
    if (precision == PF_PREC_BF16X3 && halo_shape && g.stride == 1) {
        const long B = g.M / g.N;
        const long tiles4 = B * ((g.H + 3) / 4) * ((g.W + 31) / 32), tiles8 = B * ((g.H + 7) / 8) * ((g.W + 31) / 32);
        const long wgs128 = tiles4 * ngroups * ((max_cout + 127) / 128);
        if (max_cout <= 64 && g.kh == g.kw && g.kh > 1 && tiles8 * ngroups >= 512) return 5;
        // 8: 256 px x 96 channels (TH 8, NT 3; round 6) -- the 3x3 96 -> 96 convolutions of the encoders' layer 2: no padding channels
        if (max_cout > 64 && max_cout <= 96 && g.kh == 3 && g.kw == 3 && tiles8 * ngroups >= 256) return 8;
        return (max_cout > 64 && wgs128 >= 256) ? 4 : 3;
    }
    const long m_tiles64 = ((long)g.M + 63) / 64 * ngroups;
    if (max_cout <= 32) return 0;
    // 7: 128 px x 96 channels (WM4 WN1 NT3; round 6) -- the encoders' layer 2 has 96 output channels, which the 128-channel tile
    // covers with a quarter of its weight loads and MFMAs on padding
    if (max_cout > 64 && max_cout <= 96 && ((long)g.M + 127) / 128 * ngroups >= 256) return 7;
    if (max_cout <= 64 || m_tiles64 * ((max_cout + 127) / 128) < 512) return 1;
    return 2;
}

// Pre-split operands (pf_conv_desc.in0_split): which tile the all-DMA kernel takes -- 0: not applicable (fp32 operands,
// a shape / option it does not implement), else the pf_conv2d_roles code (1: 128-px tile, 2: 256 px x 64 channels).
// (The engine's PRIORFLOW_PRESPLIT=0 is the A/B against the register-staged kernels: it hands over fp32 operands.)
static int conv_dma_choice(const ConvGroups& grp, int ngroups, const ConvGeom& g, int max_cout, int tile_id) {
    if (!grp.d[0].in0_split || tile_id < 3 || tile_id == 7) return 0;
    const bool shape = (g.kh == 3 && g.kw == 3) || (g.kh == 1 && g.kw == 5) || (g.kh == 5 && g.kw == 1);
    if (!shape || g.stride != 1) return 0;
    for (int i = 0; i < ngroups; ++i)
        if (grp.d[i].stats_out != nullptr || grp.d[i].in_scale != nullptr) return 0;
    if (tile_id == 5) return 2;            // Cout <= 64 on a big map (3x3 by conv_tile's rule): the 256 px x 64 channel tile
    const long wgs256 = (long)(g.M / g.N) * ((g.H + 7) / 8) * ((g.W + 31) / 32) * (ngroups + grp.d[0].co_groups) * ((max_cout + 63) / 64);
    // round 4: the 256 px x 64 channel tile for the 1x5 / 5x1 convolutions with Cout > 128 (the GRU's fused z|r) too: half the
    // weight bytes staged per output, twice the halo; +0.2 % at B = 1, +0.6 % at batch 32 (profiles/r4_ab_gru_tile.txt)
    if (g.kh != 3 && max_cout > 128 && wgs256 >= 256) return 2;
    return (g.kh == 3 && max_cout > 64 && wgs256 >= 256) ? 2 : 1;
}


extern "C" int pf_conv2d_tile(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8) {
    ConvGroups grp; ConvGeom g; int max_cout;
    const int rc = conv_prepare(descs, ngroups, B, H8, W8, grp, g, max_cout);
    if (rc != PF_OK) return rc;
    const int tile = conv_tile(g, ngroups + descs[0].co_groups, max_cout, descs[0].precision);
    // 6: the weights-stationary kernel of the encoders' 3x3 64 -> 64 convolutions (pf_enc_conv.hip): its statistics partials are
    // per (segment, row phase, strip) -- pf_conv2d_stats_blocks
    return ((tile == 5 || tile == 3) && pf_enc_conv64_applies(grp, ngroups, g, max_cout)) ? 6 : tile;
}

extern "C" int pf_conv2d_stats_blocks(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8) {
    ConvGroups grp; ConvGeom g; int max_cout;
    const int rc = conv_prepare(descs, ngroups, B, H8, W8, grp, g, max_cout);
    if (rc != PF_OK) return rc;
    const int tile = conv_tile(g, ngroups + descs[0].co_groups, max_cout, descs[0].precision);
    const bool split = descs[0].precision == PF_PREC_BF16X3;
    if ((tile == 5 || tile == 3) && pf_enc_conv64_applies(grp, ngroups, g, max_cout)) return pf_enc_conv64_stats_blocks(g);
    if (tile >= 3 && tile != 7) { const int th = (tile == 5 || tile == 8) ? 8 : 4; return ((g.H + th - 1) / th) * ((g.W + 31) / 32); }
    const int bm = (tile == 0 || tile == 7) ? 128 : 64;  // generic kernel: tiles of bm consecutive pixels, which must not straddle images
    return (split && g.N % bm == 0) ? g.N / bm : 0;
}

extern "C" int pf_conv2d_roles(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8) {
    ConvGroups grp; ConvGeom g; int max_cout;
    const int rc = conv_prepare(descs, ngroups, B, H8, W8, grp, g, max_cout);
    if (rc != PF_OK) return rc;
    const int tile = conv_tile(g, ngroups + descs[0].co_groups, max_cout, descs[0].precision);
    if (const int dma = conv_dma_choice(grp, ngroups, g, max_cout, tile)) return 16 + dma;
    return (tile == 3 || tile == 4) ? pf_conv_ws_choice(grp, ngroups, g, max_cout) : 0;
}

extern "C" int pf_conv2d(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8, void* stream) {
    ConvGroups grp; ConvGeom g; int max_cout;
    const int rc = conv_prepare(descs, ngroups, B, H8, W8, grp, g, max_cout);
    if (rc != PF_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool split = descs[0].precision == PF_PREC_BF16X3;
    const int tile_id = conv_tile(g, ngroups + descs[0].co_groups, max_cout, descs[0].precision);
    const bool generic = tile_id < 3 || tile_id == 7;
    for (int i = 0; i < ngroups; ++i) {     // the input affine is implemented by the halo kernel only; the fused statistics by the
        if (descs[i].in_scale && generic) return PF_ERR_BAD_SHAPE;          // halo kernel and (round 4) by the generic one when its
        if (descs[i].stats_out && descs[i].epilogue != PF_EPI_LINEAR) return PF_ERR_BAD_SHAPE;     // M tiles do not straddle images
        if (descs[i].stats_out && generic && (!split || (g.N % ((tile_id == 0 || tile_id == 7) ? 128 : 64)) != 0)) return PF_ERR_BAD_SHAPE;
    }
    if (const int roles = conv_dma_choice(grp, ngroups, g, max_cout, tile_id))      // pre-split operands: the all-DMA kernel
        return pf_conv_dma_launch(grp, ngroups, g, max_cout, tile_id == 4 ? 2 : 1, roles, s);
    for (int i = 0; i < ngroups; ++i) {
        if (!descs[i].in0 || (descs[i].c1 > 0 && !descs[i].in1)) return PF_ERR_BAD_ARG;   // fp32 operands needed from here on
        if (descs[i].pre) return PF_ERR_BAD_SHAPE;                                          // accumulator start values: all-DMA kernel only
    }
    // tile 5 (Cout <= 64 on a big map) with 64 input channels: the weights-stationary kernel, bit-identical to the halo kernel
    if ((tile_id == 5 || tile_id == 3) && pf_enc_conv64_applies(grp, ngroups, g, max_cout)) return pf_enc_conv64_launch(grp, g, s);
    switch (tile_id) {
        case 0: case 1: case 2: case 7: return pf_conv_part0_launch(tile_id, grp, ngroups, g, max_cout, split, s);
        case 3: return pf_conv_part1_launch(grp, ngroups, g, max_cout, s);
        case 4: return pf_conv_part2_launch(grp, ngroups, g, max_cout, s);
        case 8: return pf_conv_part4_launch(grp, ngroups, g, max_cout, s);
        default: return pf_conv_part3_launch(grp, ngroups, g, max_cout, s);
    }
}
