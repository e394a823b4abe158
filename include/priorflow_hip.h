/* priorflow_hip.h -- C-ABI of the MI355X-native PriOr-RAFT inner loop (libpriorflow_hip.so).
 *
 * The reference (longliangLiu/PriOr-Flow, directory PriOr-RAFT/) has no FFI / plugin
 * boundary of its own: its hot path is a chain of Python functions calling torch ops.
 * Each entry point below replaces one of those functions (cited file:line, relative to
 * PriOr-RAFT/) with a hand-written gfx950 kernel.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t on a HIP failure, or a
 *     negative PF_ERR_* code on a bad argument; nothing throws, allocates or frees;
 *   - all buffers are DEVICE pointers owned by the caller, fp32, 16-byte aligned;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) without synchronising;
 *   - "planar" = [B,2,H8*W8] (NCHW with 2 channels); "channel-last" = [B*H8*W8][ld] with the
 *     logical channels at columns [c_off, c_off+C) of each row (lets producers write straight
 *     into slices of a wider concatenation buffer: torch.cat never materialises);
 *   - n = y*W8 + x is the raster index of a 1/8-resolution pixel, N = H8*W8;
 *   - sample grids are [2,H,W] (m', n') and shared by every batch element.
 */
#ifndef PRIORFLOW_HIP_H
#define PRIORFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_ERR_BAD_ARG (-1)
#define PF_ERR_BAD_SHAPE (-2)

/* Library / build identification ("gfx950 fp32-mfma ..."). */
const char* pf_version(void);

/* ---- ERP geometry ---------------------------------------------------------------------- */

/* generate_samplegrid (core/utils/projection_prim_ortho.py:432-443): ERP pixel -> sphere ->
 * R -> ERP pixel.  R is 9 HOST floats, row-major.  grid: [2,H,W]. */
int pf_sample_grid(float* grid, int H, int W, const float* R_host, void* stream);

/* img_rotate (core/utils/projection_prim_ortho.py:507-514): wrap-x / zero-pad bilinear resample
 * of an NCHW image stack with a sample grid.  img,out: [B,C,H,W]. */
int pf_img_rotate(const float* img, const float* grid, float* out, int B, int C, int H, int W,
                  void* stream);

/* Input normalisation `2 * (image / 255.0) - 1.0` of both images (core/prior_raft.py:121-122; IEEE division, as the
 * reference's CPU path), written straight into the encoders' batches: image1 -> f1 and (optional) c1, image2 -> f2.
 * count = B*3*H*W elements per image. */
int pf_normalise_images(const float* image1, const float* image2, float* f1, float* f2, float* c1,
                        long count, void* stream);

/* The input stage of a forward in one launch (core/prior_raft.py:121-127): both images normalised as pf_normalise_images does and
 * resampled into view B as pf_img_rotate does with `grid` ([2,H,W], generate_samplegrid of R_A2B), written into the encoders'
 * batches img_f = [im1 | im2 | im1_B | im2_B] ([4B,3,H,W]) and, optionally, img_c = [im1 | im1_B] ([2B,3,H,W]; NULL to skip).
 * Bit-identical to pf_normalise_images + pf_img_rotate on the normalised pair.  image1, image2: [B,3,H,W], values 0..255. */
int pf_prepare_images(const float* image1, const float* image2, const float* grid, float* img_f, float* img_c,
                      int B, int H, int W, void* stream);

/* flow = coords1 - coords_grid (core/prior_raft.py:172,177).  coords1: planar.  flow_out
 * (planar) and the two channel-last destinations are optional (NULL to skip). */
int pf_flow_prep(const float* coords1, float* flow_out,
                 float* d0, int d0_ld, int d0_off, float* d1, int d1_ld, int d1_off,
                 int B, int H8, int W8, void* stream);

/* flo_rotate (core/utils/projection_prim_ortho.py:531-546 + core/utils/my_cycle_sample.py:6-97):
 * express a flow field of one view in the other view.  flow/out: planar. */
int pf_flo_rotate(const float* flow, const float* g_w2c, const float* g_c2w, float* out,
                  float* d0, int d0_ld, int d0_off, float* d1, int d1_ld, int d1_off,
                  int B, int H8, int W8, void* stream);

/* ---- correlation volume, pyramid, lookups ---------------------------------------------- */

/* PriOr_RAFT.corr + DCCL.build_pyramid (core/prior_raft.py:69-75, core/corr.py:99-111), fused:
 * level0[b][n1][n2] = <f1[b][n1][:], f2[b][n2][:]> / sqrt(C); level i+1 = 2x2 mean of level i
 * over (y2,x2).  f1,f2: channel-last [B*N][C] (C % 32 == 0).  lvl[i]: [B*N][(H8>>i)*(W8>>i)].
 * Any H8, W8 >= 16 (odd levels pool with floor semantics, like avg_pool2d).  Maps with W8 % 32 == 0 and H8 % 8 == 0
 * take the fused kernel that writes all four levels in one pass (at B >= 4 and W8 % 64 == 0 its LDS-DMA ring
 * form); other maps take a generic corr kernel + three pooling passes.  PF_ERR_BAD_SHAPE: C % 32 != 0 or a map
 * whose row offsets do not fit 32 bits. */
int pf_corr_pyramid(const float* f1, const float* f2, float* lvl0, float* lvl1, float* lvl2,
                    float* lvl3, int B, int H8, int W8, int C, void* stream);

/* Same as pf_corr_pyramid in the 3-pass bf16 split arithmetic (hi*hi + hi*lo + lo*hi, fp32 accumulate):
 * f1_split / f2_split are the feature rows pre-split by pf_split_bf16. */
int pf_corr_pyramid_bf16x3(const void* f1_split, const void* f2_split, float* lvl0, float* lvl1,
                           float* lvl2, float* lvl3, int B, int H8, int W8, int C, void* stream);

/* fp32 rows [rows][C] -> bf16 hi|lo split rows [rows][C/32]{hi[32], lo[32]} (C % 32 == 0): the operand
 * format of the PF_PREC_BF16X3 GEMMs.  hi = bf16(x) round-to-nearest-even, lo = bf16(x - hi). */
int pf_split_bf16(const float* in, void* out, long rows, int C, void* stream);

/* Training: the packed weight / bias gradients of several convolutions (pf_conv2d_wgrad's dw / db) added into the parameters'
 * own gradient tensors, PF_UNPACK_MAX_JOBS (16) convolutions per launch -- what optimizer-side autograd does per parameter with
 * a permute copy, a clone and two accumulation adds (train_flow.py:135 loss.backward() on the reference):
 *   gw[o][c][tap] += scale * dw[o_off + o][tap][c]   (gw: [cout][cin][kh][kw], taps = kh*kw)      gb[o] += scale * db[o_off + o]
 * o_off: first packed row of this module inside a fused convolution (convz|convr); gb / db may be NULL together. */
typedef struct pf_unpack_job {
    const float* dw; const float* db; float* gw; float* gb;
    int cout, cin, taps, cin_pad, o_off; float scale;
} pf_unpack_job;
int pf_unpack_wgrads(const pf_unpack_job* jobs, int n, void* stream);

/* nn.Conv2d weights -> the PF_PREC_BF16X3 operand format of pf_conv2d, on the device, in one launch (what a training step
 * does for every convolution after each optimizer step, train_flow.py:138-140; replaces ~9 PyTorch-ROCm kernels per pack).
 * w0 [cout0][cin][kh][kw] and optionally w1 [cout1][cin][kh][kw] concatenated on the output channels (the fused z|r
 * convolution of core/update.py:48-49); b0 / b1 their biases (NULL = zero).
 *   mode 0: forward convolution: dst_w [cout_pad][kh*kw][cin_pad/32] x {bf16 hi[32], bf16 lo[32]}, dst_b [cout_pad].
 *   mode 1: its data-gradient convolution (pf_conv2d_wgrad's comment): W'[c][o][ky][kx] = W[o][(c + cin_rot) % cin][kh-1-ky][kw-1-kx],
 *           i.e. packed output channels = forward input channels rotated by cin_rot, packed input channels = forward output
 *           channels; dst_b (optional) is zeroed.
 * Padding rows / columns are written as zeros: dst needs no initialisation. */
int pf_pack_conv_weights(const float* w0, int cout0, const float* w1, int cout1, const float* b0, const float* b1,
                         int cin, int kh, int kw, int mode, int cin_rot, void* dst_w, float* dst_b,
                         int cout_pad, int cin_pad, void* stream);
/* Several of them per launch (16 jobs each; the arguments of pf_pack_conv_weights as a struct). */
typedef struct pf_pack_job {
    const float* w0; const float* w1; const float* b0; const float* b1; void* dst_w; float* dst_b;
    int cout0, cout1, cin, kh, kw, mode, cin_rot, cout_pad, cin_pad;
} pf_pack_job;
int pf_pack_conv_weights_batch(const pf_pack_job* jobs, int n, void* stream);

/* DCCL.__call__ steps 1-2 (core/corr.py:119-137): own-view 9x9x4 lookup and the raw
 * cross-view lookup through g_w2c.  coords: planar.  own_out/raw_out: channel-last, 324
 * channels = level*81 + a*9 + b (x += a-4, y += b-4), row stride ld >= 324. */
int pf_dccl_lookup(const float* coords,
                   const float* own0, const float* own1, const float* own2, const float* own3,
                   const float* oth0, const float* oth1, const float* oth2, const float* oth3,
                   const float* g_w2c, float* own_out, float* raw_out,
                   int B, int H8, int W8, int ld, void* stream);

/* pf_dccl_lookup with an additional copy of the grid interleaved per pixel, g_w2c_il [H8*W8][2] = (x, y): the two
 * components of a bilinear row pair become one 16-byte load (the kernel is bound by its gather-instruction count).
 * g_w2c_il may be NULL (= pf_dccl_lookup).  Results are bit-identical either way. */
int pf_dccl_lookup_il(const float* coords,
                      const float* own0, const float* own1, const float* own2, const float* own3,
                      const float* oth0, const float* oth1, const float* oth2, const float* oth3,
                      const float* g_w2c, const float* g_w2c_il, float* own_out, float* raw_out,
                      int B, int H8, int W8, int ld, void* stream);


/* DCCL.__call__ step 3 + the caller's add (core/corr.py:138, core/prior_raft.py:187-188):
 * out = own + img_rotate(raw, g_back), channel-last. */
int pf_dccl_combine(const float* own, const float* raw, const float* g_back, float* out,
                    int B, int H8, int W8, int ld, int ld_out, void* stream);

/* cycle_bilinear_sampler + groupwise_corr (core/prior_raft.py:173-174,180-182,77-83):
 * dst[.., c_off+g] = mean_{c in group g} f1[c] * warp(f2, coords)[c], 4 groups.
 * coords planar; if add_grid != 0 it is a flow and coords0 is added first. */
int pf_warp_gcorr(const float* f1, const float* f2, const float* coords, int add_grid,
                  float* dst, int dst_ld, int dst_off, int B, int H8, int W8, int C, void* stream);

/* Motion inputs of one refinement iteration in one launch (core/prior_raft.py:171-182): flow_A = coords1_A - coords0,
 * flow_B = coords1_B - coords0, flow_B_A = flo_rotate(flow_B, W2C = grid(R_A2B), C2W = grid(R_B2A)), and the two
 * groupwise correlations flaw_A = gwc(f1A, warp(f2A, coords1_A)), flaw_B_A = gwc(f1A, warp(f2A, coords0 + flow_B_A)).
 * Bit-identical to pf_flow_prep x2 + pf_flo_rotate + pf_warp_gcorr x2.  Outputs: flow4_a [B*N][4] = flow_A | flow_B_A,
 * flow2_b [B*N][2]; optional GRU-input tails xa (4 columns at xa_off) / xb (2 columns at xb_off); conf [B*N][conf_ld]
 * columns 0..3 flaw_A, 4..7 flaw_B_A.  C must be 256.  xa_split / xb_split (with xa_lds / xb_lds chunks per row): optional
 * split twins (see pf_conv_desc) of the GRU-input buffers, tails written at the same channel offsets xa_off / xb_off. */
int pf_motion_prep(const float* c1a, const float* c1b, const float* g_w2c, const float* g_c2w,
                   const float* f1a, const float* f2a, float* flow4_a, float* flow2_b,
                   float* xa, int xa_ld, int xa_off, float* xb, int xb_ld, int xb_off,
                   void* xa_split, int xa_lds, void* xb_split, int xb_lds,
                   float* conf, int conf_ld, int B, int H8, int W8, int C, void* stream);

/* Confidence stem of the ODDC motion encoder in one launch (core/update.py:177-178,193-194):
 * out[.., off_out .. off_out+16) = relu(conv3x3_{32->16}(relu(conv3x3_{8->32}(in[.., off_in .. off_in+8))))), zero padding,
 * exact fp32; w1 [9*8][32], w2 [9*32][16] in the [KH*KW][Cin][Cout] packing of pf_conv2d_direct.  The 32-channel
 * intermediate map never leaves LDS.  out_split / lds_out: optional split twin of `out` (see pf_conv_desc) written at the
 * same channel offset (% 4 == 0); `out` may then be NULL. */
int pf_conf_stem(const float* in, int ld_in, int off_in, const float* w1, const float* b1,
                 const float* w2, const float* b2, float* out, int ld_out, int off_out,
                 void* out_split, int lds_out, int B, int H8, int W8, void* stream);

/* ---- update blocks ------------------------------------------------------------------------ */

/* Epilogues of pf_conv2d. */
#define PF_EPI_LINEAR 0   /* out = (acc + bias) * scale                                   */
#define PF_EPI_RELU 1     /* out = relu(acc + bias)                                       */
#define PF_EPI_GRU_ZR 2   /* cout [0,128): out = sigmoid(.) -> z; [128,256): aux_out = sigmoid(.)*h */
#define PF_EPI_GRU_Q 3    /* q = tanh(.); out = (1-z)*h + z*q                              */
#define PF_EPI_TANH_RELU 4 /* cout [0,128): out = tanh(.); [128,256): aux_out = relu(.)
                            * (net / inp split of the context features, core/prior_raft.py:136-142) */
#define PF_EPI_RELU_RES 5  /* out = relu(res + relu(acc + bias)), res = h[.., j] (ld_h): the tail of a ResidualBlock whose
                            * norm2 is an eval-mode BatchNorm folded into the weights (core/extractor.py:41-47) */
#define PF_EPI_MASK 6      /* out = h[.., j] > 0 ? (acc + bias) * scale : 0 -- the data gradient of a convolution whose INPUT was
                            * a ReLU output y (h = y, ld_h): dgrad and the ReLU's backward in one launch (training, train_flow.py:135) */
#define PF_EPI_ADD 7       /* out = (acc + bias) * scale + h[.., j] (ld_h; out may alias h): a data gradient accumulated into the
                            * gradient another consumer of the same tensor has already left there */

/* Arithmetic of pf_conv2d (pf_conv_desc.precision); the weight buffer format follows it. */
#define PF_PREC_F32 0     /* exact fp32 MFMA; weights fp32 [Cout_pad][KH*KW][Cin_pad]                 */
#define PF_PREC_BF16X3 1  /* 3-pass bf16 split (hi*hi + hi*lo + lo*hi), fp32 accumulate; weights     *
                           * pre-split: [Cout_pad][KH*KW][Cin_pad/32] x {bf16 hi[32], bf16 lo[32]}  */

/* One stride-1 "same" convolution on channel-last activations as an implicit GEMM on the
 * matrix cores (nn.Conv2d forwards of core/update.py:6-14, 35-60, 81-99, 117-136, 139-201).
 * The input is the virtual concatenation of two channel-last segments (in1 may be NULL);
 * c0 must be a multiple of 32 when in1 is used.  Weights are pre-packed
 * [Cout_pad][KH*KW][Cin_pad] (Cin_pad = c0 + c1 rounded up to 32, Cout_pad to 128, zero filled).
 */
typedef struct pf_conv_desc {
    const float* in0; int ld0; int off0; int c0;
    const float* in1; int ld1; int off1; int c1;
    const float* weight; const float* bias;
    float* out; int ld_out; int off_out; int cout;
    int kh, kw;
    int epilogue; float scale;
    const float* h; int ld_h;        /* GRU_ZR / GRU_Q: hidden state [B*N][ld_h]            */
    const float* z; int ld_z;        /* GRU_Q: update gate                                  */
    float* aux_out; int ld_aux;      /* GRU_ZR: r*h                                         */
    int precision;                   /* PF_PREC_*; identical in every group of a launch     */
    int stride;                      /* 1 or 2; input map = stride x the output map         */
    /* optional per-(image, input-channel) affine + ReLU applied to the INPUT as it is loaded
     * (the previous layer's Instance/BatchNorm folded into this conv): x' = relu?(x*s + t).
     * [B][c0+c1] floats each; NULL = none.  Halo-kernel convolutions only (3x3/1x5/5x1, bf16x3). */
    const float* in_scale; const float* in_shift; int in_relu;
    /* optional InstanceNorm statistics of THIS conv's output, fused into its epilogue (PF_EPI_LINEAR only):
     * stats_out[((image*nblk + tile)*cout + c)*2 + {0,1}] = fp64 sum / sum of squares of output channel c over one workgroup
     * tile.  nblk = pf_conv2d_stats_blocks(...).  Halo-kernel tiles 3/4/5/8: nblk = tiles per image = ceil(H8/TH)*ceil(W8/32), TH = 8 for
     * tiles 5 and 8 else 4; tile 6: one partial per (segment of rows, row phase of the 4-row step, strip).  Generic
     * kernel (pf_conv2d_tile 0/1/2/7, bf16x3; the stride-2 layers): tiles of BM = 128/64/64/128 consecutive pixels, nblk = H8*W8/BM,
     * which must divide (PF_ERR_BAD_SHAPE otherwise).  Finish with pf_channel_stats_final.  NULL = none. */
    double* stats_out;
    /* Pre-split activations (PF_PREC_BF16X3 only).  A "split twin" of a channel-last map holds, per pixel row and per
     * 32-channel chunk, the 128 bytes {bf16 hi[32], bf16 lo[32]} with hi = bf16(x) (round to nearest even) and
     * lo = bf16(x - hi): [rows][lds chunks][128 B] -- exactly what the kernels otherwise compute from fp32 while staging,
     * so results are bit-identical.  Channels past the logical width of a row are zero.
     *   in0_split / in1_split (with lds0 / lds1 chunks per row; off0 / off1 / c0 must be multiples of 32; a last segment whose
     *     width is not a multiple of 32 must end at the end of its twin's row -- the kernel copies whole chunks and relies on
     *     the zero columns past the logical width, PF_ERR_BAD_SHAPE otherwise): when EVERY group of
     *     a stride-1 3x3 / 1x5 / 5x1 launch without in_scale / stats_out provides them, the operands go global -> LDS by DMA
     *     (pf_conv_dma_kernel: no VALU split, no ds_write) and in0 / in1 may be NULL; other launches ignore them and need in0.
     *   out_split (lds_out chunks per row; off_out % 32 == 0): the epilogue also writes the split twin of `out` at the same
     *     channel offset; `out` may then be NULL (twin only).  aux_split / lds_aux: the same for `aux_out` (GRU_ZR: r*h,
     *     TANH_RELU: inp).  Any kernel form honours the output twins. */
    const void* in0_split; int lds0;
    const void* in1_split; int lds1;
    void* out_split; int lds_out;
    void* aux_split; int lds_aux;
    /* with in0_split: a block of at least 128 * max(lds0, lds1) zero bytes (16-byte aligned) -- what the all-DMA kernel
     * reads for the zero padding around the map (an LDS-DMA copies memory; it cannot write a constant) */
    const void* zeros; int zeros_bytes;
    /* Optional start value of the accumulation (all-DMA kernel only; any other kernel form answers PF_ERR_BAD_SHAPE):
     * channel-last fp32 [B*N][ld_pre], output channel j of this conv at column off_pre + j.  The accumulators start from
     * pre instead of zero, so out = epilogue(pre + conv(in) + bias).  Used for the iteration-invariant part of the GRU
     * convolutions: the context features `inp` are the same in all `iters` iterations (core/prior_raft.py:148,196), so
     * conv_{[h|inp|motion]} = conv_inp(inp) [computed once] + conv_{[h|motion]} [per iteration, 2/3 of the MFMA work]. */
    const float* pre; int ld_pre; int off_pre;
    /* Training forward (the gate values are operands of pf_gru_zr_bwd / pf_gru_q_bwd): with save_gates != 0 and aux_out set,
     * PF_EPI_GRU_ZR also writes r = sigmoid(.) to aux_out[.., 128 + j'] (j' = j - 128; r*h stays at aux_out[.., j'], so
     * ld_aux >= 256) and PF_EPI_GRU_Q also writes q = tanh(.) to aux_out[.., j] (ld_aux >= 128). */
    int save_gates;
    /* Tile-choice hint (descs[0] of a launch): the number of further launches of the same geometry the caller runs BESIDE this one
     * on other streams (round 6: branch A's and branch B's update blocks as two chains, core/prior_raft.py:196-211).  The host
     * counts work items as if those launches' groups were groups of this one, so two half-chip launches keep the tile a
     * two-group launch takes instead of falling back to the smaller tile that would fill the chip alone.  0 = none. */
    int co_groups;
} pf_conv_desc;

/* The tail of DCCL.__call__ fused with the first motion-encoder convolution (core/corr.py:138,
 * core/prior_raft.py:187-188, core/update.py:185 / :92), PF_PREC_BF16X3 arithmetic:
 *   out[.., off_out .. off_out+256) = relu(conv1x1_{324->256}(own + img_rotate(raw, g_back)))
 * own, raw: channel-last [B*N][ld] outputs of pf_dccl_lookup; weight: the pf_conv2d packing of the 1x1 conv
 * (pre-split bf16 [256][1][11] x {hi[32], lo[32]}); cout must be 256.  Bit-identical to pf_dccl_combine followed by
 * pf_conv2d, but the combined 324-channel tensor is never written. */
typedef struct pf_combine_conv_desc {
    const float* own; const float* raw; int ld;
    const float* g_back;               /* [2][N] rotate-back grid */
    const void* weight; const float* bias;
    float* out; int ld_out; int off_out; int cout;
    void* out_split; int lds_out;      /* optional split twin of `out` (see pf_conv_desc), same channel offset (% 32 == 0);
                                        * `out` may then be NULL */
} pf_combine_conv_desc;
/* ngroups = 1 | 2: branch A and branch B of an iteration in one launch (grid.y = group). */
int pf_dccl_combine_conv1x1(const pf_combine_conv_desc* descs, int ngroups, int B, int H8, int W8, void* stream);

/* Launch `ngroups` (1..4) same-geometry convolutions in ONE kernel (grid.z = group):
 * branch A and branch B of an iteration run side by side.  H8, W8 = OUTPUT map size. */
int pf_conv2d(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8, void* stream);

/* Host-only introspection: which workgroup tile pf_conv2d would use for this launch -- generic kernel 0: 128x32,
 * 1: 64x64, 2: 64x128, 7: 128x96 (pixels x channels; 7 = round 6, for the 96 output channels of the encoders' layer 2); halo kernel 3: 128x64, 4: 128x128, 5: 256x64 (8-row tile), 8: 256x96 (8-row tile, 3x3 with 64 < Cout <= 96; round 6); 6: the
 * weights-stationary kernel of the encoders' 3x3 64 -> 64 convolutions (round 5; core/extractor.py:16-17 at 1/2 resolution:
 * strips of 32 columns walked in 4-row steps, outputs bit-identical to tile 5) -- or a negative PF_ERR_* code.  Lets a profiler attribute measured time to the right kernel instantiation; launches nothing. */
int pf_conv2d_tile(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8);

/* Host-only introspection: how many fp64 partial blocks PER IMAGE this launch writes to `stats_out` (the nblk of
 * pf_channel_stats_final and of the buffer's size, [B][nblk][cout][2] doubles); 0 = this launch cannot fuse the statistics
 * (generic kernel whose pixel tiles would straddle images, or not bf16x3); negative = PF_ERR_*.  Launches nothing. */
int pf_conv2d_stats_blocks(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8);

/* Host-only introspection, companion of pf_conv2d_tile for tiles 3 / 4: which wave organisation the launch takes --
 * 0: every wave stages and multiplies (pf_conv_halo_kernel), 1: four MFMA waves + four loader waves on the same tile
 * (pf_conv_ws_kernel<NT, KH, KW, 2>), 2: the same with a 256 px x 64 channel tile (pf_conv_ws_kernel<2, KH, KW, 1>);
 * 16 + 1 | 16 + 2: the launch has pre-split operands and takes the all-DMA kernel (pf_conv_dma_kernel) with that tile. */
int pf_conv2d_roles(const pf_conv_desc* descs, int ngroups, int B, int H8, int W8);

/* The encoders' first convolution (core/extractor.py:122, :144: conv1 7x7 stride 2 pad 3, 3 -> 64) straight from the NCHW
 * image, PF_PREC_BF16X3 arithmetic on the matrix cores with K = the 7x7x3 patch (176 padded) instead of the 512 of the
 * space-to-depth form: img [Bn][3][H][W] (H, W even) -> out rows [Bn * H/2 * W/2][64] fp32 (may be NULL with out_split) and / or
 * the split twin out_split [rows][2 chunks][128 B] (see pf_conv_desc); relu != 0: ReLU epilogue (the BatchNorm-folded cnet
 * stem).  weight: 64 rows of 704 bytes -- per 8-k piece {bf16 hi[8], bf16 lo[8]}, k = ky * 24 + kx * 3 + c, zero padded
 * (engine.pack_stem7x7 builds it); bias [64].  stats_out (optional): fp64 per-tile sum / sum of squares of the stored values,
 * [Bn][ceil(H/16) * ceil(W/64)][64][2] -- the partials pf_channel_stats_final reduces (8-row x 32-column output tiles). */
int pf_enc_stem(const float* img, const void* weight, const float* bias, float* out, void* out_split, int relu,
                double* stats_out, int Bn, int H, int W, void* stream);

/* Test support (tests/test_hip_kernels.py): fills the whole 160 KB LDS of every CU with `pattern` (e.g. 0x7fc00000, a NaN), so
 * that a kernel which reads LDS it never wrote -- padding floats multiplied by zero weights -- shows up as NaNs in its output.
 * No counterpart in the reference. */
int pf_debug_dirty_lds(unsigned pattern, void* stream);

/* Tiny-Cin direct convolution (7x7 2->128, 3x3 8->32, 3x3 32->16; core/update.py:171-178,87).
 * Weights packed [KH*KW][Cin][Cout]. */
int pf_conv2d_direct(const float* in, int ld_in, int off_in, int cin,
                     const float* weight, const float* bias,
                     float* out, int ld_out, int off_out, int cout,
                     int kh, int kw, int relu, int B, int H8, int W8, void* stream);

/* The same convolution for 1..4 independent problems of one shape in ONE launch (the motion encoders' 7x7
 * stems of flow_A, flow_B_A and flow_B, core/update.py:187-188,:94, read three different 2-channel inputs).
 * Outputs must not overlap. */
typedef struct pf_direct_desc {
    const float* in;  int ld_in, off_in;
    const float* weight; const float* bias;
    float* out;       int ld_out, off_out;
    void* out_split;  int lds_out;     /* optional split twin of `out` (see pf_conv_desc; 7x7 2 -> C stems only), same
                                        * channel offset (% 8 == 0); `out` may then be NULL */
} pf_direct_desc;
int pf_conv2d_direct_group(const pf_direct_desc* descs, int n, int cin, int cout, int kh, int kw, int relu,
                           int B, int H8, int W8, void* stream);

/* Small-Cin convolution with stride 1|2 from channel-last or NCHW input (the encoders' 7x7/2 3->64
 * stem, core/extractor.py:112,144; same kernel family as pf_conv2d_direct).  Hout/Wout = OUTPUT map;
 * input map = stride x output.  nchw != 0: `in` is [B,cin,Hin,Win] planes (ld_in/off_in ignored). */
int pf_conv2d_small(const float* in, int nchw, int ld_in, int off_in, int cin,
                    const float* weight, const float* bias, float* out, int ld_out, int off_out, int cout,
                    int kh, int kw, int stride, int relu, int B, int Hout, int Wout, void* stream);

/* ---- encoder glue (core/extractor.py) ---------------------------------------------------------- */

/* Second stage of pf_channel_stats on partials some other kernel produced (pf_conv2d with
 * desc.stats_out): partials [B][nblk][C][2] doubles -> scale / shift [B][C]. */
int pf_channel_stats_final(const double* partials, int B, int Np, int C, int nblk, float eps,
                           float* scale, float* shift, void* stream);

/* Per-(image, channel) InstanceNorm statistics of a channel-last map y [B*Np][C] (C <= 256):
 * scale[b][c] = 1/sqrt(var+eps), shift[b][c] = -mean*scale (biased variance).  Deterministic
 * two-stage fp64 reduction; partials: workspace of B*nblk*C*2 doubles. */
int pf_channel_stats(const float* y, int B, int Np, int C, float eps, float* scale, float* shift,
                     double* partials, int nblk, void* stream);

/* out = relu( res' + relu(y*s + t) ), res' = res | res*rs + rt | relu(res*rs + rt) [res_relu != 0: the skip input is itself
 * a normalised + activated raw conv output that was never materialised -- the stem of the encoder] | absent (ResidualBlock
 * tail, core/extractor.py:41-47).  y,res,out channel-last [B*Np][C]; s,t,rs,rt [B][C]. */
int pf_norm_act(const float* y, const float* s, const float* t, const float* res,
                const float* rs, const float* rt, int res_relu, float* out, int B, int Np, int C, void* stream);

/* FlowHead.conv2 (3x3, C->2; core/update.py:10,13-14) fused with coords1 += delta_flow
 * (core/prior_raft.py:193,196).  x: channel-last hidden features [B*N][ld] (C channels at column 0);
 * weight packed [2][9][C] (exact fp32 dot products); coords1 planar, updated in place; delta
 * (optional, may be NULL): channel-last [B*N][ld_delta] copy of delta_flow. */
int pf_flow_head_out(const float* x, int ld, int C, const float* weight, const float* bias,
                     float* coords1, float* delta, int ld_delta, int B, int H8, int W8, void* stream);


/* coords1 += delta (core/prior_raft.py:193,196).  delta: channel-last, 2 channels at column 0. */
int pf_coords_add(float* coords1, const float* delta, int ld, int B, int H8, int W8, void* stream);
/* dst = src + delta: the same update into the next iteration's coordinates (the training loop keeps every iteration's). */
int pf_coords_add_to(const float* src, const float* delta, int ld, float* dst, int B, int H8, int W8, void* stream);

/* upsample_flow (core/prior_raft.py:58-67): convex 8x upsampling of flow = coords1 - coords0.
 * mask: channel-last [B*N][ld] (576 logits, already scaled by 0.25).  out: [B,2,8*H8,8*W8]. */
int pf_upsample_flow(const float* coords1, const float* mask, int ld, float* out,
                     int B, int H8, int W8, void* stream);

/* ---- layout plumbing ------------------------------------------------------------------------ */

/* NCHW channel slice -> channel-last slice, with activation (0 none, 1 relu, 2 tanh)
 * (core/prior_raft.py:136-142 splits/activations of the context features). */
int pf_to_channel_last(const float* in, int c_total, int c_begin, int c, float* out, int ld_out,
                       int off_out, int act, int B, int N, void* stream);

/* 2x2 space-to-depth: NCHW [B,C,H,W] -> channel-last [B*(H/2)*(W/2)][ld_out], column (py*2+px)*C + c.
 * Feeds the encoders' stem (core/extractor.py:122 `conv1 = Conv2d(3, 64, 7, stride=2, padding=3)`),
 * which pf_conv2d then runs as a 4x4 stride-1 convolution over the 12 stacked channels. */
int pf_space_to_depth2(const float* in, int C, float* out, int ld_out, int B, int H, int W, void* stream);

/* ---- evaluation counterpart ---------------------------------------------- */

/* Per-pixel EPE (evaluate.py:265 `torch.sum((flow - flow_gt)**2, dim=0).sqrt()`) and SEPE
 * (core/utils/spherical.py:20-53 `calculate_great_circle_distance`, R = 1; cosine = 0: method 'Haversine', the one
 * evaluate.py uses; cosine = 1: method 'Cosine', arccos of the spherical law of cosines as written at :40-46) of
 * pred vs gt, both NCHW [B,2,H,W].  epe / sd: [B,H,W]; either may be NULL. */
int pf_flow_metrics(const float* pred, const float* gt, float* epe, float* sd, int cosine, int B, int H, int W,
                    void* stream);

/* Region sums of evaluate.py:246-275 (All / Equator / Poles / Center ...): bit r of bits[n] puts
 * pixel n into region r (nregions <= 8).  weight: [N] cos-latitude weights (evaluate.py:208-213
 * `sd_uni`) or NULL.  partials: double [B][nblk][nregions][3] = per pixel-chunk sums of
 * epe, sd, sd*weight; the caller adds the nblk chunks (deterministic two-stage sum). */
int pf_region_sums(const float* epe, const float* sd, const float* weight, const unsigned char* bits,
                   int nregions, double* partials, int nblk, int B, int N, void* stream);

/* ---- training-step counterpart (the network's backward is NOT built) ------- */

/* One term of `uniform_loss.__call__` (train_flow.py:62-71): with m = (valid >= 0.5 && |gt| < max_flow) * weight[n],
 * partials[b][k][0] = sum m (|du|+|dv|) over pixel chunk k; [1..5] = sum epe, n_valid, n(epe<1), n(<3), n(<5)
 * over the valid pixels (the metrics of :73-79, meaningful for the last prediction).  grad (optional):
 * d(i_weight * loss_i)/d pred = i_weight * m * sign(pred - gt).  pred, gt, grad: [B,2,N]; valid [B,N]; weight [N]. */
int pf_seq_loss(const float* pred, const float* gt, const float* valid, const float* weight, float i_weight,
                float max_flow, float* grad, double* partials, int nblk, int B, int N, void* stream);

/* n <= 32 terms of the same `uniform_loss.__call__` (train_flow.py:62-71: the predictions of all iterations of one branch) in ONE
 * launch: term i = pf_seq_loss(preds[i], gt, valid, weight, i_weights[i], max_flow, grads[i], partials + i * B * nblk * 6, ...),
 * bit for bit.  preds / grads / i_weights are HOST arrays of n device pointers / floats (grads, or single entries of it, may be
 * NULL); partials is [n][B][nblk][6].  (Round 6: the training step's loss was 24 launches of ~13 us one behind the other.) */
int pf_seq_loss_batch(const float* const* preds, const float* gt, const float* valid, const float* weight,
                      const float* i_weights, float max_flow, float* const* grads, double* partials, int nblk, int n,
                      int B, int N, void* stream);

/* partials[k] = sum of squares of chunk k of x[0..n): the total norm of `clip_grad_norm_` (train_flow.py:137). */
int pf_sum_squares(const float* x, long n, double* partials, int nblk, void* stream);

/* One fused AdamW step over a flat parameter buffer (torch.optim.AdamW as built by train_flow.py:86-88):
 * g*grad_scale (the clip coefficient), decoupled weight decay, bias-corrected update; step counts from 1. */
int pf_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, float beta1, float beta2,
                  float eps, double weight_decay, int step, float grad_scale, void* stream);
/* The same update with the step-dependent scalars in device memory: hyper[4] = {1 - lr * weight_decay, lr / (1 - beta1^step),
 * sqrt(1 - beta2^step), grad_scale} (rounded from double like pf_adamw_step does).  Every launch argument is then constant
 * over the steps, so the call can sit in a captured HIP graph of the whole training step (train.GraphedTrainStep): the host
 * rewrites hyper[0..2] between replays, a kernel of the graph writes the clip coefficient hyper[3]. */
int pf_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps,
                      const float* hyper, void* stream);

/* Backward of pf_dccl_combine / pf_dccl_lookup (autograd through DCCL.__call__, core/corr.py:113-144; coords
 * are detached, core/prior_raft.py:171,176): d_corr -> d_raw (rotate-back transposed; d_own = d_corr), then
 * d_own / d_raw -> gradients of the own and the other pyramid (level i: [B*N][H_i*W_i]).  All outputs are
 * ACCUMULATED with fp32 atomics: zero them once per step. */
/* Backward of pf_warp_gcorr (autograd through cycle_bilinear_sampler + groupwise_corr, core/prior_raft.py:173-174,
 * :77-83; coords detached): d_flaw (4 channels at off_d of [B*N][ld_d]) -> d_f1, d_f2 [B*N][C], both ACCUMULATED. */
int pf_warp_gcorr_bwd(const float* f1, const float* f2, const float* coords, int add_grid, const float* d_flaw,
                      int ld_d, int off_d, float* d_f1, float* d_f2, int B, int H8, int W8, int C, void* stream);

/* Backward of pf_upsample_flow (autograd through upsample_flow, core/prior_raft.py:58-67): g [B,2,8*H8,8*W8] ->
 * d_mask [B*N][ld_d] (all 576 logits written; softmax backward) and d_flow [B,2,H8,W8] (= d coords1, ACCUMULATED). */
int pf_upsample_flow_bwd(const float* coords1, const float* mask, int ld, const float* g, float* d_mask, int ld_d,
                         float* d_flow, int B, int H8, int W8, void* stream);

/* ResidualBlock tail of the encoders on the training tape (core/extractor.py:47): out = relu(x + y); and the backward of any
 * ReLU from its forward output: dx = fwd_out > 0 ? g : 0.  n floats, any layout (elementwise). */
int pf_add_relu(const float* x, const float* y, float* out, long n, void* stream);
int pf_relu_mask(const float* g, const float* fwd_out, float* dx, long n, void* stream);

/* nn.BatchNorm2d with frozen statistics (freeze_bn, train_flow.py:107-108: the context encoder in every stage but `chairs`),
 * optionally with the ReLU behind it (core/extractor.py:41-42,144-146), on channel-last rows [rows][C], one launch:
 *   out = [relu]( x*s + t ),  s = gamma * rsqrt(var + eps),  t = beta - mean * s. */
int pf_bn_frozen_fwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                     float eps, int relu, float* out, long rows, int C, void* stream);
/* Its backward: g_m = relu ? (x*s + t > 0 ? dy : 0) : dy;  dx = s * g_m;  dgamma[c] (+)= sum g_m * (x - mean) * rsqrt(var + eps),
 * dbeta[c] (+)= sum g_m (accumulate != 0: added to what is there -- the parameters' .grad).  Deterministic fp64 sums through
 * `partials` (nblk*C*2 doubles, nblk <= rows).  Three launches. */
int pf_bn_frozen_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                     const float* var, float eps, int relu, double* partials, int nblk, float* dx,
                     float* dgamma, float* dbeta, int accumulate, long rows, int C, void* stream);

/* Backward of y = act(x * scale[b,c] + shift[b,c]) on channel-last rows [B*Np][C] (the encoders' norm + ReLU,
 * core/extractor.py:112-147; scale / shift as produced by pf_channel_stats or the folded BatchNorm affine).
 * relu != 0: dy is masked where x*scale+shift <= 0.  instance != 0: InstanceNorm backward,
 * dx = scale * (g - mean(g) - xh * mean(g*xh)) with deterministic fp64 sums through `partials` ([B][nblk][C][2]
 * doubles) and `coef` ([B][C][2] floats); instance == 0: fixed statistics (BatchNorm eval), dx = scale * g. */
int pf_norm_bwd(const float* dy, const float* x, const float* scale, const float* shift, int relu, int instance,
                double* partials, int nblk, float* coef, float* dx, int B, int Np, int C, void* stream);

/* SepConvGRU gate backward of one half-step (core/update.py:46-60), channel-last rows with leading dimensions.
 * Stage Q, before the data gradient of convq:  dq_pre = dh'*z*(1-q^2), dz = dh'*q - dh'*h, dh = dh'*(1-z).
 * Stage ZR, after it (d_rh = gradient of r*h, the first C input channels of convq):
 * dzr_pre[:, :C] = dz*(1-z)*z, dzr_pre[:, C:2C] = d_rh*h*(1-r)*r (the [z|r] gradient of the fused conv), dh += d_rh*r. */
int pf_gru_q_bwd(const float* dh_new, int ld_dhn, const float* z, int ld_z, const float* q, int ld_q,
                 const float* h, int ld_h, float* dq_pre, int ld_dq, float* dz, int ld_dz,
                 float* dh, int ld_dh, long rows, int C, void* stream);
int pf_gru_zr_bwd(const float* dz, int ld_dz, const float* d_rh, int ld_drh, const float* z, int ld_z,
                  const float* r, int ld_r, const float* h, int ld_h, float* dzr_pre, int ld_dzr,
                  float* dh, int ld_dh, long rows, int C, void* stream);

/* Gradient of a SepConvGRU's input x = [inp (C) | out (wout) | flows] once both half-steps' data gradients exist
 * (f1, f2: rows with leading dimensions, x's channel order): d_inp[., c] += f1 + f2 for c < C (inp = relu(cnet) feeds every
 * iteration, core/prior_raft.py:196); d_out[., j] = (f1 + f2)[., C + j] where x[., C + j] > 0 else 0 (out = relu(conv),
 * core/update.py:99,200). */
int pf_gru_dx_finish(const float* f1, int ld_f1, const float* f2, int ld_f2, const float* x, int ld_x,
                     float* d_inp, int ld_dinp, float* d_out, int ld_dout, long rows, int C, int wout, void* stream);

/* Backward of build_pyramid (core/corr.py:99-111): level gradients g0..g3 ([B*N][H_i*W_i]) -> the dense volume
 * gradient, written in place into g0 (avg_pool2d backward with floor semantics for odd sizes). */
int pf_pyramid_bwd(float* g0, const float* g1, const float* g2, const float* g3, int B, int H8, int W8, void* stream);
int pf_dccl_combine_bwd(const float* d_corr, int ld_in, const float* g_back, float* d_raw, int ld,
                        int B, int H8, int W8, void* stream);
/* Backward of pf_dccl_lookup (core/corr.py:113-144): the gradients of its two outputs (d_own, d_raw: [B*N][ld] rows, 324 channels)
 * scattered into the two pyramids' level gradients (accumulated).  clear_raw != 0: every d_raw value read is replaced by 0, so
 * the buffer is ready for the next pf_dccl_combine_bwd, which scatters into it (one fill launch per iteration less). */
int pf_dccl_lookup_bwd(const float* coords, const float* g_w2c, const float* d_own, float* d_raw, int ld,
                       float* own0, float* own1, float* own2, float* own3,
                       float* oth0, float* oth1, float* oth2, float* oth3, int B, int H8, int W8, int clear_raw, void* stream);

/* Weight and bias gradient of a stride-1 convolution (what autograd computes for every nn.Conv2d of
 * core/update.py / core/extractor.py in `loss.backward()`, train_flow.py:135):
 *   dw[o][tap][c] += sum_p dy[p][o] * x[p + off(tap)][c],  db[o] += sum_p dy[p][o]
 * x: channel-last, two-segment virtual concat like pf_conv_desc (in0 | in1); dy: channel-last gradient of the
 * conv's pre-activation output; dw: fp32 in the packed weight layout [Cout_pad128][KH*KW][Cin_pad32], db
 * [Cout_pad128] or NULL -- both ACCUMULATED into (zero them first).  3x3, 1x5, 5x1, 1x1; 3-pass bf16 split.
 * The data gradient needs no entry of its own: dx = pf_conv2d(dy, W') with
 * W'[c][o][ky][kx] = W[o][c][KH-1-ky][KW-1-kx]. */
int pf_conv2d_wgrad(const float* x0, int ld0, int off0, int c0, const float* x1, int ld1, int off1, int c1,
                    const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                    int kh, int kw, int B, int H8, int W8, void* stream);

/* Weight / bias gradient of a small-Cin convolution (the 7x7 stems: core/extractor.py:122 3->64 stride 2,
 * core/update.py:87,173,175 2->128 stride 1; their inputs carry no gradient): exact fp32,
 *   dw[o][c][ky][kx] += sum_p dy[p][o] * x[stride*p + (ky,kx) - pad][c],  db[o] += sum_p dy[p][o]
 * x: NCHW planes (nchw != 0) or a channel-last slice; dy: channel-last [B*Hout*Wout][ld_dy]; dw in the framework's own
 * parameter layout [Cout][Cin][KH][KW]; dw, db ACCUMULATED.  cin <= 4, kh*kw <= 52, stride 1 | 2. */
int pf_conv2d_wgrad_small(const float* x, int nchw, int ld_in, int off_in, int cin,
                          const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                          int kh, int kw, int stride, int B, int Hout, int Wout, void* stream);
/* The same through a caller-provided workspace (pf_conv2d_wgrad_small_ws_floats(...) floats, no initialisation needed; private to
 * the call until it has finished): every workgroup stores its partial sums there and a second launch adds them with 8 atomics per
 * weight.  The form above ends with one atomic per weight per workgroup on the same few hundred cache lines, which was most of its
 * time (414 -> see DESIGN.md section 7 us for the encoder stem at 384x512). */
long pf_conv2d_wgrad_small_ws_floats(int cin, int cout, int kh, int kw, int B, int Hout, int Wout);
int pf_conv2d_wgrad_small_ws(const float* x, int nchw, int ld_in, int off_in, int cin,
                             const float* dy, int ld_dy, int off_dy, int cout, float* dw, float* db,
                             int kh, int kw, int stride, int B, int Hout, int Wout,
                             float* workspace, long workspace_floats, void* stream);

/* channel-last slice -> NCHW. */
int pf_to_nchw(const float* in, int ld_in, int off_in, int c, float* out, int B, int N,
               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIORFLOW_HIP_H */
