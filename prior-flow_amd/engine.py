"""Launch plan of the PriOr-RAFT inner loop on the HIP library.

``Engine`` owns (per input shape) the device workspace -- every buffer of the loop is
allocated once through torch and stays resident in HBM -- and the packed weights, and
turns one forward into a fixed sequence of C-ABI launches on the current stream
(capturable into a HIP graph, see ``PriOr_RAFT``).  All arithmetic of the loop runs in
``libpriorflow_hip.so``; torch only provides memory and streams here.

Layouts: activations are channel-last rows ``[B*N, ld]``; concatenations of the reference
(``torch.cat`` in core/update.py:96-99,155,195-201) never materialise -- producers write
into column slices of the consumer's row buffer.

    x_a  = [ inp_A (128) | conv_A out (124) | flow_A (2) | flow_B_A (2) ]   GRU input of ODDC
    x_b  = [ inp_B (128) | conv out   (126) | flow_B (2) ]                  GRU input of update_block
    catA = [ cor (128) | floA (64) | floB (64) | conf (16) ]                input of conv_A
    catB = [ cor (192) | flo (64) ]                                         input of conv
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

import os

from ._lib import (EPI_GRU_Q, EPI_GRU_ZR, EPI_LINEAR, EPI_RELU, EPI_RELU_RES, PREC_BF16X3, PREC_F32,
                   ConvDesc, PfError, PfLib)

CORR_CH = 324


def rotation_x(theta: float) -> torch.Tensor:
    """R = Rz(0) Ry(0) Rx(theta) built in fp32 like generate_rotation_metrix
    (core/utils/projection_prim_ortho.py:23-48): cos(fp32(+-pi/2)) = -4.371139e-8, not 0."""
    c = torch.cos(torch.tensor(theta)).float().item()
    s = torch.sin(torch.tensor(theta)).float().item()
    return torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], dtype=torch.float32)


# ------------------------------------------------------------------------------------------
# weight packing (host-side plumbing; runs once per weight version)
# ------------------------------------------------------------------------------------------
def pack_mfma(w: torch.Tensor, b: torch.Tensor, cin_to: int = 0):
    """[Cout,Cin,KH,KW] -> [Cout_pad128][KH*KW][Cin_pad32] (zero filled), bias [Cout_pad128].
    cin_to > Cin pads the input channels with zero weights (lets two convs share a launch geometry)."""
    cout, cin, kh, kw = w.shape
    cp = (max(cin, cin_to) + 31) // 32 * 32
    op = (cout + 127) // 128 * 128
    wp = torch.zeros(op, kh * kw, cp, dtype=torch.float32, device=w.device)
    wp[:cout, :, :cin] = w.detach().float().permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    bp = torch.zeros(op, dtype=torch.float32, device=w.device)
    bp[:cout] = b.detach().float()
    return wp.contiguous(), bp.contiguous()


def split_bf16(wp: torch.Tensor) -> torch.Tensor:
    """fp32 [..., Cin_pad] -> bf16 [..., Cin_pad/32, 2, 32]: per 32-channel chunk the hi halves
    (bf16(w), round-to-nearest-even) followed by the lo halves (bf16(w - hi)): exactly the 128-byte
    LDS row of the PF_PREC_BF16X3 kernels, so staging the weight tile is a plain copy."""
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    shp = wp.shape[:-1] + (wp.shape[-1] // 32, 1, 32)
    return torch.cat([hi.reshape(shp), lo.reshape(shp)], dim=-2).contiguous()


def split_twin(rows: int, channels: int, device) -> torch.Tensor:
    """Zero-initialised split twin of a channel-last map [rows][channels]: bf16 [rows][ceil(channels/32)][2][32], per
    32-channel chunk the hi halves then the lo halves (include/priorflow_hip.h, pf_conv_desc).  Channels past `channels`
    stay zero: they are the zero padding of the K dimension."""
    return torch.zeros(rows, (channels + 31) // 32, 2, 32, dtype=torch.bfloat16, device=device)


_ZERO_BLOCKS: Dict[str, torch.Tensor] = {}


def _zero_block(device) -> torch.Tensor:
    """4 KB of zeros per device: pf_conv_desc.zeros (the zero padding around the map, read by the all-DMA conv kernel)."""
    key = str(device)
    if key not in _ZERO_BLOCKS:
        _ZERO_BLOCKS[key] = torch.zeros(1024, dtype=torch.float32, device=device)
    return _ZERO_BLOCKS[key]


def unsplit(twin: torch.Tensor, channels: int) -> torch.Tensor:
    """hi + lo of a split twin as fp32 rows [rows][channels] (tests / debugging; 16 mantissa bits of the original)."""
    v = twin[:, :, 0, :].float() + twin[:, :, 1, :].float()
    return v.reshape(twin.shape[0], -1)[:, :channels].contiguous()


def stem_s2d_weight(w: torch.Tensor) -> torch.Tensor:
    """7x7 stride-2 pad-3 weights [Cout,C,7,7] -> the equivalent 4x4 stride-1 weights [Cout,4C,4,4]
    over the 2x2 space-to-depth image (channel (py*2+px)*C + c, window rows Y-2..Y+1):
    input row 2y + ky - 3 = 2(y + KY - 2) + py  =>  ky = 2 KY + py - 1 (taps outside 0..6 are zero)."""
    cout, c, kh, kw = w.shape
    assert kh == 7 and kw == 7
    w2 = torch.zeros(cout, 4 * c, 4, 4, dtype=torch.float32, device=w.device)
    for py in range(2):
        for px in range(2):
            for KY in range(4):
                ky = 2 * KY + py - 1
                if not 0 <= ky < 7:
                    continue
                for KX in range(4):
                    kx = 2 * KX + px - 1
                    if 0 <= kx < 7:
                        q = py * 2 + px
                        w2[:, q * c:(q + 1) * c, KY, KX] = w.detach().float()[:, :, ky, kx]
    return w2


def pack_stem7x7(w: torch.Tensor) -> torch.Tensor:
    """7x7 3 -> 64 stem weights [64,3,7,7] -> the operand of pf_enc_stem: K axis k = ky * 24 + kx * 3 + c (a patch row's 21
    interleaved floats padded to 24), 176 padded, per 8-k piece {bf16 hi[8], bf16 lo[8]}: bf16 [64][22][2][8]."""
    assert tuple(w.shape) == (64, 3, 7, 7)
    wk = torch.zeros(64, 7, 24, dtype=torch.float32, device=w.device)
    wk[:, :, :21] = w.detach().float().permute(0, 2, 3, 1).reshape(64, 7, 21)
    wk = torch.cat([wk.reshape(64, 168), torch.zeros(64, 8, device=w.device)], 1)
    hi = wk.to(torch.bfloat16)
    lo = (wk - hi.float()).to(torch.bfloat16)
    return torch.cat([hi.view(64, 22, 1, 8), lo.view(64, 22, 1, 8)], 2).contiguous()


def default_precision() -> int:
    """PRIORFLOW_PRECISION=fp32 selects the exact-fp32 MFMA path; default is the 3-pass bf16
    split (same parity class: SURVEY.md §7 measured 2e-5 EPE for split-x3 update blocks)."""
    return PREC_F32 if os.environ.get("PRIORFLOW_PRECISION", "bf16x3").lower() in ("fp32", "f32") else PREC_BF16X3


def pack_direct(w: torch.Tensor, b: torch.Tensor):
    """[Cout,Cin,KH,KW] -> [KH*KW][Cin][Cout]."""
    cout, cin, kh, kw = w.shape
    wp = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, cin, cout).contiguous()
    return wp, b.detach().float().contiguous()


class Conv:
    """One packed convolution (MFMA implicit-GEMM path)."""

    def __init__(self, w, b, kh, kw, cin, cout, precision=PREC_F32, presplit=False):
        """presplit: `w` already is the bf16 hi|lo operand (PfLib.pack_conv_weights)."""
        self.w, self.b, self.kh, self.kw, self.cin, self.cout = w, b, kh, kw, cin, cout
        self.precision = precision
        if precision == PREC_BF16X3 and not presplit:
            self.w = split_bf16(w)

    @staticmethod
    def of(mod, precision=PREC_F32, cin_to: int = 0) -> "Conv":
        w, b = pack_mfma(mod.weight, mod.bias, cin_to)
        return Conv(w, b, mod.weight.shape[2], mod.weight.shape[3], max(mod.weight.shape[1], cin_to),
                    mod.weight.shape[0], precision)

    @staticmethod
    def folded(mod, bn, precision=PREC_F32, weight: Optional[torch.Tensor] = None) -> "Conv":
        """conv followed by an eval-mode BatchNorm as ONE convolution: BN(conv(x)) = conv'(x) with w' = w * s[cout],
        b' = b * s + t, s = gamma / sqrt(running_var + eps), t = beta - running_mean * s (core/extractor.py:41-47,112-147 with
        norm_fn='batch' in eval mode, core/prior_raft.py:43-48).  `weight`: an already re-laid-out kernel of `mod`
        (the space-to-depth form of the 7x7 stem) with the same output channels."""
        w = (mod.weight if weight is None else weight).detach().float()
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
        sh = (bn.bias - bn.running_mean * sc).detach().float()
        wf = w * sc.view(-1, 1, 1, 1)
        bf = mod.bias.detach().float() * sc + sh
        wp, bp = pack_mfma(wf, bf)
        return Conv(wp, bp, wf.shape[2], wf.shape[3], wf.shape[1], wf.shape[0], precision)

    @staticmethod
    def dgrad_of(weight: torch.Tensor, precision=PREC_F32) -> "Conv":
        """The convolution that maps dY to dX for a stride-1 conv with `weight` [Cout,Cin,KH,KW] (odd kernel):
        dX = conv(dY, W'), W'[c][o][ky][kx] = W[o][c][KH-1-ky][KW-1-kx], no bias -- runs on pf_conv2d."""
        wt = weight.detach().float().flip(2, 3).transpose(0, 1).contiguous()
        wp, bp = pack_mfma(wt, torch.zeros(wt.shape[0], device=wt.device))
        return Conv(wp, bp, wt.shape[2], wt.shape[3], wt.shape[1], wt.shape[0], precision)

    @staticmethod
    def unpack_wgrad(dw: torch.Tensor, cout: int, cin: int, kh: int, kw: int) -> torch.Tensor:
        """Packed weight gradient [Cout_pad128][KH*KW][Cin_pad32] -> [Cout,Cin,KH,KW]."""
        return dw[:cout, :, :cin].reshape(cout, kh, kw, cin).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def fused(mod_z, mod_r, precision=PREC_F32) -> "Conv":
        """convz|convr share their input (core/update.py:48-49): one 384->256 GEMM."""
        w = torch.cat([mod_z.weight, mod_r.weight], 0)
        b = torch.cat([mod_z.bias, mod_r.bias], 0)
        wp, bp = pack_mfma(w, b)
        return Conv(wp, bp, w.shape[2], w.shape[3], w.shape[1], w.shape[0], precision)

    @staticmethod
    def of_slices(mods, cin_slices, precision, with_bias: bool) -> "Conv":
        """The convolutions `mods` (same input, same kernel shape) concatenated on Cout and restricted to the input channels
        of `cin_slices` (concatenated in that order); bias of the modules, or zero."""
        w = torch.cat([m.weight.detach().float() for m in mods], 0)
        w = torch.cat([w[:, a:b] for a, b in cin_slices], 1).contiguous()
        b = torch.cat([m.bias.detach().float() for m in mods], 0)
        wp, bp = pack_mfma(w, b if with_bias else torch.zeros_like(b))
        return Conv(wp, bp, w.shape[2], w.shape[3], w.shape[1], w.shape[0], precision)

    def desc(self, in0, off0, c0, out, off_out, epilogue, in1=None, off1=0, c1=0, scale=1.0,
             h=None, z=None, aux=None, stride=1, in_scale=None, in_shift=None, in_relu=False,
             stats=None, in0s=None, in1s=None, outs=None, auxs=None, pre=None, off_pre=0, save_gates=False) -> ConvDesc:
        """in0s / in1s / outs / auxs: optional split twins (``split_twin``) of in0 / in1 / out / aux; with a twin given the
        fp32 tensor may be None (operands: the all-DMA kernel reads only the twins; outputs: twin only)."""
        assert c0 + c1 == self.cin, (c0, c1, self.cin)
        d = ConvDesc()
        # row buffers may be column-sliced views of wider ones: the row stride, not the view's width, is the leading dimension
        ld = lambda t: 0 if t is None else (t.stride(-2) if t.dim() >= 2 else t.shape[-1])       # noqa: E731
        d.in0 = in0.data_ptr() if in0 is not None else None
        d.ld0, d.off0, d.c0 = ld(in0), off0, c0
        d.in1 = in1.data_ptr() if in1 is not None else None
        d.ld1, d.off1, d.c1 = ld(in1), off1, c1
        d.weight, d.bias = self.w.data_ptr(), self.b.data_ptr()
        d.out = out.data_ptr() if out is not None else None
        d.ld_out, d.off_out, d.cout = ld(out), off_out, self.cout
        for name, t in (("in0_split", in0s), ("in1_split", in1s), ("out_split", outs), ("aux_split", auxs)):
            if t is not None:
                assert t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[2:] == (2, 32) and t.is_contiguous(), name
                setattr(d, name, t.data_ptr())
        zb = None
        if in0s is not None:           # the zero padding of an all-DMA launch is read from memory
            zb = _zero_block(in0s.device)
            d.zeros, d.zeros_bytes = zb.data_ptr(), zb.numel() * 4
        d.lds0 = in0s.shape[1] if in0s is not None else 0
        d.lds1 = in1s.shape[1] if in1s is not None else 0
        d.lds_out = outs.shape[1] if outs is not None else 0
        d.lds_aux = auxs.shape[1] if auxs is not None else 0
        d.kh, d.kw, d.epilogue, d.scale = self.kh, self.kw, epilogue, scale
        d.h = h.data_ptr() if h is not None else None
        d.ld_h = ld(h)
        d.z = z.data_ptr() if z is not None else None
        d.ld_z = ld(z)
        d.aux_out = aux.data_ptr() if aux is not None else None
        d.ld_aux = ld(aux)
        d.precision = self.precision
        d.stride = stride
        d.in_scale = in_scale.data_ptr() if in_scale is not None else None
        d.in_shift = in_shift.data_ptr() if in_shift is not None else None
        d.in_relu = int(in_relu)
        d.stats_out = stats.data_ptr() if stats is not None else None
        # start value of the accumulation (all-DMA kernel): channel-last fp32 [rows][ld_pre], this conv's channels at off_pre
        d.pre = pre.data_ptr() if pre is not None else None
        d.ld_pre, d.off_pre = ld(pre), off_pre
        d.save_gates = int(save_gates)      # training forward: GRU_ZR also stores r, GRU_Q also stores q (into aux)
        # the C struct holds raw pointers only: keep every tensor alive for as long as the descriptor is
        d._keep = (in0, in1, out, h, z, aux, in_scale, in_shift, stats, self.w, self.b, in0s, in1s, outs, auxs, zb, pre)
        return d


class DirectConv:
    def __init__(self, mod):
        self.w, self.b = pack_direct(mod.weight, mod.bias)
        self.cout, self.cin, self.kh, self.kw = mod.weight.shape


def pack_update_blocks(oddc, upd, precision: Optional[int] = None) -> Dict[str, object]:
    pr = default_precision() if precision is None else precision
    C = lambda m: Conv.of(m, pr)                      # noqa: E731
    ea, eb = oddc.encoder, upd.encoder
    P: Dict[str, object] = {
        "precision": pr,
        "a.c1": C(ea.convc1_A), "a.c2": C(ea.convc2_A),
        "a.f1a": DirectConv(ea.convf1_A), "a.f2a": C(ea.convf2_A),
        "a.f1b": DirectConv(ea.convf1_B), "a.f2b": C(ea.convf2_B),
        "a.cf1": DirectConv(ea.conv_conf1), "a.cf2": DirectConv(ea.conv_conf2),
        "a.out": C(ea.conv_A),
        "b.c1": C(eb.convc1), "b.c2": C(eb.convc2),
        # branch B's conv sees its 256-channel concat zero-padded to 272 so that it shares the
        # launch geometry (K = 9 x 272) of branch A's conv_A: one grid instead of two half-empty ones
        "b.f1": DirectConv(eb.convf1), "b.f2": C(eb.convf2), "b.out": Conv.of(eb.conv, pr, cin_to=272),
    }
    for tag, blk in (("a", oddc), ("b", upd)):
        g = blk.gru
        P[f"{tag}.zr1"] = Conv.fused(g.convz1, g.convr1, pr)
        P[f"{tag}.q1"] = C(g.convq1)
        P[f"{tag}.zr2"] = Conv.fused(g.convz2, g.convr2, pr)
        P[f"{tag}.q2"] = C(g.convq2)
        if pr == PREC_BF16X3:
            # context hoist (Engine.hoist_context): GRU input = [h 0:128 | inp 128:256 | motion 256:384]; `inp` does not change
            # over the iterations, so its part of all three gates is ONE 128 -> 384 conv per half-step, run once per forward
            # (with the biases), and the per-iteration convs see [h | motion] only
            for n, (cz, cr, cq) in (("1", (g.convz1, g.convr1, g.convq1)), ("2", (g.convz2, g.convr2, g.convq2))):
                P[f"{tag}.pre{n}"] = Conv.of_slices((cz, cr, cq), [(128, 256)], pr, with_bias=True)
                P[f"{tag}.zr{n}h"] = Conv.of_slices((cz, cr), [(0, 128), (256, 384)], pr, with_bias=False)
                P[f"{tag}.q{n}h"] = Conv.of_slices((cq,), [(0, 128), (256, 384)], pr, with_bias=False)
        P[f"{tag}.fh1"] = C(blk.flow_head.conv1)
        P[f"{tag}.fh2"] = C(blk.flow_head.conv2)
        w2 = blk.flow_head.conv2.weight.detach().float()          # [2,256,3,3] -> [2][9][256]
        P[f"{tag}.fh2w"] = w2.permute(0, 2, 3, 1).reshape(2, 9, w2.shape[1]).contiguous()
        P[f"{tag}.fh2b"] = blk.flow_head.conv2.bias.detach().float().contiguous()
        P[f"{tag}.m0"] = C(blk.mask[0])
        P[f"{tag}.m2"] = C(blk.mask[2])
    return P


# ------------------------------------------------------------------------------------------
class Workspace:
    """All device buffers of one (B, H, W) problem; allocated once, reused every call."""

    def __init__(self, lib: PfLib, B: int, H: int, W: int, device):
        if H % 8 or W % 8 or H < 128 or W < 128:
            raise PfError(f"image size {H}x{W}: H and W must be multiples of 8 (callers pad, core/utils/utils.py:7-27) "
                          "and at least 128 (the coarsest pyramid level must be 2x2 or larger)")
        self.B, self.H, self.W = B, H, W
        self.H8, self.W8 = H // 8, W // 8
        self.N = self.H8 * self.W8
        self.device = device
        N, H8, W8 = self.N, self.H8, self.W8
        rows = B * N
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)  # noqa: E731
        # ---- sample grids (constant per shape; K8).  grid(R_A2B^T) == grid(R_B2A) bit-exactly,
        # so only the two directions are generated (core/prior_raft.py:115-125 builds 8).
        self.g_a2b = z(2, H, W)
        self.g_a2b_8 = z(2, H8, W8)
        self.g_b2a_8 = z(2, H8, W8)
        r_a2b, r_b2a = rotation_x(-math.pi / 2), rotation_x(math.pi / 2)
        lib.sample_grid(self.g_a2b, r_a2b)
        lib.sample_grid(self.g_a2b_8, r_a2b)
        lib.sample_grid(self.g_b2a_8, r_b2a)
        # the lookups' cross-view grid also interleaved per pixel [N][2]: one 16-byte load per bilinear row pair (the planar
        # form stays in the C-ABI: g_w2c_il = NULL; results are bit-identical)
        self.g_a2b_8_il = self.g_a2b_8.reshape(2, -1).t().contiguous()
        self.g_b2a_8_il = self.g_b2a_8.reshape(2, -1).t().contiguous()
        # ---- encoders' side
        # fnet output of the 4 images (f1A, f2A, f1B, f2B), one row block each
        self.f_all = z(4 * rows, 256)
        self.f = {k: self.f_all[i * rows:(i + 1) * rows] for i, k in enumerate(("f1a", "f2a", "f1b", "f2b"))}
        self.f_split = torch.zeros(4 * rows, 8, 2, 32, dtype=torch.bfloat16, device=device)   # bf16 hi|lo rows
        self.img_c = z(2 * B, 3, H, W)         # cnet input  [image1_A | image1_B]
        self.img_f = z(4 * B, 3, H, W)         # fnet input  [image1_A | image2_A | image1_B | image2_B]
        # ---- pyramids: level i rows [B*N, (H8>>i)*(W8>>i)]
        self.pyr_a = [z(rows, (H8 >> i) * (W8 >> i)) for i in range(4)]
        self.pyr_b = [z(rows, (H8 >> i) * (W8 >> i)) for i in range(4)]
        # ---- loop state
        # coords0 (the pixel grid, core/utils/utils.py:75-78) is a constant of the shape: built once, copied per forward
        xs = torch.arange(W8, device=device, dtype=torch.float32).view(1, 1, 1, W8).expand(B, 1, H8, W8)
        ys = torch.arange(H8, device=device, dtype=torch.float32).view(1, 1, H8, 1).expand(B, 1, H8, W8)
        self.coords0 = torch.cat([xs, ys], 1).contiguous()
        self.c1a = z(B, 2, H8, W8)
        self.c1b = z(B, 2, H8, W8)
        self.flow_b = z(B, 2, H8, W8)
        self.flow_ba = z(B, 2, H8, W8)
        self.flow_tmp = z(B, 2, H8, W8)
        # cnet output lands here directly (view A rows first, then view B)
        self.net0_ab = z(2 * rows, 128)
        self.x_ab = z(2 * rows, 256)
        self.net_a = [self.net0_ab[:rows], z(rows, 128)]
        self.net_b = [self.net0_ab[rows:], z(rows, 128)]
        self.x_a = self.x_ab[:rows]
        self.x_b = self.x_ab[rows:]
        self.z_a, self.z_b = z(rows, 128), z(rows, 128)
        self.rh_a, self.rh_b = z(rows, 128), z(rows, 128)
        self.own, self.raw = z(rows, CORR_CH), z(rows, CORR_CH)
        self.own_b, self.raw_b = z(rows, CORR_CH), z(rows, CORR_CH)      # branch B's lookups run concurrently
        self.corr_a, self.corr_b = z(rows, CORR_CH), z(rows, CORR_CH)
        self.c1_a, self.c1_b = z(rows, 256), z(rows, 256)
        self.cat_a, self.cat_b = z(rows, 272), z(rows, 272)     # cat_b columns 256..271 stay zero
        self.flow4_a = z(rows, 4)          # [flow_A | flow_B_A]
        self.flow2_b = z(rows, 2)
        self.t_a, self.t_ba, self.t_b = z(rows, 128), z(rows, 128), z(rows, 128)
        self.conf_in, self.conf_mid = z(rows, 8), z(rows, 32)
        self.fh_a, self.fh_b, self.mh_a, self.mh_b = z(rows, 256), z(rows, 256), z(rows, 256), z(rows, 256)
        self.delta_a, self.delta_b = z(rows, 4), z(rows, 4)
        self.mask_a, self.mask_b = z(rows, 576), z(rows, 576)
        # ---- split twins (bf16 hi|lo rows, include/priorflow_hip.h): every activation an MFMA conv of the update blocks
        # consumes is written in this form by its producer, so both operands of those convs go global -> LDS by DMA.
        # Only the hidden state keeps an fp32 copy as well (the GRU epilogues blend with it).
        tw = lambda r, c: split_twin(r, c, device)  # noqa: E731
        self.net0_ab_s = tw(2 * rows, 128)
        self.net_a_s = [self.net0_ab_s[:rows], tw(rows, 128)]
        self.net_b_s = [self.net0_ab_s[rows:], tw(rows, 128)]
        self.x_ab_s = tw(2 * rows, 256)
        self.x_a_s, self.x_b_s = self.x_ab_s[:rows], self.x_ab_s[rows:]
        self.rh_a_s, self.rh_b_s = tw(rows, 128), tw(rows, 128)
        # context hoist: conv_inp(inp) + bias of [z | r | q] for the two GRU half-steps, per branch (fp32 [rows][384])
        self.pre = {(t, n): z(rows, 384) for t in "ab" for n in "12"}
        self.pre_ready = False          # set by Engine.hoist_context, cleared whenever `inp` is rewritten
        self.c1_a_s, self.c1_b_s = tw(rows, 256), tw(rows, 256)
        self.cat_a_s, self.cat_b_s = tw(rows, 272), tw(rows, 272)      # 9 chunks; cat_b's last chunk stays zero
        self.t_a_s, self.t_ba_s, self.t_b_s = tw(rows, 128), tw(rows, 128), tw(rows, 128)

    def sync_twins(self, lib: PfLib):
        """Refresh the twins of the loop's INPUT state from the fp32 buffers (tests that fill net / x by hand)."""
        lib.split_bf16(self.net0_ab, self.net0_ab_s)
        lib.split_bf16(self.net_a[1], self.net_a_s[1])
        lib.split_bf16(self.net_b[1], self.net_b_s[1])
        lib.split_bf16(self.x_ab, self.x_ab_s)

    def nbytes(self) -> int:
        tot = 0
        for v in self.__dict__.values():
            for t in (v if isinstance(v, (list, tuple)) else (v.values() if isinstance(v, dict) else [v])):
                if isinstance(t, torch.Tensor):
                    tot += t.numel() * t.element_size()
        return tot


class Engine:
    def __init__(self, lib: PfLib, side_streams=None):
        """side_streams: optional (s1, s2, s3) torch streams.  When given, the three independent chains of
        an iteration -- correlation (lookups, 1x1, 3x3), flow (prep, flo_rotate, 7x7, 3x3) and
        confidence (warps, 3x3, 3x3) -- run concurrently and join before conv_A (inside a HIP-graph
        capture they become parallel branches)."""
        self.lib = lib
        self.side = side_streams
        # debugging aid: PRIORFLOW_FORKS is a bit mask of the forks to keep (2: the three chains of
        # motion_inputs, 4: branch B's lookups, 8: the head tails; 1 is the encoders' fork in prior_raft.py)
        self.forks = int(os.environ.get("PRIORFLOW_FORKS", "15")) if side_streams is not None else 0
        self._b_pending = None      # event after branch B's deferred FlowHead tail (see iteration())
        # capture order (bit mask; see motion_inputs): 1 chains, 2 head tails, 4 lookups -- the calling stream's own chain is captured
        # first at every fork (round 2's A/B: -330 us per forward; the PRIORFLOW_ORDER knob was retired in round 6)
        self.order = 15
        # pre-split activations + all-DMA convs (bf16x3 only); PRIORFLOW_PRESPLIT=0 keeps the fp32-staged kernels (A/B knob:
        # the results are bit-identical)
        self.presplit_on = os.environ.get("PRIORFLOW_PRESPLIT", "1") != "0"
        self.hoist_on = os.environ.get("PRIORFLOW_HOIST_CTX", "1") != "0"
        # branch A's and branch B's update blocks as TWO chains of half-chip launches on two queues instead of two groups of one
        # chain of launches (iteration_split; test_mode forwards on the default pre-split path only; every batch size: +7.7 % at
        # B = 1, +6.3 % at 2, +4.8 % at 4, +4 % at 8, +0.6 % at 32 -- profiles/r6_ab_split_chains.txt).  PRIORFLOW_SPLIT_AB=0: off.
        self.split_ab = os.environ.get("PRIORFLOW_SPLIT_AB", "1") != "0"

    def presplit(self, P) -> bool:
        return self.presplit_on and P["precision"] == PREC_BF16X3

    def hoist(self, P) -> bool:
        """Context hoist on: the GRU convs take the iteration-invariant `inp` part from ws.pre (needs the all-DMA kernel)."""
        return self.hoist_on and self.presplit(P)

    def hoist_context(self, ws: "Workspace", P):
        """Once per forward, after cnet: pre = conv_{inp-part of [convz|convr|convq]}(inp) + bias for both GRU half-steps
        (core/update.py:46-60 with x = cat[inp, motion], core/prior_raft.py:196: `inp` is the same tensor in every
        iteration).  Two launches (1x5, 5x1), both branches as groups; consumed by update_blocks through pf_conv_desc.pre."""
        ws.pre_ready = False
        if not self.hoist(P):
            return
        ws.pre_ready = True
        like = ws.pre[("a", "1")]
        for n in "12":
            self.lib.conv2d([P[f"{t}.pre{n}"].desc(None, 0, 128, ws.pre[(t, n)], 0, EPI_LINEAR, in0s=xs)
                             for t, xs in (("a", ws.x_a_s), ("b", ws.x_b_s))], ws.B, ws.H8, ws.W8, like)

    # ---- stage 0: view B images --------------------------------------------------------------
    def prepare_images(self, ws: Workspace, image1: torch.Tensor, image2: torch.Tensor):
        """Input stage (core/prior_raft.py:121-127): 2 * (image / 255) - 1 of both images and img_rotate of the pair into
        view B, written straight into the encoders' batches ws.img_f = [im1 | im2 | im1_B | im2_B] and
        ws.img_c = [im1 | im1_B]: ONE launch (pf_prepare_images; round 6 -- normalise, rotate and a copy before, bit-identical)
        and no torch arithmetic."""
        B = ws.B
        if tuple(image1.shape) != tuple(ws.img_f[:B].shape) or image2.shape != image1.shape:
            raise PfError(f"prepare_images: images {tuple(image1.shape)} / {tuple(image2.shape)} do not fit the workspace")
        image1 = image1.contiguous().float()
        image2 = image2.contiguous().float()
        # rotating [im1 | im2] as 2B three-channel images gives [im1_B | im2_B] in place of the reference's six-channel stack:
        # the sample grid has no batch dimension
        self.lib.prepare_images(image1, image2, ws.g_a2b, ws.img_f, ws.img_c)

    # ---- stage 1: corr volumes + pyramids (the encoders write the channel-last features themselves) ----------
    def build_pyramids(self, ws: Workspace, precision: int = PREC_F32):
        """corr + build_pyramid for both views (core/prior_raft.py:151-159)."""
        if precision == PREC_BF16X3:
            if not getattr(ws, "f_split_ready", False):        # (round 6: fnet's last convolution writes the hi|lo rows itself)
                self.lib.split_bf16(ws.f_all, ws.f_split)      # all four feature maps at once
            rows = ws.B * ws.N
            fs = [ws.f_split[i * rows:(i + 1) * rows] for i in range(4)]
            # (the two volumes on two queues -- one launch's store phase under the other's GEMM phase -- measured 138.9 against
            # 139.1 pairs/s in round 3 and was removed)
            self.lib.corr_pyramid_bf16x3(fs[0], fs[1], ws.pyr_a, ws.B, ws.H8, ws.W8, 256)
            self.lib.corr_pyramid_bf16x3(fs[2], fs[3], ws.pyr_b, ws.B, ws.H8, ws.W8, 256)
            return
        self.lib.corr_pyramid(ws.f["f1a"], ws.f["f2a"], ws.pyr_a, ws.B, ws.H8, ws.W8)
        self.lib.corr_pyramid(ws.f["f1b"], ws.f["f2b"], ws.pyr_b, ws.B, ws.H8, ws.W8)

    def init_coords(self, ws: Workspace, init_flow: Optional[torch.Tensor]):
        """initialize_flow (+ init_flow) (core/prior_raft.py:161-165)."""
        B, H8, W8 = ws.B, ws.H8, ws.W8
        ws.c1a.copy_(ws.coords0)
        ws.c1b.copy_(ws.coords0)
        if init_flow is not None:
            fl = init_flow.to(torch.float32).contiguous()
            ws.c1a.add_(fl)
            # flo_rotate(init_flow, W2C = grid(R_A2B^T) == grid(R_B2A), C2W = grid(R_A2B))
            self.lib.flo_rotate(fl, ws.g_b2a_8, ws.g_a2b_8, ws.flow_tmp)
            ws.c1b.add_(ws.flow_tmp)

    # ---- one refinement iteration (core/prior_raft.py:170-211) --------------------------------
    def iteration(self, ws: Workspace, P: Dict[str, object], cur: int, need_b: bool, mask_a: bool,
                  mask_b: bool, defer_b_join: bool = False) -> int:
        """Runs one iteration; hidden states are read from net_x[cur] and end in net_x[cur]
        (two GRU half-steps ping-pong).  need_b=False skips branch B's update (its result is
        dead in the last test_mode iteration); mask_x selects the mask heads.
        defer_b_join: another iteration follows and nothing on the calling stream reads branch B's coords
        before it: branch B's FlowHead tail is then not joined here but awaited by its consumers in the next
        motion_inputs() (saves one cross-queue dependency hop, ~8 us, at every iteration boundary)."""
        if self.can_split(ws, P, need_b, mask_b, defer_b_join):
            return self.iteration_split(ws, P, cur, need_b, mask_a)
        self.motion_inputs(ws, P, need_b)
        return self.update_blocks(ws, P, cur, need_b, mask_a, mask_b, inputs_ready=True,
                                  defer_b_join=defer_b_join)   # incl. coords1 += delta

    # ---- round 6: the two branches as two chains ------------------------------------------------------------
    def can_split(self, ws: Workspace, P, need_b: bool, mask_b: bool, defer_b_join: bool) -> bool:
        """iteration_split applies to what the captured test_mode forward runs: side streams, the pre-split path with the hoisted
        context and the fused combine, no B mask head, and B's result either deferred to the next iteration or not needed."""
        return bool(self.split_ab and self.side is not None and self.forks == 15 and self.hoist(P) and ws.pre_ready
                    and not mask_b and (defer_b_join or not need_b))

    def iteration_split(self, ws: Workspace, P, cur: int, need_b: bool, mask_a: bool) -> int:
        """One iteration (core/prior_raft.py:170-211) with branch A's chain (lookup, combine + 1x1, motion encoder, SepConvGRU,
        FlowHead) on the calling stream and branch B's on a side stream, every convolution a launch of ONE group with
        pf_conv_desc.co_groups = 1 (the tile a two-group launch takes; half the chip's work items each).  The two chains couple
        only through the flow chain (pf_motion_prep reads both branches' coords1) at the head of an iteration, so B's chain may lag
        A's by up to A's lookup + combine + convc2, and one chain's store bursts and prologues run under the other's K loops.
        Arithmetic and launch arguments per branch are those of motion_inputs() + update_blocks(): bit-identical results."""
        lib, B, H8, W8 = self.lib, ws.B, ws.H8, ws.W8
        like = ws.x_a
        main = torch.cuda.current_stream()
        s1, s2, sb = self.side
        co = 1 if need_b else 0

        def conv(d):
            d.co_groups = co
            lib.conv2d([d], B, H8, W8, like)

        # Capture order decides the queues.  hipGraphInstantiate walks the graph depth first from its roots; a node's FIRST-captured
        # successor inherits its queue, the k-th further one gets queue + k (mod 4), and a node keeps the queue of whoever reaches
        # it first (observed; motion_inputs relies on the first half of the rule).  The walk descends branch A's chain through all
        # iterations before anything else, so what hangs off A's last kernel of an iteration decides the rest: its successors are
        # captured in the order A's next lookup (same queue), B's next lookup (queue + 1: B's whole chain follows), pf_motion_prep
        # (queue + 2: the flow chain; its second successor, the confidence stem, queue + 3).  B's lookup waits for `start` only
        # for this ordering -- B lags A, so the edge never delays it.  (With pf_motion_prep captured second the flow chain
        # shared B's queue and ran behind B's lookup + combine + convc2: 429 us per iteration instead of 405.)
        start = torch.cuda.Event()
        start.record(main)
        # ---- heads: lookup, rotate-back + add + convc1, convc2 (no dependence on the flow chain)
        lib.dccl_lookup(ws.c1a, ws.pyr_a, ws.pyr_b, ws.g_b2a_8, ws.own, ws.raw, ws.g_b2a_8_il)
        lib.dccl_combine_conv1x1([(ws.own, ws.raw, ws.g_b2a_8, P["a.c1"], None, 0, ws.c1_a_s)], B, H8, W8)
        conv(P["a.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_a_s, outs=ws.cat_a_s))
        b_prev = self._b_pending                # end of B's previous chain: the flow chain reads its coords1
        if need_b:
            sb.wait_event(start)
            with torch.cuda.stream(sb):
                lib.dccl_lookup(ws.c1b, ws.pyr_b, ws.pyr_a, ws.g_a2b_8, ws.own_b, ws.raw_b, ws.g_a2b_8_il)
                lib.dccl_combine_conv1x1([(ws.own_b, ws.raw_b, ws.g_a2b_8, P["b.c1"], None, 0, ws.c1_b_s)], B, H8, W8)
                conv(P["b.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_b_s, outs=ws.cat_b_s))
        # ---- flow chain (s1) and confidence chain (s2): need both branches' coords1
        s1.wait_event(start)
        if b_prev is not None:
            s1.wait_event(b_prev)
        with torch.cuda.stream(s1):
            self._flow_chain_head(ws, True)
            head_done = torch.cuda.Event()
            head_done.record(s1)
            self._flow_chain_tail(ws, P, need_b)
            flow_done = torch.cuda.Event()
            flow_done.record(s1)
        s2.wait_event(head_done)
        with torch.cuda.stream(s2):
            self._conf_chain(ws, P)
        # ---- branch A's tail (calling stream)
        main.wait_event(flow_done)
        # B's tail needs the flow chain (s1) too, and s1 waits for B's previous tail (sb): two forked streams that wait on EACH
        # OTHER's events crash hipStreamEndCapture on ROCm 7.2 (segmentation fault after "[hipGraph] Add EmptyNode"; one
        # direction alone, or either stream against the capture's origin stream, is fine) -- so the flow chain's completion
        # reaches sb through an event recorded on the calling stream, which has to wait for it anyway
        relay = torch.cuda.Event()
        relay.record(main)
        main.wait_stream(s2)
        conv(P["a.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_a_s, outs=ws.x_a_s))
        self._gru_branch(ws, P, "a", cur, conv)
        nas = ws.net_a_s[cur]
        if mask_a:              # the mask head's stem shares its input with the FlowHead's: two groups of one launch
            d = [P["a.fh1"].desc(None, 0, 128, ws.fh_a, 0, EPI_RELU, in0s=nas), P["a.m0"].desc(None, 0, 128, ws.mh_a, 0, EPI_RELU, in0s=nas)]
            d[0].co_groups = 0
            lib.conv2d(d, B, H8, W8, like)
            heads_done = torch.cuda.Event()
            heads_done.record(main)
            lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
            s2.wait_event(heads_done)
            with torch.cuda.stream(s2):
                lib.conv2d([P["a.m2"].desc(ws.mh_a, 0, 256, ws.mask_a, 0, EPI_LINEAR, scale=0.25)], B, H8, W8, like)
            main.wait_stream(s2)
        else:
            conv(P["a.fh1"].desc(None, 0, 128, ws.fh_a, 0, EPI_RELU, in0s=nas))
            lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
        # ---- branch B's tail (sb)
        if need_b:
            sb.wait_event(relay)
            with torch.cuda.stream(sb):
                conv(P["b.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_b_s, outs=ws.x_b_s))
                self._gru_branch(ws, P, "b", cur, conv)
                conv(P["b.fh1"].desc(None, 0, 128, ws.fh_b, 0, EPI_RELU, in0s=ws.net_b_s[cur]))
                lib.flow_head_out(ws.fh_b, 256, P["b.fh2w"], P["b.fh2b"], ws.c1b, ws.delta_b)
                self._b_pending = torch.cuda.Event()
                self._b_pending.record(sb)
        else:                   # the last iteration of a test_mode forward: every side queue rejoins the calling stream
            self._await_b(main)
            main.wait_stream(s1)
        return cur

    def _gru_branch(self, ws: Workspace, P, t: str, cur: int, conv):
        """SepConvGRU of ONE branch (core/update.py:46-60) on twins with the hoisted context: z|r, q horizontally, then vertically;
        the hidden state ends in net_x[cur] again."""
        net, zb, ns, xs, rhs = ((ws.net_a, ws.z_a, ws.net_a_s, ws.x_a_s, ws.rh_a_s) if t == "a" else
                                (ws.net_b, ws.z_b, ws.net_b_s, ws.x_b_s, ws.rh_b_s))
        c = cur
        for tag in ("1", "2"):
            conv(P[f"{t}.zr{tag}h"].desc(None, 0, 128, zb, 0, EPI_GRU_ZR, off1=128, c1=128, h=net[c], in0s=ns[c], in1s=xs,
                                          auxs=rhs, pre=ws.pre[(t, tag)], off_pre=0))
            conv(P[f"{t}.q{tag}h"].desc(None, 0, 128, net[c ^ 1], 0, EPI_GRU_Q, off1=128, c1=128, h=net[c], z=zb,
                                         in0s=rhs, in1s=xs, outs=ns[c ^ 1], pre=ws.pre[(t, tag)], off_pre=256))
            c ^= 1
        assert c == cur

    def _await_b(self, stream, keep: bool = False):
        """Make `stream` wait for branch B's deferred FlowHead tail of the previous iteration, if any."""
        if self._b_pending is not None:
            stream.wait_event(self._b_pending)
            if not keep:
                self._b_pending = None

    # -- the three independent chains of an iteration ---------------------------------------------
    def _flow_chain_head(self, ws: Workspace, ps: bool = False):
        """flows + flo_rotate + the two feature warps / groupwise correlations (:171-182) in ONE launch: produces
        flow4_a, flow2_b, the flow tails of x_a / x_b and conf_in.  flo_rotate(flow_B, W2C = grid(R_B2A^T) ==
        grid(R_A2B), C2W = grid(R_B2A)) (:179).  (The five separate kernels it replaced stay in the library for the training
        tape; tests/test_hip_kernels.py pins the fused launch to them bit for bit.)"""
        lib = self.lib
        if ps:          # the GRU-input tails go to the split twins of x_a / x_b only
            lib.motion_prep(ws.c1a, ws.c1b, ws.g_a2b_8, ws.g_b2a_8, ws.f["f1a"], ws.f["f2a"], ws.flow4_a, ws.flow2_b,
                            ws.conf_in, None, 252, None, 254, xa_split=ws.x_a_s, xb_split=ws.x_b_s)
            return
        lib.motion_prep(ws.c1a, ws.c1b, ws.g_a2b_8, ws.g_b2a_8, ws.f["f1a"], ws.f["f2a"], ws.flow4_a, ws.flow2_b,
                        ws.conf_in, ws.x_a, 252, ws.x_b, 254)

    def _flow_chain_tail(self, ws: Workspace, P, need_b: bool):
        """7x7 flow stems + 3x3 (core/update.py:187-191, :94-95) -> cat_a[128:256], cat_b[192:256]."""
        lib, B, H8, W8 = self.lib, ws.B, ws.H8, ws.W8
        ps = self.presplit(P)

        # the 7x7 stems of flow_A, flow_B_A (and flow_B) are independent and of one shape: one launch
        stems = [(P["a.f1a"], ws.flow4_a, 0, ws.t_a, ws.t_a_s), (P["a.f1b"], ws.flow4_a, 2, ws.t_ba, ws.t_ba_s)]
        f2 = [(P["a.f2a"], ws.t_a, ws.t_a_s, ws.cat_a, ws.cat_a_s, 128), (P["a.f2b"], ws.t_ba, ws.t_ba_s, ws.cat_a, ws.cat_a_s, 192)]
        if need_b:
            stems.append((P["b.f1"], ws.flow2_b, 0, ws.t_b, ws.t_b_s))
            f2.append((P["b.f2"], ws.t_b, ws.t_b_s, ws.cat_b, ws.cat_b_s, 192))
        dc = stems[0][0]
        assert all((s[0].cin, s[0].cout, s[0].kh, s[0].kw) == (dc.cin, dc.cout, dc.kh, dc.kw) for s in stems)
        if ps:
            lib.conv2d_direct_group([(x, off_in, c.w, c.b, None, 0, tw) for c, x, off_in, out, tw in stems],
                                    dc.cin, dc.cout, dc.kh, dc.kw, True, B, H8, W8)
            d = [c.desc(None, 0, 128, None, off, EPI_RELU, in0s=ts, outs=cs) for c, t, ts, cat, cs, off in f2]
        else:
            lib.conv2d_direct_group([(x, off_in, c.w, c.b, out, 0) for c, x, off_in, out, tw in stems],
                                    dc.cin, dc.cout, dc.kh, dc.kw, True, B, H8, W8)
            d = [c.desc(t, 0, 128, cat, off, EPI_RELU) for c, t, ts, cat, cs, off in f2]
        lib.conv2d(d, B, H8, W8, ws.x_a)

    def _conf_chain(self, ws: Workspace, P):
        """confidence stem on conf_in = [flaw_A | flaw_B_A] (core/update.py:193-194) -> cat_a[256:272]."""
        self._conf_stem(ws, P)

    def _conf_stem(self, ws: Workspace, P):
        """relu(conv_conf2(relu(conv_conf1(cat[flaw_A, flaw_B_A])))) (core/update.py:193-194): one launch, the 32-channel
        intermediate map stays in LDS."""
        c1, c2 = P["a.cf1"], P["a.cf2"]
        assert (c1.cin, c1.cout, c1.kh, c1.kw, c2.cin, c2.cout, c2.kh, c2.kw) == (8, 32, 3, 3, 32, 16, 3, 3)
        if self.presplit(P):
            self.lib.conf_stem(ws.conf_in, 0, c1.w, c1.b, c2.w, c2.b, None, 256, ws.B, ws.H8, ws.W8, out_split=ws.cat_a_s)
        else:
            self.lib.conf_stem(ws.conf_in, 0, c1.w, c1.b, c2.w, c2.b, ws.cat_a, 256, ws.B, ws.H8, ws.W8)

    def _corr_chain(self, ws: Workspace, P, need_b: bool, fork_from=None):
        """DCCL lookups (K3+K4; :185-188) + 1x1 + 3x3 of the motion encoders -> cat_a[0:128], cat_b[0:192].
        A looks into B through grid(R_A2B^T)==grid(R_B2A) and rotates back with grid(R_B2A); B the other way."""
        lib, B, H8, W8 = self.lib, ws.B, ws.H8, ws.W8
        # bf16x3: rotate-back + add + convc1 are ONE launch (pf_dccl_combine_conv1x1): corr_a / corr_b never exist
        fused = P["precision"] == PREC_BF16X3
        ps = self.presplit(P)

        def look_a():
            lib.dccl_lookup(ws.c1a, ws.pyr_a, ws.pyr_b, ws.g_b2a_8, ws.own, ws.raw, ws.g_b2a_8_il)
            if not fused:
                lib.dccl_combine(ws.own, ws.raw, ws.g_b2a_8, ws.corr_a, B, H8, W8)

        def look_b():
            lib.dccl_lookup(ws.c1b, ws.pyr_b, ws.pyr_a, ws.g_a2b_8, ws.own_b, ws.raw_b, ws.g_a2b_8_il)
            if not fused:
                lib.dccl_combine(ws.own_b, ws.raw_b, ws.g_a2b_8, ws.corr_b, B, H8, W8)

        # (both branches' lookups as ONE grid, pf_dccl_lookup_pair, measured 132.0 / 131.9 against 133.3 / 132.3 pairs/s in round 3:
        # the cross-queue join it removes is paid for by B's lookup no longer starting before A's chain needs the chip.  The
        # entry point stays in the library -- bit-identical, tested -- but the engine no longer carries the switch.)
        if need_b and self.forks & 4:
            # the two branches' lookups are independent gather chains: B's runs beside A's
            main, sb = torch.cuda.current_stream(), self.side[2]
            a_first = bool(self.order & 4)
            if a_first:
                ev = fork_from
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(main)
                look_a()
                sb.wait_event(ev)
            elif fork_from is not None:
                sb.wait_event(fork_from)
            else:
                sb.wait_stream(main)
            self._await_b(sb, keep=True)
            with torch.cuda.stream(sb):
                look_b()
            if not a_first:
                look_a()
            main.wait_stream(sb)
        else:
            look_a()
            if need_b:
                self._await_b(torch.cuda.current_stream(), keep=True)
                look_b()
        c1o = (lambda t: None) if ps else (lambda t: t)       # presplit: convc1's output exists as a twin only
        if fused:
            items = [(ws.own, ws.raw, ws.g_b2a_8, P["a.c1"], c1o(ws.c1_a), 0, ws.c1_a_s if ps else None)]
            if need_b:
                items.append((ws.own_b, ws.raw_b, ws.g_a2b_8, P["b.c1"], c1o(ws.c1_b), 0, ws.c1_b_s if ps else None))
            lib.dccl_combine_conv1x1(items, B, H8, W8)
        else:
            d = [P["a.c1"].desc(ws.corr_a, 0, CORR_CH, c1o(ws.c1_a), 0, EPI_RELU, outs=ws.c1_a_s if ps else None)]
            if need_b:
                d.append(P["b.c1"].desc(ws.corr_b, 0, CORR_CH, c1o(ws.c1_b), 0, EPI_RELU, outs=ws.c1_b_s if ps else None))
            lib.conv2d(d, B, H8, W8, ws.x_a)
        lib.conv2d(self._c2_descs(ws, P, need_b), B, H8, W8, ws.x_a)

    def _c2_descs(self, ws: Workspace, P, need_b: bool):
        """convc2 (3x3 256 -> 128 / 192) of both motion encoders -> cat_a[0:128], cat_b[0:192]."""
        if self.presplit(P):
            d = [P["a.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_a_s, outs=ws.cat_a_s)]
            if need_b:
                d.append(P["b.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_b_s, outs=ws.cat_b_s))
            return d
        d = [P["a.c2"].desc(ws.c1_a, 0, 256, ws.cat_a, 0, EPI_RELU)]
        if need_b:
            d.append(P["b.c2"].desc(ws.c1_b, 0, 256, ws.cat_b, 0, EPI_RELU))
        return d

    def prep_and_lookup(self, ws: Workspace, need_b: bool):
        """flows, flo_rotate and the DCCL lookups of one iteration, single stream (tests)."""
        lib, B, H8, W8 = self.lib, ws.B, ws.H8, ws.W8
        self._flow_chain_head(ws)
        lib.dccl_lookup(ws.c1a, ws.pyr_a, ws.pyr_b, ws.g_b2a_8, ws.own, ws.raw, ws.g_b2a_8_il)
        lib.dccl_combine(ws.own, ws.raw, ws.g_b2a_8, ws.corr_a, B, H8, W8)
        if need_b:
            lib.dccl_lookup(ws.c1b, ws.pyr_b, ws.pyr_a, ws.g_a2b_8, ws.own, ws.raw, ws.g_a2b_8_il)
            lib.dccl_combine(ws.own, ws.raw, ws.g_a2b_8, ws.corr_b, B, H8, W8)

    def motion_inputs(self, ws: Workspace, P, need_b: bool):
        """Everything of an iteration up to (excluding) conv_A / conv: fills cat_a, cat_b and the flow
        tails of x_a, x_b from coords1 and the pyramids.  Three concurrent chains when side streams exist."""
        if not self.forks & 2:
            self._await_b(torch.cuda.current_stream())
            self._flow_chain_head(ws, self.presplit(P))
            self._corr_chain(ws, P, need_b)
            self._flow_chain_tail(ws, P, need_b)
            self._conf_chain(ws, P)
            return
        main = torch.cuda.current_stream()
        s1, s2 = self.side[0], self.side[1]
        # Enqueue order matters inside a captured graph: the runtime keeps a node's FIRST-captured successor on its
        # queue and moves the others to other queues (a cross-queue dependency costs ~10 us).  The calling stream's
        # own chain is therefore enqueued first and the side chains fork from an event recorded at its start
        # (same dependencies as forking first; -330 us per forward in a same-box A/B).
        order = "main-first" if self.order & 1 else "forks-first"
        if order == "main-first":
            start = torch.cuda.Event()
            start.record(main)
            self._corr_chain(ws, P, need_b, fork_from=start)
            s1.wait_event(start)
        else:
            s1.wait_stream(main)
        with torch.cuda.stream(s1):
            self._flow_chain_head(ws, self.presplit(P))
            head_done = torch.cuda.Event()
            head_done.record(s1)
            self._flow_chain_tail(ws, P, need_b)
        s2.wait_event(head_done)            # the confidence stem needs conf_in
        with torch.cuda.stream(s2):
            self._conf_chain(ws, P)
        if order != "main-first":
            self._corr_chain(ws, P, need_b)
        main.wait_stream(s1)
        main.wait_stream(s2)
        self._b_pending = None      # s1 (which ran the deferred tail) has been joined

    def update_blocks(self, ws: Workspace, P: Dict[str, object], cur: int, need_b: bool, mask_a: bool,
                      mask_b: bool, inputs_ready: bool = False, defer_b_join: bool = False) -> int:
        """ODDC (branch A) and update_block (branch B) (core/update.py:152-159, :129-136).
        Inputs: corr_x, flow4_a / flow2_b, conf_in, x_x (inp + flow tail), net_x[cur], c1x.
        Outputs: net_x[cur], delta_x, mask_x, and c1x += delta_x (core/prior_raft.py:193,196).
        inputs_ready: cat_a / cat_b were already filled by motion_inputs()."""
        lib, B, H8, W8 = self.lib, ws.B, ws.H8, ws.W8
        like = ws.x_a

        def conv(descs):
            lib.conv2d(descs, B, H8, W8, like)

        ps = self.presplit(P)
        if not inputs_ready:
            # motion encoders (core/update.py:183-201, :91-99) from corr_x / flows / conf_in
            d = [P["a.c1"].desc(ws.corr_a, 0, CORR_CH, None if ps else ws.c1_a, 0, EPI_RELU, outs=ws.c1_a_s if ps else None)]
            if need_b:
                d.append(P["b.c1"].desc(ws.corr_b, 0, CORR_CH, None if ps else ws.c1_b, 0, EPI_RELU,
                                        outs=ws.c1_b_s if ps else None))
            conv(d)
            conv(self._c2_descs(ws, P, need_b))
            self._flow_chain_tail(ws, P, need_b)
            self._conf_stem(ws, P)
        if ps:
            d = [P["a.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_a_s, outs=ws.x_a_s)]
            if need_b:
                d.append(P["b.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_b_s, outs=ws.x_b_s))
        else:
            d = [P["a.out"].desc(ws.cat_a, 0, 272, ws.x_a, 128, EPI_RELU)]
            if need_b:
                d.append(P["b.out"].desc(ws.cat_b, 0, 272, ws.x_b, 128, EPI_RELU))
        conv(d)

        # SepConvGRU (core/update.py:46-60): z|r fused GEMM with sigmoid + r*h epilogue, then q
        # with the tanh + blend epilogue; horizontal (1x5) then vertical (5x1)
        branches = [("a", ws.net_a, ws.x_a, ws.z_a, ws.rh_a, ws.net_a_s, ws.x_a_s, ws.rh_a_s)]
        if need_b:
            branches.append(("b", ws.net_b, ws.x_b, ws.z_b, ws.rh_b, ws.net_b_s, ws.x_b_s, ws.rh_b_s))
        c = cur
        for tag in ("1", "2"):
            if ps and self.hoist(P) and ws.pre_ready:     # [h | motion] only: the inp part (and the bias) is the accumulators' start value
                conv([P[f"{t}.zr{tag}h"].desc(None, 0, 128, zb, 0, EPI_GRU_ZR, off1=128, c1=128, h=net[c], in0s=ns[c], in1s=xs,
                                               auxs=rhs, pre=ws.pre[(t, tag)], off_pre=0)
                      for t, net, x, zb, rh, ns, xs, rhs in branches])
                conv([P[f"{t}.q{tag}h"].desc(None, 0, 128, net[c ^ 1], 0, EPI_GRU_Q, off1=128, c1=128, h=net[c], z=zb,
                                              in0s=rhs, in1s=xs, outs=ns[c ^ 1], pre=ws.pre[(t, tag)], off_pre=256)
                      for t, net, x, zb, rh, ns, xs, rhs in branches])
            elif ps:    # operands as twins; z stays fp32 (epilogue operand of q), r*h exists as a twin only, h' in both forms
                conv([P[f"{t}.zr{tag}"].desc(None, 0, 128, zb, 0, EPI_GRU_ZR, off1=0, c1=256, h=net[c], in0s=ns[c], in1s=xs,
                                              auxs=rhs) for t, net, x, zb, rh, ns, xs, rhs in branches])
                conv([P[f"{t}.q{tag}"].desc(None, 0, 128, net[c ^ 1], 0, EPI_GRU_Q, off1=0, c1=256, h=net[c], z=zb,
                                             in0s=rhs, in1s=xs, outs=ns[c ^ 1]) for t, net, x, zb, rh, ns, xs, rhs in branches])
            else:
                conv([P[f"{t}.zr{tag}"].desc(net[c], 0, 128, zb, 0, EPI_GRU_ZR, in1=x, off1=0, c1=256,
                                              h=net[c], aux=rh) for t, net, x, zb, rh, ns, xs, rhs in branches])
                conv([P[f"{t}.q{tag}"].desc(rh, 0, 128, net[c ^ 1], 0, EPI_GRU_Q, in1=x, off1=0, c1=256,
                                             h=net[c], z=zb) for t, net, x, zb, rh, ns, xs, rhs in branches])
            c ^= 1
        assert c == cur

        # heads (core/update.py:13-14, :124-136): the 3x3 128->256 stems share their input `net`
        na, nb = (None, None) if ps else (ws.net_a[c], ws.net_b[c])
        nas, nbs = (ws.net_a_s[c], ws.net_b_s[c]) if ps else (None, None)
        d = [P["a.fh1"].desc(na, 0, 128, ws.fh_a, 0, EPI_RELU, in0s=nas)]
        if need_b:
            d.append(P["b.fh1"].desc(nb, 0, 128, ws.fh_b, 0, EPI_RELU, in0s=nbs))
        if mask_a:
            d.append(P["a.m0"].desc(na, 0, 128, ws.mh_a, 0, EPI_RELU, in0s=nas))
        if mask_b and need_b:
            d.append(P["b.m0"].desc(nb, 0, 128, ws.mh_b, 0, EPI_RELU, in0s=nbs))
        conv(d)
        # FlowHead.conv2 (256 -> 2) + coords1 += delta_flow in one wave-per-pixel kernel
        d = []
        if mask_a:
            d.append(P["a.m2"].desc(ws.mh_a, 0, 256, ws.mask_a, 0, EPI_LINEAR, scale=0.25))
        if mask_b and need_b:
            d.append(P["b.m2"].desc(ws.mh_b, 0, 256, ws.mask_b, 0, EPI_LINEAR, scale=0.25))
        # (both FlowHead tails as one launch on the calling stream, pf_flow_head_out_pair, measured 128.8 / 129.4 against 130.8 /
        # 130.8 pairs/s in round 3; the entry point stays in the library, the engine switch is gone)
        if self.forks & 8 and (need_b or d):
            # three independent tails of the heads: flow_out A | flow_out B | mask convs
            main = torch.cuda.current_stream()
            s1, s2 = self.side[0], self.side[1]
            tail_first = bool(self.order & 2)
            if tail_first:              # branch A's tail stays on the calling stream's queue: enqueue it first
                heads_done = torch.cuda.Event()
                heads_done.record(main)
                lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
            if need_b:
                if tail_first:
                    s1.wait_event(heads_done)
                else:
                    s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    lib.flow_head_out(ws.fh_b, 256, P["b.fh2w"], P["b.fh2b"], ws.c1b, ws.delta_b)
                    if defer_b_join:
                        self._b_pending = torch.cuda.Event()
                        self._b_pending.record(s1)
            if d:
                if tail_first:
                    s2.wait_event(heads_done)
                else:
                    s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    lib.conv2d(d, B, H8, W8, like)
            if not tail_first:
                lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
            if need_b and not defer_b_join:
                main.wait_stream(s1)
            if d:
                main.wait_stream(s2)
        else:
            lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
            if need_b:
                lib.flow_head_out(ws.fh_b, 256, P["b.fh2w"], P["b.fh2b"], ws.c1b, ws.delta_b)
            if d:
                conv(d)

        return c

    def upsample(self, ws: Workspace, branch: str, out: torch.Tensor):
        """upsample_flow (core/prior_raft.py:58-67) for branch 'a' or 'b' into out [B,2,H,W]."""
        if branch == "a":
            self.lib.upsample_flow(ws.c1a, ws.mask_a, out)
        else:
            self.lib.upsample_flow(ws.c1b, ws.mask_b, out)
        return out


# ------------------------------------------------------------------------------------------
# Encoders (BasicEncoder, core/extractor.py:98-158) on the HIP kernels
# ------------------------------------------------------------------------------------------
class EncoderPlan:
    """fnet (InstanceNorm) or cnet (BatchNorm, eval) as a launch plan.

    conv1 7x7/2 (3->64): in bf16x3 mode the image is 2x2 space-to-depth'd (12 channels) and the
    stem runs on the halo kernel as the equivalent 4x4 stride-1 conv (5x faster than the small-Cin
    kernel: 351 -> ~70 us per launch); in exact-fp32 mode it runs on the small-Cin fp32 MFMA kernel
    straight from the NCHW image.  Every 3x3 stride-1 conv runs on the bf16x3 halo kernel; the stride-2 3x3 / 1x1 convs
    and the final 1x1 on the generic bf16x3 kernel.  A conv writes its RAW output (+bias); the
    following norm is never applied as a pass of its own: it is folded, together with the ReLU,
    into the next conv's input load (`in_scale/in_shift`) or into the residual-tail kernel
    (`pf_norm_act`) that materialises a block's output.  InstanceNorm statistics come from
    `pf_channel_stats`; BatchNorm (always eval, core/prior_raft.py:43-48) is a constant affine.
    """

    def __init__(self, lib: PfLib, enc, precision: int):
        self.lib = lib
        self.kind = enc.norm_fn
        self.precision = precision
        dev = enc.conv1.weight.device
        self.dev = dev
        self.stem = DirectConv(enc.conv1)                      # [49][3][64]  (exact-fp32 mode)
        # bf16x3 mode: the stem as a 4x4 stride-1 conv over the space-to-depth image (halo kernel)
        self.stem_s2d = None
        # round 4: the stem straight from the NCHW image (pf_enc_stem, K = 176 instead of 512 and no space-to-depth pass);
        # (exact-fp32 mode keeps the space-to-depth form on the generic kernel)
        self.stem_direct = precision == PREC_BF16X3
        if precision == PREC_BF16X3:
            wp, bp = pack_mfma(stem_s2d_weight(enc.conv1.weight), enc.conv1.bias)
            self.stem_s2d = Conv(wp, bp, 4, 4, 12, 64, precision)
            self.stem_w7 = pack_stem7x7(enc.conv1.weight)
            self.stem_b7 = enc.conv1.bias.detach().float().contiguous()
        self.blocks = []
        for layer, stride in ((enc.layer1, 1), (enc.layer2, 2), (enc.layer3, 2)):
            for i, blk in enumerate(layer):
                st = stride if i == 0 else 1
                item = {"stride": st, "c1": Conv.of(blk.conv1, precision), "c2": Conv.of(blk.conv2, precision),
                        "cin": blk.conv1.weight.shape[1], "cout": blk.conv1.weight.shape[0],
                        "n1": blk.norm1, "n2": blk.norm2}
                if st != 1:
                    item["ds"] = Conv.of(blk.downsample[0], precision)
                    item["n3"] = blk.norm3
                self.blocks.append(item)
        self.norm1 = enc.norm1
        self.final = Conv.of(enc.conv2, precision)             # 1x1 128 -> 256
        self._bn_cache: Dict[int, tuple] = {}
        self._bufs = None                           # the buffers of the shape being run
        self._bufs_by_key: Dict[tuple, dict] = {}   # every resident shape's (captured HIP graphs hold pointers into them)
        # cnet in bf16x3 mode: every BatchNorm (eval) is folded into the convolution in front of it, ReLU and the residual add
        # move into the conv epilogues (PF_EPI_RELU / PF_EPI_RELU_RES), and the stride-1 3x3 convs read split twins through the
        # all-DMA kernel -- no normalisation pass, no statistics kernel, no input affine.  PRIORFLOW_FOLD_BN=0: the unfolded plan.
        self.fold = (self.kind == "batch" and precision == PREC_BF16X3 and os.environ.get("PRIORFLOW_FOLD_BN", "1") != "0"
                     and os.environ.get("PRIORFLOW_PRESPLIT", "1") != "0")
        if self.fold:
            self.f_stem = Conv.folded(enc.conv1, enc.norm1, precision, weight=stem_s2d_weight(enc.conv1.weight))
            sc = (enc.norm1.weight / torch.sqrt(enc.norm1.running_var + enc.norm1.eps)).detach().float()
            self.f_stem_w7 = pack_stem7x7(enc.conv1.weight.detach().float() * sc.view(-1, 1, 1, 1))      # Conv.folded's arithmetic
            self.f_stem_b7 = self.f_stem.b[:64].contiguous()
            self.f_blocks = []
            for blk_mod, item in zip([b for layer in (enc.layer1, enc.layer2, enc.layer3) for b in layer], self.blocks):
                f = {"stride": item["stride"], "cin": item["cin"], "cout": item["cout"],
                     "c1": Conv.folded(blk_mod.conv1, blk_mod.norm1, precision), "c2": Conv.folded(blk_mod.conv2, blk_mod.norm2, precision)}
                if item["stride"] != 1:
                    f["ds"] = Conv.folded(blk_mod.downsample[0], blk_mod.norm3, precision)
                self.f_blocks.append(f)

    # BatchNorm(eval) -> constant per-channel affine, replicated per image
    def _bn_affine(self, bn, Bn):
        key = (id(bn), Bn)
        if key not in self._bn_cache:
            with torch.no_grad():
                sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
                sh = (bn.bias - bn.running_mean * sc).float()
                self._bn_cache[key] = (sc[None].repeat(Bn, 1).contiguous(), sh[None].repeat(Bn, 1).contiguous())
        return self._bn_cache[key]

    def _conv_norm(self, cv, norm, x, cin, y, Bn, h, w, cout, slot, **kw):
        """Launch conv `cv` (x -> raw y) and return the (scale, shift) of the norm that follows it.
        InstanceNorm statistics ride in the conv's epilogue when the launch runs on a halo tile
        (per-tile fp64 partials + pf_channel_stats_final); otherwise a separate pass reads y."""
        lib, bufs = self.lib, self._bufs
        d = cv.desc(x, 0, cin, y, 0, EPI_LINEAR, **kw)
        fused = False
        if self.kind != "batch" and self.precision == PREC_BF16X3:
            nblk = lib.conv2d_stats_blocks([d], Bn, h, w)        # per-tile fp64 partials of the kernel this launch takes (0: none)
            if nblk > 0 and Bn * nblk * cout * 2 <= bufs["part"].numel():
                d.stats_out = bufs["part"].data_ptr()
                fused = True
        lib.conv2d([d], Bn, h, w, x)
        if not fused:
            return self._affine(norm, y, Bn, h * w, cout, slot)
        sc, sh = bufs["sc"][slot][: Bn * cout].view(Bn, cout), bufs["sh"][slot][: Bn * cout].view(Bn, cout)
        lib.channel_stats_final(bufs["part"], Bn, h * w, cout, nblk, sc, sh)
        return sc, sh

    def _affine(self, norm, y, Bn, Np, C, slot):
        """(scale, shift) [Bn][C] that normalises the raw conv output y."""
        if self.kind == "batch":
            return self._bn_affine(norm, Bn)
        sc, sh = self._bufs["sc"][slot][: Bn * C].view(Bn, C), self._bufs["sh"][slot][: Bn * C].view(Bn, C)
        self.lib.channel_stats(y, Bn, Np, C, sc, sh, self._bufs["part"], 128)
        return sc, sh

    def _alloc(self, Bn, H, W):
        key = (Bn, H, W)
        if self._select(key):
            return
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        b = {"key": key}
        # three rotating activation buffers per resolution (x / y1 / y2 / out rotate through them)
        for lvl, (h, w, c) in enumerate(((H // 2, W // 2, 64), (H // 4, W // 4, 96), (H // 8, W // 8, 128))):
            b[f"act{lvl}"] = [z(Bn * h * w, c) for _ in range(4)]
        b["s2d"] = z(Bn * (H // 2) * (W // 2), 12)
        b["sc"] = [z(Bn * 128) for _ in range(3)]
        b["sh"] = [z(Bn * 128) for _ in range(3)]
        # [image][<= 1024 tiles or 128 chunks][C][2] (pf_conv2d_stats_blocks of the launch; a launch whose partials do not fit takes the separate pass)
        b["part"] = z(Bn * 1024 * 128 * 2, dt=torch.float64)
        self._bufs = self._bufs_by_key[key] = b

    def _select(self, key) -> bool:
        """Makes the buffers of `key` current when they exist.  One set per resident shape: PriOr_RAFT keeps several workspaces
        and the graphs captured on them, and a graph holds pointers into these buffers too (round 5: with a single cached set a
        batch-32 forward freed the buffers the B = 1 graph replays into)."""
        b = self._bufs_by_key.get(key)
        if b is not None:
            self._bufs = b
        return b is not None

    def release(self, H: int, W: int, batches) -> None:
        """Drops the buffers of the shapes (Bn in `batches`, H, W): their workspace (and its graphs) left the model's cache."""
        for key in [k for k in self._bufs_by_key if k[-2:] == (H, W) and k[-3] in batches]:
            if self._bufs is self._bufs_by_key[key]:
                self._bufs = None
            del self._bufs_by_key[key]

    def run(self, images: torch.Tensor, out: torch.Tensor, epilogue: int, aux: Optional[torch.Tensor] = None,
            outs: Optional[torch.Tensor] = None, auxs: Optional[torch.Tensor] = None):
        """images: NCHW [Bn,3,H,W] in [-1,1]; out: channel-last [Bn*N, ld] (fnet: 256 features;
        cnet with EPI_TANH_RELU: out = net [.,128], aux = x buffer [.,256] whose first 128 columns get inp).
        outs / auxs: optional split twins of out / aux (with auxs given, aux may be None)."""
        lib = self.lib
        Bn, _, H, W = images.shape
        if self.fold:
            return self._run_folded(images, out, epilogue, aux, outs, auxs)
        self._alloc(Bn, H, W)
        bufs = self._bufs
        h, w = H // 2, W // 2
        a0 = bufs["act0"]
        # stem: raw conv -> a0[0]; x0 = relu(norm1(.)) materialised -> a0[1]
        # bf16x3: x0 = relu(norm1(stem)) is never materialised -- the first block's conv1 applies it while staging and its tail
        # applies it to the skip input (one 2 x 134 MB pass less per 4 images); the stem's statistics live in slot 2 until then
        stem_folded = self.precision == PREC_BF16X3 and self.kind != "batch" and os.environ.get("PRIORFLOW_FOLD_STEM", "1") != "0"
        slot0 = 2 if stem_folded else 0
        if self.stem_direct and self.kind != "batch" and Bn * ((h + 7) // 8) * ((w + 31) // 32) * 64 * 2 <= bufs["part"].numel():
            nblk = ((h + 7) // 8) * ((w + 31) // 32)
            lib.enc_stem(images, self.stem_w7, self.stem_b7, out=a0[0], stats=bufs["part"])
            sc, sh = bufs["sc"][slot0][: Bn * 64].view(Bn, 64), bufs["sh"][slot0][: Bn * 64].view(Bn, 64)
            lib.channel_stats_final(bufs["part"], Bn, h * w, 64, nblk, sc, sh)
        elif self.stem_direct:
            lib.enc_stem(images, self.stem_w7, self.stem_b7, out=a0[0])
            sc, sh = self._affine(self.norm1, a0[0], Bn, h * w, 64, slot0)
        elif self.stem_s2d is not None:
            lib.space_to_depth2(images, bufs["s2d"])
            sc, sh = self._conv_norm(self.stem_s2d, self.norm1, bufs["s2d"], 12, a0[0], Bn, h, w, 64, slot0)
        else:
            lib.conv2d_small(images, True, 0, 3, self.stem.w, self.stem.b, a0[0], 0, 64, 7, 7, 2, False, Bn, h, w)
            sc, sh = self._affine(self.norm1, a0[0], Bn, h * w, 64, slot0)
        if stem_folded:
            s0, t0 = sc, sh
            x = a0[0]
        else:
            lib.norm_act(a0[0], sc, sh, a0[1], Bn, h * w, 64)
            x = a0[1]
        lvl = 0
        for bi, blk in enumerate(self.blocks):
            cin, cout, st = blk["cin"], blk["cout"], blk["stride"]
            if st != 1:
                lvl += 1
                h, w = h // 2, w // 2
            acts = bufs[f"act{lvl}"]
            free = [t for t in acts if t.data_ptr() != x.data_ptr()]
            y1, y2, o = free[0], free[1], free[2]
            Np = h * w
            # conv1 (input x is a materialised activation: no affine -- except the folded stem in front of the first block)
            first = stem_folded and bi == 0
            kw1 = dict(in_scale=s0, in_shift=t0, in_relu=True) if first else {}
            s1, t1 = self._conv_norm(blk["c1"], blk["n1"], x, cin, y1, Bn, h, w, cout, 0, stride=st, **kw1)
            if self.precision == PREC_BF16X3:
                # conv2 consumes relu(norm1(y1)) folded into its load
                s2, t2 = self._conv_norm(blk["c2"], blk["n2"], y1, cout, y2, Bn, h, w, cout, 1,
                                         in_scale=s1, in_shift=t1, in_relu=True)
            else:
                # exact-fp32 mode (generic MFMA kernel: no input affine): relu(norm1(y1)) is materialised into `o`
                lib.norm_act(y1, s1, t1, o, Bn, Np, cout)
                s2, t2 = self._conv_norm(blk["c2"], blk["n2"], o, cout, y2, Bn, h, w, cout, 1)
            if st != 1:
                # shortcut: norm3(conv1x1/2(x)); reuse y1 (conv2 has consumed it) for the raw shortcut
                s3, t3 = self._conv_norm(blk["ds"], blk["n3"], x, cin, y1, Bn, h, w, cout, 2, stride=st)
                lib.norm_act(y2, s2, t2, o, Bn, Np, cout, res=y1, rs=s3, rt=t3)
            elif first:
                lib.norm_act(y2, s2, t2, o, Bn, Np, cout, res=x, rs=s0, rt=t0, res_relu=True)
            else:
                lib.norm_act(y2, s2, t2, o, Bn, Np, cout, res=x)
            x = o
        # final 1x1 conv 128 -> 256 (core/extractor.py:151)
        d = self.final.desc(x, 0, 128, out, 0, epilogue, aux=aux, outs=outs, auxs=auxs)
        lib.conv2d([d], Bn, h, w, x)

    # ---- cnet with folded BatchNorm (bf16x3) ---------------------------------------------------------------------------
    def _alloc_folded(self, Bn, H, W):
        key = ("fold", Bn, H, W)
        if self._select(key):
            return
        z = lambda *s_: torch.zeros(*s_, dtype=torch.float32, device=self.dev)  # noqa: E731
        b = {"key": key, "s2d": z(Bn * (H // 2) * (W // 2), 12)}
        for lvl, (h, w, c) in enumerate(((H // 2, W // 2, 64), (H // 4, W // 4, 96), (H // 8, W // 8, 128))):
            rows = Bn * h * w
            # block inputs / outputs exist in both forms (fp32: residual operand, stride-2 convs, final 1x1; twin: the 3x3 convs);
            # the intermediate y1 as a twin only; r = the folded downsample branch of a stride-2 block
            b[f"x{lvl}"] = [z(rows, c), z(rows, c)]
            # a level's twins (xs: block outputs, y: conv1's output) or its fp32 intermediate (yr) are allocated by _run_folded
            # when it knows which form the level's 3x3 convolutions take (level 0's twins were allocated and never used)
            b[f"xs{lvl}"] = None
            b[f"y{lvl}"] = None
            b[f"yr{lvl}"] = None
            b[f"r{lvl}"] = z(rows, c)
        self._bufs = self._bufs_by_key[key] = b

    def nbytes(self) -> int:
        """Bytes of every resident activation set of this plan (PriOr_RAFT._workspace counts them against its byte cap)."""
        def size(v):
            if isinstance(v, torch.Tensor):
                return v.numel() * v.element_size()
            if isinstance(v, (list, tuple)):
                return sum(size(t) for t in v)
            return 0
        return sum(size(v) for b in self._bufs_by_key.values() for v in b.values())

    def _run_folded(self, images, out, epilogue, aux, outs, auxs):
        lib = self.lib
        Bn, _, H, W = images.shape
        self._alloc_folded(Bn, H, W)
        bufs = self._bufs
        h, w = H // 2, W // 2
        x = bufs["x0"][0]
        # Layer 1 (3x3 64 -> 64 at 1/2 resolution) on fp32 rows when pf_conv2d hands those launches to the weights-stationary
        # kernel (pf_conv2d_tile 6, round 5): no twins on this level at all -- the stem writes 67 MB less, conv1's output and the
        # residual tail stay fp32 rows, the stride-2 convs of layer 2 read fp32 anyway.  Same products, same order: same bits.
        f0 = self.f_blocks[0]
        l0_rows = (f0["stride"] == 1 and
                   lib.conv2d_tile([f0["c1"].desc(x, 0, f0["cin"], bufs["r0"], 0, EPI_RELU)], Bn, h, w) == 6 and
                   lib.conv2d_tile([f0["c2"].desc(bufs["r0"], 0, f0["cout"], x, 0, EPI_RELU_RES, h=x)], Bn, h, w) == 6)
        if not l0_rows and bufs["xs0"] is None:
            rows0 = Bn * h * w
            bufs["xs0"] = [split_twin(rows0, 64, self.dev), split_twin(rows0, 64, self.dev)]
            bufs["y0"] = split_twin(rows0, 64, self.dev)
        xs = None if l0_rows else bufs["xs0"][0]
        if self.stem_direct:
            lib.enc_stem(images, self.f_stem_w7, self.f_stem_b7, out=x, out_split=None if l0_rows else xs, relu=True)
        else:
            lib.space_to_depth2(images, bufs["s2d"])
            lib.conv2d([self.f_stem.desc(bufs["s2d"], 0, 12, x, 0, EPI_RELU, outs=None if l0_rows else xs)], Bn, h, w, x)
        lvl, cur = 0, 0
        rows_lvl = l0_rows                                  # this level's 3x3 convolutions read and write fp32 rows only
        for f in self.f_blocks:
            cin, cout, st = f["cin"], f["cout"], f["stride"]
            if st == 1 and lvl == 0 and l0_rows:
                yf, o = bufs["r0"], bufs["x0"][cur ^ 1]
                lib.conv2d([f["c1"].desc(x, 0, cin, yf, 0, EPI_RELU)], Bn, h, w, x)
                lib.conv2d([f["c2"].desc(yf, 0, cout, o, 0, EPI_RELU_RES, h=x)], Bn, h, w, o)
                x, cur = o, cur ^ 1
                continue
            if st != 1:
                x_in = x                                    # fp32 input of the two stride-2 convs
                lvl += 1
                h, w = h // 2, w // 2
                cur = 0
                r, o = bufs[f"r{lvl}"], bufs[f"x{lvl}"][0]
                # Layer 2 (96 channels) on fp32 rows when pf_conv2d gives its 3x3 convolutions the halo kernel's 256 px x 96 channel
                # tile (pf_conv2d_tile 8, round 6): the all-DMA kernel that twins would select has 64-channel tiles -- one and a
                # half of them, a quarter of the MFMAs on padding: 48 against 35 us per launch on cnet's two images
                # (profiles/r6_ab_96_channel_tiles.txt).  Same products, same order: same bits.
                rows_lvl = lib.conv2d_tile([f["c2"].desc(o, 0, cout, o, 0, EPI_RELU_RES, h=r)], Bn, h, w) == 8
                rows = Bn * h * w
                if rows_lvl and bufs[f"yr{lvl}"] is None:
                    bufs[f"yr{lvl}"] = torch.zeros(rows, cout, dtype=torch.float32, device=self.dev)
                if not rows_lvl and bufs[f"y{lvl}"] is None:
                    bufs[f"xs{lvl}"] = [split_twin(rows, cout, self.dev), split_twin(rows, cout, self.dev)]
                    bufs[f"y{lvl}"] = split_twin(rows, cout, self.dev)
                if rows_lvl:
                    lib.conv2d([f["c1"].desc(x_in, 0, cin, bufs[f"yr{lvl}"], 0, EPI_RELU, stride=st)], Bn, h, w, x_in)
                else:
                    lib.conv2d([f["c1"].desc(x_in, 0, cin, None, 0, EPI_RELU, stride=st, outs=bufs[f"y{lvl}"])], Bn, h, w, x_in)
                lib.conv2d([f["ds"].desc(x_in, 0, cin, r, 0, EPI_LINEAR, stride=st)], Bn, h, w, x_in)
                res = r
            else:
                o = bufs[f"x{lvl}"][cur ^ 1]
                if rows_lvl:
                    lib.conv2d([f["c1"].desc(x, 0, cin, bufs[f"yr{lvl}"], 0, EPI_RELU)], Bn, h, w, x)
                else:
                    lib.conv2d([f["c1"].desc(None, 0, cin, None, 0, EPI_RELU, in0s=xs, outs=bufs[f"y{lvl}"])], Bn, h, w, x)
                res = x
                cur ^= 1
            # conv2 + folded norm2 + ReLU, residual add + ReLU in the epilogue (core/extractor.py:44-47)
            if rows_lvl:
                lib.conv2d([f["c2"].desc(bufs[f"yr{lvl}"], 0, cout, o, 0, EPI_RELU_RES, h=res)], Bn, h, w, o)
                x, xs = o, None
            else:
                os_ = bufs[f"xs{lvl}"][cur]
                lib.conv2d([f["c2"].desc(None, 0, cout, o, 0, EPI_RELU_RES, in0s=bufs[f"y{lvl}"], h=res, outs=os_)], Bn, h, w, o)
                x, xs = o, os_
        d = self.final.desc(x, 0, 128, out, 0, epilogue, aux=aux, outs=outs, auxs=auxs)
        lib.conv2d([d], Bn, h, w, x)
