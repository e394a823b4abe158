"""ctypes binding of ``libpriorflow_hip.so`` (C-ABI declared in ``include/priorflow_hip.h``).

``load()`` fails loudly when the HIP library has not been built: there is no CPU or
PyTorch fallback anywhere in the product path.  torch is plumbing only (device memory,
streams): every wrapper passes ``tensor.data_ptr()`` and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRIORFLOW_LIB") or os.path.join(_HERE, "lib", "libpriorflow_hip.so")

EPI_LINEAR, EPI_RELU, EPI_GRU_ZR, EPI_GRU_Q, EPI_TANH_RELU, EPI_RELU_RES, EPI_MASK, EPI_ADD = 0, 1, 2, 3, 4, 5, 6, 7
PREC_F32, PREC_BF16X3 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2

_fp = C.c_void_p
_i = C.c_int


def _norm_bwd_blocks(Np: int) -> int:
    """Pixel chunks of pf_norm_bwd's two-stage reduction: >= 16 pixels per partial sum, at most 1024 chunks per image (the
    partial pass runs one thread per chunk and channel; the final pass sums the chunks of one (image, channel) with one wave)."""
    return max(1, min(Np // 16, 1024))


class UnpackJob(C.Structure):
    """Mirror of ``pf_unpack_job`` (include/priorflow_hip.h)."""
    _fields_ = [("dw", _fp), ("db", _fp), ("gw", _fp), ("gb", _fp),
                ("cout", _i), ("cin", _i), ("taps", _i), ("cin_pad", _i), ("o_off", _i), ("scale", C.c_float)]


class PackJob(C.Structure):
    """Mirror of ``pf_pack_job`` (include/priorflow_hip.h)."""
    _fields_ = [("w0", _fp), ("w1", _fp), ("b0", _fp), ("b1", _fp), ("dst_w", _fp), ("dst_b", _fp),
                ("cout0", _i), ("cout1", _i), ("cin", _i), ("kh", _i), ("kw", _i), ("mode", _i), ("cin_rot", _i),
                ("cout_pad", _i), ("cin_pad", _i)]


class ConvDesc(C.Structure):
    """Mirror of ``pf_conv_desc`` (include/priorflow_hip.h)."""
    _fields_ = [
        ("in0", _fp), ("ld0", _i), ("off0", _i), ("c0", _i),
        ("in1", _fp), ("ld1", _i), ("off1", _i), ("c1", _i),
        ("weight", _fp), ("bias", _fp),
        ("out", _fp), ("ld_out", _i), ("off_out", _i), ("cout", _i),
        ("kh", _i), ("kw", _i),
        ("epilogue", _i), ("scale", C.c_float),
        ("h", _fp), ("ld_h", _i),
        ("z", _fp), ("ld_z", _i),
        ("aux_out", _fp), ("ld_aux", _i),
        ("precision", _i), ("stride", _i),
        ("in_scale", _fp), ("in_shift", _fp), ("in_relu", _i),
        ("stats_out", _fp),
        ("in0_split", _fp), ("lds0", _i),
        ("in1_split", _fp), ("lds1", _i),
        ("out_split", _fp), ("lds_out", _i),
        ("aux_split", _fp), ("lds_aux", _i),
        ("zeros", _fp), ("zeros_bytes", _i),
        ("pre", _fp), ("ld_pre", _i), ("off_pre", _i),
        ("save_gates", _i),
        ("co_groups", _i),
    ]


class CombineConvDesc(C.Structure):
    """Mirror of ``pf_combine_conv_desc`` (include/priorflow_hip.h)."""
    _fields_ = [("own", _fp), ("raw", _fp), ("ld", _i), ("g_back", _fp), ("weight", _fp), ("bias", _fp),
                ("out", _fp), ("ld_out", _i), ("off_out", _i), ("cout", _i), ("out_split", _fp), ("lds_out", _i)]


class DirectDesc(C.Structure):
    """Mirror of ``pf_direct_desc`` (include/priorflow_hip.h)."""
    _fields_ = [("in_", _fp), ("ld_in", _i), ("off_in", _i), ("weight", _fp), ("bias", _fp),
                ("out", _fp), ("ld_out", _i), ("off_out", _i), ("out_split", _fp), ("lds_out", _i)]


_SIGNATURES = {
    "pf_sample_grid": [_fp, _i, _i, C.POINTER(C.c_float), _fp],
    "pf_img_rotate": [_fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_normalise_images": [_fp, _fp, _fp, _fp, _fp, C.c_long, _fp],
    "pf_prepare_images": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp],
    "pf_flow_prep": [_fp, _fp, _fp, _i, _i, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_flo_rotate": [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_corr_pyramid": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_corr_pyramid_bf16x3": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_split_bf16": [_fp, _fp, C.c_long, _i, _fp],
    "pf_enc_stem": [_fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _fp],
    "pf_debug_dirty_lds": [C.c_uint, _fp],
    "pf_gru_dx_finish": [_fp, _i, _fp, _i, _fp, _i, _fp, _i, _fp, _i, C.c_long, _i, _i, _fp],
    "pf_pack_conv_weights": [_fp, _i, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _i, _i, _fp],
    "pf_unpack_wgrads": [C.POINTER(UnpackJob), _i, _fp],
    "pf_pack_conv_weights_batch": [C.POINTER(PackJob), _i, _fp],
    "pf_add_relu": [_fp, _fp, _fp, C.c_long, _fp],
    "pf_relu_mask": [_fp, _fp, _fp, C.c_long, _fp],
    "pf_bn_frozen_fwd": [_fp, _fp, _fp, _fp, _fp, C.c_float, _i, _fp, C.c_long, _i, _fp],
    "pf_bn_frozen_bwd": [_fp, _fp, _fp, _fp, _fp, _fp, C.c_float, _i, _fp, _i, _fp, _fp, _fp, _i, C.c_long, _i, _fp],
    "pf_dccl_lookup": [_fp] * 12 + [_i, _i, _i, _i, _fp],
    "pf_dccl_lookup_il": [_fp] * 13 + [_i, _i, _i, _i, _fp],
    "pf_dccl_combine": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_warp_gcorr": [_fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _i, _i, _fp],
    "pf_conf_stem": [_fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp],
    "pf_motion_prep": [_fp] * 9 + [_i, _i, _fp, _i, _i, _fp, _i, _fp, _i, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_conv2d": [C.POINTER(ConvDesc), _i, _i, _i, _i, _fp],
    "pf_dccl_combine_conv1x1": [C.POINTER(CombineConvDesc), _i, _i, _i, _i, _fp],
    "pf_conv2d_tile": [C.POINTER(ConvDesc), _i, _i, _i, _i],
    "pf_conv2d_stats_blocks": [C.POINTER(ConvDesc), _i, _i, _i, _i],
    "pf_conv2d_roles": [C.POINTER(ConvDesc), _i, _i, _i, _i],
    "pf_conv2d_direct": [_fp, _i, _i, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pf_conv2d_direct_group": [C.POINTER(DirectDesc), _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pf_conv2d_small": [_fp, _i, _i, _i, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pf_channel_stats": [_fp, _i, _i, _i, C.c_float, _fp, _fp, _fp, _i, _fp],
    "pf_channel_stats_final": [_fp, _i, _i, _i, _i, C.c_float, _fp, _fp, _fp],
    "pf_norm_act": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _fp],
    "pf_flow_head_out": [_fp, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_coords_add": [_fp, _fp, _i, _i, _i, _i, _fp],
    "pf_coords_add_to": [_fp, _fp, _i, _fp, _i, _i, _i, _fp],
    "pf_upsample_flow": [_fp, _fp, _i, _fp, _i, _i, _i, _fp],
    "pf_to_channel_last": [_fp, _i, _i, _i, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_space_to_depth2": [_fp, _i, _fp, _i, _i, _i, _i, _fp],
    "pf_conv2d_wgrad_small": [_fp, _i, _i, _i, _i, _fp, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp],
    "pf_conv2d_wgrad_small_ws": [_fp, _i, _i, _i, _i, _fp, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, C.c_long, _fp],
    "pf_conv2d_wgrad_small_ws_floats": [_i, _i, _i, _i, _i, _i, _i],
    "pf_to_nchw": [_fp, _i, _i, _i, _fp, _i, _i, _fp],
    "pf_warp_gcorr_bwd": [_fp, _fp, _fp, _i, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_upsample_flow_bwd": [_fp, _fp, _i, _fp, _fp, _i, _fp, _i, _i, _i, _fp],
    "pf_pyramid_bwd": [_fp, _fp, _fp, _fp, _i, _i, _i, _fp],
    "pf_norm_bwd": [_fp, _fp, _fp, _fp, _i, _i, _fp, _i, _fp, _fp, _i, _i, _i, _fp],
    "pf_gru_q_bwd": [_fp, _i] * 7 + [C.c_long, _i, _fp],
    "pf_gru_zr_bwd": [_fp, _i] * 7 + [C.c_long, _i, _fp],
    "pf_dccl_combine_bwd": [_fp, _i, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_dccl_lookup_bwd": [_fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_conv2d_wgrad": [_fp, _i, _i, _i, _fp, _i, _i, _i, _fp, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _fp],
    "pf_seq_loss": [_fp, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _i, _i, _i, _fp],
    "pf_seq_loss_batch": [_fp, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_sum_squares": [_fp, C.c_long, _fp, _i, _fp],
    "pf_adamw_step": [_fp, _fp, _fp, _fp, C.c_long, C.c_double, C.c_float, C.c_float, C.c_float, C.c_double, _i,
                      C.c_float, _fp],
    "pf_adamw_step_dev": [_fp, _fp, _fp, _fp, C.c_long, C.c_float, C.c_float, C.c_float, _fp, _fp],
    "pf_flow_metrics": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pf_region_sums": [_fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _fp],
}
EXPORTS = ["pf_version"] + list(_SIGNATURES)


class PfError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _twin(t: Optional[torch.Tensor]):
    """(pointer, chunks per row) of a split twin [rows, chunks, 2, 32] bfloat16 (``engine.split_twin``), or (None, 0)."""
    if t is None:
        return None, 0
    if t.dtype != torch.bfloat16 or t.dim() != 4 or tuple(t.shape[2:]) != (2, 32) or not t.is_contiguous():
        raise PfError("a split twin is a contiguous bfloat16 tensor [rows, chunks, 2, 32]")
    return C.c_void_p(t.data_ptr()), t.shape[1]


class PfLib:
    """Typed wrappers; one instance per loaded shared object."""

    def __init__(self, path: str, require_cuda: bool = True, optional: Sequence[str] = ()):
        if not os.path.exists(path):
            raise PfError(
                f"HIP library not found: {path}\n"
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        if require_cuda:
            # bind against the HIP runtime torch already loaded (same SONAME libamdhip64.so.7)
            torch_hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(torch_hip):
                C.CDLL(torch_hip, mode=C.RTLD_GLOBAL)
        self.path = path
        self.require_cuda = require_cuda
        self._dll = C.CDLL(path)
        self._dll.pf_version.restype = C.c_char_p
        self.missing = []
        self.pack_queue = None
        for name, args in _SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                if name in optional:
                    self.missing.append(name)
                    continue
                raise PfError(f"{path} does not export {name}")
            fn.argtypes = args
            fn.restype = C.c_long if name.endswith("_ws_floats") else _i

    # ---- helpers -----------------------------------------------------------------------------
    def version(self) -> str:
        return self._dll.pf_version().decode()

    def _chk(self, *tensors: Optional[torch.Tensor]):
        for t in tensors:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise PfError(f"expected contiguous fp32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
            if self.require_cuda and not t.is_cuda:
                raise PfError("the HIP path needs tensors on a cuda (ROCm) device; there is no CPU fallback")

    def _chk_rows(self, *tensors: torch.Tensor):
        """2-D fp32 row views: unit stride along the channels, any leading dimension (column slices are fine)."""
        for t in tensors:
            if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
                raise PfError(f"expected a 2-D fp32 view with unit channel stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")
            if self.require_cuda and not t.is_cuda:
                raise PfError("the HIP path needs tensors on a cuda (ROCm) device; there is no CPU fallback")

    def _stream(self, t: torch.Tensor):
        if t.is_cuda:
            return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        return None

    @staticmethod
    def _rc(rc: int, what: str):
        if rc != 0:
            kind = "bad argument/shape" if rc < 0 else "hipError_t"
            raise PfError(f"{what} failed: {kind} {rc}")

    # ---- geometry ----------------------------------------------------------------------------
    def sample_grid(self, grid: torch.Tensor, R: torch.Tensor):
        self._chk(grid)
        H, W = grid.shape[-2:]
        r = (C.c_float * 9)(*[float(v) for v in R.reshape(-1).tolist()])
        self._rc(self._dll.pf_sample_grid(_ptr(grid), H, W, r, self._stream(grid)), "pf_sample_grid")
        return grid

    def img_rotate(self, img, grid, out):
        self._chk(img, grid, out)
        B, Cc, H, W = img.shape
        self._rc(self._dll.pf_img_rotate(_ptr(img), _ptr(grid), _ptr(out), B, Cc, H, W,
                                         self._stream(img)), "pf_img_rotate")
        return out

    def normalise_images(self, image1, image2, f1, f2, c1=None):
        """2 * (image / 255.0) - 1.0 of both images into the encoders' batches (image1 -> f1 and c1, image2 -> f2)."""
        self._chk(image1, image2, f1, f2, c1)
        n = image1.numel()
        if not (image2.numel() == f1.numel() == f2.numel() == n and (c1 is None or c1.numel() == n)):
            raise PfError("normalise_images: operand sizes differ")
        self._rc(self._dll.pf_normalise_images(_ptr(image1), _ptr(image2), _ptr(f1), _ptr(f2), _ptr(c1), n,
                                               self._stream(image1)), "pf_normalise_images")

    def prepare_images(self, image1, image2, grid, img_f, img_c=None):
        """Normalise + rotate into view B in one launch: img_f = [im1 | im2 | im1_B | im2_B], img_c = [im1 | im1_B] (optional)."""
        self._chk(image1, image2, grid, img_f, img_c)
        B, Cc, H, W = image1.shape
        if not (Cc == 3 and image2.shape == image1.shape and tuple(grid.shape) == (2, H, W) and
                tuple(img_f.shape) == (4 * B, 3, H, W) and (img_c is None or tuple(img_c.shape) == (2 * B, 3, H, W))):
            raise PfError("prepare_images: operand shapes do not fit")
        self._rc(self._dll.pf_prepare_images(_ptr(image1), _ptr(image2), _ptr(grid), _ptr(img_f), _ptr(img_c), B, H, W,
                                             self._stream(image1)), "pf_prepare_images")

    def flow_prep(self, coords1, flow_out=None, d0=None, d0_off=0, d1=None, d1_off=0):
        self._chk(coords1, flow_out, d0, d1)
        B, _, H, W = coords1.shape
        self._rc(self._dll.pf_flow_prep(
            _ptr(coords1), _ptr(flow_out),
            _ptr(d0), 0 if d0 is None else d0.shape[-1], d0_off,
            _ptr(d1), 0 if d1 is None else d1.shape[-1], d1_off,
            B, H, W, self._stream(coords1)), "pf_flow_prep")

    def flo_rotate(self, flow, g_w2c, g_c2w, out=None, d0=None, d0_off=0, d1=None, d1_off=0):
        self._chk(flow, g_w2c, g_c2w, out, d0, d1)
        B, _, H, W = flow.shape
        self._rc(self._dll.pf_flo_rotate(
            _ptr(flow), _ptr(g_w2c), _ptr(g_c2w), _ptr(out),
            _ptr(d0), 0 if d0 is None else d0.shape[-1], d0_off,
            _ptr(d1), 0 if d1 is None else d1.shape[-1], d1_off,
            B, H, W, self._stream(flow)), "pf_flo_rotate")
        return out

    def conf_stem(self, x, off_in, w1, b1, w2, b2, out, off_out, B, H8, W8, out_split=None):
        """relu(conv3x3 32->16(relu(conv3x3 8->32(x)))) in one launch (core/update.py:193-194); w*: [9*Cin][Cout].
        out_split: optional split twin of `out` written at the same channel offset (`out` may then be None)."""
        self._chk(x, w1, b1, w2, b2, out)
        sp, lds = _twin(out_split)
        self._rc(self._dll.pf_conf_stem(_ptr(x), x.shape[-1], off_in, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                                        _ptr(out), 0 if out is None else out.shape[-1], off_out, sp, lds,
                                        B, H8, W8, self._stream(x)), "pf_conf_stem")

    def motion_prep(self, c1a, c1b, g_w2c, g_c2w, f1a, f2a, flow4_a, flow2_b, conf, xa=None, xa_off=0, xb=None, xb_off=0,
                    xa_split=None, xb_split=None):
        """One launch for flow_prep x2 + flo_rotate + warp_gcorr x2 (core/prior_raft.py:171-182), bit-identical to them.
        xa_split / xb_split: optional split twins of the GRU input buffers (tails written at xa_off / xb_off)."""
        self._chk(c1a, c1b, g_w2c, g_c2w, f1a, f2a, flow4_a, flow2_b, conf, xa, xb)
        B, _, H, W = c1a.shape
        (sa, la), (sb, lb) = _twin(xa_split), _twin(xb_split)
        self._rc(self._dll.pf_motion_prep(
            _ptr(c1a), _ptr(c1b), _ptr(g_w2c), _ptr(g_c2w), _ptr(f1a), _ptr(f2a), _ptr(flow4_a), _ptr(flow2_b),
            _ptr(xa), 0 if xa is None else xa.shape[-1], xa_off, _ptr(xb), 0 if xb is None else xb.shape[-1], xb_off,
            sa, la, sb, lb, _ptr(conf), conf.shape[-1], B, H, W, f1a.shape[-1], self._stream(c1a)), "pf_motion_prep")

    # ---- correlation -------------------------------------------------------------------------
    def corr_pyramid(self, f1, f2, levels, B, H8, W8):
        """f1,f2: channel-last [B*N, C]; levels: 4 tensors [B*N, (H8>>i)*(W8>>i)]."""
        self._chk(f1, f2, *levels)
        self._rc(self._dll.pf_corr_pyramid(_ptr(f1), _ptr(f2), *[_ptr(l) for l in levels],
                                           B, H8, W8, f1.shape[-1], self._stream(f1)), "pf_corr_pyramid")

    def split_bf16(self, x, out):
        """x fp32 [rows, C] -> out bfloat16 [rows, C/32, 2, 32] (hi | lo halves per 32-channel chunk)."""
        self._chk(x)
        if out.dtype != torch.bfloat16 or out.numel() != 2 * x.numel() or not out.is_contiguous():
            raise PfError("split_bf16: out must be contiguous bfloat16 with 2x the elements of x")
        self._rc(self._dll.pf_split_bf16(_ptr(x), C.c_void_p(out.data_ptr()), x.shape[0], x.shape[-1],
                                         self._stream(x)), "pf_split_bf16")
        return out

    def pack_conv_weights(self, w0, b0=None, w1=None, b1=None, mode=0, cin_rot=0, cout_pad_to=128):
        """[Cout,Cin,KH,KW] weights (optionally two tensors concatenated on Cout) -> (bf16 split operand of pf_conv2d
        [Cout_pad, KH*KW, Cin_pad/32, 2, 32], bias [Cout_pad]); mode 1: the data-gradient convolution's operand
        (packed Cout = Cin of the forward conv rotated by cin_rot; packed Cin = forward Cout rounded up to 4, then 32)."""
        self._chk(w0, w1, b0, b1)
        cout0, cin, kh, kw = w0.shape
        cout1 = 0 if w1 is None else w1.shape[0]
        oc, ic = (cout0 + cout1, cin) if mode == 0 else (cin, cout0 + cout1)
        op = (oc + cout_pad_to - 1) // cout_pad_to * cout_pad_to
        cp = (ic + 31) // 32 * 32
        dst_w = torch.empty(op, kh * kw, cp // 32, 2, 32, dtype=torch.bfloat16, device=w0.device)
        dst_b = torch.empty(op, dtype=torch.float32, device=w0.device)
        if self.pack_queue is not None:         # inside ``batched_packs()``: launched together when the block ends
            self.pack_queue.append((w0, w1, b0, b1, dst_w, dst_b, cout0, cout1, cin, kh, kw, mode, cin_rot, op, cp))
            return dst_w, dst_b
        self._rc(self._dll.pf_pack_conv_weights(_ptr(w0), cout0, _ptr(w1), cout1, _ptr(b0), _ptr(b1), cin, kh, kw, mode, cin_rot,
                                                C.c_void_p(dst_w.data_ptr()), _ptr(dst_b), op, cp, self._stream(w0)),
                 "pf_pack_conv_weights")
        return dst_w, dst_b

    def batched_packs(self):
        """Context: every ``pack_conv_weights`` inside returns its (still unwritten) destination tensors at once and the packs run
        as ceil(n / 16) launches of pf_pack_conv_weights_batch when the block ends -- before anything may read them."""
        lib = self

        class _Batch:
            def __enter__(self):
                self.outer = lib.pack_queue
                lib.pack_queue = []
                return self

            def __exit__(self, *exc):
                jobs, lib.pack_queue = lib.pack_queue, self.outer
                if exc[0] is None and jobs:
                    arr = (PackJob * len(jobs))()
                    for q, (w0, w1, b0, b1, dw, db, c0, c1, cin, kh, kw, mode, rot, op, cp) in zip(arr, jobs):
                        q.w0, q.w1, q.b0, q.b1 = _ptr(w0), _ptr(w1), _ptr(b0), _ptr(b1)
                        q.dst_w, q.dst_b = C.c_void_p(dw.data_ptr()), _ptr(db)
                        q.cout0, q.cout1, q.cin, q.kh, q.kw, q.mode, q.cin_rot, q.cout_pad, q.cin_pad = c0, c1, cin, kh, kw, mode, rot, op, cp
                    lib._rc(lib._dll.pf_pack_conv_weights_batch(arr, len(jobs), lib._stream(jobs[0][0])), "pf_pack_conv_weights_batch")
                return False
        return _Batch()

    def unpack_wgrads(self, jobs):
        """jobs: [(dw, db, gw, gb | None, cout, cin, taps, cin_pad, o_off, scale)]: gw += scale * unpacked(dw), gb += scale * db."""
        if not jobs:
            return
        arr = (UnpackJob * len(jobs))()
        for q, (dw, db, gw, gb, cout, cin, taps, cin_pad, o_off, scale) in zip(arr, jobs):
            self._chk(dw, db, gw, gb)
            if gw.numel() != cout * cin * taps or (gb is not None and gb.numel() != cout):
                raise PfError("pf_unpack_wgrads: gradient tensor does not match the convolution's shape")
            q.dw, q.db, q.gw, q.gb = _ptr(dw), _ptr(db), _ptr(gw), _ptr(gb)
            q.cout, q.cin, q.taps, q.cin_pad, q.o_off, q.scale = cout, cin, taps, cin_pad, o_off, scale
        self._rc(self._dll.pf_unpack_wgrads(arr, len(jobs), self._stream(jobs[0][0])), "pf_unpack_wgrads")

    def corr_pyramid_bf16x3(self, f1s, f2s, levels, B, H8, W8, Cch):
        self._chk(*levels)
        self._rc(self._dll.pf_corr_pyramid_bf16x3(C.c_void_p(f1s.data_ptr()), C.c_void_p(f2s.data_ptr()),
                                                  *[_ptr(l) for l in levels], B, H8, W8, Cch,
                                                  self._stream(levels[0])), "pf_corr_pyramid_bf16x3")

    def dccl_lookup(self, coords, pyr_own, pyr_other, g_w2c, own_out, raw_out, g_il=None):
        """g_il: optional interleaved copy [H8*W8, 2] of g_w2c (``interleave_grid``): fewer gather instructions."""
        self._chk(coords, g_w2c, own_out, raw_out, g_il, *pyr_own, *pyr_other)
        B, _, H, W = coords.shape
        if g_il is not None and g_il.numel() != g_w2c.numel():
            raise PfError("dccl_lookup: g_il must hold the same grid as g_w2c")
        self._rc(self._dll.pf_dccl_lookup_il(
            _ptr(coords), *[_ptr(p) for p in pyr_own], *[_ptr(p) for p in pyr_other],
            _ptr(g_w2c), _ptr(g_il), _ptr(own_out), _ptr(raw_out), B, H, W, own_out.shape[-1],
            self._stream(coords)), "pf_dccl_lookup_il")

    def dccl_combine(self, own, raw, g_back, out, B, H8, W8):
        self._chk(own, raw, g_back, out)
        self._rc(self._dll.pf_dccl_combine(_ptr(own), _ptr(raw), _ptr(g_back), _ptr(out), B, H8, W8,
                                           own.shape[-1], out.shape[-1], self._stream(own)),
                 "pf_dccl_combine")

    def dccl_combine_conv1x1(self, items, B, H8, W8):
        """items: 1 or 2 tuples (own, raw, g_back, conv, out, off_out[, out_split]) with `conv` a packed bf16x3 1x1 324->256
        engine.Conv: out[.., off_out:off_out+256] = relu(conv(own + rotate_back(raw))) without materialising the sum.
        out_split: optional split twin of `out` (`out` may then be None)."""
        arr = (CombineConvDesc * len(items))()
        keep = []
        for d, item in zip(arr, items):
            own, raw, g_back, conv, out, off_out = item[:6]
            out_split = item[6] if len(item) > 6 else None
            d.out_split, d.lds_out = _twin(out_split)
            self._chk(own, raw, g_back, conv.b, out)
            if conv.precision != PREC_BF16X3 or (conv.kh, conv.kw, conv.cin, conv.cout) != (1, 1, 324, 256):
                raise PfError("dccl_combine_conv1x1 needs the bf16x3 packing of a 1x1 324(352)->256 convolution")
            d.own, d.raw, d.ld, d.g_back = own.data_ptr(), raw.data_ptr(), own.shape[-1], g_back.data_ptr()
            d.weight, d.bias = conv.w.data_ptr(), conv.b.data_ptr()
            d.out = out.data_ptr() if out is not None else None
            d.ld_out, d.off_out, d.cout = (out.shape[-1] if out is not None else 0), off_out, conv.cout
            keep.append((own, raw, g_back, conv, out, out_split))
        self._rc(self._dll.pf_dccl_combine_conv1x1(arr, len(items), B, H8, W8, self._stream(items[0][0])),
                 "pf_dccl_combine_conv1x1")

    def warp_gcorr(self, f1, f2, coords, add_grid, dst, dst_off):
        self._chk(f1, f2, coords, dst)
        B, _, H, W = coords.shape
        self._rc(self._dll.pf_warp_gcorr(_ptr(f1), _ptr(f2), _ptr(coords), int(add_grid), _ptr(dst),
                                         dst.shape[-1], dst_off, B, H, W, f1.shape[-1],
                                         self._stream(f1)), "pf_warp_gcorr")

    # ---- update blocks -----------------------------------------------------------------------
    def conv2d(self, descs: Sequence[ConvDesc], B, H8, W8, like: torch.Tensor):
        arr = (ConvDesc * len(descs))(*descs)
        self._rc(self._dll.pf_conv2d(arr, len(descs), B, H8, W8, self._stream(like)), "pf_conv2d")

    def conv2d_tile(self, descs: Sequence[ConvDesc], B, H8, W8) -> int:
        arr = (ConvDesc * len(descs))(*descs)
        rc = self._dll.pf_conv2d_tile(arr, len(descs), B, H8, W8)
        if rc < 0:
            self._rc(rc, "pf_conv2d_tile")
        return rc

    def conv2d_stats_blocks(self, descs: Sequence[ConvDesc], B, H8, W8) -> int:
        """fp64 partial blocks per image a launch with ``stats_out`` writes (0: the launch cannot fuse the statistics)."""
        arr = (ConvDesc * len(descs))(*descs)
        rc = self._dll.pf_conv2d_stats_blocks(arr, len(descs), B, H8, W8)
        if rc < 0:
            self._rc(rc, "pf_conv2d_stats_blocks")
        return rc

    def conv2d_roles(self, descs: Sequence[ConvDesc], B, H8, W8) -> int:
        arr = (ConvDesc * len(descs))(*descs)
        rc = self._dll.pf_conv2d_roles(arr, len(descs), B, H8, W8)
        if rc < 0:
            self._rc(rc, "pf_conv2d_roles")
        return rc

    def conv2d_direct(self, x, off_in, cin, weight, bias, out, off_out, cout, kh, kw, relu, B, H8, W8):
        self._chk(x, weight, bias, out)
        self._rc(self._dll.pf_conv2d_direct(_ptr(x), x.shape[-1], off_in, cin, _ptr(weight), _ptr(bias),
                                            _ptr(out), out.shape[-1], off_out, cout, kh, kw, int(relu),
                                            B, H8, W8, self._stream(x)), "pf_conv2d_direct")

    def conv2d_direct_group(self, problems, cin, cout, kh, kw, relu, B, H8, W8):
        """problems: 1..4 tuples (x, off_in, weight, bias, out, off_out[, out_split]) of one shape -> one launch.
        out_split: optional split twin of `out` (7x7 2 -> C stems; `out` may then be None)."""
        arr = (DirectDesc * len(problems))()
        for d, item in zip(arr, problems):
            x, off_in, weight, bias, out, off_out = item[:6]
            self._chk(x, weight, bias, out)
            d.in_, d.ld_in, d.off_in = _ptr(x), x.shape[-1], off_in
            d.weight, d.bias = _ptr(weight), _ptr(bias)
            d.out, d.ld_out, d.off_out = _ptr(out), (out.shape[-1] if out is not None else 0), off_out
            d.out_split, d.lds_out = _twin(item[6] if len(item) > 6 else None)
        self._rc(self._dll.pf_conv2d_direct_group(arr, len(problems), cin, cout, kh, kw, int(relu), B, H8, W8,
                                                  self._stream(problems[0][0])), "pf_conv2d_direct_group")

    def conv2d_small(self, x, nchw, off_in, cin, weight, bias, out, off_out, cout, kh, kw, stride, relu,
                     B, Hout, Wout):
        self._chk(x, weight, bias, out)
        self._rc(self._dll.pf_conv2d_small(_ptr(x), int(nchw), 0 if nchw else x.shape[-1], off_in, cin,
                                           _ptr(weight), _ptr(bias), _ptr(out), out.shape[-1], off_out, cout,
                                           kh, kw, stride, int(relu), B, Hout, Wout, self._stream(x)),
                 "pf_conv2d_small")

    def enc_stem(self, images, weight, bias, out=None, out_split=None, relu=False, stats=None):
        """The encoders' 7x7 / 2 stem from NCHW images [Bn,3,H,W] (pf_enc_stem); weight from engine.pack_stem7x7."""
        self._chk(images, bias, out)
        Bn, Cc, H, W = images.shape
        if Cc != 3 or weight.dtype != torch.bfloat16 or weight.numel() != 64 * 352 or not weight.is_contiguous():
            raise PfError("enc_stem: images must be [Bn,3,H,W] and weight the 64 x 704-byte pack of engine.pack_stem7x7")
        sp, _ = _twin(out_split)
        self._rc(self._dll.pf_enc_stem(_ptr(images), C.c_void_p(weight.data_ptr()), _ptr(bias), _ptr(out), sp, int(relu),
                                       None if stats is None else C.c_void_p(stats.data_ptr()), Bn, H, W, self._stream(images)),
                 "pf_enc_stem")

    def debug_dirty_lds(self, pattern: int = 0x7fc00000, like=None):
        """Test support: every CU's whole LDS is filled with `pattern` (default: a NaN) on the current stream (pf_debug_dirty_lds)."""
        stream = self._stream(like) if like is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._rc(self._dll.pf_debug_dirty_lds(C.c_uint(pattern & 0xffffffff), stream), "pf_debug_dirty_lds")

    def conv2d_wgrad_small(self, x, nchw, off_in, cin, dy, off_dy, cout, dw, db, kh, kw, stride, B, Hout, Wout):
        """dw [Cout,Cin,KH,KW] (+= ), db [Cout] (+= or None) of a small-Cin convolution; x NCHW planes or channel-last."""
        self._chk(x, dy, dw, db)
        n = int(self._dll.pf_conv2d_wgrad_small_ws_floats(cin, cout, kh, kw, B, Hout, Wout))
        ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)      # per call: two streams may run stems side by side
        self._rc(self._dll.pf_conv2d_wgrad_small_ws(_ptr(x), int(nchw), 0 if nchw else x.shape[-1], off_in, cin,
                                                    _ptr(dy), dy.shape[-1], off_dy, cout, _ptr(dw), _ptr(db),
                                                    kh, kw, stride, B, Hout, Wout, _ptr(ws), n, self._stream(x)),
                 "pf_conv2d_wgrad_small_ws")

    def channel_stats(self, y, B, Np, Cch, scale, shift, partials, nblk, eps=1e-5):
        self._chk(y, scale, shift)
        if partials.dtype != torch.float64 or partials.numel() < B * nblk * Cch * 2:
            raise PfError("channel_stats: partials must be float64 with >= B*nblk*C*2 elements")
        self._rc(self._dll.pf_channel_stats(_ptr(y), B, Np, Cch, eps, _ptr(scale), _ptr(shift),
                                            C.c_void_p(partials.data_ptr()), nblk, self._stream(y)),
                 "pf_channel_stats")

    def channel_stats_final(self, partials, B, Np, Cch, nblk, scale, shift, eps=1e-5):
        """Second stage only: partials [B][nblk][C][2] float64 written by a conv launched with stats_out."""
        self._chk(scale, shift)
        if partials.dtype != torch.float64 or partials.numel() < B * nblk * Cch * 2:
            raise PfError("channel_stats_final: partials must be float64 with >= B*nblk*C*2 elements")
        self._rc(self._dll.pf_channel_stats_final(C.c_void_p(partials.data_ptr()), B, Np, Cch, nblk, eps,
                                                  _ptr(scale), _ptr(shift), self._stream(scale)),
                 "pf_channel_stats_final")

    def norm_act(self, y, s, t, out, B, Np, Cc, res=None, rs=None, rt=None, res_relu=False):
        self._chk(y, s, t, out, res, rs, rt)
        self._rc(self._dll.pf_norm_act(_ptr(y), _ptr(s), _ptr(t), _ptr(res), _ptr(rs), _ptr(rt), int(res_relu), _ptr(out),
                                       B, Np, Cc, self._stream(y)), "pf_norm_act")

    def flow_head_out(self, x, Cch, weight, bias, coords1, delta=None):
        self._chk(x, weight, bias, coords1, delta)
        B, _, H, W = coords1.shape
        self._rc(self._dll.pf_flow_head_out(_ptr(x), x.shape[-1], Cch, _ptr(weight), _ptr(bias), _ptr(coords1),
                                            _ptr(delta), 0 if delta is None else delta.shape[-1], B, H, W,
                                            self._stream(x)), "pf_flow_head_out")

    def coords_add(self, coords1, delta, src=None):
        """coords1 += delta, or (src given) coords1 = src + delta."""
        self._chk(coords1, delta, src)
        B, _, H, W = coords1.shape
        if src is None:
            self._rc(self._dll.pf_coords_add(_ptr(coords1), _ptr(delta), delta.shape[-1], B, H, W,
                                             self._stream(coords1)), "pf_coords_add")
        else:
            if src.shape != coords1.shape:
                raise PfError("pf_coords_add_to: src and dst differ in shape")
            self._rc(self._dll.pf_coords_add_to(_ptr(src), _ptr(delta), delta.shape[-1], _ptr(coords1), B, H, W,
                                                self._stream(coords1)), "pf_coords_add_to")

    def upsample_flow(self, coords1, mask, out):
        self._chk(coords1, mask, out)
        B, _, H, W = coords1.shape
        self._rc(self._dll.pf_upsample_flow(_ptr(coords1), _ptr(mask), mask.shape[-1], _ptr(out),
                                            B, H, W, self._stream(coords1)), "pf_upsample_flow")
        return out

    # ---- layout ------------------------------------------------------------------------------
    def to_channel_last(self, x, c_begin, c, out, off_out, act=ACT_NONE):
        """x: NCHW [B,Ct,H,W] -> out rows [B*H*W, ld] columns [off_out, off_out+c)."""
        self._chk(x, out)
        B, Ct = x.shape[:2]
        N = x.shape[2] * x.shape[3]
        self._rc(self._dll.pf_to_channel_last(_ptr(x), Ct, c_begin, c, _ptr(out), out.shape[-1], off_out,
                                              act, B, N, self._stream(x)), "pf_to_channel_last")
        return out

    def space_to_depth2(self, x, out):
        """x: NCHW [B,C,H,W] -> out rows [B*(H/2)*(W/2), ld], columns (py*2+px)*C + c."""
        self._chk(x, out)
        B, C, H, W = x.shape
        self._rc(self._dll.pf_space_to_depth2(_ptr(x), C, _ptr(out), out.shape[-1], B, H, W, self._stream(x)),
                 "pf_space_to_depth2")
        return out

    # ---- training-step pieces ---------------------------------------------------------------
    def warp_gcorr_bwd(self, f1, f2, coords, add_grid, d_flaw, off_d, d_f1, d_f2):
        """d_flaw: channel-last rows with the 4 group gradients at column off_d; d_f1 / d_f2 accumulated."""
        self._chk(f1, f2, coords, d_flaw, d_f1, d_f2)
        B, _, H, W = coords.shape
        self._rc(self._dll.pf_warp_gcorr_bwd(_ptr(f1), _ptr(f2), _ptr(coords), int(add_grid), _ptr(d_flaw),
                                             d_flaw.shape[-1], off_d, _ptr(d_f1), _ptr(d_f2), B, H, W, f1.shape[-1],
                                             self._stream(f1)), "pf_warp_gcorr_bwd")

    def upsample_flow_bwd(self, coords1, mask, g, d_mask, d_flow):
        """g [B,2,8H,8W] -> d_mask [B*N, >=576] (written), d_flow [B,2,H,W] (accumulated)."""
        self._chk(coords1, mask, g, d_mask, d_flow)
        B, _, H, W = coords1.shape
        self._rc(self._dll.pf_upsample_flow_bwd(_ptr(coords1), _ptr(mask), mask.shape[-1], _ptr(g), _ptr(d_mask),
                                                d_mask.shape[-1], _ptr(d_flow), B, H, W, self._stream(g)),
                 "pf_upsample_flow_bwd")

    def pyramid_bwd(self, g_levels, B, H8, W8):
        """g_levels: 4 level gradients [B*N, H_i*W_i]; level 0 becomes the dense volume gradient (in place)."""
        self._chk(*g_levels)
        self._rc(self._dll.pf_pyramid_bwd(*[_ptr(t) for t in g_levels], B, H8, W8, self._stream(g_levels[0])),
                 "pf_pyramid_bwd")
        return g_levels[0]

    def norm_bwd(self, dy, x, scale, shift, relu, instance, dx, B, Np, Cc, nblk=None):
        """Backward of act(x*scale+shift) (InstanceNorm when `instance`, else fixed statistics); rows [B*Np, C]."""
        self._chk(dy, x, scale, shift, dx)
        part = coef = None
        if instance:
            nblk = nblk or _norm_bwd_blocks(Np)
            part = torch.empty(B * nblk * Cc * 2, dtype=torch.float64, device=dy.device)
            coef = torch.empty(B * Cc * 2, dtype=torch.float32, device=dy.device)
        self._rc(self._dll.pf_norm_bwd(_ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), int(relu), int(instance),
                                       _ptr(part), nblk or 0, _ptr(coef), _ptr(dx), B, Np, Cc, self._stream(dy)),
                 "pf_norm_bwd")

    def norm_bwd_sums(self, dy, x, scale, shift, B, Np, Cc, nblk=None):
        """Per-(image, channel) means of g and g * (x*scale+shift) over the Np pixels: [B, C, 2] (the reduction stage of
        pf_norm_bwd's InstanceNorm branch; its dx is written to a scratch buffer and discarded)."""
        self._chk(dy, x, scale, shift)
        nblk = nblk or _norm_bwd_blocks(Np)
        part = torch.empty(B * nblk * Cc * 2, dtype=torch.float64, device=dy.device)
        coef = torch.empty(B * Cc * 2, dtype=torch.float32, device=dy.device)
        scratch = torch.empty_like(x)
        self._rc(self._dll.pf_norm_bwd(_ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), 0, 1, _ptr(part), nblk, _ptr(coef),
                                       _ptr(scratch), B, Np, Cc, self._stream(dy)), "pf_norm_bwd")
        return coef.view(B, Cc, 2)

    def add_relu(self, x, y, out):
        """out = relu(x + y), same-shape contiguous fp32 tensors."""
        self._chk(x, y, out)
        if x.numel() != y.numel() or x.numel() != out.numel():
            raise PfError("add_relu: shapes differ")
        self._rc(self._dll.pf_add_relu(_ptr(x), _ptr(y), _ptr(out), x.numel(), self._stream(x)), "pf_add_relu")
        return out

    def relu_mask(self, g, fwd_out, dx):
        """dx = fwd_out > 0 ? g : 0 (the backward of a ReLU from its output)."""
        self._chk(g, fwd_out, dx)
        if g.numel() != fwd_out.numel() or g.numel() != dx.numel():
            raise PfError("relu_mask: shapes differ")
        self._rc(self._dll.pf_relu_mask(_ptr(g), _ptr(fwd_out), _ptr(dx), g.numel(), self._stream(g)), "pf_relu_mask")
        return dx

    def bn_frozen_fwd(self, x, gamma, beta, mean, var, eps, relu, out):
        """out = [relu](frozen BatchNorm(x)) on channel-last rows [rows, C]."""
        self._chk(x, gamma, beta, mean, var, out)
        rows, Cc = x.shape
        self._rc(self._dll.pf_bn_frozen_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps), int(bool(relu)),
                                            _ptr(out), rows, Cc, self._stream(x)), "pf_bn_frozen_fwd")
        return out

    def bn_frozen_bwd(self, dy, x, gamma, beta, mean, var, eps, relu, dx, dgamma, dbeta, accumulate):
        self._chk(dy, x, gamma, beta, mean, var, dx, dgamma, dbeta)
        rows, Cc = x.shape
        nblk = int(max(1, min(rows // 16, 2048)))        # >= 16 rows per partial sum; 2048 x C threads fill the chip at the encoder sizes
        part = torch.empty(nblk * Cc * 2, dtype=torch.float64, device=dy.device)
        self._rc(self._dll.pf_bn_frozen_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps),
                                            int(bool(relu)), C.c_void_p(part.data_ptr()), nblk, _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                            int(bool(accumulate)), rows, Cc, self._stream(dy)), "pf_bn_frozen_bwd")
        return dx

    def gru_q_bwd(self, dh_new, z, q, h, dq_pre, dz, dh):
        """Stage Q of the GRU gate backward; every argument a channel-last [rows, >=C] view (C = dh_new.shape[-1])."""
        self._chk_rows(dh_new, z, q, h, dq_pre, dz, dh)
        rows, Cc = dh_new.shape
        args = []
        for t in (dh_new, z, q, h, dq_pre, dz, dh):
            args += [_ptr(t), t.stride(0)]
        self._rc(self._dll.pf_gru_q_bwd(*args, rows, Cc, self._stream(dh_new)), "pf_gru_q_bwd")

    def gru_zr_bwd(self, dz, d_rh, z, r, h, dzr_pre, dh):
        """Stage ZR; dzr_pre is [rows, >=2C] (z half | r half), dh is accumulated."""
        self._chk_rows(dz, d_rh, z, r, h, dzr_pre, dh)
        rows, Cc = dz.shape
        args = []
        for t in (dz, d_rh, z, r, h, dzr_pre, dh):
            args += [_ptr(t), t.stride(0)]
        self._rc(self._dll.pf_gru_zr_bwd(*args, rows, Cc, self._stream(dz)), "pf_gru_zr_bwd")

    def gru_dx_finish(self, f1, f2, x, d_inp, d_out, Cc, wout):
        """d_inp += (f1 + f2)[:, :C]; d_out[:, :wout] = (f1 + f2)[:, C:C+wout] masked by x[:, C:C+wout] > 0 (row views)."""
        self._chk_rows(f1, f2, x, d_inp, d_out)
        self._rc(self._dll.pf_gru_dx_finish(_ptr(f1), f1.stride(0), _ptr(f2), f2.stride(0), _ptr(x), x.stride(0),
                                            _ptr(d_inp), d_inp.stride(0), _ptr(d_out), d_out.stride(0), f1.shape[0], Cc, wout,
                                            self._stream(f1)), "pf_gru_dx_finish")

    def dccl_combine_bwd(self, d_corr, g_back, d_raw, B, H8, W8):
        self._chk(d_corr, g_back, d_raw)
        self._rc(self._dll.pf_dccl_combine_bwd(_ptr(d_corr), d_corr.shape[-1], _ptr(g_back), _ptr(d_raw), d_raw.shape[-1],
                                               B, H8, W8, self._stream(d_corr)), "pf_dccl_combine_bwd")

    def dccl_lookup_bwd(self, coords, g_w2c, d_own, d_raw, g_own, g_other, clear_raw=False):
        """g_own / g_other: lists of 4 level gradients [B*N, H_i*W_i], accumulated into.  clear_raw: d_raw is left all zero."""
        self._chk(coords, g_w2c, d_own, d_raw, *g_own, *g_other)
        B, _, H, W = coords.shape
        if d_raw.shape[-1] != d_own.shape[-1]:
            raise PfError("pf_dccl_lookup_bwd: d_own and d_raw must share their row length")
        self._rc(self._dll.pf_dccl_lookup_bwd(_ptr(coords), _ptr(g_w2c), _ptr(d_own), _ptr(d_raw), d_own.shape[-1],
                                              *[_ptr(t) for t in g_own], *[_ptr(t) for t in g_other], B, H, W, int(clear_raw),
                                              self._stream(coords)), "pf_dccl_lookup_bwd")

    def conv2d_wgrad(self, x0, off0, c0, dy, off_dy, cout, dw, db, kh, kw, B, H8, W8, x1=None, off1=0, c1=0):
        """dw [Cout_pad128, kh*kw, Cin_pad32] and db [Cout_pad128] (or None) are accumulated into."""
        self._chk(x0, x1, dy, dw, db)
        cin_pad = (c0 + c1 + 31) // 32 * 32
        if dw.shape[-1] != cin_pad or dw.shape[-2] != kh * kw or dw.shape[0] < cout:
            raise PfError(f"conv2d_wgrad: dw must be [>= {cout}, {kh * kw}, {cin_pad}], got {tuple(dw.shape)}")
        self._rc(self._dll.pf_conv2d_wgrad(_ptr(x0), x0.shape[-1], off0, c0, _ptr(x1), 0 if x1 is None else x1.shape[-1],
                                           off1, c1, _ptr(dy), dy.shape[-1], off_dy, cout, _ptr(dw), _ptr(db),
                                           kh, kw, B, H8, W8, self._stream(x0)), "pf_conv2d_wgrad")

    def seq_loss(self, pred, gt, valid, weight, i_weight, max_flow, grad, partials):
        """pred, gt [B,2,H,W]; valid [B,H,W]; weight [H*W]; grad [B,2,H,W] or None; partials float64 [B,nblk,6]."""
        self._chk(pred, gt, valid, weight, grad)
        B = pred.shape[0]
        N = pred.shape[2] * pred.shape[3]
        if partials.dtype != torch.float64 or partials.dim() != 3 or partials.shape[0] != B or partials.shape[2] != 6:
            raise PfError("seq_loss: partials must be float64 [B, nblk, 6]")
        if gt.shape != pred.shape or valid.numel() != B * N or weight.numel() != N:
            raise PfError("seq_loss: shape mismatch")
        self._rc(self._dll.pf_seq_loss(_ptr(pred), _ptr(gt), _ptr(valid), _ptr(weight), i_weight, max_flow, _ptr(grad),
                                       C.c_void_p(partials.data_ptr()), partials.shape[1], B, N, self._stream(pred)),
                 "pf_seq_loss")

    def seq_loss_batch(self, preds, gt, valid, weight, i_weights, max_flow, grads, partials):
        """The n <= 32 terms seq_loss(preds[i], ..., i_weights[i], grads[i], partials[i]) in one launch; grads: list of tensors or
        None; partials float64 [n, B, nblk, 6]."""
        n = len(preds)
        self._chk(gt, valid, weight, *preds, *(grads or ()))
        B = gt.shape[0]
        N = gt.shape[2] * gt.shape[3]
        if partials.dtype != torch.float64 or tuple(partials.shape[:2]) != (n, B) or partials.shape[3] != 6 or not partials.is_contiguous():
            raise PfError("seq_loss_batch: partials must be contiguous float64 [n, B, nblk, 6]")
        if any(p.shape != gt.shape or not p.is_contiguous() or p.dtype != torch.float32 for p in preds) or valid.numel() != B * N \
                or weight.numel() != N or (grads is not None and (len(grads) != n or any(g.shape != gt.shape for g in grads))):
            raise PfError("seq_loss_batch: shape mismatch")
        pa = (C.c_void_p * n)(*[p.data_ptr() for p in preds])
        ga = (C.c_void_p * n)(*[g.data_ptr() for g in grads]) if grads is not None else None
        wa = (C.c_float * n)(*[float(w) for w in i_weights])
        self._rc(self._dll.pf_seq_loss_batch(pa, _ptr(gt), _ptr(valid), _ptr(weight), wa, max_flow, ga,
                                             C.c_void_p(partials.data_ptr()), partials.shape[2], n, B, N, self._stream(gt)),
                 "pf_seq_loss_batch")

    def sum_squares(self, x, partials):
        self._chk(x)
        if partials.dtype != torch.float64 or not partials.is_contiguous():
            raise PfError("sum_squares: partials must be contiguous float64")
        self._rc(self._dll.pf_sum_squares(_ptr(x), x.numel(), C.c_void_p(partials.data_ptr()), partials.numel(),
                                          self._stream(x)), "pf_sum_squares")

    def adamw_step(self, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        self._chk(p, g, m, v)
        if not (p.numel() == g.numel() == m.numel() == v.numel()):
            raise PfError("adamw_step: buffers must have the same size")
        self._rc(self._dll.pf_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps,
                                         weight_decay, step, grad_scale, self._stream(p)), "pf_adamw_step")

    def adamw_step_dev(self, p, g, m, v, beta1, beta2, eps, hyper):
        """pf_adamw_step with {decay, step_size, sqrt_bc2, grad_scale} read from the device tensor `hyper` (4 floats)."""
        self._chk(p, g, m, v, hyper)
        if not (p.numel() == g.numel() == m.numel() == v.numel()) or hyper.numel() < 4:
            raise PfError("adamw_step_dev: buffers must have the same size and hyper 4 floats")
        self._rc(self._dll.pf_adamw_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), beta1, beta2, eps, _ptr(hyper),
                                             self._stream(p)), "pf_adamw_step_dev")

    def flow_metrics(self, pred, gt, epe=None, sd=None, cosine=False):
        """pred, gt: NCHW [B,2,H,W]; epe / sd: [B,H,W] outputs (either optional); cosine: the 'Cosine' form of the distance."""
        self._chk(pred, gt, epe, sd)
        B, _, H, W = pred.shape
        if gt.shape != pred.shape:
            raise PfError("flow_metrics: pred and gt must have the same shape")
        self._rc(self._dll.pf_flow_metrics(_ptr(pred), _ptr(gt), _ptr(epe), _ptr(sd), int(cosine), B, H, W, self._stream(pred)),
                 "pf_flow_metrics")

    def region_sums(self, epe, sd, weight, bits, nregions, partials):
        """epe, sd: [B,H,W]; weight: [H*W] or None; bits: uint8 [H*W]; partials: float64 [B,nblk,nregions,3]."""
        self._chk(epe, sd, weight)
        if bits.dtype != torch.uint8 or partials.dtype != torch.float64 or not bits.is_contiguous() \
                or not partials.is_contiguous():
            raise PfError("region_sums: bits must be contiguous uint8, partials contiguous float64")
        B = epe.shape[0]
        N = epe[0].numel()
        if bits.numel() != N or partials.shape[0] != B or partials.shape[2] != nregions or partials.shape[3] != 3:
            raise PfError("region_sums: shape mismatch")
        self._rc(self._dll.pf_region_sums(_ptr(epe), _ptr(sd), _ptr(weight), C.c_void_p(bits.data_ptr()), nregions,
                                          C.c_void_p(partials.data_ptr()), partials.shape[1], B, N,
                                          self._stream(epe)), "pf_region_sums")

    def to_nchw(self, x, off_in, c, out):
        self._chk(x, out)
        B = out.shape[0]
        N = out.shape[2] * out.shape[3]
        self._rc(self._dll.pf_to_nchw(_ptr(x), x.shape[-1], off_in, c, _ptr(out), B, N,
                                      self._stream(x)), "pf_to_nchw")
        return out


_cached: Optional[PfLib] = None


_WEIGHTS_EPOCH = [0]


def weights_epoch() -> int:
    """Counter of in-place parameter updates made through raw pointers (pf_adamw_step): torch's per-tensor
    version counters do not see those, so every cache of packed weights keys on this as well."""
    return _WEIGHTS_EPOCH[0]


def bump_weights_epoch() -> None:
    _WEIGHTS_EPOCH[0] += 1


def load() -> PfLib:
    """The product's library handle (built in-tree under prior-flow_amd/lib/)."""
    global _cached
    if _cached is None:
        _cached = PfLib(LIB_PATH, require_cuda=True)
    return _cached
