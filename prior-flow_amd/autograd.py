"""Differentiable (training-mode) forward of PriOr-RAFT on the HIP kernels.

``train_forward(model, image1, image2, iters, init_flow)`` follows core/prior_raft.py:107-215 and
core/update.py / core/extractor.py with torch autograd as the tape and the HIP library as the arithmetic,
in BOTH directions, of every heavy operator:

  operator (reference)                           forward                          backward
  ---------------------------------------------  -------------------------------  ----------------------------------------
  nn.Conv2d 3x3 / 1x5 / 5x1 / 1x1, stride 1      pf_conv2d (bf16x3 MFMA)          pf_conv2d on flipped weights (dgrad),
                                                                                  pf_conv2d_wgrad (+ bias column sums)
  nn.Conv2d 3x3 / 1x1, stride 2 (extractor.py)   pf_conv2d, stride 2              the same two on the zero-stuffed dY
  nn.Conv2d 7x7 stems (3->64 /2, 2->128)         pf_conv2d_small (exact fp32)     pf_conv2d_wgrad_small (inputs carry no gradient)
  nn.BatchNorm2d, frozen (freeze_bn)             elementwise affine               pf_norm_bwd (dx, and the sums for d gamma / d beta)
  corr + build_pyramid (prior_raft.py:69-75)     pf_corr_pyramid_bf16x3           pf_pyramid_bwd + two MFMA GEMMs (pf_conv2d 1x1)
  DCCL.__call__ own + cross (corr.py:113-144)    pf_dccl_lookup + pf_dccl_combine pf_dccl_combine_bwd + pf_dccl_lookup_bwd
  warp + groupwise_corr (prior_raft.py:173-182)  pf_warp_gcorr                    pf_warp_gcorr_bwd
  upsample_flow (prior_raft.py:58-67)            pf_upsample_flow                 pf_upsample_flow_bwd
  SepConvGRU gates (update.py:49-60)             elementwise                      pf_gru_zr_bwd, pf_gru_q_bwd
  InstanceNorm2d (extractor.py:112-113)          elementwise                      pf_norm_bwd
  img_rotate / flo_rotate / sample grids         pf_img_rotate / pf_flo_rotate    (inputs are detached in the reference)

No convolution, normalisation or GEMM of a training step runs on PyTorch-ROCm kernels (STATS["torch"] stays 0 for the
reference's configuration: BatchNorm frozen, train_flow.py:107-108).  What torch still does is plumbing on device
tensors: ReLU / tanh / sigmoid forward, torch.cat / slicing / transposes, the zero-stuffing copy of a stride-2
gradient, and autograd's own bookkeeping.  A BatchNorm left in training mode (batch statistics, the `chairs` stage) runs on
pf_channel_stats / pf_norm_bwd over the whole batch (HipBatchNormTrain); a convolution geometry without a HIP pair raises
PfError -- there is no PyTorch-ROCm convolution or normalisation fallback (STATS["torch"] only counts what a test injects).
There is no CPU path: every tensor must live on a ROCm device and ``_lib.load()`` raises when the HIP library
is missing.  ``args.mixed_precision`` (CUDA autocast + GradScaler in the reference, train_flow.py:112,131-139) has no
counterpart here: the convolutions run the 3-pass bf16 split with fp32 accumulation and fp32 storage, which needs no
loss scaling.  The weight and pyramid gradients are accumulated in side buffers and handed to autograd once per
backward pass by token-ordered nodes (``WeightGate``, ``HipCorrPyramid``), so ``.backward()``, ``torch.autograd.backward``
and ``torch.autograd.grad`` all see ordinary gradients.
"""
from __future__ import annotations

import math
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F          # F.pad only (layout plumbing): no convolution / normalisation goes through torch

from . import _lib
from ._lib import EPI_LINEAR, PREC_BF16X3
from .engine import Conv, pack_mfma, rotation_x

class _LaunchStats:
    """Launch counters {"hip": launches routed to the HIP library, "torch": launches left to torch} -- PER DEVICE (round 6; a
    process-wide dict before): ``STATS[key]`` reads / writes the counter of the CURRENT device, so nn.DataParallel's worker threads or
    two models training on two cards never add into one another's count (no process-global mutable state).  Tests and
    profiles read ``STATS["hip"]`` on the device they drive; ``for_device`` / ``as_dict`` give the others."""

    def __init__(self):
        self._by_dev = {}

    def _cur(self):
        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        d = self._by_dev.get(dev)
        if d is None:
            d = self._by_dev.setdefault(dev, {"hip": 0, "torch": 0})
        return d

    def __getitem__(self, key):
        return self._cur()[key]

    def __setitem__(self, key, value):
        self._cur()[key] = value

    def for_device(self, dev) -> dict:
        return self._by_dev.setdefault(torch.device("cuda", dev).index if not isinstance(dev, int) else dev, {"hip": 0, "torch": 0})

    def as_dict(self) -> dict:
        return {d: dict(c) for d, c in self._by_dev.items()}

    def __repr__(self):
        return repr(dict(self._cur()))


STATS = _LaunchStats()


def _rows(x: torch.Tensor) -> torch.Tensor:
    """NCHW (logical) -> channel-last rows [B*H*W, C]; a view when x already has channels_last strides."""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _nchw(rows: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """Channel-last rows -> logical NCHW with channels_last strides (no copy): the kernels' layout IS the
    tape's memory format, torch's elementwise ops / cat preserve it, and _rows() of the result is free."""
    return rows.view(B, H, W, -1).permute(0, 3, 1, 2)


# weight packs of one training step, keyed by (storage, version): a conv that runs 2 x iters times packs once
_PACKS: Dict[tuple, object] = {}


def _src_key(*params: torch.Tensor) -> tuple:
    """Identity of the parameters a (possibly derived) weight tensor was built from."""
    return tuple((p.data_ptr(), p._version) for p in params)


def _pack(w: torch.Tensor, b: Optional[torch.Tensor], kind: str, src: Optional[tuple] = None, w1: Optional[torch.Tensor] = None,
          b1: Optional[torch.Tensor] = None, rot: int = 0):
    """Packed operand of pf_conv2d for weights `w` (optionally `w1` concatenated on Cout: the fused z|r convolution):
    kind "fwd" = the forward convolution (with bias b [| b1]), "dgrad" = its data-gradient convolution (flipped taps,
    transposed channels, forward input channels rotated by `rot`, Cout padded to 4).  One pf_pack_conv_weights launch
    (round 3 built the same bits with ~9 PyTorch-ROCm kernels per pack)."""
    if src is None:
        src = _src_key(*[t for t in (w, w1, b, b1) if t is not None])
    cout = w.shape[0] + (0 if w1 is None else w1.shape[0])
    _, cin, kh, kw = w.shape
    key = (kind, rot, src, cout, tuple(w.shape[1:]), _lib.weights_epoch())
    hit = _PACKS.get(key)
    if hit is not None:
        return hit
    if len(_PACKS) > 256:           # one training step needs ~140 entries; stale ones go with the next refill
        _PACKS.clear()
    lib = _lib.load()
    det = lambda t: None if t is None else t.detach().contiguous()      # noqa: E731
    if kind == "fwd":
        wp, bp = lib.pack_conv_weights(det(w), det(b), det(w1), det(b1), mode=0)
        val = Conv(wp, bp, kh, kw, cin, cout, PREC_BF16X3, presplit=True)
    else:           # the forward kernel on flipped / transposed weights (its input channels = Cout padded to 4 with zero rows)
        wp, bp = lib.pack_conv_weights(det(w), None, det(w1), None, mode=1, cin_rot=rot)
        val = Conv(wp, bp, kh, kw, (cout + 3) // 4 * 4, cin, PREC_BF16X3, presplit=True)
    _PACKS[key] = val
    return val


class GradSink:
    """Direct route of the convolutions' weight / bias gradients into the parameters' ``.grad`` tensors (the views of
    FlatAdamW's flat gradient buffer), switched on by ``train.train_step`` / ``GraphedTrainStep`` around forward + backward.
    Without it every convolution hands autograd a fresh (dW, db): a zero fill of the packed buffer, a permute copy, a clone and two
    accumulation adds -- five PyTorch kernels per convolution, ~290 per step.  With it the packed buffers are slices of ONE arena
    (one fill per step), the autograd nodes return None for the parameters, and ``flush()`` adds all of them into the ``.grad``
    tensors with ceil(n / 16) launches of pf_unpack_wgrads after the backward pass."""

    def __init__(self):
        self.active = False
        self.arena: Optional[torch.Tensor] = None
        self.used = 0
        self.want = 0
        self.jobs: list = []
        self.slow: list = []                      # convolutions with a frozen weight or bias (torch ops at flush)
        self.keep: list = []
        self.join_streams: list = []              # streams that still write packed gradients (train_loop's deferred launches)

    def begin(self, params) -> bool:
        """Starts a step; False (and stays off) unless every parameter carries a contiguous fp32 ``.grad`` already."""
        self.active = all(p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in params)
        self.jobs, self.slow, self.keep, self.used = [], [], [], 0
        self.join_streams = []
        if not self.active:
            return False
        dev = params[0].device
        if not self.reserve(dev) and self.arena is not None:
            self.arena.zero_()
        self.want = 0
        return True

    def reserve(self, dev) -> bool:
        """Sizes the arena for the step about to run from what the previous step asked for; True when a new (zeroed) one was
        allocated.  ``GraphedTrainStep`` calls this BEFORE it starts capturing, so the arena is an ordinary allocation and not part
        of the graph's private pool, and keeps a reference to the arena its graph was captured on: a later step of a larger model
        on this device REPLACES ``self.arena`` (this method never frees or resizes in place), so the captured graph's pointer stays
        valid for as long as its stepper lives."""
        if self.arena is None or self.arena.numel() < self.want or self.arena.device != dev:
            self.arena = torch.zeros(max(self.want, 1), device=dev) if self.want else None
            return True
        return False

    def packed(self, op: int, taps: int, cin_pad: int, device):
        """Zeroed (dw [op, taps, cin_pad], db [op]) for pf_conv2d_wgrad to accumulate into."""
        n = op * taps * cin_pad + op
        self.want += n
        if self.arena is not None and self.used + n <= self.arena.numel() and self.arena.device == device:
            both = self.arena[self.used:self.used + n]
            self.used += n
        else:                                   # first step (arena not sized yet) or a model that grew: a buffer of its own
            both = torch.zeros(n, device=device)
            self.keep.append(both)
        return both[:op * taps * cin_pad].view(op, taps, cin_pad), both[op * taps * cin_pad:]

    def add(self, dw, db, w, b, o_off: int = 0, scale: float = 1.0):
        """Registers the packed gradient of one convolution.  A frozen parameter (requires_grad=False: not in the optimizer, no
        ``.grad`` -- e.g. frozen encoders under a trainable update block) is skipped, as autograd drops a gradient nobody asked
        for; a convolution of which only the weight or only the bias is trainable takes the slow path of ``flush``."""
        cout, cin, kh, kw = w.shape
        wg = w.grad if w.requires_grad else None
        bg = b.grad if (b is not None and b.requires_grad) else None
        if wg is None and bg is None:
            return
        if wg is None or (b is not None and bg is None):
            self.slow.append((dw, db, wg, bg, cout, cin, kh, kw, o_off, float(scale)))
            return
        self.jobs.append((dw, db, wg, bg, cout, cin, kh * kw, dw.shape[2], o_off, float(scale)))

    def flush(self):
        """Adds every registered packed gradient into its parameters' ``.grad``; ends the step."""
        try:
            for st in self.join_streams:
                torch.cuda.current_stream().wait_stream(st)
            if self.active and self.jobs:
                _lib.load().unpack_wgrads(self.jobs)
                STATS["hip"] += (len(self.jobs) + 15) // 16
            if self.active:
                for dw, db, wg, bg, cout, cin, kh, kw, o, scale in self.slow:
                    if wg is not None:
                        wg.add_(Conv.unpack_wgrad(dw[o:o + cout], cout, cin, kh, kw), alpha=scale)
                    if bg is not None:
                        bg.add_(db[o:o + cout], alpha=scale)
        finally:
            self.abort()

    def abort(self):
        """Ends the step without touching the gradients (an exception is on its way out)."""
        for st in self.join_streams:          # (also on the way out of a failed step: later work must not overtake the side stream)
            try:
                torch.cuda.current_stream().wait_stream(st)
            except RuntimeError:              # e.g. inside an invalidated graph capture: the original exception must not be masked
                pass
        self.jobs, self.slow, self.keep, self.join_streams, self.active = [], [], [], [], False


class _SinkByDevice:
    """``autograd.SINK``: one GradSink per device, selected by the CURRENT device (the reference's caller is
    ``nn.DataParallel`` -- worker threads, one device each --, so the training path may hold no state that two replicas share;
    autograd runs a backward node with its forward's device current, so the autograd Functions, ``train_step`` and
    ``GraphedTrainStep`` of one replica all see the same sink and another replica's step never touches it)."""

    def __init__(self):
        object.__setattr__(self, "_by_device", {})

    def for_device(self, index=None) -> GradSink:
        idx = torch.cuda.current_device() if index is None else int(index)
        sinks = self._by_device
        if idx not in sinks:
            sinks[idx] = GradSink()
        return sinks[idx]

    def __getattr__(self, name):
        return getattr(self.for_device(), name)

    def __setattr__(self, name, value):
        setattr(self.for_device(), name, value)


SINK = _SinkByDevice()


class WeightGrad:
    """Packed weight / bias gradient of ONE convolution, accumulated over all its uses in a backward pass
    (pf_conv2d_wgrad accumulates): a conv of the update blocks runs `iters` times, and handing autograd a fresh
    gradient per use cost two zero fills, an unpack copy, a clone and two accumulation adds each time."""

    def __init__(self):
        self.dw: Optional[torch.Tensor] = None
        self.db: Optional[torch.Tensor] = None
        self.scale = 1.0                                  # applied when the gradient is handed over (mask head: 0.25)

    def buffers(self, op: int, taps: int, cin_pad: int, device):
        if self.dw is None:
            if SINK.active:
                self.dw, self.db = SINK.packed(op, taps, cin_pad, device)
            else:
                both = torch.zeros(op * taps * cin_pad + op, device=device)         # one fill for the pair
                self.dw = both[:op * taps * cin_pad].view(op, taps, cin_pad)
                self.db = both[op * taps * cin_pad:]
        return self.dw, self.db

    def take(self):
        got = (self.dw, self.db) if self.dw is not None else None
        self.dw = self.db = None
        return got


class WeightGate(torch.autograd.Function):
    """Returns a one-element token for the parameters (w1, b1[, w2, b2 ...]) of one (possibly fused) convolution.  Every
    HipConv that uses them takes the token as an input, which orders this node's backward after all of theirs; it then
    unpacks the accumulated gradient ONCE and splits it over the parameters by output-channel range."""

    @staticmethod
    def forward(ctx, acc, *params):
        ctx.acc = acc
        ctx.shapes = [tuple(p.shape) for p in params[0::2]]
        ctx.params = params                                  # python references (for GradSink), not saved tensors
        ctx.set_materialize_grads(False)
        return torch.empty(1, device=params[0].device)      # a token: its value is never read

    @staticmethod
    def backward(ctx, _):
        scale = ctx.acc.scale
        got = ctx.acc.take()
        if got is None:
            return (None,) * (1 + 2 * len(ctx.shapes))
        dw, db = got
        if SINK.active:
            o = 0
            for w, b in zip(ctx.params[0::2], ctx.params[1::2]):
                SINK.add(dw, db, w, b, o_off=o, scale=scale)
                o += w.shape[0]
            return (None,) * (1 + 2 * len(ctx.shapes))
        if scale != 1.0:
            torch._foreach_mul_([dw, db], scale)
        grads, o = [], 0
        for cout, cin, kh, kw in ctx.shapes:
            grads += [Conv.unpack_wgrad(dw[o:o + cout], cout, cin, kh, kw), db[o:o + cout].clone()]
            o += cout
        return (None, *grads)


class HipConv(torch.autograd.Function):
    """Stride-1 'same' convolution with bias: pf_conv2d forward, pf_conv2d (dgrad) + pf_conv2d_wgrad backward.
    With a (token, WeightGrad) pair the weight / bias gradient goes to the accumulator (see WeightGate) and autograd
    gets None for w and b; without one it is returned per call."""

    @staticmethod
    def forward(ctx, x, w, b, src=None, tok=None, acc=None):
        lib = _lib.load()
        B, C, H, W = x.shape
        cout, _, kh, kw = w.shape
        xr = _rows(x.detach())
        cv = _pack(w, b, "fwd", src)
        ctx.src = src
        ctx.acc = acc if tok is not None else None
        cp = (cout + 3) // 4 * 4
        out = (torch.empty if cp == cout else torch.zeros)(B * H * W, cp, device=x.device)
        lib.conv2d([cv.desc(xr, 0, C, out, 0, EPI_LINEAR)], B, H, W, xr)
        ctx.save_for_backward(xr, w)
        ctx.wb = (w, b)
        ctx.shape = (B, C, H, W, cout, kh, kw, cp)
        STATS["hip"] += 1
        return _nchw(out if cp == cout else out[:, :cout].contiguous(), B, H, W)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xr, w = ctx.saved_tensors
        B, C, H, W, cout, kh, kw, cp = ctx.shape
        if cp == cout:
            dy = _rows(gy)
        else:
            dy = torch.zeros(B * H * W, cp, device=gy.device)
            dy[:, :cout] = _rows(gy)
        dx = None
        if ctx.needs_input_grad[0]:
            dg = _pack(w, None, "dgrad", ctx.src)
            dxr = torch.empty(B * H * W, C, device=gy.device)
            lib.conv2d([dg.desc(dy, 0, cp, dxr, 0, EPI_LINEAR)], B, H, W, dy)
            dx = _nchw(dxr, B, H, W)
        op = (cp + 127) // 128 * 128
        STATS["hip"] += 2
        if ctx.acc is not None:
            dw, db = ctx.acc.buffers(op, kh * kw, (C + 31) // 32 * 32, gy.device)
            lib.conv2d_wgrad(xr, 0, C, dy, 0, cp, dw, db, kh, kw, B, H, W)
            return dx, None, None, None, None, None
        if SINK.active:
            dw, db = SINK.packed(op, kh * kw, (C + 31) // 32 * 32, gy.device)
            lib.conv2d_wgrad(xr, 0, C, dy, 0, cp, dw, db, kh, kw, B, H, W)
            SINK.add(dw, db, *ctx.wb)
            return dx, None, None, None, None, None
        dw = torch.zeros(op, kh * kw, (C + 31) // 32 * 32, device=gy.device)
        db = torch.zeros(op, device=gy.device)
        lib.conv2d_wgrad(xr, 0, C, dy, 0, cp, dw, db, kh, kw, B, H, W)
        return dx, Conv.unpack_wgrad(dw, cout, C, kh, kw), db[:cout].clone(), None, None, None


class HipConvS2(torch.autograd.Function):
    """Stride-2 convolution (3x3 pad 1 / 1x1 pad 0: the encoders' down-sampling layers, core/extractor.py:16-17,30-33):
    pf_conv2d(stride = 2) forward.  Backward on the ZERO-STUFFED gradient dYu (dYu[2p] = dY[p], zero elsewhere, input
    resolution): dX = stride-1 conv of dYu with the flipped weights, dW = stride-1 weight gradient of (X, dYu) --
    out[p] = sum_k w[k] x[2p + k - pad]  =>  dX[q] = sum_k w[k] dYu[q - k + pad],  dW[k] = sum_q dYu[q] x[q + k - pad]."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        B, C, H, W = x.shape
        cout, _, kh, kw = w.shape
        Ho, Wo = H // 2, W // 2
        xr = _rows(x.detach())
        cv = _pack(w, b, "fwd")
        out = torch.empty(B * Ho * Wo, cout, device=x.device)
        lib.conv2d([cv.desc(xr, 0, C, out, 0, EPI_LINEAR, stride=2)], B, Ho, Wo, xr)
        ctx.save_for_backward(xr, w)
        ctx.wb = (w, b)
        ctx.shape = (B, C, H, W, cout, kh, kw)
        STATS["hip"] += 1
        return _nchw(out, B, Ho, Wo)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xr, w = ctx.saved_tensors
        B, C, H, W, cout, kh, kw = ctx.shape
        dyu = torch.zeros(B, H, W, cout, device=gy.device)
        dyu[:, ::2, ::2] = gy.permute(0, 2, 3, 1)                 # zero-stuffing (a strided copy)
        dyu = dyu.view(B * H * W, cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dg = _pack(w, None, "dgrad")
            dxr = torch.empty(B * H * W, C, device=gy.device)
            lib.conv2d([dg.desc(dyu, 0, cout, dxr, 0, EPI_LINEAR)], B, H, W, dyu)
            dx = _nchw(dxr, B, H, W)
        op = (cout + 127) // 128 * 128
        STATS["hip"] += 2
        if SINK.active:
            dw, db = SINK.packed(op, kh * kw, (C + 31) // 32 * 32, gy.device)
            lib.conv2d_wgrad(xr, 0, C, dyu, 0, cout, dw, db, kh, kw, B, H, W)
            SINK.add(dw, db, *ctx.wb)
            return dx, None, None
        dw = torch.zeros(op, kh * kw, (C + 31) // 32 * 32, device=gy.device)
        db = torch.zeros(op, device=gy.device)
        lib.conv2d_wgrad(xr, 0, C, dyu, 0, cout, dw, db, kh, kw, B, H, W)
        # every zero of dYu adds 0 to db, so the column sums over the stuffed map are the bias gradient
        return dx, Conv.unpack_wgrad(dw, cout, C, kh, kw), db[:cout].clone()


class HipSmallConv(torch.autograd.Function):
    """The 7x7 stems (3 -> 64 stride 2 of the encoders, core/extractor.py:122; 2 -> 128 stride 1 of the motion encoders,
    core/update.py:87,173,175) on the exact-fp32 small-Cin kernel; their inputs -- the images, the detached flows --
    need no gradient, so the backward is the weight / bias gradient alone (pf_conv2d_wgrad_small)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        lib = _lib.load()
        B, C, H, W = x.shape
        cout, _, kh, kw = w.shape
        Ho, Wo = H // stride, W // stride
        xin = x.detach().contiguous()                              # NCHW planes
        wp = w.detach().permute(2, 3, 1, 0).reshape(kh * kw * C, cout).contiguous()      # [KH*KW][Cin][Cout]
        out = torch.empty(B * Ho * Wo, cout, device=x.device)
        lib.conv2d_small(xin, True, 0, C, wp, b.detach().contiguous(), out, 0, cout, kh, kw, stride, False, B, Ho, Wo)
        ctx.save_for_backward(xin)
        ctx.wb = (w, b)
        ctx.shape = (B, C, Ho, Wo, cout, kh, kw, stride)
        STATS["hip"] += 1
        return _nchw(out, B, Ho, Wo)

    @staticmethod
    def backward(ctx, gy):
        if ctx.needs_input_grad[0]:
            raise _lib.PfError("the small-Cin stem convolutions have no data gradient (their inputs are detached in the reference)")
        lib = _lib.load()
        (xin,) = ctx.saved_tensors
        B, C, Ho, Wo, cout, kh, kw, stride = ctx.shape
        STATS["hip"] += 1
        w, b = ctx.wb
        if SINK.active and w.requires_grad and b.requires_grad and w.grad is not None and b.grad is not None:
            # the kernel accumulates in the parameter layout: straight into .grad (a frozen stem takes the path below and
            # autograd drops what was not asked for)
            lib.conv2d_wgrad_small(xin, True, 0, C, _rows(gy), 0, cout, w.grad, b.grad, kh, kw, stride, B, Ho, Wo)
            return None, None, None, None
        dw = torch.zeros(cout, C, kh, kw, device=gy.device)
        db = torch.zeros(cout, device=gy.device)
        lib.conv2d_wgrad_small(xin, True, 0, C, _rows(gy), 0, cout, dw, db, kh, kw, stride, B, Ho, Wo)
        return None, dw, db, None


class HipFrozenBnAct(torch.autograd.Function):
    """nn.BatchNorm2d with frozen statistics (freeze_bn, train_flow.py:107-108: y = x * s + t with s = gamma * rstd,
    t = beta - mean * s) [+ the ReLU behind it] in one launch forward (pf_bn_frozen_fwd) and three backward (pf_bn_frozen_bwd:
    masked sums -> d gamma / d beta, dx), instead of ~8 + ~15 elementwise PyTorch kernels per layer.  With the GradSink on,
    d gamma / d beta are added straight into the parameters' .grad."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, relu):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        xr = _rows(x.detach())
        out = torch.empty_like(xr)
        lib.bn_frozen_fwd(xr, gamma.detach(), beta.detach(), mean, var, eps, relu, out)
        ctx.save_for_backward(xr, mean, var)
        ctx.gb = (gamma, beta)
        ctx.cfg = (x.shape, float(eps), bool(relu))
        STATS["hip"] += 1
        return _nchw(out, B, H, W)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xr, mean, var = ctx.saved_tensors
        gamma, beta = ctx.gb
        (B, Cc, H, W), eps, relu = ctx.cfg
        dx = torch.empty_like(xr)
        sink = SINK.active and gamma.grad is not None and beta.grad is not None
        dgamma = gamma.grad if sink else torch.empty_like(gamma)
        dbeta = beta.grad if sink else torch.empty_like(beta)
        lib.bn_frozen_bwd(_rows(g), xr, gamma.detach(), beta.detach(), mean, var, eps, relu, dx, dgamma, dbeta, sink)
        STATS["hip"] += 3
        if sink:
            return _nchw(dx, B, H, W), None, None, None, None, None, None
        return _nchw(dx, B, H, W), dgamma, dbeta, None, None, None, None


class HipAddRelu(torch.autograd.Function):
    """relu(x + y) of a ResidualBlock (core/extractor.py:47) in one launch (pf_add_relu); backward = one pf_relu_mask, the same
    gradient for both inputs.  Works on the storage order of its (identically laid out) inputs."""

    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        xr, yr = _rows(x.detach()), _rows(y.detach())
        out = torch.empty_like(xr)
        lib.add_relu(xr, yr, out)
        ctx.save_for_backward(out)
        ctx.shape = x.shape
        STATS["hip"] += 1
        B, Cc, H, W = x.shape
        return _nchw(out, B, H, W)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        dx = torch.empty_like(out)
        lib.relu_mask(_rows(g), out, dx)
        STATS["hip"] += 1
        d = _nchw(dx, B, H, W)
        return d, d


class HipBatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in training mode (batch statistics; the reference's `chairs` stage leaves BatchNorm unfrozen,
    train_flow.py:107-108).  The statistics run over the whole batch, i.e. the channel-last rows [B*H*W][C] are ONE image of
    B*H*W pixels for pf_channel_stats (forward) and for pf_norm_bwd's InstanceNorm branch (backward:
    dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))); d gamma = sum g * xhat and d beta = sum g come from the same
    kernel's per-channel sums.  The module's running statistics are updated like torch does (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, module):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        n = B * H * W
        xr = _rows(x.detach())
        scale = torch.empty(1, Cc, dtype=torch.float32, device=x.device)       # rstd
        shift = torch.empty(1, Cc, dtype=torch.float32, device=x.device)       # -mean * rstd
        nblk = 128
        part = torch.empty(nblk * Cc * 2, dtype=torch.float64, device=x.device)
        lib.channel_stats(xr, 1, n, Cc, scale, shift, part, nblk, eps=module.eps)
        STATS["hip"] += 1
        with torch.no_grad():
            if module.track_running_stats and module.running_mean is not None:
                mean = -shift[0] / scale[0]
                var = 1.0 / (scale[0] * scale[0]) - module.eps
                mom = module.momentum if module.momentum is not None else 1.0 / float(module.num_batches_tracked + 1)
                module.running_mean.mul_(1.0 - mom).add_(mom * mean)
                module.running_var.mul_(1.0 - mom).add_(mom * var * (n / max(n - 1, 1)))
                module.num_batches_tracked += 1
        g_ = gamma.detach()
        ctx.save_for_backward(xr, scale, shift, g_)
        ctx.shape = x.shape
        s = (g_ * scale[0]).view(1, -1, 1, 1)
        t = (beta.detach() + g_ * shift[0]).view(1, -1, 1, 1)
        return x * s + t

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xr, scale, shift, gamma = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        n = B * H * W
        gr = _rows(g)
        coef = lib.norm_bwd_sums(gr, xr, scale, shift, 1, n, Cc)              # [1, C, 2]: mean g, mean g * xhat
        dx = torch.empty_like(xr)
        lib.norm_bwd((gr * gamma.view(1, Cc)).contiguous(), xr, scale, shift, False, True, dx, 1, n, Cc)
        STATS["hip"] += 2
        return _nchw(dx, B, H, W), coef[0, :, 1] * n, coef[0, :, 0] * n, None


_TAPE = threading.local()       # .gates: id(conv module) -> (token, WeightGrad) of the forward being recorded


def conv2d(x: torch.Tensor, m: nn.Conv2d) -> torch.Tensor:
    """nn.Conv2d.forward of module ``m`` on the HIP pair that covers its geometry (PfError when none does)."""
    kh, kw = m.kernel_size
    hip = (m.stride == (1, 1) and (kh, kw) in ((3, 3), (1, 5), (5, 1), (1, 1)) and m.padding == (kh // 2, kw // 2)
           and x.shape[1] % 4 == 0 and m.bias is not None and m.dilation == (1, 1) and m.groups == 1)
    if hip:
        gates = getattr(_TAPE, "gates", None)
        if gates is None or not m.weight.requires_grad:
            return HipConv.apply(x, m.weight, m.bias)
        g = gates.get(id(m))
        if g is None:
            acc = WeightGrad()
            g = gates[id(m)] = (WeightGate.apply(acc, m.weight, m.bias), acc)
        return HipConv.apply(x, m.weight, m.bias, None, g[0], g[1])
    plain = m.bias is not None and m.dilation == (1, 1) and m.groups == 1 and m.padding == (kh // 2, kw // 2)
    if plain and m.stride == (2, 2) and (kh, kw) in ((3, 3), (1, 1)) and x.shape[1] % 4 == 0 \
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and m.weight.shape[0] % 4 == 0:
        return HipConvS2.apply(x, m.weight, m.bias)
    if plain and kh == kw and kh * kw <= 52 and x.shape[1] <= 4 and m.stride in ((1, 1), (2, 2)) \
            and m.weight.shape[0] % 4 == 0 and not x.requires_grad:
        return HipSmallConv.apply(x, m.weight, m.bias, m.stride[0])
    # No second backend inside the product path: a geometry none of the three HIP pairs covers is an error, not a silent
    # PyTorch-ROCm convolution (every nn.Conv2d of PriOr-RAFT is covered: tests/test_hip_train_step.py).
    raise _lib.PfError(f"training forward: no HIP kernel pair for Conv2d(kernel={m.kernel_size}, stride={m.stride}, "
                       f"padding={m.padding}, dilation={m.dilation}, groups={m.groups}, bias={m.bias is not None}) on "
                       f"{tuple(x.shape)} input")


def gate_of(m: nn.Conv2d):
    """(token, WeightGrad) of convolution module ``m`` for the forward being recorded (created on first use): the
    accumulator the weight gradient of every use of ``m`` goes to, and the token that orders its WeightGate's backward
    after all of them."""
    gates = _TAPE.gates
    g = gates.get(id(m))
    if g is None:
        acc = WeightGrad()
        g = gates[id(m)] = (WeightGate.apply(acc, m.weight, m.bias), acc)
    return g


class PyramidGrad:
    """Gradient accumulator of one branch's pyramid for one backward pass.  Every DCCL lookup of every iteration
    scatters straight into these four buffers (pf_dccl_lookup_bwd accumulates) instead of handing autograd a
    pyramid-sized tensor per call to add up: at 384x512 that was 8 zero fills and 8 adds of up to 38 MB per lookup."""

    def __init__(self):
        self.g: Optional[List[torch.Tensor]] = None

    def buffers(self, like: List[torch.Tensor]) -> List[torch.Tensor]:
        if self.g is None:
            # ONE zero fill for the four levels (round 6; a fill per level before: 8 launches per step)
            flat = torch.zeros(sum(t.numel() for t in like), dtype=like[0].dtype, device=like[0].device)
            self.g, o = [], 0
            for t in like:
                self.g.append(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
        return self.g

    def take(self) -> Optional[List[torch.Tensor]]:
        g, self.g = self.g, None
        return g


class HipCorrPyramid(torch.autograd.Function):
    """corr + build_pyramid (core/prior_raft.py:69-75, core/corr.py:99-111).  Returns the four levels
    ([B*N, H_i*W_i] rows, not differentiable as tensors) and a one-element token: the lookups take the token as an
    input, which orders this node's backward after all of theirs, and leave the levels' gradient in ``acc``."""

    @staticmethod
    def forward(ctx, f1, f2, acc):
        lib = _lib.load()
        B, C, H, W = f1.shape
        n = H * W
        r1, r2 = _rows(f1.detach()), _rows(f2.detach())
        sp = [lib.split_bf16(r, torch.empty(B * n, C // 32, 2, 32, dtype=torch.bfloat16, device=f1.device))
              for r in (r1, r2)]
        lv = [torch.empty(B * n, (H >> i) * (W >> i), device=f1.device) for i in range(4)]
        lib.corr_pyramid_bf16x3(sp[0], sp[1], lv, B, H, W, C)
        ctx.save_for_backward(r1, r2)
        ctx.shape = (B, C, H, W)
        ctx.acc = acc
        ctx.mark_non_differentiable(*lv)
        ctx.set_materialize_grads(False)         # backward ignores its arguments: no zero tensors of the levels' sizes (8 fills per step)
        STATS["hip"] += 1
        return (*lv, torch.zeros(1, device=f1.device))

    @staticmethod
    def backward(ctx, *_):
        lib = _lib.load()
        r1, r2 = ctx.saved_tensors
        B, C, H, W = ctx.shape
        n = H * W
        gl = ctx.acc.take()
        if gl is None:                       # no lookup contributed a gradient
            return torch.zeros(B, C, H, W, device=r1.device), torch.zeros(B, C, H, W, device=r1.device), None
        dv = lib.pyramid_bwd(gl, B, H, W).view(B, n, n)       # level 0 becomes the dense volume gradient, in place
        s = 1.0 / math.sqrt(C)
        # d f1 = dV f2 / sqrt(C), d f2 = dV^T f1 / sqrt(C): two GEMMs per image on the MFMA 1x1 convolution -- rows = the n
        # pixels of a feature map, "input channels" = the n pixels of the other one, weights = the other feature map^T
        d1, d2 = torch.empty(B * n, C, device=r1.device), torch.empty(B * n, C, device=r1.device)
        n4 = (n + 3) // 4 * 4                                 # the kernel reads 16 bytes at a time: K padded with zero columns
        for b in range(B):
            dvb = dv[b]
            dvt = dvb.t()
            for src, feat, dst in ((dvb, r2, d1), (dvt, r1, d2)):
                src = F.pad(src, (0, n4 - n)) if n4 != n else src.contiguous()
                # the operand [Cout = C][Cin = n] is the other feature map TRANSPOSED: its rows [n][C], read as the weights of a
                # 1x1 convolution n <- C, are packed by the device-side packer in its data-gradient mode (packed Cout = C, packed
                # Cin = n rounded up: transpose, hi | lo split and zero padding in one launch; round 6 -- the torch packing
                # before it cost ~8 launches per GEMM: a transposing copy, two fills, a copy and the split's elementwise kernels)
                wp, bp = lib.pack_conv_weights(feat[b * n:(b + 1) * n].view(n, C, 1, 1), None, mode=1)
                cv = Conv(wp, bp, 1, 1, n4, C, PREC_BF16X3, presplit=True)
                lib.conv2d([cv.desc(src, 0, n4, dst[b * n:(b + 1) * n], 0, EPI_LINEAR, scale=s)], 1, H, W, src)
        STATS["hip"] += 1 + 4 * B
        return _nchw(d1, B, H, W), _nchw(d2, B, H, W), None


class HipDccl(torch.autograd.Function):
    """DCCL.__call__ (core/corr.py:113-144), own + rotated-back cross lookup summed (prior_raft.py:187-188).
    Pyramid levels are [B*N, H_i*W_i] rows; the coordinates carry no gradient (detached, prior_raft.py:171,176).
    ``own`` / ``other`` are (levels, token, PyramidGrad) triples of HipCorrPyramid."""

    @staticmethod
    def forward(ctx, coords, g_w2c, g_back, tok_own, tok_other, own, other):
        lib = _lib.load()
        B, _, H, W = coords.shape
        n = H * W
        own_out, raw, out = (torch.empty(B * n, 324, device=coords.device) for _ in range(3))
        co = coords.detach().contiguous()
        lib.dccl_lookup(co, own[0], other[0], g_w2c, own_out, raw)
        lib.dccl_combine(own_out, raw, g_back, out, B, H, W)
        ctx.save_for_backward(co, g_w2c, g_back)
        ctx.dims = (B, H, W)
        ctx.own, ctx.other = own, other
        STATS["hip"] += 2
        return _nchw(out, B, H, W)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        co, g_w2c, g_back = ctx.saved_tensors
        B, H, W = ctx.dims
        n = H * W
        d_corr = _rows(g)
        d_raw = torch.zeros(B * n, 324, device=g.device)
        lib.dccl_combine_bwd(d_corr, g_back, d_raw, B, H, W)
        lib.dccl_lookup_bwd(co, g_w2c, d_corr, d_raw, ctx.own[2].buffers(ctx.own[0]), ctx.other[2].buffers(ctx.other[0]))
        STATS["hip"] += 2
        zero = torch.zeros(1, device=g.device)
        return None, None, None, zero, zero, None, None


class SplitBatch(torch.autograd.Function):
    """x[k * B:(k + 1) * B] for k = 0 .. n - 1 as n outputs (fnet's batch [f1_A | f2_A | f1_B | f2_B], core/prior_raft.py:144-149).
    As plain slices autograd's backward was, per slice, a zero fill of the whole batch and a copy, then n - 1 adds; here it is one
    concatenation (round 6: 11 PyTorch kernels -> 1)."""

    @staticmethod
    def forward(ctx, x, n):
        B = x.shape[0] // n
        ctx.piece = (B,) + tuple(x.shape[1:])
        return tuple(x[k * B:(k + 1) * B] for k in range(n))

    @staticmethod
    def backward(ctx, *gs):
        like = next(g for g in gs if g is not None)
        return torch.cat([g if g is not None else like.new_zeros(ctx.piece) for g in gs], 0), None


class ContextSplit(torch.autograd.Function):
    """cnet's output [im1 | im1_B] x 256 channels -> net = tanh(first 128), inp = relu(last 128) of both views
    (core/prior_raft.py:135-142), four outputs.  Written out with slices and torch.tanh / torch.relu the backward was four
    zero fills, four copies, three adds and four elementwise kernels; here: two concatenations and two elementwise products
    into one gradient tensor (round 6)."""

    @staticmethod
    def forward(ctx, cnet, B):
        t = torch.tanh(cnet[:, :128])
        r = torch.relu(cnet[:, 128:])
        ctx.save_for_backward(t, r)
        ctx.B = B
        return t[:B], r[:B], t[B:], r[B:]

    @staticmethod
    def backward(ctx, g_net_a, g_inp_a, g_net_b, g_inp_b):
        t, r = ctx.saved_tensors
        z = lambda g, like: g if g is not None else torch.zeros_like(like)          # noqa: E731
        B = ctx.B
        g = torch.empty(t.shape[0], 256, t.shape[2], t.shape[3], dtype=t.dtype, device=t.device)
        gt = torch.cat([z(g_net_a, t[:B]), z(g_net_b, t[B:])], 0)
        gr = torch.cat([z(g_inp_a, r[:B]), z(g_inp_b, r[B:])], 0)
        torch.addcmul(gt, gt * t, t, value=-1.0, out=g[:, :128])                     # g * (1 - t^2) = g - (g t) t
        torch.mul(gr, (r > 0).to(gr.dtype), out=g[:, 128:])
        return g, None


def corr_pyramid(f1: torch.Tensor, f2: torch.Tensor):
    """-> (levels, token, PyramidGrad) of one branch."""
    acc = PyramidGrad()
    *lv, tok = HipCorrPyramid.apply(f1, f2, acc)
    return list(lv), tok, acc


def dccl(coords, g_w2c, g_back, own, other) -> torch.Tensor:
    return HipDccl.apply(coords, g_w2c, g_back, own[1], other[1], own, other)


class HipUpsample(torch.autograd.Function):
    """upsample_flow (core/prior_raft.py:58-67) of flow = coords1 - coords0 with the 0.25-scaled mask logits."""

    @staticmethod
    def forward(ctx, flow, mask, coords0):
        lib = _lib.load()
        B, _, H, W = flow.shape
        coords1 = (coords0 + flow.detach()).contiguous()
        mrows = _rows(mask.detach())
        out = torch.empty(B, 2, 8 * H, 8 * W, device=flow.device)
        lib.upsample_flow(coords1, mrows, out)
        ctx.save_for_backward(coords1, mrows)
        STATS["hip"] += 1
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        coords1, mrows = ctx.saved_tensors
        B, _, H, W = coords1.shape
        d_mask = torch.empty(B * H * W, 576, device=g.device)
        d_flow = torch.zeros(B, 2, H, W, device=g.device)
        lib.upsample_flow_bwd(coords1, mrows, g.contiguous(), d_mask, d_flow)
        STATS["hip"] += 1
        return d_flow, _nchw(d_mask, B, H, W), None


class HipWarpGcorr(torch.autograd.Function):
    """cycle_bilinear_sampler + groupwise_corr (core/prior_raft.py:173-174, :77-83); coords carry no gradient."""

    @staticmethod
    def forward(ctx, f1, f2, coords):
        lib = _lib.load()
        B, C, H, W = f1.shape
        r1, r2, co = _rows(f1.detach()), _rows(f2.detach()), coords.detach().contiguous()
        out = torch.empty(B * H * W, 4, device=f1.device)
        lib.warp_gcorr(r1, r2, co, False, out, 0)
        ctx.save_for_backward(r1, r2, co)
        STATS["hip"] += 1
        return _nchw(out, B, H, W)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r1, r2, co = ctx.saved_tensors
        B, _, H, W = co.shape
        d1, d2 = torch.zeros_like(r1), torch.zeros_like(r2)
        lib.warp_gcorr_bwd(r1, r2, co, False, _rows(g), 0, d1, d2)
        STATS["hip"] += 1
        return _nchw(d1, B, H, W), _nchw(d2, B, H, W), None


class HipGruGates(torch.autograd.Function):
    """z = sigmoid(az), r = sigmoid(ar), rh = r * h (core/update.py:49-51, :56-58) on the output [az | ar] of the
    fused z|r convolution; backward = pf_gru_zr_bwd, which writes the gradient of [az | ar] in one piece."""

    @staticmethod
    def forward(ctx, azr, h):
        Cc = h.shape[1]
        zr = torch.sigmoid(azr)
        z, r = zr[:, :Cc], zr[:, Cc:]
        zr_rows = _rows(zr)
        ctx.save_for_backward(zr_rows, _rows(h))
        ctx.shape = h.shape
        return z, r * h

    @staticmethod
    def backward(ctx, dz, d_rh):
        lib = _lib.load()
        zr, h = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        dzr = torch.empty(B * H * W, 2 * Cc, device=dz.device)
        dh = torch.zeros(B * H * W, Cc, device=dz.device)
        lib.gru_zr_bwd(_rows(dz), _rows(d_rh), zr[:, :Cc], zr[:, Cc:], h, dzr, dh)
        STATS["hip"] += 1
        return _nchw(dzr, B, H, W), _nchw(dh, B, H, W)


class HipGruBlend(torch.autograd.Function):
    """q = tanh(aq), h' = (1 - z) * h + z * q (core/update.py:52-53, :59-60); backward = pf_gru_q_bwd."""

    @staticmethod
    def forward(ctx, z, aq, h):
        q = torch.tanh(aq)
        ctx.save_for_backward(_rows(z), _rows(q), _rows(h))
        ctx.shape = z.shape
        return (1 - z) * h + z * q

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        z, q, h = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        dq_pre, dz, dh = (torch.empty(B * H * W, Cc, device=g.device) for _ in range(3))
        lib.gru_q_bwd(_rows(g), z, q, h, dq_pre, dz, dh)
        STATS["hip"] += 1
        return _nchw(dz, B, H, W), _nchw(dq_pre, B, H, W), _nchw(dh, B, H, W)


class HipInstanceNorm(torch.autograd.Function):
    """nn.InstanceNorm2d without affine / running statistics (core/extractor.py:112-113), optionally with the ReLU that
    follows it in BasicEncoder / ResidualBlock (core/extractor.py:41-42,144-146) in the same pass: statistics by
    pf_channel_stats (deterministic two-stage fp64 reduction), y = [relu]((x - mean) * rstd) by pf_norm_act (ReLU form) or one
    fused multiply-add; backward = pf_norm_bwd (its `relu` flag masks by the sign of the normalised value)."""

    @staticmethod
    def forward(ctx, x, relu=False):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        xr = _rows(x.detach())
        scale = torch.empty(B, Cc, dtype=torch.float32, device=x.device)        # rstd
        shift = torch.empty(B, Cc, dtype=torch.float32, device=x.device)        # -mean * rstd
        nblk = 128
        part = torch.empty(B * nblk * Cc * 2, dtype=torch.float64, device=x.device)
        lib.channel_stats(xr, B, H * W, Cc, scale, shift, part, nblk, eps=1e-5)
        ctx.save_for_backward(xr, scale, shift)
        ctx.shape, ctx.relu = x.shape, bool(relu)
        STATS["hip"] += 1
        if relu:
            out = torch.empty_like(xr)
            lib.norm_act(xr, scale, shift, out, B, H * W, Cc)
            STATS["hip"] += 1
            return _nchw(out, B, H, W)
        return torch.addcmul(shift.view(B, Cc, 1, 1), x.detach(), scale.view(B, Cc, 1, 1))

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xr, scale, shift = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        dx = torch.empty_like(xr)
        lib.norm_bwd(_rows(g), xr, scale, shift, ctx.relu, True, dx, B, H * W, Cc)
        STATS["hip"] += 1
        return _nchw(dx, B, H, W), None


# ---- module forwards (the parameter containers of modules.py carry no arithmetic of their own) ---------------

def _norm(m: nn.Module, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """norm layer `m` on x; relu=True: followed by the ReLU (one pass for InstanceNorm)."""
    if isinstance(m, nn.InstanceNorm2d):
        return HipInstanceNorm.apply(x, relu)
    if isinstance(m, nn.BatchNorm2d) and not m.training and m.weight is not None:
        # frozen statistics (freeze_bn): the reference's configuration
        return HipFrozenBnAct.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.eps, relu)
    if relu:
        return torch.relu(_norm(m, x))
    if isinstance(m, nn.BatchNorm2d):
        if m.training and m.weight is not None:               # batch statistics (the `chairs` stage)
            return HipBatchNormTrain.apply(x, m.weight, m.bias, m)
        raise _lib.PfError("BatchNorm2d without affine parameters has no HIP path (the reference's encoders always carry them)")
    raise _lib.PfError(f"unsupported norm layer {type(m).__name__}")


def encoder_forward(enc, x: torch.Tensor) -> torch.Tensor:
    """BasicEncoder.forward (core/extractor.py:136-158) on one concatenated batch."""
    x = _norm(enc.norm1, conv2d(x, enc.conv1), relu=True)
    for layer in (enc.layer1, enc.layer2, enc.layer3):
        for blk in layer:                                     # ResidualBlock.forward (core/extractor.py:39-47)
            y = _norm(blk.norm1, conv2d(x, blk.conv1), relu=True)
            y = _norm(blk.norm2, conv2d(y, blk.conv2), relu=True)
            if blk.downsample is not None:
                x = _norm(blk.norm3, conv2d(x, blk.downsample[0]))
            x = HipAddRelu.apply(x, y)
    x = conv2d(x, enc.conv2)
    if enc.training and enc.dropout is not None:
        x = enc.dropout(x)
    return x


def fuse_zr(gru, need_cat: bool = True) -> Dict[str, tuple]:
    """convz | convr of a SepConvGRU half share their input (core/update.py:48-49, :55-56): one convolution with
    the output channels concatenated.  The gradient comes back through ONE WeightGate over the four parameters, which splits
    the accumulated packed gradient by output-channel range.  Entry: (w, b, source key, token, WeightGrad, (convz, convr));
    w / b (the concatenated tensors) only when `need_cat` -- the per-node tape (HipConv) takes them, the loop node packs
    straight from the two modules."""
    out = {}
    for tag in ("1", "2"):
        cz, cr = getattr(gru, "convz" + tag), getattr(gru, "convr" + tag)
        w = b = None
        if need_cat:
            with torch.no_grad():
                w, b = torch.cat([cz.weight, cr.weight], 0), torch.cat([cz.bias, cr.bias], 0)
        acc = WeightGrad()
        tok = WeightGate.apply(acc, cz.weight, cz.bias, cr.weight, cr.bias)
        out[tag] = (w, b, _src_key(cz.weight, cr.weight, cz.bias, cr.bias), tok if tok.requires_grad else None, acc, (cz, cr))
    return out


def sepconv_gru(gru, zr: Dict[str, tuple], h: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """SepConvGRU.forward (core/update.py:45-60)."""
    for tag in ("1", "2"):
        hx = torch.cat([h, x], 1)
        z, rh = HipGruGates.apply(HipConv.apply(hx, *zr[tag][:5]), h)
        h = HipGruBlend.apply(z, conv2d(torch.cat([rh, x], 1), getattr(gru, "convq" + tag)), h)
    return h


def _heads(blk, net: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    delta = conv2d(torch.relu(conv2d(net, blk.flow_head.conv1)), blk.flow_head.conv2)        # update.py:13-14
    mask = 0.25 * conv2d(torch.relu(conv2d(net, blk.mask[0])), blk.mask[2])                  # update.py:134,157
    return mask, delta


def update_block_b(blk, zr, net, inp, corr, flow):
    """BasicUpdateBlock.forward + BasicMotionEncoder.forward (core/update.py:91-99, :129-136)."""
    e = blk.encoder
    cor = torch.relu(conv2d(torch.relu(conv2d(corr, e.convc1)), e.convc2))
    flo = torch.relu(conv2d(torch.relu(conv2d(flow, e.convf1)), e.convf2))
    out = torch.relu(conv2d(torch.cat([cor, flo], 1), e.conv))
    net = sepconv_gru(blk.gru, zr, net, torch.cat([inp, out, flow], 1))
    mask, delta = _heads(blk, net)
    return net, mask, delta


def update_block_a(blk, zr, net, inp, flow_a, corr_a, flaw_a, flow_ba, flaw_ba):
    """BasicMultiUpdateBlock.forward + BasicMultiMotionEncoder.forward (core/update.py:152-159, :183-201)."""
    e = blk.encoder
    cor = torch.relu(conv2d(torch.relu(conv2d(corr_a, e.convc1_A)), e.convc2_A))
    flo_a = torch.relu(conv2d(torch.relu(conv2d(flow_a, e.convf1_A)), e.convf2_A))
    flo_b = torch.relu(conv2d(torch.relu(conv2d(flow_ba, e.convf1_B)), e.convf2_B))
    conf = torch.relu(conv2d(torch.relu(conv2d(torch.cat([flaw_a, flaw_ba], 1), e.conv_conf1)), e.conv_conf2))
    out = torch.relu(conv2d(torch.cat([cor, flo_a, flo_b, conf], 1), e.conv_A))
    net = sepconv_gru(blk.gru, zr, net, torch.cat([inp, out, flow_a, flow_ba], 1))
    mask, delta = _heads(blk, net)
    return net, mask, delta


_GRIDS: Dict[tuple, tuple] = {}


def _grids(H: int, W: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """grid(R_A2B) at full size and grid(R_A2B), grid(R_B2A) at 1/8 (core/prior_raft.py:115-125; the W2C grids
    are these same two: grid(R^T_A2B) == grid(R_B2A) bit for bit)."""
    key = (H, W, str(device))
    if key not in _GRIDS:
        lib = _lib.load()
        g_a2b = torch.empty(2, H, W, device=device)
        g_a2b_8 = torch.empty(2, H // 8, W // 8, device=device)
        g_b2a_8 = torch.empty(2, H // 8, W // 8, device=device)
        lib.sample_grid(g_a2b, rotation_x(-math.pi / 2))
        lib.sample_grid(g_a2b_8, rotation_x(-math.pi / 2))
        lib.sample_grid(g_b2a_8, rotation_x(math.pi / 2))
        _GRIDS[key] = (g_a2b, g_a2b_8, g_b2a_8)
    return _GRIDS[key]


_COORDS0: dict = {}


def _coords0(B: int, H8: int, W8: int, device) -> torch.Tensor:
    """coords_grid (core/utils/utils.py:50-56) -- constant per shape: built once (it was two aranges, a cat and a copy per step)."""
    key = (B, H8, W8, str(device))
    c = _COORDS0.get(key)
    if c is None:
        if len(_COORDS0) > 16:
            _COORDS0.clear()
        xs = torch.arange(W8, device=device, dtype=torch.float32).view(1, 1, 1, W8).expand(B, 1, H8, W8)
        ys = torch.arange(H8, device=device, dtype=torch.float32).view(1, 1, H8, 1).expand(B, 1, H8, W8)
        c = _COORDS0[key] = torch.cat([xs, ys], 1).contiguous()
    return c


def train_forward(model, image1: torch.Tensor, image2: torch.Tensor, iters: int = 12,
                  init_flow: Optional[torch.Tensor] = None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """PriOr_RAFT.forward(test_mode=False) with an autograd graph (core/prior_raft.py:107-215)."""
    lib = _lib.load()
    if not image1.is_cuda:
        raise _lib.PfError("train_forward needs inputs on a cuda/ROCm device; there is no CPU path")
    B, _, H, W = image1.shape
    if H % 8 or W % 8 or H < 128 or W < 128:
        raise _lib.PfError(f"image size {H}x{W}: H and W must be multiples of 8 and at least 128")
    dev = image1.device
    H8, W8 = H // 8, W // 8
    g_a2b, g_a2b_8, g_b2a_8 = _grids(H, W, dev)
    with torch.no_grad():
        # Input stage as in the inference engine (Engine.prepare_images; round 6 -- eleven elementwise / cat / copy kernels of
        # PyTorch's before): 2 * (image / 255) - 1 of both images straight into the encoders' batches img_f = [im1 | im2 |
        # im1_B | im2_B] and img_c = [im1 | im1_B] (numpy's IEEE arithmetic bit for bit) and img_rotate of [im1 | im2] as 2B
        # three-channel images into the other half of fnet's batch (:121-127): one launch, pf_prepare_images.
        img_f = torch.empty(4 * B, 3, H, W, device=dev)
        img_c = torch.empty(2 * B, 3, H, W, device=dev)
        lib.prepare_images(image1.float().contiguous(), image2.float().contiguous(), g_a2b, img_f, img_c)
        coords0 = _coords0(B, H8, W8, dev)                                                      # :50-56

    _TAPE.gates = {}
    # The pack cache serves ONE tape (a convolution that runs 2 x iters times packs once; the backward's data-gradient packs are
    # made on demand): its keys -- parameter address + version + weights epoch -- cannot tell a new model apart whose parameters
    # the allocator placed where a freed model's were, so nothing is carried from one forward to the next.
    _PACKS.clear()
    try:
        return _train_forward_body(model, lib, B, img_c, img_f, coords0, g_a2b_8, g_b2a_8, iters, init_flow)
    finally:
        _TAPE.gates = None


def _prepack_encoders(lib, *encoders):
    """Forward and data-gradient operands of every MFMA convolution of the encoders, packed together (ceil(n / 16) launches
    instead of one per operand in front of its first use); conv2d() / the backward nodes then find them in the pack cache."""
    with lib.batched_packs():
        for enc in encoders:
            for m in enc.modules():
                if isinstance(m, nn.Conv2d) and m.weight.shape[1] % 4 == 0 and m.bias is not None:
                    _pack(m.weight, m.bias, "fwd")
                    _pack(m.weight, None, "dgrad")


def _train_forward_body(model, lib, B, img_c, img_f, coords0, g_a2b_8, g_b2a_8, iters, init_flow):
    import os
    _prepack_encoders(lib, model.cnet, model.fnet)
    # cnet beside fnet on a side stream (round 4): at the training crop an encoder launch fills a fraction of the chip, and
    # autograd runs every backward node on its forward node's stream, so the two encoders' backwards overlap as well
    # (inside train.GraphedTrainStep the fork / join become parallel branches of the captured graph)
    side = getattr(model, "_train_side_stream", None)
    if side is None:
        side = model._train_side_stream = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        cnet = encoder_forward(model.cnet, img_c)                                               # :133-142
        net_a, inp_a, net_b, inp_b = ContextSplit.apply(cnet, B)
    fm = encoder_forward(model.fnet, img_f).float()                                             # :144-149
    torch.cuda.current_stream().wait_stream(side)
    for t in (net_a, inp_a, net_b, inp_b):
        t.record_stream(torch.cuda.current_stream())
    f1a, f2a, f1b, f2b = SplitBatch.apply(fm, 4)
    pyr_a = corr_pyramid(f1a, f2a)                                                              # :151-159
    pyr_b = corr_pyramid(f1b, f2b)

    import os
    use_loop = (os.environ.get("PRIORFLOW_TRAIN_LOOP", "1") != "0" and all(p.requires_grad for p in model.ODDC.parameters())
                and all(p.requires_grad for p in model.update_block.parameters()))
    zr_a, zr_b = fuse_zr(model.ODDC.gru, not use_loop), fuse_zr(model.update_block.gru, not use_loop)
    c1a, c1b = coords0.clone(), coords0.clone()
    if init_flow is not None:                                                                   # :162-165
        with torch.no_grad():
            fl = init_flow.float().contiguous()
            c1a = c1a + fl
            c1b = c1b + lib.flo_rotate(fl, g_b2a_8, g_a2b_8, torch.empty_like(fl))
    if use_loop:
        # round 4: the refinement iterations as ONE autograd node with a hand-written backward and deferred weight gradients
        from .train_loop import run_loop
        return run_loop(model, lib, zr_a, zr_b, gate_of, net_a, net_b, inp_a, inp_b, f1a, f2a, pyr_a, pyr_b, coords0,
                        c1a, c1b, g_a2b_8, g_b2a_8, iters)
    preds_a: List[torch.Tensor] = []
    preds_b: List[torch.Tensor] = []
    for _ in range(iters):
        c1a, c1b = c1a.detach(), c1b.detach()                                                   # :171, :176
        with torch.no_grad():
            flow_a = (c1a - coords0).contiguous()
            flow_b = (c1b - coords0).contiguous()
            flow_ba = lib.flo_rotate(flow_b, g_a2b_8, g_b2a_8, torch.empty_like(flow_b))        # :179
            c_ba = (coords0 + flow_ba).contiguous()
        flaw_a = HipWarpGcorr.apply(f1a, f2a, c1a)                                              # :173-174
        flaw_ba = HipWarpGcorr.apply(f1a, f2a, c_ba)                                            # :181-182
        corr_a = dccl(c1a, g_b2a_8, g_b2a_8, pyr_a, pyr_b)                                      # :185, :187
        corr_b = dccl(c1b, g_a2b_8, g_a2b_8, pyr_b, pyr_a)                                      # :186, :188
        net_a, mask_a, delta_a = update_block_a(model.ODDC, zr_a, net_a, inp_a, flow_a, corr_a, flaw_a, flow_ba, flaw_ba)
        net_b, mask_b, delta_b = update_block_b(model.update_block, zr_b, net_b, inp_b, corr_b, flow_b)
        c1a = c1a + delta_a                                                                     # :193-197
        c1b = c1b + delta_b
        preds_a.append(HipUpsample.apply(c1a - coords0, mask_a, coords0))                       # :200-208
        preds_b.append(HipUpsample.apply(c1b - coords0, mask_b, coords0))
    return preds_a, preds_b
