"""Drop-in ``PriOr_RAFT`` for MI355X: same constructor, ``forward`` signature, return types
and ``state_dict`` as the reference module (PriOr-RAFT/core/prior_raft.py:27-215), with the
inner loop running in hand-written gfx950 kernels (``libpriorflow_hip.so``).

    model = PriOr_RAFT(args).cuda().eval()
    flow  = model(image1, image2, iters=12, test_mode=True)      # [B,2,H,W]

Callers covered: demo.py:11-19, demo_image.py:30-39, evaluate.py:208,266,318,350,382 (inference:
the workspace / HIP-graph engine) and the training caller train_flow.py:131: in ``train()`` mode with
autograd enabled the forward runs ``autograd.train_forward`` -- torch autograd as the tape, the HIP
kernels as the arithmetic in both directions (``autograd.py`` lists what is still a PyTorch-ROCm op).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .engine import EncoderPlan, Engine, Workspace, default_precision, pack_update_blocks
from .modules import BasicEncoder, build_tree, state_dict_shapes  # noqa: F401


class PriOr_RAFT(nn.Module):
    """``args.mixed_precision`` is accepted and has no effect: the reference wraps its encoders and update blocks in CUDA
    autocast (core/prior_raft.py:133,146,190) and scales the loss (train_flow.py:112,136-139); here every GEMM-shaped op runs
    the 3-pass bf16 split with fp32 accumulation and storage (or exact fp32, ``PRIORFLOW_PRECISION=fp32``), which needs neither.
    ``args.dropout`` is honoured by the training forward."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.hidden_dim = 128
        self.context_dim = 128
        args.corr_levels = 4          # the reference writes these back (core/prior_raft.py:34-35)
        args.corr_radius = 4
        dropout = float(getattr(args, "dropout", 0.0))
        self.fnet, self.cnet, self.ODDC, self.update_block = build_tree(dropout)
        # non-module state (not part of state_dict)
        self._packed: Optional[Dict[str, object]] = None
        self._packed_sig = None
        self._ws: Dict[Tuple, Workspace] = {}
        self._graphs: Dict[Tuple, object] = {}
        self._enc_plans = None
        self._enc_sig = None
        self.use_graph = os.environ.get("PRIORFLOW_GRAPH", "1") != "0"
        self.precision: Optional[int] = None      # None -> PRIORFLOW_PRECISION env (default bf16x3)
        self.use_streams = os.environ.get("PRIORFLOW_STREAMS", "1") != "0"
        self._side_streams = None

    # ---- reference API surface ----------------------------------------------------------------
    def freeze_bn(self):
        """core/prior_raft.py:43-48."""
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                m.eval()

    def load_things_ckpt(self, ckpt_path):
        """RAFT-things import with the update_block.* -> ODDC.* remap (core/prior_raft.py:85-104)."""
        raw = torch.load(ckpt_path, map_location=torch.device("cpu"))
        ckpt = {k[7:]: v for k, v in raw.items() if k.startswith("module.")}
        state = self.state_dict()
        for key in state.keys():
            if key in ckpt and state[key].shape == ckpt[key].shape:
                state[key] = ckpt[key]
                continue
            alt = key.replace("ODDC", "update_block")
            if ("ODDC" in key and any(s in key for s in (".gru.", ".flow_head.", ".mask."))
                    and alt in ckpt and state[key].shape == ckpt[alt].shape):
                state[key] = ckpt[alt]
            else:
                print(f"Skip loading parameter: {key}, not found in checkpoint")
        self.load_state_dict(state, strict=True)

    # ---- internals ----------------------------------------------------------------------------
    def _lib(self) -> _lib.PfLib:
        return _lib.load()       # raises loudly when the HIP library is missing

    def _weights(self):
        params = list(self.ODDC.parameters()) + list(self.update_block.parameters())
        sig = tuple((p.data_ptr(), p._version) for p in params) + (self.precision, _lib.weights_epoch())
        if self._packed is None or sig != self._packed_sig:
            with torch.no_grad():
                self._packed = pack_update_blocks(self.ODDC, self.update_block, self.precision)
            self._packed_sig = sig
            self._graphs.clear()
        return self._packed

    WS_KEEP = 3                 # shapes kept resident (least recently used goes first); PRIORFLOW_WS_KEEP overrides
    WS_BYTES = 128 << 30        # ... as long as their buffers stay under this many bytes (B = 32 at 512x1024 is ~27 GB)

    def _workspace(self, B, H, W, device) -> Workspace:
        """The buffers of one (B, H, W) problem.  A few shapes stay resident, least recently used first out, together with the
        HIP graphs captured on them: an evaluation loop over mixed sizes (or B = 1 / B = 32 in turn) re-uses its workspaces and
        graphs instead of re-allocating ~1 GB and re-capturing 258 kernels per switch."""
        key = (B, H, W, str(device))
        ws = self._ws.pop(key, None)
        if ws is None:
            ws = Workspace(self._lib(), B, H, W, device)
            ws.nbytes = sum(t.numel() * t.element_size() for t in vars(ws).values() if isinstance(t, torch.Tensor)) + \
                sum(t.numel() * t.element_size() for v in vars(ws).values() if isinstance(v, (list, tuple))
                    for t in v if isinstance(t, torch.Tensor))
        self._ws[key] = ws          # most recently used last
        keep = max(1, int(os.environ.get("PRIORFLOW_WS_KEEP", self.WS_KEEP)))
        # the encoders keep one activation set per resident shape as well (tens of GB at B = 32): they count against the cap
        resident = lambda: sum(w.nbytes for w in self._ws.values()) + sum(p.nbytes() for p in (self._enc_plans or ()))  # noqa: E731
        while len(self._ws) > 1 and (len(self._ws) > keep or resident() > self.WS_BYTES):
            old = next(iter(self._ws))
            del self._ws[old]
            for gk in [gk for gk in self._graphs if (gk[0], gk[1], gk[2], gk[4]) == old]:
                del self._graphs[gk]      # a graph holds pointers into its workspace
            for plan in (self._enc_plans or ()):
                plan.release(old[1], old[2], (2 * old[0], 4 * old[0]))      # ... and into the encoders' activation buffers
        return ws

    def _encoder_plans(self):
        """HIP launch plans of cnet / fnet (either precision), rebuilt when their weights or the precision change."""
        precision = default_precision() if self.precision is None else self.precision
        params = list(self.fnet.parameters()) + list(self.cnet.parameters()) + list(self.cnet.buffers())
        sig = tuple((p.data_ptr(), p._version) for p in params) + (_lib.weights_epoch(), precision)
        if self._enc_plans is None or sig != self._enc_sig:
            with torch.no_grad():
                self._enc_plans = (EncoderPlan(self._lib(), self.cnet, precision),
                                   EncoderPlan(self._lib(), self.fnet, precision))
            self._enc_sig = sig
            self._graphs.clear()
        return self._enc_plans

    def _encode(self, image1, image2, ws: Workspace, eng: Engine, init_flow=None):
        """Normalise, rotate to view B, run cnet / fnet (core/prior_raft.py:109-149) and leave the
        features in the workspace (channel-last): f1A,f2A,f1B,f2B; net = tanh(cnet[:128]),
        inp = relu(cnet[128:]) for both views; then the two corr volumes and their pyramids (:150-158)."""
        from ._lib import EPI_LINEAR, EPI_TANH_RELU
        B = ws.B
        ws.pre_ready = False                        # `inp` is about to be rewritten (Engine.hoist_context)
        if image1 is not None:                      # (None: the caller has run the input stage already -- _run_graph)
            eng.prepare_images(ws, image1, image2)  # normalise + rotate, straight into ws.img_f / ws.img_c
        cplan, fplan = self._encoder_plans()       # both precisions run on the HIP library: there is no PyTorch-ROCm branch
        # context features: net (fp32 + split twin) and inp (first 128 columns of the GRU input x; twin only when the
        # update blocks run on pre-split activations)
        ctx = dict(outs=ws.net0_ab_s, auxs=ws.x_ab_s) if eng.presplit(self._weights()) else dict(aux=ws.x_ab)
        # fnet's last convolution (1x1 128 -> 256) writes the features twice: fp32 rows (the warps' operand) and the bf16 hi|lo
        # rows the corr GEMM multiplies -- the epilogue's twin equals pf_split_bf16 of the fp32 rows bit for bit, and the
        # separate split launch (21-37 us of kernel time between the encoders and the corr build) is gone (round 6; the forward's
        # wall time did not move: 147.1 against 147.2 pairs/s, profiles/r6_ab_head.txt -- the stretch overlaps cnet's tail)
        from ._lib import PREC_BF16X3
        fsplit = dict(outs=ws.f_split) if self._weights()["precision"] == PREC_BF16X3 and fplan.precision == PREC_BF16X3 else {}
        ws.f_split_ready = bool(fsplit)
        P = self._weights()
        if self.use_streams and int(os.environ.get("PRIORFLOW_FORKS", "15")) & 1:
            # cnet and fnet are independent: two queues (inside the HIP graph: parallel branches).  Round 6 order: the calling stream
            # takes cnet and, right behind it, the hoisted context terms of the GRU convolutions (they need only cnet's `inp`); fnet forks from an event.  cnet's
            # chain is short, so the 90 us of hoisted convolutions run inside the encoder phase, which is bound by the SUM of the
            # two encoders' work whatever their order (rounds 4 and 5: profiles/r4_encoder_order.txt), and the two corr + pyramid
            # launches then have the chip to themselves: 134 / 142 -> 90 / 87 us per launch in the replay, first lookup 26 us earlier,
            # 154.0 -> 155.1 pairs/s same box (profiles/r6_ab_cnet_first.txt).  Before, fnet came first, cnet only got CUs at
            # t = 1.3 ms, and the hoisted convolutions shared the chip with the corr build: both were slowed (157 us for a 45 us launch).
            cur = torch.cuda.current_stream()
            s1, s2 = self._streams()[:2]
            ev = torch.cuda.Event()
            ev.record(cur)
            cplan.run(ws.img_c, ws.net0_ab, EPI_TANH_RELU, **ctx)
            eng.hoist_context(ws, P)
            s1.wait_event(ev)
            with torch.cuda.stream(s1):
                fplan.run(ws.img_f, ws.f_all, EPI_LINEAR, **fsplit)
                # the corr volumes + pyramids need fnet's features only: on fnet's queue, right behind its last convolution (the
                # hop back to the calling stream cost 14 us in front of the first corr launch; cnet and the hoisted
                # convolutions are long done by then, so the launches still have the chip to themselves)
                eng.build_pyramids(ws, P["precision"])
            # coords1 = coords0 (+ init_flow): two small copies nobody needs before the first lookup -- on a queue of their own,
            # not in front of cnet's stem (19 us of an otherwise idle chip at the head of every forward)
            s2.wait_event(ev)
            with torch.cuda.stream(s2):
                eng.init_coords(ws, init_flow)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        else:
            eng.init_coords(ws, init_flow)
            cplan.run(ws.img_c, ws.net0_ab, EPI_TANH_RELU, **ctx)
            fplan.run(ws.img_f, ws.f_all, EPI_LINEAR, **fsplit)
            eng.hoist_context(ws, P)               # iteration-invariant part of the GRU convs (pre-split path only)
            eng.build_pyramids(ws, P["precision"])

    def _streams(self):
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
        return self._side_streams

    def _run(self, ws: Workspace, image1, image2, iters, init_flow, test_mode, out_a, out_b, upsample_last=True):
        eng = Engine(self._lib(), self._streams() if self.use_streams else None)
        P = self._weights()
        # (joining cnet in front of the iterations instead of after the encoders -- its tail beside the corr build -- measured
        # 133.1 / 133.3 against 133.7 / 133.3 pairs/s in round 3 and was removed)
        self._encode(image1, image2, ws, eng, init_flow)       # incl. coords1 = coords0 (+ init_flow), the hoisted context terms
        #                                                         and the corr volumes + pyramids
        cur = 0
        for it in range(iters):
            last = it == iters - 1
            if test_mode:
                # dead-work elimination: only the last prediction of branch A is returned
                # (core/prior_raft.py:212-213), so 23 of 24 upsamples, every B mask head and
                # branch B's last update are never observable.
                cur = eng.iteration(ws, P, cur, need_b=not last, mask_a=last, mask_b=False,
                                    defer_b_join=not last)
                if last and upsample_last:
                    eng.upsample(ws, "a", out_a[0])
            else:
                cur = eng.iteration(ws, P, cur, need_b=True, mask_a=True, mask_b=True)
                eng.upsample(ws, "a", out_a[it])
                eng.upsample(ws, "b", out_b[it])

    def forward(self, image1, image2, iters=12, init_flow=None, test_mode=False, flow_init=None):
        """Estimate optical flow between a pair of ERP frames (core/prior_raft.py:107).
        ``flow_init`` is accepted as an alias of ``init_flow`` (BASELINE.json spells it so)."""
        if init_flow is None:
            init_flow = flow_init
        if not image1.is_cuda:
            raise _lib.PfError("PriOr_RAFT (MI355X build) needs inputs on a cuda/ROCm device; "
                               "there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only)")
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            # training caller (train_flow.py:131): autograd tape over the HIP forward / backward kernels
            from .autograd import train_forward
            with torch.cuda.device(image1.device):
                preds_a, preds_b = train_forward(self, image1, image2, iters, init_flow)
            return preds_a[-1] if test_mode else (preds_a, preds_b)
        B, _, H, W = image1.shape
        device = image1.device
        with torch.no_grad(), torch.cuda.device(device):
            ws = self._workspace(B, H, W, device)
            n_out = 1 if test_mode else iters
            out_a = [torch.empty(B, 2, H, W, dtype=torch.float32, device=device) for _ in range(n_out)]
            out_b = [] if test_mode else [torch.empty(B, 2, H, W, dtype=torch.float32, device=device)
                                           for _ in range(n_out)]
            image1 = image1.float().contiguous()
            image2 = image2.float().contiguous()
            if self.use_graph and test_mode and init_flow is None and not self.training:
                return self._run_graph(ws, image1, image2, iters)
            self._run(ws, image1, image2, iters, init_flow, test_mode, out_a, out_b)
        if test_mode:
            return out_a[0]
        return out_a, out_b

    # ---- HIP-graph replay of the whole test_mode forward --------------------------------------
    def _run_graph(self, ws: Workspace, image1, image2, iters):
        # both signature checks clear self._graphs when a parameter (or a cnet BatchNorm statistic) changed in place:
        # the captured launches hold pointers to PACKED copies of the weights, of the encoders' too
        self._weights()
        self._encoder_plans()
        key = (ws.B, ws.H, ws.W, iters, str(image1.device))
        # The input stage (pf_prepare_images: normalise + rotate into the encoders' batches) runs OUTSIDE the graph, straight from the
        # caller's tensors into the workspace, and the capture starts behind it: a replay needs no copy of the images into static
        # buffers (round 6: two copies and their gaps, 16 us per forward).  The last launch, the convex upsampling, runs BEHIND the replay
        # into a fresh tensor for the same reason at the other end: no static output buffer to clone from.
        eng = Engine(self._lib(), None)
        graph = self._graphs.get(key)
        if graph is None:
            scratch_out = [torch.empty(ws.B, 2, ws.H, ws.W, dtype=torch.float32, device=image1.device)]
            # warm-up on a side stream (lazy initialisations -- function attributes, module loads -- happen here)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._run(ws, image1, image2, iters, None, True, scratch_out, [])
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._run(ws, None, None, iters, None, True, None, [], upsample_last=False)
            self._graphs[key] = graph
        eng.prepare_images(ws, image1, image2)
        graph.replay()
        out = torch.empty(ws.B, 2, ws.H, ws.W, dtype=torch.float32, device=image1.device)
        eng.upsample(ws, "a", out)
        return out
