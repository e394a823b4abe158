"""The refinement loop of the TRAINING step as ONE autograd node with a hand-written backward (round 4).

``autograd.train_forward`` keeps torch autograd as the tape of the encoders and of the correlation pyramids, but the
``iters`` refinement iterations (core/prior_raft.py:170-211: flows, warps, DCCL lookups, both update blocks, convex
upsampling) no longer put ~60 nodes per iteration on that tape.  ``LoopFn`` runs them on a preallocated workspace
(``LoopBuffers``: every activation of every iteration is kept, [iters][B*N][C] channel-last rows, because the backward needs
them) and walks the iterations backwards itself:

  * every ReLU lives in a convolution epilogue (forward: PF_EPI_RELU; backward: PF_EPI_MASK = data gradient x ReLU mask),
    the GRU gates in PF_EPI_GRU_ZR / PF_EPI_GRU_Q with ``save_gates`` (r and q are operands of pf_gru_*_bwd), gradient
    accumulation into a tensor with several consumers in PF_EPI_ADD; ``torch.cat`` is a column slice of a wider row buffer;
  * the WEIGHT gradients are deferred: a convolution that runs in all ``iters`` iterations gets ONE pf_conv2d_wgrad launch
    over the iters * B stored (input, output-gradient) images at the end of the backward instead of one launch per
    iteration on a 48 x 64 map (core/update.py's 19 + 17 convolutions: 36 launches instead of 432 per step, each with
    12x the pixels to split over the chip);
  * coords1 is detached at every iteration (core/prior_raft.py:171,176), so an iteration's backward needs only the
    gradient of its two predictions and of the hidden states it hands on.

Reference lines: core/update.py:81-99 (BasicMotionEncoder), :117-136 (BasicUpdateBlock), :139-159 + :162-201 (ODDC),
:35-60 (SepConvGRU), :6-14 (FlowHead); train_flow.py:131-135 (forward + loss.backward()).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import EPI_ADD, EPI_GRU_Q, EPI_GRU_ZR, EPI_LINEAR, EPI_MASK, EPI_RELU
from .engine import Conv


class LoopBuffers:
    """Activations and gradients of all iterations of one (B, H8, W8, iters) problem; allocated once per shape."""

    def __init__(self, B: int, H8: int, W8: int, iters: int, device):
        self.key = (B, H8, W8, iters, str(device))
        self.generation = 0           # bumped by every forward that fills the buffers; a backward checks it still owns them
        N = B * H8 * W8
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)      # noqa: E731
        it = iters
        self.N = N
        for t in "ab":
            s: Dict[str, torch.Tensor] = {}
            # ---- forward activations, one slice per iteration
            s["corr"] = z(it, N, 324)
            s["c1"] = z(it, N, 256)
            s["cat"] = z(it, N, 272 if t == "a" else 256)          # a: [cor 128 | floA 64 | floB 64 | conf 16]; b: [cor 192 | flo 64]
            s["x"] = z(it, N, 256)                                  # GRU input [inp | out | flow tails]
            s["h"] = z(it + 1, N, 128)                              # hidden state entering iteration i; [it] = the last one
            s["h1"] = z(it, N, 128)                                 # after the horizontal half-step
            for k in "12":
                s["z" + k] = z(it, N, 128)
                s["rhr" + k] = z(it, N, 256)                        # [r*h | r]
                s["q" + k] = z(it, N, 128)
            s["fh"] = z(it, N, 256)
            s["mh"] = z(it, N, 256)
            s["mask"] = z(it, N, 576)
            s["delta"] = z(it, N, 4)
            s["c"] = z(it + 1, B, 2, H8, W8)                        # coords1 entering iteration i; [i + 1] = leaving it
            # ---- output gradients of the convolutions (operands of the deferred weight gradients)
            s["d_mask"] = z(it, N, 576)
            s["d_mh"] = z(it, N, 256)
            s["d_delta"] = z(it, N, 4)                              # columns 2, 3 stay zero (Cout 2 padded to 4)
            s["d_fh"] = z(it, N, 256)
            for k in "12":
                s["d_q" + k] = z(it, N, 128)
                s["d_zr" + k] = z(it, N, 256)
            s["d_out"] = z(it, N, 128)                              # 124 / 126 live columns
            s["d_cat"] = z(it, N, 272 if t == "a" else 256)
            s["d_c1"] = z(it, N, 256)
            # ---- scratch of one iteration's backward
            s["F"] = [z(N, 512) for _ in range(4)]                  # [d(r*h) | d x | d h] of a half-step; two pairs, ping-pong
            s["dz"] = z(N, 128)
            s["d_corr"] = z(N, 324)
            s["d_raw"] = z(N, 324)
            s["d_flow"] = z(iters, B, 2, H8, W8)          # accumulated into by pf_upsample_flow_bwd: one zero fill per backward
            s["own"], s["raw"] = z(N, 324), z(N, 324)
            setattr(self, t, s)
        a, b = self.a, self.b
        a["flow4"] = z(it, N, 4)                                    # [flow_A | flow_B_A]
        a["t_a"], a["t_ba"] = z(it, N, 128), z(it, N, 128)          # 7x7 stem outputs
        a["conf_in"], a["cf1"] = z(it, N, 8), z(it, N, 32)
        a["d_t_a"], a["d_t_ba"], a["d_cf1"] = z(it, N, 128), z(it, N, 128), z(it, N, 32)
        a["d_conf"] = z(N, 8)
        a["flow_ba"] = z(B, 2, H8, W8)
        b["flow2"] = z(it, N, 2)                                    # flow_B (pf_motion_prep writes 2-float rows)
        b["t"], b["d_t"] = z(it, N, 128), z(it, N, 128)


def _side_stream(model) -> torch.cuda.Stream:
    s = getattr(model, "_loop_side_stream", None)
    if s is None:
        s = model._loop_side_stream = torch.cuda.Stream()
    return s


def _b_stream(model) -> torch.cuda.Stream:
    s = getattr(model, "_loop_b_stream", None)
    if s is None:
        s = model._loop_b_stream = torch.cuda.Stream()
    return s


def _split_on() -> bool:
    """Round 6: branch A's and branch B's chains of the loop on two queues (forward and backward), as in the inference engine's
    Engine.iteration_split.  PRIORFLOW_TRAIN_SPLIT=0: both branches as groups of one chain of launches (round 4)."""
    return os.environ.get("PRIORFLOW_TRAIN_SPLIT", "1") != "0"


def _flat(t: torch.Tensor) -> torch.Tensor:
    """[iters][N][C] -> [iters * N][C] (the iterations as extra images of the deferred weight gradient)."""
    return t.view(-1, t.shape[-1])


class _Packs:
    """Forward / data-gradient packs of the update blocks' convolutions for one forward (cached per weight version by
    autograd._pack) plus the (token, WeightGrad) pair of each."""

    def __init__(self, model, zr_a, zr_b, gate_of):
        from .autograd import _pack
        self.fwd: Dict[str, Conv] = {}
        self.dg: Dict[str, Conv] = {}
        self.acc: Dict[str, object] = {}
        self.mods: Dict[str, object] = {}
        ea, eb = model.ODDC.encoder, model.update_block.encoder
        named = {"a.c1": ea.convc1_A, "a.c2": ea.convc2_A, "a.f2a": ea.convf2_A, "a.f2b": ea.convf2_B,
                 "a.cf1": ea.conv_conf1, "a.cf2": ea.conv_conf2, "a.out": ea.conv_A,
                 "b.c1": eb.convc1, "b.c2": eb.convc2, "b.f2": eb.convf2, "b.out": eb.conv}
        for t, blk in (("a", model.ODDC), ("b", model.update_block)):
            named[t + ".q1"], named[t + ".q2"] = blk.gru.convq1, blk.gru.convq2
            named[t + ".fh1"], named[t + ".fh2"] = blk.flow_head.conv1, blk.flow_head.conv2
            named[t + ".m0"], named[t + ".m2"] = blk.mask[0], blk.mask[2]
        with _lib.load().batched_packs():          # ~54 operands: 4 launches
            for name, m in named.items():
                self.mods[name] = m
                self.fwd[name] = _pack(m.weight, m.bias, "fwd")
                self.dg[name] = _pack(m.weight, None, "dgrad")
                self.acc[name] = gate_of(m)[1]
            # fused z|r convolutions: forward pack of the concatenated weights; the data gradient with its OUTPUT channels
            # reordered [x 256 | h 128] (so that d x and d h land next to each other in the half-step's scratch rows)
            for t, zr in (("a", zr_a), ("b", zr_b)):
                for k in "12":
                    _w, _b, src, _tok, acc, (cz, cr) = zr[k]
                    self.fwd[f"{t}.zr{k}"] = _pack(cz.weight, cz.bias, "fwd", src, w1=cr.weight, b1=cr.bias)
                    self.dg[f"{t}.zr{k}"] = _pack(cz.weight, None, "dgrad", src, w1=cr.weight, rot=128)
                    self.acc[f"{t}.zr{k}"] = acc
        self.stems = {"a.f1a": ea.convf1_A, "a.f1b": ea.convf1_B, "b.f1": eb.convf1}
        self.stem_w = {n: (m.weight.detach().permute(2, 3, 1, 0).reshape(-1, m.weight.shape[0]).contiguous(),
                           m.bias.detach().contiguous()) for n, m in self.stems.items()}


class LoopFn(torch.autograd.Function):
    """(net_A, net_B, inp_A, inp_B, f1_A, f2_A, pyramid tokens, stem parameters, weight tokens) -> 2 * iters flow predictions."""

    @staticmethod
    def forward(ctx, cfg, net_a, net_b, inp_a, inp_b, f1a, f2a, tok_pa, tok_pb, *rest):
        from .autograd import _rows, STATS
        lib = _lib.load()
        model, P, pyr_a, pyr_b, coords0, c1a, c1b, g_a2b_8, g_b2a_8, iters = (
            cfg["model"], cfg["packs"], cfg["pyr_a"], cfg["pyr_b"], cfg["coords0"], cfg["c1a"], cfg["c1b"],
            cfg["g_a2b_8"], cfg["g_b2a_8"], cfg["iters"])
        B, _, H8, W8 = coords0.shape
        dev = coords0.device
        bufs: Optional[LoopBuffers] = getattr(model, "_loop_bufs", None)
        if bufs is None or bufs.key != (B, H8, W8, iters, str(dev)):
            bufs = model._loop_bufs = LoopBuffers(B, H8, W8, iters, dev)
        bufs.generation += 1
        A, Bb = bufs.a, bufs.b
        N = bufs.N
        f1r, f2r = _rows(f1a.detach()), _rows(f2a.detach())
        A["h"][0].copy_(_rows(net_a.detach()))
        Bb["h"][0].copy_(_rows(net_b.detach()))
        A["x"][:, :, :128] = _rows(inp_a.detach())                  # the context features enter every iteration's GRU input
        Bb["x"][:, :, :128] = _rows(inp_b.detach())
        A["c"][0].copy_(c1a)
        Bb["c"][0].copy_(c1b)
        H, W = 8 * H8, 8 * W8
        preds_a = [torch.empty(B, 2, H, W, device=dev) for _ in range(iters)]
        preds_b = [torch.empty(B, 2, H, W, device=dev) for _ in range(iters)]

        def cv(*items):
            """One launch: the same-geometry convolutions `items` = (name, x, off0, c0, out, off_out, epilogue, kwargs) as groups."""
            lib.conv2d([P.fwd[n].desc(x, o0, c0, out, oo, epi, **kw) for n, x, o0, c0, out, oo, epi, kw in items], B, H8, W8, items[0][1])

        # Two chains per iteration (round 4, like the inference engine's forks): the flow / confidence stems run on a side stream
        # beside the lookups and the correlation convolutions, and the convex upsampling of iteration i beside iteration i + 1.
        # Eagerly that is host overhead for little; inside train.GraphedTrainStep they are parallel branches of the captured graph,
        # and at the training crop a launch fills a fraction of the chip.
        main, side = torch.cuda.current_stream(), _side_stream(model)
        side.wait_stream(main)
        n_launch = 0
        if _split_on():
            # ---- two chains (round 6).  Per iteration: A's lookup -> combine -> convc1 -> convc2 on the calling stream, B's on its own
            # stream, the flow / confidence chain (pf_motion_prep, 7x7 stems, 3x3s) and the previous iteration's upsampling on the
            # side stream; both motion-encoder output convolutions wait for the flow chain; then each branch's conv -> SepConvGRU ->
            # heads -> coords1 update as launches of ONE group with pf_conv_desc.co_groups = 1.  Capture order = queue assignment
            # (engine.Engine.iteration_split): A's head, B's head (ordered behind A's previous tail so that it is that node's
            # SECOND-captured successor), the side chain (third), then the tails.  B's tail learns of the flow chain through an
            # event recorded on the calling stream: two forked streams must not wait on each other's events under capture.
            sb = _b_stream(model)

            def cv1(name, x, o0, c0, out, oo, epi, **kw):
                d = P.fwd[name].desc(x, o0, c0, out, oo, epi, **kw)
                d.co_groups = 1
                lib.conv2d([d], B, H8, W8, x)

            def head(t, S, pyr_own, pyr_other, g_w2c, i):
                lib.dccl_lookup(S["c"][i], pyr_own[0], pyr_other[0], g_w2c, S["own"], S["raw"])
                lib.dccl_combine(S["own"], S["raw"], g_w2c, S["corr"][i], B, H8, W8)
                cv1(t + ".c1", S["corr"][i], 0, 324, S["c1"][i], 0, EPI_RELU)
                cv1(t + ".c2", S["c1"][i], 0, 256, S["cat"][i], 0, EPI_RELU)

            def tail(t, S, i):
                cv1(t + ".out", S["cat"][i], 0, 272 if t == "a" else 256, S["x"][i], 128, EPI_RELU)
                for k, hin, hout in (("1", "h", "h1"), ("2", "h1", "h")):
                    hi = S[hin][i]
                    ho = S[hout][i + 1] if hout == "h" else S[hout][i]
                    cv1(t + ".zr" + k, hi, 0, 128, S["z" + k][i], 0, EPI_GRU_ZR, in1=S["x"][i], off1=0, c1=256, h=hi,
                       aux=S["rhr" + k][i], save_gates=True)
                    cv1(t + ".q" + k, S["rhr" + k][i], 0, 128, ho, 0, EPI_GRU_Q, in1=S["x"][i], off1=0, c1=256, h=hi, z=S["z" + k][i],
                       aux=S["q" + k][i], save_gates=True)
                cv1(t + ".fh1", S["h"][i + 1], 0, 128, S["fh"][i], 0, EPI_RELU)
                cv1(t + ".fh2", S["fh"][i], 0, 256, S["delta"][i], 0, EPI_LINEAR)
                cv1(t + ".m0", S["h"][i + 1], 0, 128, S["mh"][i], 0, EPI_RELU)
                cv1(t + ".m2", S["mh"][i], 0, 256, S["mask"][i], 0, EPI_LINEAR, scale=0.25)
                lib.coords_add(S["c"][i + 1], S["delta"][i], src=S["c"][i])

            ev_b_end = None
            for i in range(iters):
                ev_start = torch.cuda.Event()
                ev_start.record(main)                       # A's previous tail (iteration 0: everything in front of the loop)
                head("a", A, pyr_a, pyr_b, g_b2a_8, i)
                sb.wait_event(ev_start)
                with torch.cuda.stream(sb):
                    head("b", Bb, pyr_b, pyr_a, g_a2b_8, i)
                side.wait_event(ev_start)
                if ev_b_end is not None:
                    side.wait_event(ev_b_end)
                with torch.cuda.stream(side):
                    lib.motion_prep(A["c"][i], Bb["c"][i], g_a2b_8, g_b2a_8, f1r, f2r, A["flow4"][i], Bb["flow2"][i], A["conf_in"][i],
                                    A["x"][i], 252, Bb["x"][i], 254)
                    for name, src, off, dst in (("a.f1a", A["flow4"][i], 0, A["t_a"][i]), ("a.f1b", A["flow4"][i], 2, A["t_ba"][i]),
                                                ("b.f1", Bb["flow2"][i], 0, Bb["t"][i])):
                        w, bias = P.stem_w[name]
                        lib.conv2d_small(src, False, off, 2, w, bias, dst, 0, 128, 7, 7, 1, True, B, H8, W8)
                    cv(("a.f2a", A["t_a"][i], 0, 128, A["cat"][i], 128, EPI_RELU, {}), ("a.f2b", A["t_ba"][i], 0, 128, A["cat"][i], 192, EPI_RELU, {}),
                       ("b.f2", Bb["t"][i], 0, 128, Bb["cat"][i], 192, EPI_RELU, {}))
                    cv(("a.cf1", A["conf_in"][i], 0, 8, A["cf1"][i], 0, EPI_RELU, {}))
                    cv(("a.cf2", A["cf1"][i], 0, 32, A["cat"][i], 256, EPI_RELU, {}))
                    stems_done = torch.cuda.Event()
                    stems_done.record(side)
                    if i > 0:                               # the previous iteration's predictions (prior_raft.py:200-208)
                        lib.upsample_flow(A["c"][i], A["mask"][i - 1], preds_a[i - 1])
                        lib.upsample_flow(Bb["c"][i], Bb["mask"][i - 1], preds_b[i - 1])
                main.wait_event(stems_done)
                relay = torch.cuda.Event()
                relay.record(main)
                tail("a", A, i)
                sb.wait_event(relay)
                with torch.cuda.stream(sb):
                    tail("b", Bb, i)
                    ev_b_end = torch.cuda.Event()
                    ev_b_end.record(sb)
                n_launch += 5 + 2 + 3 + 5 + 4 + 4 + 4 + 9
            ev_end = torch.cuda.Event()
            ev_end.record(main)
            side.wait_event(ev_end)
            side.wait_event(ev_b_end)
            with torch.cuda.stream(side):
                lib.upsample_flow(A["c"][iters], A["mask"][iters - 1], preds_a[iters - 1])
                lib.upsample_flow(Bb["c"][iters], Bb["mask"][iters - 1], preds_b[iters - 1])
            main.wait_stream(side)
            main.wait_stream(sb)
            STATS["hip"] += n_launch
            ctx.cfg, ctx.bufs, ctx.generation = cfg, bufs, bufs.generation
            ctx.save_for_backward(f1r, f2r)
            ctx.n_rest = len(rest)
            return (*preds_a, *preds_b)
        for i in range(iters):
            c1a, c1b = A["c"][i], Bb["c"][i]
            # flows, flo_rotate(flow_B), both feature warps + groupwise correlations: one launch (prior_raft.py:171-182)
            lib.motion_prep(c1a, c1b, g_a2b_8, g_b2a_8, f1r, f2r, A["flow4"][i], Bb["flow2"][i], A["conf_in"][i],
                            A["x"][i], 252, Bb["x"][i], 254)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for name, src, off, dst in (("a.f1a", A["flow4"][i], 0, A["t_a"][i]), ("a.f1b", A["flow4"][i], 2, A["t_ba"][i]),
                                            ("b.f1", Bb["flow2"][i], 0, Bb["t"][i])):
                    w, bias = P.stem_w[name]
                    lib.conv2d_small(src, False, off, 2, w, bias, dst, 0, 128, 7, 7, 1, True, B, H8, W8)
                cv(("a.f2a", A["t_a"][i], 0, 128, A["cat"][i], 128, EPI_RELU, {}), ("a.f2b", A["t_ba"][i], 0, 128, A["cat"][i], 192, EPI_RELU, {}),
                   ("b.f2", Bb["t"][i], 0, 128, Bb["cat"][i], 192, EPI_RELU, {}))
                cv(("a.cf1", A["conf_in"][i], 0, 8, A["cf1"][i], 0, EPI_RELU, {}))
                cv(("a.cf2", A["cf1"][i], 0, 32, A["cat"][i], 256, EPI_RELU, {}))
                stems_done = torch.cuda.Event()
                stems_done.record(side)
            # DCCL lookups (prior_raft.py:185-188)
            lib.dccl_lookup(c1a, pyr_a[0], pyr_b[0], g_b2a_8, A["own"], A["raw"])
            lib.dccl_combine(A["own"], A["raw"], g_b2a_8, A["corr"][i], B, H8, W8)
            lib.dccl_lookup(c1b, pyr_b[0], pyr_a[0], g_a2b_8, Bb["own"], Bb["raw"])
            lib.dccl_combine(Bb["own"], Bb["raw"], g_a2b_8, Bb["corr"][i], B, H8, W8)
            # ---- motion encoders (update.py:183-201, :91-99); branch A and branch B of a shape are groups of one launch
            cv(("a.c1", A["corr"][i], 0, 324, A["c1"][i], 0, EPI_RELU, {}), ("b.c1", Bb["corr"][i], 0, 324, Bb["c1"][i], 0, EPI_RELU, {}))
            cv(("a.c2", A["c1"][i], 0, 256, A["cat"][i], 0, EPI_RELU, {}), ("b.c2", Bb["c1"][i], 0, 256, Bb["cat"][i], 0, EPI_RELU, {}))
            main.wait_event(stems_done)
            cv(("a.out", A["cat"][i], 0, 272, A["x"][i], 128, EPI_RELU, {}))
            cv(("b.out", Bb["cat"][i], 0, 256, Bb["x"][i], 128, EPI_RELU, {}))
            # ---- SepConvGRU (update.py:46-60), heads (update.py:13-14, :124-136)
            SS = (("a", A), ("b", Bb))
            for k, hin, hout in (("1", "h", "h1"), ("2", "h1", "h")):
                hi = lambda S: S[hin][i]                                            # noqa: E731
                ho = lambda S: S[hout][i + 1] if hout == "h" else S[hout][i]        # noqa: E731
                cv(*[(t + ".zr" + k, hi(S), 0, 128, S["z" + k][i], 0, EPI_GRU_ZR,
                      dict(in1=S["x"][i], off1=0, c1=256, h=hi(S), aux=S["rhr" + k][i], save_gates=True)) for t, S in SS])
                cv(*[(t + ".q" + k, S["rhr" + k][i], 0, 128, ho(S), 0, EPI_GRU_Q,
                      dict(in1=S["x"][i], off1=0, c1=256, h=hi(S), z=S["z" + k][i], aux=S["q" + k][i], save_gates=True)) for t, S in SS])
            cv(*[(t + ".fh1", S["h"][i + 1], 0, 128, S["fh"][i], 0, EPI_RELU, {}) for t, S in SS])
            cv(*[(t + ".fh2", S["fh"][i], 0, 256, S["delta"][i], 0, EPI_LINEAR, {}) for t, S in SS])
            cv(*[(t + ".m0", S["h"][i + 1], 0, 128, S["mh"][i], 0, EPI_RELU, {}) for t, S in SS])
            cv(*[(t + ".m2", S["mh"][i], 0, 256, S["mask"][i], 0, EPI_LINEAR, dict(scale=0.25)) for t, S in SS])
            for S in (A, Bb):
                lib.coords_add(S["c"][i + 1], S["delta"][i], src=S["c"][i])     # coords1 += delta_flow (prior_raft.py:193,196), every iteration's kept
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                lib.upsample_flow(A["c"][i + 1], A["mask"][i], preds_a[i])      # prior_raft.py:200-208
                lib.upsample_flow(Bb["c"][i + 1], Bb["mask"][i], preds_b[i])
            n_launch += 5 + 2 + 3 + 5 + 4 + 4 + 4
        main.wait_stream(side)
        STATS["hip"] += n_launch
        ctx.cfg, ctx.bufs, ctx.generation = cfg, bufs, bufs.generation
        ctx.save_for_backward(f1r, f2r)
        ctx.n_rest = len(rest)
        return (*preds_a, *preds_b)

    @staticmethod
    def backward(ctx, *g_preds):
        from .autograd import _nchw, SINK, STATS
        lib = _lib.load()
        cfg, bufs = ctx.cfg, ctx.bufs
        if bufs.generation != ctx.generation:
            raise _lib.PfError("the training loop's activation workspace was refilled by a later forward before this backward ran "
                               "(one workspace per model and shape): run backward before the next forward, or set "
                               "PRIORFLOW_TRAIN_LOOP=0 for the per-node tape, which keeps its activations per call")
        P, pyr_a, pyr_b, coords0, g_a2b_8, g_b2a_8, iters = (cfg["packs"], cfg["pyr_a"], cfg["pyr_b"], cfg["coords0"],
                                                             cfg["g_a2b_8"], cfg["g_b2a_8"], cfg["iters"])
        f1r, f2r = ctx.saved_tensors
        B, _, H8, W8 = coords0.shape
        dev = coords0.device
        A, Bb = bufs.a, bufs.b
        N = bufs.N
        def dg(*items):
            """One launch: the same-geometry data-gradient convolutions `items` = (name, dy, off, c, out, off_out, epilogue, kwargs)."""
            lib.conv2d([P.dg[n].desc(x, o0, c0, out, oo, epi, **kw) for n, x, o0, c0, out, oo, epi, kw in items], B, H8, W8, items[0][1])

        d_f1, d_f2 = torch.zeros_like(f1r), torch.zeros_like(f2r)
        d_inp = {"a": torch.zeros(N, 128, device=dev), "b": torch.zeros(N, 128, device=dev)}
        gp = {"a": g_preds[:iters], "b": g_preds[iters:]}
        pg_a, pg_b = pyr_a[2].buffers(pyr_a[0]), pyr_b[2].buffers(pyr_b[0])
        SS = (("a", A), ("b", Bb))
        for _t, S in SS:
            S["d_flow"].zero_()
            S["d_raw"].zero_()           # once per backward: every pf_dccl_lookup_bwd below leaves it zero again (clear_raw)
        gh: Dict[str, Optional[torch.Tensor]] = {"a": None, "b": None}        # gradient of the hidden state an iteration hands on
        # The hidden-state chain (heads, GRU) is the only dependency between iterations; the motion encoders' data gradients, the
        # DCCL and warp backwards of iteration i hang off it and run on a side stream beside the chain of iteration i - 1.
        main, side = torch.cuda.current_stream(), _side_stream(cfg["model"])
        side.wait_stream(main)
        n_launch = 0
        split = _split_on()
        if split:
            # ---- two chains (round 6): the hidden-state chain of branch A on the calling stream, branch B's on its own stream -- they
            # share nothing --, every data-gradient convolution a launch of ONE group with co_groups = 1; the motion encoders', DCCL
            # and warp backwards of iteration i on the side stream behind BOTH chains of that iteration.  Capture order = queue
            # assignment (forward): the chains of iteration i - 1 are captured BEFORE the side work of iteration i, and B's chain of
            # iteration i - 1 is ordered behind A's chain of iteration i, so that A's last kernel of an iteration has the successors
            # A's next kernel (same queue), B's next chain (queue + 1), the side work (queue + 2), in that order.
            sb = _b_stream(cfg["model"])

            def dg1(name, x, o0, c0, out, oo, epi, **kw):
                d = P.dg[name].desc(x, o0, c0, out, oo, epi, **kw)
                d.co_groups = 1
                lib.conv2d([d], B, H8, W8, x)

            def chain(t, S, i):
                """heads -> GRU (vertical, then horizontal half-step) of one branch and one iteration; leaves gh[t] for iteration i - 1."""
                Fq2, Fq1 = S["F"][2 * (i & 1)], S["F"][2 * (i & 1) + 1]
                if gh[t] is None:
                    Fq1[:, 384:].zero_()
                    ghv = Fq1[:, 384:]
                else:
                    ghv = gh[t]
                if gp[t][i] is None:
                    S["d_mask"][i].zero_(); S["d_mh"][i].zero_(); S["d_delta"][i].zero_(); S["d_fh"][i].zero_()
                else:
                    lib.upsample_flow_bwd(S["c"][i + 1], S["mask"][i], gp[t][i].contiguous(), S["d_mask"][i], S["d_flow"][i])
                    lib.to_channel_last(S["d_flow"][i], 0, 2, S["d_delta"][i], 0)
                    dg1(t + ".m2", S["d_mask"][i], 0, 576, S["d_mh"][i], 0, EPI_MASK, h=S["mh"][i], scale=0.25)
                    dg1(t + ".m0", S["d_mh"][i], 0, 256, ghv, 0, EPI_ADD, h=ghv)
                    dg1(t + ".fh2", S["d_delta"][i], 0, 4, S["d_fh"][i], 0, EPI_MASK, h=S["fh"][i])
                    dg1(t + ".fh1", S["d_fh"][i], 0, 256, ghv, 0, EPI_ADD, h=ghv)
                for k, F, hname in (("2", Fq2, "h1"), ("1", Fq1, "h")):
                    g_in = ghv if k == "2" else Fq2[:, 384:]
                    lib.gru_q_bwd(g_in, S["z" + k][i], S["q" + k][i], S[hname][i], S["d_q" + k][i], S["dz"], F[:, 384:])
                    dg1(f"{t}.q{k}", S["d_q" + k][i], 0, 128, F, 0, EPI_LINEAR)
                    lib.gru_zr_bwd(S["dz"], F[:, :128], S["z" + k][i], S["rhr" + k][i][:, 128:], S[hname][i], S["d_zr" + k][i], F[:, 384:])
                    dg1(f"{t}.zr{k}", S["d_zr" + k][i], 0, 256, F, 128, EPI_ADD, h=F[:, 128:])
                gh[t] = Fq1[:, 384:]
                lib.gru_dx_finish(Fq1[:, 128:384], Fq2[:, 128:384], S["x"][i], d_inp[t], S["d_out"][i], 128, 124 if t == "a" else 126)

            def side_work(i):
                dg(("a.out", A["d_out"][i], 0, 124, A["d_cat"][i], 0, EPI_MASK, dict(h=A["cat"][i])))
                dg(("b.out", Bb["d_out"][i], 0, 128, Bb["d_cat"][i], 0, EPI_MASK, dict(h=Bb["cat"][i])))
                dg(("a.c2", A["d_cat"][i], 0, 128, A["d_c1"][i], 0, EPI_MASK, dict(h=A["c1"][i])))
                dg(("b.c2", Bb["d_cat"][i], 0, 192, Bb["d_c1"][i], 0, EPI_MASK, dict(h=Bb["c1"][i])))
                dg(*[(t + ".c1", S["d_c1"][i], 0, 256, S["d_corr"], 0, EPI_LINEAR, {}) for t, S in SS])
                dg(("a.f2a", A["d_cat"][i], 128, 64, A["d_t_a"][i], 0, EPI_MASK, dict(h=A["t_a"][i])),
                   ("a.f2b", A["d_cat"][i], 192, 64, A["d_t_ba"][i], 0, EPI_MASK, dict(h=A["t_ba"][i])),
                   ("b.f2", Bb["d_cat"][i], 192, 64, Bb["d_t"][i], 0, EPI_MASK, dict(h=Bb["t"][i])))
                dg(("a.cf2", A["d_cat"][i], 256, 16, A["d_cf1"][i], 0, EPI_MASK, dict(h=A["cf1"][i])))
                dg(("a.cf1", A["d_cf1"][i], 0, 32, A["d_conf"], 0, EPI_LINEAR, {}))
                for t, S in SS:
                    g_back = g_b2a_8 if t == "a" else g_a2b_8
                    lib.dccl_combine_bwd(S["d_corr"], g_back, S["d_raw"], B, H8, W8)
                    lib.dccl_lookup_bwd(S["c"][i], g_back, S["d_corr"], S["d_raw"], pg_a if t == "a" else pg_b, pg_b if t == "a" else pg_a,
                                        clear_raw=True)
                lib.warp_gcorr_bwd(f1r, f2r, A["c"][i], False, A["d_conf"], 0, d_f1, d_f2)
                lib.to_nchw(A["flow4"][i], 2, 2, A["flow_ba"])
                lib.warp_gcorr_bwd(f1r, f2r, A["flow_ba"], True, A["d_conf"], 4, d_f1, d_f2)

            sb.wait_stream(main)
            pending = None                          # (iteration, event after A's chain, event after B's chain) whose side work is not captured yet
            for i in range(iters - 1, -1, -1):
                chain("a", A, i)
                ev_a = torch.cuda.Event()
                ev_a.record(main)
                if pending is not None:
                    sb.wait_event(pending[1])       # queue steering only: B's chain of iteration i behind A's chain of iteration i + 1
                with torch.cuda.stream(sb):
                    chain("b", Bb, i)
                    ev_b = torch.cuda.Event()
                    ev_b.record(sb)
                if pending is not None:
                    side.wait_event(pending[1])
                    side.wait_event(pending[2])
                    with torch.cuda.stream(side):
                        side_work(pending[0])
                pending = (i, ev_a, ev_b)
                n_launch += 2 * (2 + 4 + 6 + 1) + 8 + 4 + 3 - 2
            side.wait_event(pending[1])
            side.wait_event(pending[2])
            with torch.cuda.stream(side):
                side_work(pending[0])
            main.wait_stream(sb)
        for i in (range(iters - 1, -1, -1) if not split else ()):
            Fq2 = {t: S["F"][2 * (i & 1)] for t, S in SS}             # ping-pong: gh of iteration i + 1 lives in the other pair
            Fq1 = {t: S["F"][2 * (i & 1) + 1] for t, S in SS}
            ghv = {}
            for t, S in SS:
                if gh[t] is None:
                    Fq1[t][:, 384:].zero_()                           # borrowed as the zero gradient of the last state
                    ghv[t] = Fq1[t][:, 384:]
                else:
                    ghv[t] = gh[t]
            # ---- heads (the two consumers of h2 add into the gradient the next iteration left for it)
            live = [(t, S) for t, S in SS if gp[t][i] is not None]
            for t, S in SS:
                if gp[t][i] is None:
                    S["d_mask"][i].zero_(); S["d_mh"][i].zero_(); S["d_delta"][i].zero_(); S["d_fh"][i].zero_()
                    continue
                lib.upsample_flow_bwd(S["c"][i + 1], S["mask"][i], gp[t][i].contiguous(), S["d_mask"][i], S["d_flow"][i])
                lib.to_channel_last(S["d_flow"][i], 0, 2, S["d_delta"][i], 0)
                n_launch += 2
            if live:
                # mask = 0.25 * conv (update.py:134,157): the factor rides in the data gradient's epilogue; the weight
                # gradient of that conv is scaled once, after the deferred launch
                dg(*[(t + ".m2", S["d_mask"][i], 0, 576, S["d_mh"][i], 0, EPI_MASK, dict(h=S["mh"][i], scale=0.25)) for t, S in live])
                dg(*[(t + ".m0", S["d_mh"][i], 0, 256, ghv[t], 0, EPI_ADD, dict(h=ghv[t])) for t, S in live])
                dg(*[(t + ".fh2", S["d_delta"][i], 0, 4, S["d_fh"][i], 0, EPI_MASK, dict(h=S["fh"][i])) for t, S in live])
                dg(*[(t + ".fh1", S["d_fh"][i], 0, 256, ghv[t], 0, EPI_ADD, dict(h=ghv[t])) for t, S in live])
                n_launch += 4
            # ---- GRU, vertical then horizontal half-step (update.py:55-60, :48-53 in reverse)
            for k, F, hname in (("2", Fq2, "h1"), ("1", Fq1, "h")):
                for t, S in SS:
                    g_in = ghv[t] if k == "2" else Fq2[t][:, 384:]     # d h2 from the heads / d h1 of the vertical half-step
                    lib.gru_q_bwd(g_in, S["z" + k][i], S["q" + k][i], S[hname][i], S["d_q" + k][i], S["dz"], F[t][:, 384:])
                dg(*[(f"{t}.q{k}", S["d_q" + k][i], 0, 128, F[t], 0, EPI_LINEAR, {}) for t, S in SS])               # [d(r*h) | d x]
                for t, S in SS:
                    lib.gru_zr_bwd(S["dz"], F[t][:, :128], S["z" + k][i], S["rhr" + k][i][:, 128:], S[hname][i], S["d_zr" + k][i], F[t][:, 384:])
                dg(*[(f"{t}.zr{k}", S["d_zr" + k][i], 0, 256, F[t], 128, EPI_ADD, dict(h=F[t][:, 128:])) for t, S in SS])   # [d x | d h] +=
                n_launch += 6
            for t, S in SS:
                gh[t] = Fq1[t][:, 384:]
                # d x = [d inp | d out | flows]: inp feeds every iteration, out = relu(conv) -> mask
                lib.gru_dx_finish(Fq1[t][:, 128:384], Fq2[t][:, 128:384], S["x"][i], d_inp[t], S["d_out"][i], 128, 124 if t == "a" else 126)
            # ---- motion encoders, DCCL, warps: beside the next (earlier) iteration's chain
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                dg(("a.out", A["d_out"][i], 0, 124, A["d_cat"][i], 0, EPI_MASK, dict(h=A["cat"][i])))
                dg(("b.out", Bb["d_out"][i], 0, 128, Bb["d_cat"][i], 0, EPI_MASK, dict(h=Bb["cat"][i])))
                dg(("a.c2", A["d_cat"][i], 0, 128, A["d_c1"][i], 0, EPI_MASK, dict(h=A["c1"][i])))
                dg(("b.c2", Bb["d_cat"][i], 0, 192, Bb["d_c1"][i], 0, EPI_MASK, dict(h=Bb["c1"][i])))
                dg(*[(t + ".c1", S["d_c1"][i], 0, 256, S["d_corr"], 0, EPI_LINEAR, {}) for t, S in SS])
                dg(("a.f2a", A["d_cat"][i], 128, 64, A["d_t_a"][i], 0, EPI_MASK, dict(h=A["t_a"][i])),
                   ("a.f2b", A["d_cat"][i], 192, 64, A["d_t_ba"][i], 0, EPI_MASK, dict(h=A["t_ba"][i])),
                   ("b.f2", Bb["d_cat"][i], 192, 64, Bb["d_t"][i], 0, EPI_MASK, dict(h=Bb["t"][i])))
                dg(("a.cf2", A["d_cat"][i], 256, 16, A["d_cf1"][i], 0, EPI_MASK, dict(h=A["cf1"][i])))
                dg(("a.cf1", A["d_cf1"][i], 0, 32, A["d_conf"], 0, EPI_LINEAR, {}))
                for t, S in SS:
                    g_back = g_b2a_8 if t == "a" else g_a2b_8
                    lib.dccl_combine_bwd(S["d_corr"], g_back, S["d_raw"], B, H8, W8)       # scatters into the zeroed d_raw
                    lib.dccl_lookup_bwd(S["c"][i], g_back, S["d_corr"], S["d_raw"], pg_a if t == "a" else pg_b, pg_b if t == "a" else pg_a,
                                        clear_raw=True)
                lib.warp_gcorr_bwd(f1r, f2r, A["c"][i], False, A["d_conf"], 0, d_f1, d_f2)
                lib.to_nchw(A["flow4"][i], 2, 2, A["flow_ba"])
                lib.warp_gcorr_bwd(f1r, f2r, A["flow_ba"], True, A["d_conf"], 4, d_f1, d_f2)
            n_launch += 2 + 8 + 4 + 3
        # ---- deferred weight gradients: one launch per convolution over the iters * B stored images
        Bi = iters * B

        def wg(name, x0, off0, c0, dy, off_dy, cout, x1=None, off1=0, c1=0):
            m = P.fwd[name]
            dw, db = P.acc[name].buffers((max(cout, m.cout) + 127) // 128 * 128, m.kh * m.kw, (c0 + c1 + 31) // 32 * 32, dev)
            lib.conv2d_wgrad(_flat(x0), off0, c0, _flat(dy), off_dy, cout, dw, db, m.kh, m.kw, Bi, H8, W8,
                             x1=None if x1 is None else _flat(x1), off1=off1, c1=c1)

        # Round 5: with the gradient sink on, nothing on this stream reads a weight gradient before the sink's flush (the autograd
        # nodes only register their packed buffers), so the deferred launches -- 3.5 ms of chip-filling kernels -- go to the side
        # stream and run beside the correlation pyramids' and the encoders' backward (6 ms of mostly small kernels); the sink joins
        # the stream before it unpacks.  Without the sink the gradients are handed to autograd right away: same stream, as before.
        defer_side = bool(SINK.active)
        wg_stream = side if defer_side else main
        if defer_side:
            ev_iter = torch.cuda.Event()
            ev_iter.record(side)                   # the iterations' side chains (d f1 / d f2, pyramid and motion-encoder gradients) ...
            main.wait_event(ev_iter)               # ... are what this node returns: the calling stream waits for them, not for the launches below
            side.wait_stream(main)                 # every output gradient of the chain has been produced
        with torch.cuda.stream(wg_stream):
            # (the chain's convolutions first: their output gradients were produced on this stream; the side stream is joined in
            # front of the motion encoders' weight gradients)
            for t, S in (("a", A), ("b", Bb)):
                hs = S["h"]
                wg(t + ".m2", S["mh"], 0, 256, S["d_mask"], 0, 576)
                wg(t + ".m0", hs[1:], 0, 128, S["d_mh"], 0, 256)
                wg(t + ".fh2", S["fh"], 0, 256, S["d_delta"], 0, 4)
                wg(t + ".fh1", hs[1:], 0, 128, S["d_fh"], 0, 256)
                wg(t + ".q2", S["rhr2"], 0, 128, S["d_q2"], 0, 128, x1=S["x"], off1=0, c1=256)
                wg(t + ".zr2", S["h1"], 0, 128, S["d_zr2"], 0, 256, x1=S["x"], off1=0, c1=256)
                wg(t + ".q1", S["rhr1"], 0, 128, S["d_q1"], 0, 128, x1=S["x"], off1=0, c1=256)
                wg(t + ".zr1", hs[:iters], 0, 128, S["d_zr1"], 0, 256, x1=S["x"], off1=0, c1=256)
            wg("a.out", A["cat"], 0, 272, A["d_out"], 0, 124)
            wg("b.out", Bb["cat"], 0, 256, Bb["d_out"], 0, 128)
            if not defer_side:
                main.wait_stream(side)
            for t, S in (("a", A), ("b", Bb)):
                wg(t + ".c1", S["corr"], 0, 324, S["d_c1"], 0, 256)
            wg("a.c2", A["c1"], 0, 256, A["d_cat"], 0, 128)
            wg("b.c2", Bb["c1"], 0, 256, Bb["d_cat"], 0, 192)
            wg("a.f2a", A["t_a"], 0, 128, A["d_cat"], 128, 64)
            wg("a.f2b", A["t_ba"], 0, 128, A["d_cat"], 192, 64)
            wg("b.f2", Bb["t"], 0, 128, Bb["d_cat"], 192, 64)
            wg("a.cf2", A["cf1"], 0, 32, A["d_cat"], 256, 16)
            wg("a.cf1", A["conf_in"], 0, 8, A["d_cf1"], 0, 32)
            for t in "ab":          # mask = 0.25 * conv: the output gradient stored for m2 is the un-scaled one
                P.acc[t + ".m2"].scale = 0.25
            stem_grads, stem_on_side = [], False
            for name, x, off, dy in (("a.f1a", A["flow4"], 0, A["d_t_a"]), ("a.f1b", A["flow4"], 2, A["d_t_ba"]), ("b.f1", Bb["flow2"], 0, Bb["d_t"])):
                m = P.stems[name]
                if SINK.active and m.weight.grad is not None and m.bias.grad is not None and m.weight.requires_grad and m.bias.requires_grad:
                    # accumulated in the parameter layout: straight into .grad (a frozen stem: autograd drops the returned gradient)
                    lib.conv2d_wgrad_small(_flat(x), False, off, 2, _flat(dy), 0, 128, m.weight.grad, m.bias.grad, 7, 7, 1, Bi, H8, W8)
                    stem_grads += [None, None]
                    continue
                if not (m.weight.requires_grad or m.bias.requires_grad):
                    stem_grads += [None, None]         # a frozen stem: nothing to compute (the launch was wasted)
                    continue
                dw, db = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
                lib.conv2d_wgrad_small(_flat(x), False, off, 2, _flat(dy), 0, 128, dw, db, 7, 7, 1, Bi, H8, W8)
                stem_grads += [dw, db]
                stem_on_side = defer_side             # handed to autograd, which consumes it on the CALLING stream
        if defer_side:
            SINK.join_streams.append(side)
            if stem_on_side:
                # a trainable stem outside the sink (its .grad not allocated yet): dw / db were produced on the side stream after
                # the event the calling stream waited for -- join, or autograd's accumulation races the launch
                main.wait_stream(side)
                for t in stem_grads:
                    if t is not None:
                        t.record_stream(main)
        n_launch += 29 + 3
        STATS["hip"] += n_launch
        d_net_a, d_net_b = _nchw(gh["a"].contiguous(), B, H8, W8), _nchw(gh["b"].contiguous(), B, H8, W8)
        zero = torch.zeros(1, device=dev)
        return (None, d_net_a, d_net_b, _nchw(d_inp["a"], B, H8, W8), _nchw(d_inp["b"], B, H8, W8),
                _nchw(d_f1, B, H8, W8), _nchw(d_f2, B, H8, W8), zero, zero, *stem_grads, *([zero] * (ctx.n_rest - 6)))


def run_loop(model, lib, zr_a, zr_b, gate_of, net_a, net_b, inp_a, inp_b, f1a, f2a, pyr_a, pyr_b, coords0, c1a, c1b,
             g_a2b_8, g_b2a_8, iters: int):
    """The ``iters`` refinement iterations as one autograd node; returns (preds_A, preds_B)."""
    P = _Packs(model, zr_a, zr_b, gate_of)
    cfg = dict(model=model, packs=P, pyr_a=pyr_a, pyr_b=pyr_b, coords0=coords0, c1a=c1a, c1b=c1b, g_a2b_8=g_a2b_8,
               g_b2a_8=g_b2a_8, iters=iters)
    stems = []
    for m in P.stems.values():
        stems += [m.weight, m.bias]
    # the weight tokens order this node's backward in front of every WeightGate's (which unpacks the accumulated gradients)
    toks = [gate_of(m)[0] for m in P.mods.values()] + [zr[k][3] for zr in (zr_a, zr_b) for k in "12" if zr[k][3] is not None]
    out = LoopFn.apply(cfg, net_a, net_b, inp_a, inp_b, f1a, f2a, pyr_a[1], pyr_b[1], *stems, *toks)
    return list(out[:iters]), list(out[iters:])
