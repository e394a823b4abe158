"""Training-step counterpart of the reference on the HIP kernels.  The network's
forward / backward in ``train()`` mode is ``autograd.train_forward`` (reached through ``PriOr_RAFT.forward``);
``train_step`` below is the loop body of ``train_flow.py:120-141`` built from the pieces of this module.

Mirrors, with the reference's names and argument meaning:
  * ``uniform_loss(H, W)(flow_preds, flow_gt, valid, gamma, extro_info, max_flow)`` (train_flow.py:55-79)
    -> ``(loss, metrics)``; there being no autograd graph, the gradient of the loss with respect to every
    prediction (what ``loss.backward()`` would feed the network) is left in ``.grads``;
  * ``rotate_gt(flow_gt)`` -> ``(flow_gt_B, valid_B)`` (train_flow.py:123-126, ``flo_A2B`` + validity);
  * ``fetch_optimizer(args, model)`` -> ``(FlatAdamW, OneCycleLinearLR)`` (train_flow.py:86-91): AdamW over
    ONE flat parameter / gradient buffer (the parameters become views of it, so the single all-reduce of
    ``parallel.all_reduce_sum_`` and the fused ``pf_adamw_step`` both see one array; with more than one rank
    the replicas are synchronised from rank 0 here, see ``FlatAdamW.sync_replicas``)
    and OneCycleLR(max_lr, num_steps + 100, pct_start=0.05, linear, cycle_momentum=False) in closed form;
  * ``clip_grad_norm_(optimizer, max_norm)`` (train_flow.py:137): total norm by ``pf_sum_squares``; the
    clip coefficient is applied inside the AdamW kernel instead of rewriting the gradients.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Sequence, Tuple

import torch

from . import _lib, parallel
from .engine import rotation_x
from .evaluate import spherical_mask

MAX_FLOW = 400      # train_flow.py:46


class uniform_loss:
    NBLK = 64

    def __init__(self, H: int, W: int, device=None):
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.H, self.W = H, W
        self.uniform_mask = torch.from_numpy(spherical_mask(H, W)).float().to(self.device).contiguous()   # H x W
        self.grads: List[torch.Tensor] = []
        self._w: Dict[tuple, torch.Tensor] = {}        # gamma ** (n - i - 1) per (n, gamma), on the device

    @torch.no_grad()
    def __call__(self, flow_preds: Sequence[torch.Tensor], flow_gt: torch.Tensor, valid: torch.Tensor,
                 gamma: float = 0.8, extro_info: str = "", max_flow: float = MAX_FLOW,
                 need_grads: bool = True, lazy: bool = False) -> Tuple[torch.Tensor, Dict[str, float]]:
        """lazy: the metrics stay 0-dim device tensors (no host synchronisation: what a captured HIP graph of the step needs)."""
        n = len(flow_preds)
        gt = flow_gt.float().contiguous()
        vd = valid.float().contiguous()
        B = gt.shape[0]
        part = torch.empty(n, B, self.NBLK, 6, dtype=torch.float64, device=gt.device)
        weights = [gamma ** (n - i - 1) for i in range(n)]
        preds = [p.float().contiguous() for p in flow_preds]
        self.grads = [torch.empty_like(gt) for _ in range(n)] if need_grads else []
        # all terms of a branch in ONE launch (round 6; the 24 launches of a step were 0.5 ms one behind the other), 32 at a time
        for lo in range(0, n, 32):
            hi = min(n, lo + 32)
            self.lib.seq_loss_batch(preds[lo:hi], gt, vd, self.uniform_mask.view(-1), weights[lo:hi], float(max_flow),
                                    self.grads[lo:hi] if need_grads else None, part[lo:hi])
        tot = part.sum(dim=(1, 2))                                     # [n, 6]
        wk = (n, float(gamma), str(gt.device))
        if wk not in self._w:                                          # (built outside any graph capture: a host -> device copy)
            self._w[wk] = torch.tensor(weights, dtype=torch.float64, device=gt.device)
        flow_loss = (self._w[wk] * tot[:, 0]).sum().float()
        if lazy:
            last = tot[-1]
            nv = last[2].clamp(min=1.0)
            return flow_loss, {extro_info + "epe": last[1] / nv, extro_info + "1px": last[3] / nv,
                               extro_info + "3px": last[4] / nv, extro_info + "5px": last[5] / nv}
        last = tot[-1].tolist()
        nv = max(last[2], 1.0)
        metrics = {extro_info + "epe": last[1] / nv, extro_info + "1px": last[3] / nv,
                   extro_info + "3px": last[4] / nv, extro_info + "5px": last[5] / nv}
        return flow_loss, metrics


_GT_GRIDS: Dict[tuple, tuple] = {}


@torch.no_grad()
def rotate_gt(flow_gt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """flow_gt_B = flo_A2B(flow_gt) (core/utils/projection_prim_ortho.py:563-565) and
    valid_B = (|u| < 1000) & (|v| < 1000) as float (train_flow.py:123-126)."""
    lib = _lib.load()
    fl = flow_gt.float().contiguous()
    B, _, H, W = fl.shape
    key = (H, W, fl.device)
    if key not in _GT_GRIDS:
        g_a2b = torch.empty(2, H, W, device=fl.device)
        g_b2a = torch.empty(2, H, W, device=fl.device)
        lib.sample_grid(g_a2b, rotation_x(-math.pi / 2))
        lib.sample_grid(g_b2a, rotation_x(math.pi / 2))
        _GT_GRIDS[key] = (g_a2b, g_b2a)
    g_a2b, g_b2a = _GT_GRIDS[key]
    out = torch.empty_like(fl)
    # flo_rotate(flow, W2C = grid(R_A2B^T) == grid(R_B2A), C2W = grid(R_A2B))
    lib.flo_rotate(fl, g_b2a, g_a2b, out)
    valid_b = ((out[:, 0].abs() < 1000) & (out[:, 1].abs() < 1000)).float()
    return out, valid_b


class OneCycleLinearLR:
    """OneCycleLR(optimizer, max_lr, total_steps, pct_start=0.05, cycle_momentum=False, anneal_strategy='linear')."""

    def __init__(self, optimizer: "FlatAdamW", max_lr: float, total_steps: int, pct_start: float = 0.05,
                 div_factor: float = 25.0, final_div_factor: float = 1e4):
        self.optimizer, self.max_lr, self.total = optimizer, max_lr, total_steps
        self.initial = max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.end1 = float(pct_start * total_steps) - 1
        self.step_num = 0
        self._apply()

    def lr_at(self, step: int) -> float:
        if step <= self.end1:
            return (self.max_lr - self.initial) * (step / self.end1) + self.initial
        return (self.min_lr - self.max_lr) * ((step - self.end1) / ((self.total - 1) - self.end1)) + self.max_lr

    def _apply(self):
        for grp in self.optimizer.param_groups:
            grp["lr"] = self.lr_at(self.step_num)

    def step(self):
        self.step_num += 1
        if self.step_num > self.total:
            raise ValueError(f"Tried to step {self.step_num} times. The specified number of total steps is {self.total}")
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


class FlatAdamW:
    """AdamW(model.parameters(), lr, weight_decay, eps) with every parameter, gradient and moment in one
    flat fp32 buffer each; ``p.data`` / ``p.grad`` of the model become views of ``flat`` / ``grad``.

    Use ``optimizer.zero_grad()`` (it zeroes the flat buffer).  Calls that re-bind the tensors behind the
    optimizer's back -- ``model.zero_grad()`` / torch-style ``zero_grad(set_to_none=True)``, ``model.to()`` /
    ``.cuda()`` / ``.half()`` after ``fetch_optimizer`` -- are detected by ``step()`` / ``total_grad_norm()``,
    which copy the stray tensors back into the flat buffers and re-bind them (a parameter that changed
    dtype or device cannot be repaired and raises ``PfError``)."""

    def __init__(self, params, lr: float, weight_decay: float, eps: float, betas=(0.9, 0.999)):
        self.lib = _lib.load()
        self.params, self.flat, self.grad = parallel.flatten_parameters(params)
        dev = self.flat.device
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "eps": eps, "betas": betas}]
        self.step_count = 0
        self.grad_scale = 1.0
        self.realiased = 0                       # how many tensors step() / total_grad_norm() had to re-bind so far
        self._norm_part = torch.empty(256, dtype=torch.float64, device=dev)

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient buffer.  ``set_to_none`` is accepted for torch compatibility and ignored:
        the gradients must stay views of the flat buffer."""
        self.grad.zero_()
        self.grad_scale = 1.0

    def _check_aliases(self):
        for p in self.params:
            if p.device != self.flat.device or p.dtype != torch.float32:
                raise _lib.PfError(f"a parameter moved to {p.device}/{p.dtype} after fetch_optimizer(); "
                                   "create the optimizer after model.cuda() and keep the model in fp32")
        self.realiased += parallel.realias(self.params, self.flat, self.grad, keep_values=True)

    def sync_replicas(self, model=None, group=None, src: int = 0):
        """Data-parallel start-up: broadcast rank ``src``'s parameters (the flat buffer) and the model's
        buffers (cnet's BatchNorm running statistics) to every rank.  No-op for a single process."""
        self._check_aliases()
        parallel.sync_replicas(self.flat, list(model.buffers()) if model is not None else (), group, src)
        _lib.bump_weights_epoch()

    def assert_in_sync(self, group=None):
        parallel.assert_replicas_in_sync(self.flat, group)

    @torch.no_grad()
    def total_grad_norm(self) -> float:
        self._check_aliases()
        self.lib.sum_squares(self.grad, self._norm_part)
        return float(self._norm_part.sum().sqrt())

    @torch.no_grad()
    def step(self):
        self._check_aliases()
        g = self.param_groups[0]
        self.step_count += 1
        self.lib.adamw_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1],
                            g["eps"], g["weight_decay"], self.step_count, self.grad_scale)
        self.grad_scale = 1.0
        _lib.bump_weights_epoch()       # the kernel wrote the parameters through raw pointers: drop packed copies


def clip_grad_norm_(optimizer: FlatAdamW, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: returns the total norm; the coefficient min(1, max_norm/(norm+1e-6))
    is applied to the gradients by the next ``optimizer.step()`` (inside the AdamW kernel)."""
    total = optimizer.total_grad_norm()
    optimizer.grad_scale = min(1.0, float(max_norm) / (total + 1e-6))
    return total


def fetch_optimizer(args, model, group=None):
    """Create the optimizer and learning rate scheduler (train_flow.py:86-91).  When torch.distributed runs
    with more than one rank, every rank's weights and BatchNorm buffers are overwritten with rank 0's here
    (``nn.DataParallel`` re-broadcasts its module every step, train_flow.py:96; one process per GPU needs it
    once): ranks may construct the model unseeded."""
    net = getattr(model, "module", model)
    optimizer = FlatAdamW(net.parameters(), lr=args.lr, weight_decay=args.wdecay, eps=args.epsilon)
    optimizer.sync_replicas(net, group)
    scheduler = OneCycleLinearLR(optimizer, args.lr, args.num_steps + 100, pct_start=0.05)
    return optimizer, scheduler


SYNC_CHECK_EVERY = 500      # steps between replica checksum comparisons (16-byte all-gather)


def train_step(model, optimizer: FlatAdamW, scheduler: OneCycleLinearLR, criterion: uniform_loss,
               image1: torch.Tensor, image2: torch.Tensor, flow_gt: torch.Tensor, valid: torch.Tensor,
               iters: int = 12, gamma: float = 0.8, clip: float = 1.0, group=None, add_noise: bool = False):
    """One optimisation step, the loop body of train_flow.py:120-141 (without GradScaler: fp32 / bf16x3 math
    needs no loss scaling, so `--mixed_precision` has nothing to switch): zero_grad, GT rotation, [`add_noise`
    (train_flow.py:127-130): Gaussian noise of a standard deviation drawn uniformly from [0, 5) -- numpy's global
    generator, like the reference -- added to both images, clamped to [0, 255]], forward of both branches, sequence loss of both, backward,
    [one SUM all-reduce of the flat gradient buffer when torch.distributed runs with more than one rank --
    RCCL over xGMI, replacing DataParallel's reduce_add (train_flow.py:96)], clip, AdamW, scheduler.
    ``model`` may be the bare module or an ``nn.DataParallel``-style wrapper exposing ``.module``.
    Returns ``(loss, metrics)``; ``metrics['grad_norm']`` is the pre-clip total norm."""
    net = getattr(model, "module", model)
    optimizer.zero_grad()
    flow_gt_b, valid_b = rotate_gt(flow_gt)
    if add_noise:
        import numpy as np
        stdv = float(np.random.uniform(0.0, 5.0))
        image1 = (image1 + stdv * torch.randn(*image1.shape, device=image1.device)).clamp(0.0, 255.0)
        image2 = (image2 + stdv * torch.randn(*image2.shape, device=image2.device)).clamp(0.0, 255.0)
    sink = _grad_sink(optimizer)
    try:
        preds_a, preds_b = net(image1, image2, iters=iters)
        loss_a, metrics_a = criterion(preds_a, flow_gt, valid, gamma, extro_info="A-")
        seeds = list(criterion.grads)
        loss_b, metrics_b = criterion(preds_b, flow_gt_b, valid_b, gamma, extro_info="B-")
        seeds += list(criterion.grads)
        # loss.backward(): the criterion has already produced d loss / d prediction for every prediction
        torch.autograd.backward(list(preds_a) + list(preds_b), seeds)
    except BaseException:
        sink.abort()            # nothing half-accumulated is added to the gradients; the sink is off for whoever runs next
        raise
    sink.flush()
    parallel.all_reduce_sum_(optimizer.grad, group)
    norm = clip_grad_norm_(optimizer, clip)
    optimizer.step()
    scheduler.step()
    if optimizer.step_count % SYNC_CHECK_EVERY == 1:      # first step and every SYNC_CHECK_EVERY after it
        optimizer.assert_in_sync(group)
    return loss_a + loss_b, {**metrics_a, **metrics_b, "grad_norm": norm}


def _grad_sink(optimizer):
    """Starts autograd.GradSink for one step (the convolutions' weight gradients go straight into the ``.grad`` views of the flat
    buffer; PRIORFLOW_GRAD_SINK=0: through autograd's own accumulation as in round 3).  Call ``.flush()`` after backward."""
    from .autograd import SINK
    dev = optimizer.flat.device
    sink = SINK.for_device(dev.index if dev.index is not None else torch.cuda.current_device())     # this replica's own
    if os.environ.get("PRIORFLOW_GRAD_SINK", "1") != "0":
        sink.begin(optimizer.params)
    else:
        sink.active = False
    return sink


class GraphedTrainStep:
    """The whole optimisation step -- zero_grad, GT rotation, forward of both branches, sequence loss, backward, [gradient
    all-reduce], clip, fused AdamW (train_flow.py:120-141) -- captured ONCE into HIP graphs and replayed: at the reference's
    training crop the step is ~1 400 launches of 5-80 us each, and the host (autograd's Python nodes, ctypes calls) is as slow as
    the GPU; a replay costs the host nothing.  What changes from step to step lives in device memory: the four inputs (static
    buffers copied into), the step-dependent AdamW scalars and the clip coefficient (``pf_adamw_step_dev``: hyper = {1 - lr * wd,
    lr / bc1, sqrt(bc2), clip coefficient}; the first three are uploaded by the host before a replay -- from a ring of pinned
    slots, each guarded by an event, so a host that runs several replays ahead of the GPU never rewrites a slot whose copy is
    still pending --, the last by the graph itself from ``pf_sum_squares``).

    One rank: ONE graph.  More ranks (``torch.distributed`` initialised, or ``group=``): graph A = zero_grad ... backward + sink
    flush, then the ONE eager collective of training -- ``parallel.all_reduce_sum_`` of the flat gradient buffer (RCCL over xGMI,
    replacing DataParallel's reduce_add, train_flow.py:96) --, then graph B = gradient norm, clip coefficient, AdamW; the replica
    checksum comparison of ``train_step`` runs every ``SYNC_CHECK_EVERY`` steps.  ``add_noise`` (a host-side numpy draw,
    train_flow.py:127-130) stays with the eager ``train_step``.

        step = GraphedTrainStep(model, optimizer, scheduler, criterion, iters=12)
        loss, metrics = step(image1, image2, flow_gt, valid)        # 0-dim device tensors; float(...) them when needed

    The returned tensors are views of ONE fresh copy of the step's outputs (a later replay does not overwrite them).  The first
    ``warmup`` calls run the eager ``train_step`` (lazy initialisations, allocator warm-up); the next call captures (recording
    executes nothing) and replays.  If the capture fails nothing is kept: the next call captures again."""

    HYPER_SLOTS = 8

    def __init__(self, model, optimizer: FlatAdamW, scheduler: OneCycleLinearLR, criterion: uniform_loss, iters: int = 12,
                 gamma: float = 0.8, clip: float = 1.0, warmup: int = 2, group=None, split: bool = None):
        self.model, self.opt, self.sched, self.crit = model, optimizer, scheduler, criterion
        self.iters, self.gamma, self.clip, self.warmup = iters, gamma, clip, max(1, int(warmup))   # >= 1: first-use allocations cannot be captured
        self.group = group
        self.world = parallel._world(group)
        # two graphs with the all-reduce between them; a single rank can ask for the split too (what it costs: profiles/)
        self.split = (self.world > 1 or os.environ.get("PRIORFLOW_TRAIN_SPLIT_GRAPH", "0") == "1") if split is None else bool(split) or self.world > 1
        self.calls = 0
        self.graphs = None               # (graph,) or (graph A, graph B)
        self.static = None
        self.out_vec = None              # [loss, metrics..., grad_norm] of the last replay (static, device)
        self.out_keys = None
        dev = optimizer.flat.device
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(self.HYPER_SLOTS, 3, dtype=torch.float32).pin_memory()
        self._hyper_ev = [None] * self.HYPER_SLOTS
        self._hyper_n = 0

    def _body_a(self):
        """zero_grad ... backward + sink flush on the static inputs; every value that leaves it is a device tensor."""
        net = getattr(self.model, "module", self.model)
        opt, crit = self.opt, self.crit
        i1, i2, gt, valid = self.static
        opt.grad.zero_()
        gt_b, valid_b = rotate_gt(gt)
        sink = _grad_sink(opt)
        try:
            preds_a, preds_b = net(i1, i2, iters=self.iters)
            loss_a, m_a = crit(preds_a, gt, valid, self.gamma, extro_info="A-", lazy=True)
            seeds = list(crit.grads)
            loss_b, m_b = crit(preds_b, gt_b, valid_b, self.gamma, extro_info="B-", lazy=True)
            seeds += list(crit.grads)
            torch.autograd.backward(list(preds_a) + list(preds_b), seeds)
        except BaseException:
            sink.abort()
            raise
        sink.flush()
        return loss_a + loss_b, {**m_a, **m_b}

    def _body_b(self, loss, metrics):
        """Gradient norm, clip coefficient, AdamW on the (all-reduced) flat gradient; packs the step's outputs."""
        opt = self.opt
        with torch.no_grad():
            opt.lib.sum_squares(opt.grad, opt._norm_part)
            norm = opt._norm_part.sum().sqrt()
            # clip_grad_norm_: coefficient min(1, max_norm / (norm + 1e-6)), applied inside the AdamW kernel
            self.hyper[3:4].copy_((self.clip / (norm + 1e-6)).clamp(max=1.0).float().reshape(1))
            g = opt.param_groups[0]
            opt.lib.adamw_step_dev(opt.flat, opt.grad, opt.exp_avg, opt.exp_avg_sq, g["betas"][0], g["betas"][1], g["eps"], self.hyper)
            keys = list(metrics) + ["grad_norm"]
            vec = torch.stack([loss.float()] + [metrics[k].float() for k in metrics] + [norm.float()])
        return vec, keys

    def _set_hyper(self):
        g = self.opt.param_groups[0]
        lr, wd, (b1, b2) = float(g["lr"]), float(g["weight_decay"]), g["betas"]
        t = self.opt.step_count + 1
        k = self._hyper_n % self.HYPER_SLOTS
        self._hyper_n += 1
        if self._hyper_ev[k] is not None:
            self._hyper_ev[k].synchronize()          # the copy that last read this slot has run (HYPER_SLOTS steps ago)
        h = self._hyper_host[k]
        h[0] = 1.0 - lr * wd
        h[1] = lr / (1.0 - b1 ** t)
        h[2] = math.sqrt(1.0 - b2 ** t)
        self.hyper[:3].copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[k] = ev

    def _capture(self):
        graphs = []
        if not self.split:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._body_b(*self._body_a())
            graphs.append(g)
        else:
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                la = self._body_a()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool()):
                out = self._body_b(*la)
            graphs += [ga, gb]
        return tuple(graphs), out

    def __call__(self, image1, image2, flow_gt, valid):
        if parallel._world(self.group) != self.world:
            raise _lib.PfError("GraphedTrainStep: the process group changed after construction "
                               f"({self.world} -> {parallel._world(self.group)} ranks); build the stepper after init_process_group")
        self.calls += 1
        if self.calls <= self.warmup:
            loss, m = train_step(self.model, self.opt, self.sched, self.crit, image1, image2, flow_gt, valid, iters=self.iters,
                                 gamma=self.gamma, clip=self.clip, group=self.group)
            return loss, m
        self.opt._check_aliases()
        if self.graphs is None:
            self.static = tuple(t.detach().float().contiguous().clone() for t in (image1, image2, flow_gt, valid))
            self._set_hyper()
            from .autograd import SINK
            fdev = self.opt.flat.device
            sink = SINK.for_device(fdev.index if fdev.index is not None else torch.cuda.current_device())
            sink.reserve(fdev)                       # the gradient arena: allocated HERE, not inside the capture's private pool
            torch.cuda.synchronize()
            try:
                graphs, (vec, keys) = self._capture()
            except BaseException:
                self.static = None                   # nothing half-captured is kept (the sink was aborted by _body_a)
                raise
            self.graphs, self.out_vec, self.out_keys = graphs, vec, keys
            self._arena = sink.arena                 # the graph holds its raw pointer: alive for as long as this stepper is
        else:
            for dst, src in zip(self.static, (image1, image2, flow_gt, valid)):
                dst.copy_(src)
            self._set_hyper()
        self.graphs[0].replay()
        if self.split:
            parallel.all_reduce_sum_(self.opt.grad, self.group)      # no-op with one rank
            self.graphs[1].replay()
        self.opt.step_count += 1
        self.opt.grad_scale = 1.0
        _lib.bump_weights_epoch()       # the graph wrote the parameters through raw pointers: drop packed copies
        self.sched.step()
        if self.world > 1 and self.opt.step_count % SYNC_CHECK_EVERY == 1:
            self.opt.assert_in_sync(self.group)
        out = self.out_vec.clone()
        return out[0], {k: out[i + 1] for i, k in enumerate(self.out_keys)}
