"""GPU tests of the training-step counterpart (the network's backward: test_hip_train_step.py):
uniform_loss, rotate_gt, fetch_optimizer (FlatAdamW + OneCycleLinearLR), clip_grad_norm_ against the
reference-generated goldens in tests/golden/train.npz."""
import argparse

import numpy as np
import pytest
import torch

import golden_cases as gc
from gen_golden_train import adam_case, loss_case

pytestmark = pytest.mark.gpu


def test_uniform_loss_and_gt_rotation_vs_reference():
    from prior_flow_amd import train as tr
    g = gc.load("train")
    preds, gt, valid = loss_case()
    crit = tr.uniform_loss(64, 128)
    loss, metrics = crit([p.cuda() for p in preds], gt.cuda(), valid.cuda(), 0.8, extro_info="A-")
    assert loss.dtype == torch.float32 and abs(float(loss) - float(g["loss"])) < 2e-6 * float(g["loss"])
    for j, key in enumerate(("A-epe", "A-1px", "A-3px", "A-5px")):
        assert abs(metrics[key] - g["metrics"][j]) < 1e-6
    assert float((crit.grads[0].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["grad0"])).abs().max()) < 1e-9
    assert float((crit.grads[2].cpu()[:, :, ::2, ::2] - torch.from_numpy(g["grad2"])).abs().max()) < 1e-9
    gt_b, valid_b = tr.rotate_gt(gt.cuda())
    # the grids are generated on the device (fp32 trig differs from the CPU's by ulps) and this GT has
    # flows of up to 350 px sampled near the poles, where the grid is steep: a few 1e-4 px at worst
    err = (gt_b.cpu()[:, :, ::2, ::2] - torch.from_numpy(g["gt_b"])).abs()
    assert float(err.max()) < 3e-3 and float(err.mean()) < 2e-5
    assert float(valid_b.sum()) == float(g["valid_b_sum"])


def test_fetch_optimizer_trajectory_vs_reference():
    from prior_flow_amd import train as tr
    g = gc.load("train")
    args = argparse.Namespace(lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=60000, clip=1.0)
    p0, grads = adam_case()
    model = torch.nn.ParameterList([torch.nn.Parameter(p0[:1000].clone().cuda()), torch.nn.Parameter(p0[1000:].clone().cuda())])
    opt, sched = tr.fetch_optimizer(args, model)
    assert model[0].data.data_ptr() == opt.flat.data_ptr() and model[1].grad.data_ptr() == opt.grad[1000:].data_ptr()
    for k, gk in enumerate(grads):
        opt.zero_grad()
        model[0].grad.copy_(gk[:1000]); model[1].grad.copy_(gk[1000:])
        norm = tr.clip_grad_norm_(opt, args.clip)
        assert abs(norm - g["norms"][k]) < 2e-6 * norm
        opt.step(); sched.step()
        assert abs(opt.param_groups[0]["lr"] - g["lrs"][k]) < 1e-15
    got = torch.cat([model[0].detach().cpu(), model[1].detach().cpu()])
    assert float((got - torch.from_numpy(g["p_final"])).abs().max()) < 5e-8
    for i, lr in zip(g["sched_idx"], g["sched_lr"]):
        assert abs(sched.lr_at(int(i)) - lr) <= 1e-15 + 1e-12 * lr
    with pytest.raises(ValueError):
        s2 = tr.OneCycleLinearLR(opt, 1e-4, 3)
        for _ in range(4):
            s2.step()


def test_train_mode_forward_needs_a_device():
    """No CPU path: the training forward fails loudly on CPU tensors instead of falling back."""
    from prior_flow_amd import _lib
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    m.load_state_dict(gc.det_state_dict(state_dict_shapes()), strict=True)
    m = m.train()
    i1, i2 = gc.synthetic_pair(1, 128, 256)
    with pytest.raises(_lib.PfError):
        m(i1, i2, iters=2)


def test_batchnorm_batch_statistics_on_hip_matches_torch():
    """BatchNorm2d left in training mode (the reference's `chairs` stage, train_flow.py:107-108) runs on pf_channel_stats /
    pf_norm_bwd over the whole batch: output, input / gamma / beta gradients and the running statistics against torch's own
    batch_norm; and the training forward refuses a convolution geometry it has no HIP pair for instead of calling torch."""
    import torch.nn as nn
    import torch.nn.functional as F
    from prior_flow_amd import autograd as ag
    from prior_flow_amd._lib import PfError
    torch.manual_seed(3)
    x = (torch.randn(3, 64, 24, 40, device="cuda") * 2.0 + 0.5).requires_grad_(True)
    bn = nn.BatchNorm2d(64).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    ref = nn.BatchNorm2d(64).cuda().train()
    ref.load_state_dict(bn.state_dict())
    w = torch.randn_like(x)
    ag.STATS["hip"] = ag.STATS["torch"] = 0
    y = ag._norm(bn, x)
    (y * w).sum().backward()
    gx, gw, gb = x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()
    x.grad = None
    y2 = F.batch_norm(x, ref.running_mean, ref.running_var, ref.weight, ref.bias, True, ref.momentum, ref.eps)
    (y2 * w).sum().backward()
    assert ag.STATS["torch"] == 0 and ag.STATS["hip"] >= 3
    assert float((y - y2).abs().max()) < 2e-5
    assert float((gx - x.grad).abs().max()) < 2e-4 * float(x.grad.abs().max() + 1)
    assert float((gw - ref.weight.grad).abs().max()) < 2e-3 * float(ref.weight.grad.abs().max())
    assert float((gb - ref.bias.grad).abs().max()) < 2e-3 * float(ref.bias.grad.abs().max())
    assert float((bn.running_mean - ref.running_mean).abs().max()) < 1e-5
    assert float((bn.running_var - ref.running_var).abs().max()) < 1e-4
    conv = nn.Conv2d(64, 64, 3, padding=2, dilation=2).cuda()
    with pytest.raises(PfError):
        ag.conv2d(x.detach(), conv)


@pytest.mark.gpu
def test_seq_loss_batch_equals_the_single_launches_bitwise():
    """Round 6: uniform_loss hands all predictions of a branch to ONE pf_seq_loss_batch launch (train_flow.py:62-71; 24 launches of
    ~13 us per step before).  Term i must equal pf_seq_loss of that prediction bit for bit: gradient seeds and fp64 partial sums --
    with invalid pixels, flows above max_flow, exact zeros of pred - gt (sign 0) and a ragged pixel count."""
    from prior_flow_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, H, W, n = 2, 37, 53, 5
    g = torch.Generator().manual_seed(5)
    gt = (torch.rand(B, 2, H, W, generator=g) * 40 - 20)
    gt[0, :, :3] = 500.0                                      # |gt| >= max_flow: masked
    valid = (torch.rand(B, H, W, generator=g) > 0.2).float()
    weight = torch.rand(H * W, generator=g)
    preds = [gt + torch.randn(B, 2, H, W, generator=g) * (i + 1) for i in range(n)]
    preds[2][1, :, 5:9] = gt[1, :, 5:9]                       # pred == gt: the gradient seed is exactly 0 there
    gt, valid, weight = gt.to(dev), valid.to(dev), weight.to(dev)
    preds = [p.to(dev).contiguous() for p in preds]
    ws = [0.8 ** (n - i - 1) for i in range(n)]
    part1 = torch.full((n, B, 64, 6), float("nan"), dtype=torch.float64, device=dev)
    grads1 = [torch.full_like(gt, float("nan")) for _ in range(n)]
    for i in range(n):
        lib.seq_loss(preds[i], gt, valid, weight, ws[i], 400.0, grads1[i], part1[i])
    part2 = torch.full_like(part1, float("nan"))
    grads2 = [torch.full_like(gt, float("nan")) for _ in range(n)]
    lib.seq_loss_batch(preds, gt, valid, weight, ws, 400.0, grads2, part2)
    torch.cuda.synchronize()
    assert torch.isfinite(part2).all() and float(part2[:, :, :, 0].sum()) > 0
    assert torch.equal(part1, part2)
    for a, b in zip(grads1, grads2):
        assert torch.equal(a, b)
    part3 = torch.full_like(part1, float("nan"))
    lib.seq_loss_batch(preds, gt, valid, weight, ws, 400.0, None, part3)          # the gradient seeds are optional
    torch.cuda.synchronize()
    assert torch.equal(part1, part3)
