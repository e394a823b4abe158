"""Pin the CPU oracle against every golden vector generated from the reference
(oracle/gen_golden.py).  CPU only; runs in the build container and on the GPU box."""
import math

import numpy as np
import pytest
import torch

import golden_cases as gc
import priorflow_oracle as po

T = torch.from_numpy
H8, W8 = gc.H8, gc.W8


def close(a, b, atol, rtol=0.0, what=""):
    a = a if isinstance(a, torch.Tensor) else T(np.asarray(a))
    b = b if isinstance(b, torch.Tensor) else T(np.asarray(b))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e} (atol {atol})"


@pytest.fixture(scope="module")
def grids():
    g = gc.load("grids")
    return {k: T(g[k]) for k in g.files}


@pytest.fixture(scope="module")
def params():
    # shapes come from the product's own module (same state_dict contract as the reference)
    from prior_flow_amd.prior_raft import state_dict_shapes
    return gc.det_state_dict(state_dict_shapes())


def test_rotation_matrices(grids):
    close(po.rotation_x(-math.pi / 2), grids["r_a2b"], 0.0, what="R_A2B")
    close(po.rotation_x(math.pi / 2), grids["r_b2a"], 0.0, what="R_B2A")
    assert grids["r_a2b"][1, 1].item() == pytest.approx(-4.371139e-8, rel=1e-6)


@pytest.mark.parametrize("tag,h,w", [("16x32", 16, 32), ("64x128", 64, 128), ("80x160", 80, 160)])
def test_sample_grids(grids, tag, h, w):
    ra, rb = po.rotation_x(-math.pi / 2), po.rotation_x(math.pi / 2)
    close(po.sample_grid(h, w, ra), grids[f"a2b_{tag}"], 2e-4, what="a2b")
    close(po.sample_grid(h, w, rb), grids[f"b2a_{tag}"], 2e-4, what="b2a")
    # grid(R^T_A2B) == grid(R_B2A) bit-exactly
    assert torch.equal(po.sample_grid(h, w, ra.T), po.sample_grid(h, w, rb))


def test_identity_rotation_gives_identity_grid():
    g = po.sample_grid(16, 32, torch.eye(3))
    xs = torch.arange(32).view(1, 32).expand(16, 32).float()
    ys = torch.arange(16).view(16, 1).expand(16, 32).float()
    close(g[0], xs, 2e-5)
    close(g[1], ys, 2e-5)


def test_sampler_seam_semantics():
    g = gc.load("sampler")
    img = gc.uni("sampler/img", (2, 3, H8, W8), -2, 2)
    co = gc.nasty_coords("sampler", B=2)
    out = po.cycle_bilinear_sampler(img, co[:, 0], co[:, 1])
    close(out, g["out"], 2e-6, what="cycle_bilinear_sampler")
    pts = torch.tensor([[W8 - 0.5, 3.0], [5.0, -0.5], [W8 - 1.0, 2.0], [-0.25, 2.0], [3.0, H8 - 0.5]])
    ones = po.cycle_bilinear_sampler(torch.ones(1, 1, H8, W8), pts[None, :, 0], pts[None, :, 1])
    close(ones.reshape(-1), g["ones"].reshape(-1), 1e-6, what="ones")
    # known answers: x = W-0.5 blends column W-1 with ZERO (not column 0); y=-0.5 fades to 0.5
    close(ones.reshape(-1)[:2], torch.tensor([0.5, 0.5]), 1e-5)


def test_img_rotate(grids):
    im6 = gc.uni("img_rotate/img", (1, 6, 64, 128), -1, 1)
    close(po.img_rotate(im6, grids["a2b_64x128"]), gc.load("img_rotate")["out"], 2e-6)


def test_flo_rotate(grids):
    g = gc.load("flo_rotate")
    fl = gc.flows("flo_rotate", B=2)
    b2a = po.flo_rotate(fl, grids["b2aT_16x32"], grids["b2a_16x32"])
    a2b = po.flo_rotate(fl, grids["a2bT_16x32"], grids["a2b_16x32"])
    close(b2a, g["b2a"], 1e-5, what="b2a")
    close(a2b, g["a2b"], 1e-5, what="a2b")


def test_flo_rotate_zero_flow_is_zero(grids):
    z = torch.zeros(1, 2, 16, 32)
    out = po.flo_rotate(z, grids["b2aT_16x32"], grids["b2a_16x32"])
    assert float(out.abs().max()) == 0.0


def test_corr_pyramid():
    g = gc.load("corr_pyramid")
    f1, f2 = gc.fmaps("corr", B=2)
    pyr = po.build_pyramid(po.corr_volume(f1, f2))
    rows = g["rows"]
    for i, p in enumerate(pyr):
        close(p[rows], g[f"l{i}"], 2e-5, what=f"level {i}")
        assert float(p.double().sum()) == pytest.approx(float(g["checksum"][i]), abs=1e-2 * (1 + i))
        assert float(p.double().abs().sum()) == pytest.approx(float(g["abssum"][i]), rel=1e-6)


def test_corr_known_answers():
    f1, _ = gc.fmaps("corr")
    vol = po.corr_volume(f1, f1).reshape(H8 * W8, H8 * W8)
    diag = (f1.reshape(256, -1) ** 2).sum(0) / 16.0
    close(torch.diagonal(vol), diag, 1e-4, what="diag = |f|^2/16")
    # pyramid level i == corr(f1, avgpool_i(f2)) by linearity
    f1, f2 = gc.fmaps("corr")
    pyr = po.build_pyramid(po.corr_volume(f1, f2))
    f2p = torch.nn.functional.avg_pool2d(f2, 2)
    lvl1 = torch.matmul(f1.reshape(1, 256, -1).transpose(1, 2), f2p.reshape(1, 256, -1)) / 16.0
    close(pyr[1].reshape(1, H8 * W8, -1), lvl1, 2e-5, what="linearity")


def test_dccl(grids):
    g = gc.load("dccl")
    va, vb = gc.volumes("dccl")
    pa, pb = po.build_pyramid(va), po.build_pyramid(vb)
    co = gc.nasty_coords("dccl")
    own, cross = po.dccl_lookup(co, pa, pb, grids["a2bT_16x32"], grids["b2a_16x32"])
    own2, cross2 = po.dccl_lookup(co, pb, pa, grids["b2aT_16x32"], grids["a2b_16x32"])
    close(own[:, :, :8], g["own_a"], 2e-5, what="own_a")
    # white-noise volume sampled through an fp32-rounded grid: 1 ulp of coordinate
    # (6e-6 px) times |dV/dx| <= 8 -> up to ~2e-4
    close(cross[:, :, :8], g["cross_a"], 1e-3, what="cross_a")
    close((own2 + cross2)[:, :, :8], g["corr_b"], 1e-3, what="corr_b")
    close(cross2[:, :, 8:, ::4], g["cross_b_tail"], 1e-3, what="cross_b_tail")
    assert float((cross[:, :, :8] - T(g["cross_a"])).abs().mean()) < 2e-5


def test_dccl_centre_tap_is_volume_diagonal():
    va, vb = gc.volumes("dccl")
    pa, pb = po.build_pyramid(va), po.build_pyramid(vb)
    g = po.grids_for(128, 256)
    own, _ = po.dccl_lookup(po.coords_grid(1, H8, W8), pa, pb, g["a2b_w2c_8"], g["b2a_8"])
    diag = torch.diagonal(va.reshape(H8 * W8, H8 * W8)).reshape(H8, W8)
    close(own[0, 40], diag, 1e-5, what="centre tap of level 0")


def test_warp_gcorr():
    f1, f2 = gc.fmaps("gwc")
    co = gc.nasty_coords("gwc")
    close(po.warp_groupwise_corr(f1, f2, co), gc.load("warp_gcorr")["flaw"], 2e-6)


def test_update_blocks(params):
    ui = gc.update_inputs("upd")
    ga, gb = gc.load("update_A"), gc.load("update_B")
    mf = po.motion_encoder_A(params, "ODDC.encoder.", ui["flow_a"], ui["corr"], ui["flaw_a"],
                             ui["flow_ba"], ui["flaw_ba"])
    close(mf, ga["motion"], 2e-5, what="motion A")
    net, mask, delta = po.update_A(params, ui["net"], ui["inp"], ui["flow_a"], ui["corr"],
                                   ui["flaw_a"], ui["flow_ba"], ui["flaw_ba"])
    close(net, ga["net"], 2e-5, what="net A")
    close(mask[:, 3::8], ga["mask"], 2e-5, what="mask A")
    close(delta, ga["delta"], 2e-5, what="delta A")
    close(po.sepconv_gru(params, "ODDC.gru.", ui["net"], torch.cat([ui["inp"], mf], 1)),
          gc.load("gru")["out"], 2e-5, what="gru")
    mfb = po.motion_encoder_B(params, "update_block.encoder.", ui["flow_a"], ui["corr"])
    close(mfb, gb["motion"], 2e-5, what="motion B")
    net, mask, delta = po.update_B(params, ui["net"], ui["inp"], ui["corr"], ui["flow_a"])
    close(net, gb["net"], 2e-5, what="net B")
    close(mask[:, 3::8], gb["mask"], 2e-5, what="mask B")
    close(delta, gb["delta"], 2e-5, what="delta B")


def test_upsample():
    fl8 = gc.uni("up/flow", (1, 2, H8, W8), -6, 6)
    mk = gc.uni("up/mask", (1, 576, H8, W8), -2, 2)
    close(po.upsample_flow(fl8, mk), gc.load("upsample")["out"], 1e-5)


def test_encoders(params):
    im = gc.uni("enc/img", (2, 3, 128, 256), -1, 1)
    g = gc.load("encoders")
    close(po.encoder(params, "fnet.", im, "instance")[:, ::4], g["fnet"], 5e-5, what="fnet")
    close(po.encoder(params, "cnet.", im, "batch")[:, 1::4], g["cnet"], 5e-5, what="cnet")


NOISE = 2e-5   # mean-EPE bound, ~10x the measured oracle-vs-reference fp32 noise (see below)


def _epe_stats(a, b):
    e = po.epe(a, b if isinstance(b, torch.Tensor) else T(b))
    return float(e.mean()), float(e.max())


def test_forward_128x256(params):
    i1, i2 = gc.synthetic_pair(1, 128, 256)
    g = gc.load("forward_128x256_it12")
    pa, pb = po.forward(params, i1, i2, iters=12)
    sub = lambda t: t[:, :, ::2, ::2]
    # Noise floor: oracle vs reference on this input is 2.3e-6 mean / 1.2e-5 max EPE at iteration
    # 11 (both branches); the reference against itself (1 vs 8 CPU threads) shows the same
    # order (DESIGN.md "Parity").  NOISE = ~10x that floor.
    for i in (0, 2, 6):
        assert _epe_stats(sub(pa[i]), g[f"a{i}"])[0] < NOISE, i
        assert _epe_stats(sub(pb[i]), g[f"b{i}"])[0] < NOISE, i
    for pred, key in ((pa[11], "a11"), (pb[11], "b11")):
        mean, mx = _epe_stats(pred, g[key])
        assert mean < NOISE and mx < 1e-3, (key, mean, mx)
    tm = po.forward(params, i1, i2, iters=12, test_mode=True)
    assert torch.equal(tm, pa[11])


def test_forward_short_and_init_flow(params):
    i1, i2 = gc.synthetic_pair(1, 128, 256)
    sub = lambda t: t[:, :, ::2, ::2]
    g3, g1 = gc.load("forward_128x256_it3"), gc.load("forward_128x256_it1")
    pa, pb = po.forward(params, i1, i2, iters=3)
    assert _epe_stats(sub(pa[2]), g3["a2"])[0] < NOISE
    assert _epe_stats(sub(pb[2]), g3["b2"])[0] < NOISE
    pa, pb = po.forward(params, i1, i2, iters=1)
    assert _epe_stats(sub(pa[0]), g1["a0"])[0] < NOISE
    assert _epe_stats(sub(pb[0]), g1["b0"])[0] < NOISE
    init = gc.uni("fwd/init_flow", (1, 2, 16, 32), -3, 3)
    out = po.forward(params, i1, i2, iters=3, init_flow=init, test_mode=True)
    assert _epe_stats(out, gc.load("forward_128x256_init")["out"])[0] < NOISE


def test_forward_batch2(params):
    j1, j2 = gc.synthetic_pair(2, 128, 256, seed=77)
    out = po.forward(params, j1, j2, iters=2, test_mode=True)
    assert _epe_stats(out[:, :, ::2, ::2], gc.load("forward_128x256_b2")["out"])[0] < NOISE
    # batch independence: sample 1 alone gives the same flow
    solo = po.forward(params, j1[1:], j2[1:], iters=2, test_mode=True)
    assert _epe_stats(solo, out[1:])[0] < NOISE


def test_forward_demo_config1(params):
    """BASELINE.json configs[0]: demo.py-style randn 'images', 256x512, iters=4 (inputs stored in the fixture)."""
    g = gc.load("forward_256x512_demo")
    out = po.forward(params, T(g["image1"]).float(), T(g["image2"]).float(), iters=4, test_mode=True)
    mean, mx = _epe_stats(out[:, :, ::2, ::2], g["out"])
    assert mean < NOISE, (mean, mx)


def test_config4_region_metrics_and_ckpt_fixture():
    """BASELINE.json configs[4] (640x1280, iters=32, EPE by region): the oracle's region arithmetic on the reference's
    own (sub-sampled) flow reproduces the order of the reference's region numbers; the full-resolution check is the GPU
    test (tests/test_hip_forward.py::test_forward_640x1280_iters32) -- the 38 s oracle forward is not run here."""
    from gen_golden_configs import CFG4, config4_gt
    g = gc.load("forward_640x1280_it32")
    assert g["out"].shape == (1, 2, CFG4["h"] // 4, CFG4["w"] // 4) and g["regions"].shape == (4, 3)
    gt = config4_gt()
    assert tuple(gt.shape) == (2, CFG4["h"], CFG4["w"])
    # every 4th pixel of flow and ground truth: region means of a smooth field move by a few percent at most
    sub = po.region_metrics([T(g["out"])[0]], [gt[:, ::4, ::4]])
    for r, name in enumerate(("All", "Equator", "Poles", "Center")):
        assert abs(sub[name]["epe"] - g["regions"][r, 0]) < 0.05 * g["regions"][r, 0], (name, sub[name], g["regions"][r])


# ---- evaluation counterpart: oracle vs the reference's own helpers -------------
def test_eval_sepe_and_masks():
    g = gc.load("eval")
    pre, gt = gc.flows("eval/pre", 2), gc.flows("eval/gt", 2)
    close(po.great_circle_distance(pre, gt), g["sd_rand"], 0.0, what="SEPE random flows")
    kat = torch.zeros(1, 2, 64, 128)
    kat[:, 0] = 4.0
    sd = po.great_circle_distance(kat, torch.zeros_like(kat))[0, :, 0]
    close(sd, g["sd_kat"], 0.0, what="SEPE +4 px u-flow")
    assert abs(float(sd[31]) - 0.19629) < 1e-5            # SURVEY.md 8c known answer at the equator row
    for h, w, na, nb in ((16, 32, 256, 104), (64, 128, 32 * 128, 1648)):
        a, b = po.generate_polemask(h, w)
        assert np.array_equal(a.numpy().astype(np.uint8), g[f"pole_a_{h}x{w}"])
        assert np.array_equal(b.numpy().astype(np.uint8), g[f"pole_b_{h}x{w}"])
        assert int(a.sum()) == na and int(b.sum()) == nb
    u = po.spherical_mask(64, 128)
    close(u[:, 0], g["uni_col"], 0.0, what="spherical_mask")
    assert abs(float(u.double().sum()) - 1.0) < 1e-6 and abs(float(g["uni_sum"]) - 1.0) < 1e-6


def test_eval_region_metrics():
    """The arithmetic of evaluate.py:196-227 / :234-282 on fixed flows (reference numbers in eval.npz)."""
    from gen_golden_eval import eval_samples
    g = gc.load("eval")
    s = eval_samples()
    r = po.region_metrics([p for p, _ in s], [q for _, q in s])
    for i, name in enumerate(("All", "Equator", "Poles", "Center")):
        for j, key in enumerate(("epe", "sd", "sd_uni")):
            assert abs(r[name][key] - g["regions"][i, j]) <= 2e-6 * abs(g["regions"][i, j]), (name, key)


# ---- training-step counterpart: oracle vs the reference's train_flow.py helpers -------
def test_train_loss_schedule_optimizer():
    from gen_golden_train import adam_case, loss_case
    g = gc.load("train")
    preds, gt, valid = loss_case()
    loss, met, grads = po.uniform_loss(preds, gt, valid, gamma=0.8)
    assert abs(loss - float(g["loss"])) < 2e-7 * abs(loss)                     # reference sums in fp32
    for j, key in enumerate(("epe", "1px", "3px", "5px")):
        assert abs(met[key] - g["metrics"][j]) < 1e-6
    close(grads[0][:, :, ::4, ::4], g["grad0"], 0.0, what="autograd grad of prediction 0")
    close(grads[2][:, :, ::2, ::2], g["grad2"], 0.0, what="autograd grad of prediction 2")
    for i, lr in zip(g["sched_idx"], g["sched_lr"]):
        assert abs(po.one_cycle_lr(int(i), 1e-4, 60000) - lr) <= 1e-15 + 1e-12 * lr
    assert abs(po.one_cycle_lr(1, 1e-4, 60000) - 4.032e-6) < 1e-9           # SURVEY.md 8c known answers
    assert abs(po.one_cycle_lr(2, 1e-4, 60000) - 4.064e-6) < 1e-9
    p0, grads = adam_case()
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for k, gk in enumerate(grads):
        norm = float((gk.double() ** 2).sum().sqrt())
        assert abs(norm - g["norms"][k]) < 2e-6 * norm
        p, m, v = po.adamw_step(p, gk * np.float32(po.clip_coef(norm, 1.0)), m, v, po.one_cycle_lr(k, 1e-4, 60000), k + 1, 5e-5)
        assert abs(po.one_cycle_lr(k + 1, 1e-4, 60000) - g["lrs"][k]) < 1e-15
    close(p, g["p_final"], 5e-8, what="AdamW + clip trajectory")


def test_forward_odd_sizes(params):
    """1/8 maps that are not multiples of 8 (17x27, 20x45): avg_pool2d floors the odd pyramid levels.
    With an odd W8 the reference has a knife edge of its own: the cross-view grid interpolated across
    its wrap seam equals W8 >> 1 exactly (e.g. (0.064 + 43.936) / 2 = 22.0 at W8 = 45), the modulus
    boundary of pyramid level 1, so fp32 rounding decides whether ~0.5 % of that level's samples wrap
    to column 0 or fall off the right edge.  Branch B of the 160x360 case sits on it (measured: the
    two restatements differ by 1.8e-3 mean EPE there while a 1e-6 input perturbation moves either by
    2e-6), so that one output is pinned loosely."""
    g = gc.load("forward_odd")
    for (h, w), tol_a, tol_b in (((136, 216), 2e-5, 2e-5), ((160, 360), 1e-4, 5e-3)):
        i1, i2 = gc.synthetic_pair(1, h, w, seed=31)
        pa, pb = po.forward(params, i1, i2, iters=3)
        ea = po.epe(pa[-1], T(g[f"a_{h}x{w}"]))
        eb = po.epe(pb[-1][:, :, ::2, ::2], T(g[f"b_{h}x{w}"]))
        assert float(ea.mean()) < tol_a and float(eb.mean()) < tol_b, (h, w, float(ea.mean()), float(eb.mean()))


def test_flow_rotation_round_trip():
    """SURVEY.md 8c known answer: flo_B2A(flo_A2B(f)) ~= f away from the poles of both views (bilinear
    resampling of a smooth flow: measured 2e-3 px mean / 1e-2 px max at 64x128)."""
    H, W = 64, 128
    ga, gb = po.sample_grid(H, W, po.rotation_x(-math.pi / 2)), po.sample_grid(H, W, po.rotation_x(math.pi / 2))
    ys = torch.arange(H).view(1, H, 1).float()
    xs = torch.arange(W).view(1, 1, W).float()
    f = torch.stack([1.5 * torch.sin(2 * math.pi * xs / W) * torch.ones(1, H, W) + 0.5,
                     0.8 * torch.cos(2 * math.pi * ys / H) * torch.ones(1, H, W)], 1)
    back = po.flo_rotate(po.flo_rotate(f, gb, ga), ga, gb)
    pole_a, pole_b = po.generate_polemask(H, W)
    keep = ((pole_a == 0) & (pole_b == 0))[0]
    e = po.epe(back, f)[0][keep]
    assert float(e.mean()) < 5e-3 and float(e.max()) < 3e-2


def test_train_step_gradients_vs_reference(params):
    """One training step of the reference (forward in train mode with frozen BN, uniform_loss on both
    branches, backward; B=2, 128x256, iters=3; oracle/gen_golden_train_step.py): autograd through the
    oracle's forward reproduces the loss, the total gradient norm and the gradient slices -- the pin for
    the backward kernels."""
    from gen_golden_train_step import SLICES, step_inputs
    g = gc.load("train_step")
    p = {k: v.clone() for k, v in params.items()}
    leaf = [k for k, v in p.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    for k in leaf:
        p[k].requires_grad_(True)
    i1, i2, gt, valid = step_inputs()
    with torch.no_grad():
        gt_b = po.flo_rotate(gt, po.sample_grid(128, 256, po.rotation_x(math.pi / 2)), po.sample_grid(128, 256, po.rotation_x(-math.pi / 2)))
        valid_b = ((gt_b[:, 0].abs() < 1000) & (gt_b[:, 1].abs() < 1000)).float()
    uni = po.spherical_mask(128, 256)[None]

    def loss_fn(preds, tgt, v, gamma=0.8):
        ok = (v >= 0.5) & (torch.sum(tgt ** 2, dim=1).sqrt() < 400)
        n = len(preds)
        return sum(gamma ** (n - i - 1) * torch.sum(ok * uni * torch.sum((preds[i] - tgt).abs(), dim=1)) for i in range(n))

    pa, pb = po.forward_with_grad(p, i1, i2, iters=3)
    loss = loss_fn(pa, gt, valid) + loss_fn(pb, gt_b, valid_b)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    total = math.sqrt(sum(float((p[k].grad.double() ** 2).sum()) for k in leaf if p[k].grad is not None))
    assert abs(total - float(g["grad_norm"])) < 1e-5 * float(g["grad_norm"])
    for k, sl in SLICES.items():
        want = T(g["g:" + k])
        close(p[k].grad[sl], want, 2e-3 * float(want.abs().max()) + 1e-9, what="grad " + k)
