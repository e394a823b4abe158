"""Per-kernel parity cases: C-ABI library (through prior-flow_amd/_lib.py) vs the CPU oracle
and the reference-generated golden vectors, on the same seeded inputs.

The same cases run (a) on the GPU against libpriorflow_hip.so  -- tests/test_hip_kernels.py,
the parity gate proper -- and (b) on the CPU against the host-emulation build of the
per-element kernels -- tests/test_emu_kernels.py, which only checks index / wrap / padding
logic in the GPU-less build container.
"""
import math

import numpy as np
import torch

import golden_cases as gc
import priorflow_oracle as po

T = torch.from_numpy
H8, W8 = gc.H8, gc.W8
N = H8 * W8


def maxerr(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max())


def check(a, b, atol, what):
    a = a.detach().cpu()
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else T(np.asarray(b))
    assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a).all(), f"{what}: non-finite output"
    err = float((a - b).abs().max())
    assert err <= atol, f"{what}: max err {err:.3e} > {atol:.1e}"
    return err


def cl(x):
    """NCHW -> channel-last rows [B*H*W, C]."""
    B, C = x.shape[:2]
    return x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()


def uncl(rows, B, H, W):
    return rows.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def grids16(dev):
    g = gc.load("grids")
    return {k: T(g[k]).to(dev).contiguous() for k in g.files if k.endswith("16x32")}


# ------------------------------------------------------------------------------------------
def case_sample_grid(lib, dev):
    g = gc.load("grids")
    for tag, (h, w) in (("16x32", (16, 32)), ("64x128", (64, 128)), ("80x160", (80, 160))):
        for name, theta in (("a2b", -math.pi / 2), ("b2a", math.pi / 2)):
            R = po.rotation_x(theta)
            out = torch.empty(2, h, w, device=dev)
            lib.sample_grid(out, R)
            # device libm vs torch-CPU libm: atan2 near the poles amplifies 1-ulp
            # differences of sin/cos to ~1e-4 px (DESIGN.md "Sample grids")
            check(out, g[f"{name}_{tag}"], 5e-4 * (w / 32) ** 0.5, f"grid {name} {tag}")
            assert float((out.cpu() - T(g[f"{name}_{tag}"])).abs().mean()) < 2e-5
    # identity rotation -> identity grid
    out = torch.empty(2, 16, 32, device=dev)
    lib.sample_grid(out, torch.eye(3))
    check(out, po.coords_grid(1, 16, 32)[0], 3e-5, "identity grid")


def case_img_rotate(lib, dev):
    g = gc.load("grids")
    im6 = gc.uni("img_rotate/img", (1, 6, 64, 128), -1, 1)
    out = torch.empty_like(im6, device=dev)
    lib.img_rotate(im6.to(dev), T(g["a2b_64x128"]).to(dev), out)
    check(out, gc.load("img_rotate")["out"], 2e-6, "img_rotate vs reference")
    # generic sampler semantics incl. seams: resample with "nasty" coordinates as the grid
    img = gc.uni("sampler/img", (2, 3, H8, W8), -2, 2)
    co = gc.nasty_coords("sampler", B=2)
    ref = T(gc.load("sampler")["out"])
    for b in range(2):
        o = torch.empty(1, 3, H8, W8, device=dev)
        lib.img_rotate(img[b:b + 1].to(dev).contiguous(), co[b].to(dev).contiguous(), o)
        check(o, ref[b:b + 1], 2e-6, f"seam sampler b={b}")


def case_normalise_images(lib, dev):
    """pf_normalise_images against numpy's fp32 `2 * (image / 255.0) - 1.0` (IEEE division, the reference's CPU arithmetic,
    core/prior_raft.py:121-122) bit for bit, on every integer pixel value, fractional values and values outside [0, 255]."""
    g = torch.Generator().manual_seed(3)
    n = 2 * 3 * 16 * 24
    i1 = torch.cat([torch.arange(256, dtype=torch.float32), torch.rand(n - 256, generator=g) * 300 - 20]).reshape(2, 3, 16, 24)
    i2 = (torch.rand(2, 3, 16, 24, generator=g) * 255).round()
    f = torch.full((8, 3, 16, 24), 7.0, device=dev)
    c = torch.full((4, 3, 16, 24), 7.0, device=dev)
    lib.normalise_images(i1.to(dev), i2.to(dev), f[:2], f[2:4], c[:2])
    r1 = (np.float32(2) * (i1.numpy() / np.float32(255)) - np.float32(1)).astype(np.float32)
    r2 = (np.float32(2) * (i2.numpy() / np.float32(255)) - np.float32(1)).astype(np.float32)
    assert r1.dtype == np.float32
    assert np.array_equal(f[:2].cpu().numpy(), r1) and np.array_equal(c[:2].cpu().numpy(), r1), "image1"
    assert np.array_equal(f[2:4].cpu().numpy(), r2), "image2"
    assert float(f[4:].min()) == 7.0 and float(c[2:].max()) == 7.0, "wrote outside its slices"
    lib.normalise_images(i1.to(dev), i2.to(dev), f[4:6], f[6:], None)           # the context destination is optional
    assert np.array_equal(f[4:6].cpu().numpy(), r1) and np.array_equal(f[6:].cpu().numpy(), r2)


def case_prepare_images(lib, dev):
    """pf_prepare_images (the input stage in one launch) against pf_normalise_images + pf_img_rotate + the copy it replaces, bit
    for bit: on the real A->B grid and on a grid of nasty coordinates (seam crossers, out-of-range rows, negatives), with and
    without the context batch."""
    g = gc.load("grids")
    gen = torch.Generator().manual_seed(11)
    Bn, Hh, Ww = 2, 64, 128
    i1 = (torch.rand(Bn, 3, Hh, Ww, generator=gen) * 255).round().to(dev)
    i2 = (torch.rand(Bn, 3, Hh, Ww, generator=gen) * 300 - 20).to(dev)
    grids = [T(g["a2b_64x128"]).to(dev).contiguous()]
    grids.append(gc.nasty_coords("prep_img", B=1, h=Hh, w=Ww)[0].to(dev).contiguous())    # pixel coordinates, some far outside
    for grid in grids:
        f_ref = torch.full((4 * Bn, 3, Hh, Ww), 7.0, device=dev)
        c_ref = torch.full((2 * Bn, 3, Hh, Ww), 7.0, device=dev)
        lib.normalise_images(i1, i2, f_ref[:Bn], f_ref[Bn:2 * Bn], c_ref[:Bn])
        lib.img_rotate(f_ref[:2 * Bn], grid, f_ref[2 * Bn:])
        c_ref[Bn:].copy_(f_ref[2 * Bn:3 * Bn])
        f = torch.full_like(f_ref, 5.0)
        c = torch.full_like(c_ref, 5.0)
        lib.prepare_images(i1, i2, grid, f, c)
        assert torch.equal(f, f_ref), "img_f"
        assert torch.equal(c, c_ref), "img_c"
        f2 = torch.full_like(f_ref, 5.0)
        lib.prepare_images(i1, i2, grid, f2, None)
        assert torch.equal(f2, f_ref), "img_f without the context batch"


def case_flow_prep(lib, dev):
    co = gc.nasty_coords("prep", B=2)
    flow = torch.empty(2, 2, H8, W8, device=dev)
    d0 = torch.zeros(2 * N, 6, device=dev)
    d1 = torch.zeros(2 * N, 2, device=dev)
    lib.flow_prep(co.to(dev), flow, d0, 3, d1, 0)
    want = co - po.coords_grid(2, H8, W8)
    check(flow, want, 0.0, "flow planar")
    check(d0[:, 3:5], cl(want), 0.0, "flow dst0")
    check(d1, cl(want), 0.0, "flow dst1")
    assert float(d0[:, :3].abs().max()) == 0.0 and float(d0[:, 5].abs().max()) == 0.0


def case_flo_rotate(lib, dev):
    g = grids16(dev)
    gold = gc.load("flo_rotate")
    fl = gc.flows("flo_rotate", B=2)
    for name, w2c, c2w in (("b2a", "b2aT_16x32", "b2a_16x32"), ("a2b", "a2bT_16x32", "a2b_16x32")):
        out = torch.empty(2, 2, H8, W8, device=dev)
        d0 = torch.zeros(2 * N, 4, device=dev)
        lib.flo_rotate(fl.to(dev), g[w2c], g[c2w], out, d0, 2)
        check(out, gold[name], 1e-5, f"flo_rotate {name} vs reference")
        check(d0[:, 2:4], cl(out.cpu()), 0.0, "flo_rotate dst")
    z = torch.zeros(1, 2, H8, W8, device=dev)
    out = torch.empty_like(z)
    lib.flo_rotate(z, g["b2aT_16x32"], g["b2a_16x32"], out)
    assert float(out.abs().max()) == 0.0, "zero flow must rotate to exactly zero"
    # known answer: flo_B2A(flo_A2B(f)) ~= f away from the poles of both views
    import math
    H, W = 64, 128
    ga = torch.empty(2, H, W, device=dev)
    gb = torch.empty(2, H, W, device=dev)
    lib.sample_grid(ga, po.rotation_x(-math.pi / 2))
    lib.sample_grid(gb, po.rotation_x(math.pi / 2))
    ys = torch.arange(H).view(1, H, 1).float()
    xs = torch.arange(W).view(1, 1, W).float()
    f = torch.stack([1.5 * torch.sin(2 * math.pi * xs / W) * torch.ones(1, H, W) + 0.5,
                     0.8 * torch.cos(2 * math.pi * ys / H) * torch.ones(1, H, W)], 1).contiguous()
    fb, back = torch.empty(1, 2, H, W, device=dev), torch.empty(1, 2, H, W, device=dev)
    lib.flo_rotate(f.to(dev), gb, ga, fb)
    lib.flo_rotate(fb, ga, gb, back)
    pole_a, pole_b = po.generate_polemask(H, W)
    keep = ((pole_a == 0) & (pole_b == 0))[0]
    e = po.epe(back.cpu(), f)[0][keep]
    assert float(e.mean()) < 5e-3 and float(e.max()) < 3e-2, "flow rotation round trip"


def _pyr_rows(pyr):
    return [p.reshape(p.shape[0], -1).contiguous() for p in pyr]


def case_dccl(lib, dev):
    g = grids16(dev)
    gold = gc.load("dccl")
    va, vb = gc.volumes("dccl")
    pa, pb = _pyr_rows(po.build_pyramid(va)), _pyr_rows(po.build_pyramid(vb))
    pa_d, pb_d = [p.to(dev) for p in pa], [p.to(dev) for p in pb]
    co = gc.nasty_coords("dccl").to(dev)
    LD = 324
    own = torch.empty(N, LD, device=dev)
    raw = torch.empty(N, LD, device=dev)
    out = torch.empty(N, LD, device=dev)
    lib.dccl_lookup(co, pa_d, pb_d, g["a2bT_16x32"], own, raw)
    lib.dccl_combine(own, raw, g["b2a_16x32"], out, 1, H8, W8)
    own_n = uncl(own.cpu(), 1, H8, W8)
    corr_n = uncl(out.cpu(), 1, H8, W8)
    check(own_n[:, :, :8], gold["own_a"], 2e-5, "own_a vs reference")
    want = T(gold["own_a"]) + T(gold["cross_a"])
    check(corr_n[:, :, :8], want, 1e-3, "corr_a vs reference")          # see test_oracle_golden
    assert float((corr_n[:, :, :8] - want).abs().mean()) < 2e-5
    # interleaved-grid variant (pf_dccl_lookup_il): bit-identical to the planar-grid launch
    gw = g["a2bT_16x32"]
    g_il = gw.reshape(2, -1).t().contiguous()
    own_i, raw_i = torch.empty(N, LD, device=dev), torch.empty(N, LD, device=dev)
    lib.dccl_lookup(co, pa_d, pb_d, gw, own_i, raw_i, g_il)
    assert torch.equal(own_i, own) and torch.equal(raw_i, raw), "interleaved grid changes the lookup"
    # other direction (B looks into A)
    lib.dccl_lookup(co, pb_d, pa_d, g["b2aT_16x32"], own, raw)
    lib.dccl_combine(own, raw, g["a2b_16x32"], out, 1, H8, W8)
    corr_n = uncl(out.cpu(), 1, H8, W8)
    check(corr_n[:, :, :8], gold["corr_b"], 1e-3, "corr_b vs reference")
    # and against the oracle on ALL pixels
    o_own, o_cross = po.dccl_lookup(co.cpu(), po.build_pyramid(vb), po.build_pyramid(va),
                                    g["b2aT_16x32"].cpu(), g["a2b_16x32"].cpu())
    check(corr_n, o_own + o_cross, 1e-3, "corr_b vs oracle (all pixels)")
    # padded row stride
    own2 = torch.full((N, 336), 7.0, device=dev)
    raw2 = torch.full((N, 336), 7.0, device=dev)
    lib.dccl_lookup(co, pb_d, pa_d, g["b2aT_16x32"], own2, raw2)
    check(own2[:, :324], own, 0.0, "ld=336 own")
    assert float((own2[:, 324:] - 7.0).abs().max()) == 0.0, "padding columns must stay untouched"


def case_warp_gcorr(lib, dev):
    f1, f2 = gc.fmaps("gwc")
    co = gc.nasty_coords("gwc")
    dst = torch.zeros(N, 8, device=dev)
    f1c, f2c = cl(f1).to(dev), cl(f2).to(dev)
    lib.warp_gcorr(f1c, f2c, co.to(dev), False, dst, 0)
    flow = (co - po.coords_grid(1, H8, W8)).contiguous()
    lib.warp_gcorr(f1c, f2c, flow.to(dev), True, dst, 4)
    want = T(gc.load("warp_gcorr")["flaw"])
    check(uncl(dst[:, :4].cpu(), 1, H8, W8), want, 3e-6, "warp_gcorr vs reference")
    # coords0 + (coords1 - coords0) differs from coords1 by an ulp of the coordinate
    check(uncl(dst[:, 4:].cpu(), 1, H8, W8), want, 2e-4, "warp_gcorr add_grid")


def case_motion_prep(lib, dev):
    """pf_motion_prep (core/prior_raft.py:171-182 in one launch) against the oracle: flows of both branches,
    flo_rotate(flow_B) into view A, groupwise correlations at coords1_A and at coords0 + flow_B_A."""
    import math
    B = 2
    c1a, c1b = gc.nasty_coords("mprep/a", B), gc.nasty_coords("mprep/b", B)
    f1, f2 = gc.fmaps("mprep", B)
    g = po.grids_for(8 * H8, 8 * W8)
    flow4, flow2 = torch.zeros(B * N, 4, device=dev), torch.zeros(B * N, 2, device=dev)
    xa, xb = torch.full((B * N, 8), 7.25, device=dev), torch.full((B * N, 4), 7.25, device=dev)
    conf = torch.zeros(B * N, 8, device=dev)
    ga = torch.empty(2, H8, W8, device=dev)
    gb = torch.empty(2, H8, W8, device=dev)
    lib.sample_grid(ga, po.rotation_x(-math.pi / 2))
    lib.sample_grid(gb, po.rotation_x(math.pi / 2))
    lib.motion_prep(c1a.to(dev), c1b.to(dev), ga, gb, cl(f1).to(dev), cl(f2).to(dev), flow4, flow2, conf, xa, 4, xb, 2)
    c0 = po.coords_grid(B, H8, W8)
    flow_a, flow_b = c1a - c0, c1b - c0
    flow_ba = po.flo_rotate(flow_b, g["b2a_w2c_8"], g["b2a_8"])      # W2C = grid(R_B2A^T), C2W = grid(R_B2A) (:179)
    check(flow4[:, :2], cl(flow_a), 0.0, "flow_A")
    check(flow2, cl(flow_b), 0.0, "flow_B")
    check(flow4[:, 2:], cl(flow_ba), 1e-4, "flow_B_A")      # coordinates up to 2.5 W: an ulp of the wrapped end point is 1e-5
    check(xa[:, 4:], flow4.cpu(), 0.0, "x_a tail")
    check(xb[:, 2:], flow2.cpu(), 0.0, "x_b tail")
    assert float(xa[:, :4].min()) == 7.25 and float(xb[:, :2].max()) == 7.25
    check(uncl(conf[:, :4].cpu(), B, H8, W8), po.warp_groupwise_corr(f1, f2, c1a), 3e-6, "flaw_A")
    got_ba = uncl(flow4[:, 2:].cpu(), B, H8, W8)
    check(uncl(conf[:, 4:].cpu(), B, H8, W8), po.warp_groupwise_corr(f1, f2, c0 + got_ba), 3e-6, "flaw_B_A")


def case_conf_stem(lib, dev):
    """pf_conf_stem: relu(conv3x3 32->16(relu(conv3x3 8->32(x)))) in one launch (core/update.py:193-194) against torch
    conv2d, on a ragged map (17 x 27: partial tiles on both axes), with padded row strides and column offsets."""
    import torch.nn.functional as F
    B, h, w = 2, 17, 27
    x = gc.uni("confstem/x", (B, 8, h, w), -1, 1)
    w1, b1 = gc.uni("confstem/w1", (32, 8, 3, 3), -0.3, 0.3), gc.uni("confstem/b1", (32,), -0.2, 0.2)
    w2, b2 = gc.uni("confstem/w2", (16, 32, 3, 3), -0.2, 0.2), gc.uni("confstem/b2", (16,), -0.2, 0.2)
    want = F.relu(F.conv2d(F.relu(F.conv2d(x, w1, b1, padding=1)), w2, b2, padding=1))
    pack = lambda t: t.permute(2, 3, 1, 0).reshape(-1, t.shape[0]).contiguous().to(dev)       # [KH*KW*Cin][Cout]
    xin = torch.full((B * h * w, 12), 9.5, device=dev)
    xin[:, 4:] = cl(x).to(dev)
    out = torch.full((B * h * w, 24), 7.25, device=dev)
    lib.conf_stem(xin, 4, pack(w1), b1.to(dev), pack(w2), b2.to(dev), out, 8, B, h, w)
    check(uncl(out[:, 8:].cpu(), B, h, w), want, 5e-6, "conf stem")      # 288-term fp32 sums of O(1) values
    assert float(out[:, :8].min()) == 7.25 and float(out[:, :8].max()) == 7.25, "columns outside the slice were written"
    assert float((want == 0).float().mean()) > 0.05            # the ReLUs are active


def case_upsample(lib, dev):
    fl8 = gc.uni("up/flow", (1, 2, H8, W8), -6, 6)
    mk = gc.uni("up/mask", (1, 576, H8, W8), -2, 2)
    coords1 = (po.coords_grid(1, H8, W8) + fl8).contiguous()
    out = torch.empty(1, 2, 8 * H8, 8 * W8, device=dev)
    lib.upsample_flow(coords1.to(dev), cl(mk).to(dev), out)
    # coords0 + flow - coords0 rounds at the coordinate's ulp (<= 32 * 2^-24 * 8)
    check(out, gc.load("upsample")["out"], 3e-5, "upsample vs reference")


def case_warp_gcorr_backward(lib, dev):
    """pf_warp_gcorr_bwd vs autograd through the oracle's warp + groupwise correlation."""
    f1, f2 = gc.fmaps("wgb", 2)
    f1, f2 = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    co = gc.nasty_coords("wgb", 2)
    G = gc.uni("wgb/G", (2, 4, H8, W8), -1, 1)
    (po.warp_groupwise_corr(f1, f2, co) * G).sum().backward()
    d_flaw = torch.zeros(2 * N, 8, device=dev)
    d_flaw[:, 2:6] = cl(G).to(dev)
    d1, d2 = torch.zeros(2 * N, 256, device=dev), torch.zeros(2 * N, 256, device=dev)
    lib.warp_gcorr_bwd(cl(f1.detach()).to(dev).contiguous(), cl(f2.detach()).to(dev).contiguous(), co.to(dev), False,
                       d_flaw, 2, d1, d2)
    check(d1, cl(f1.grad), 2e-6, "d f1")
    check(d2, cl(f2.grad), 5e-6, "d f2 (scatter)")


def case_upsample_backward(lib, dev):
    """pf_upsample_flow_bwd vs autograd through the oracle's convex upsampling."""
    fl = gc.uni("upb/flow", (2, 2, H8, W8), -6, 6).requires_grad_(True)
    mk = gc.uni("upb/mask", (2, 576, H8, W8), -2, 2).requires_grad_(True)
    G = gc.uni("upb/G", (2, 2, 8 * H8, 8 * W8), -1, 1)
    (po.upsample_flow(fl, mk) * G).sum().backward()
    coords1 = (po.coords_grid(2, H8, W8) + fl.detach()).contiguous().to(dev)
    d_mask = torch.full((2 * N, 580), 7.0, device=dev)
    d_flow = torch.zeros(2, 2, H8, W8, device=dev)
    lib.upsample_flow_bwd(coords1, cl(mk.detach()).to(dev).contiguous(), G.to(dev), d_mask, d_flow)
    check(d_mask[:, :576], cl(mk.grad), 2e-5, "d mask logits")
    check(d_flow, fl.grad, 1e-5 * float(fl.grad.abs().max()), "d coarse flow")      # 576 fp32 atomic contributions per pixel
    assert float((d_mask[:, 576:] - 7.0).abs().max()) == 0.0


def case_coords_add(lib, dev):
    co = gc.nasty_coords("cadd", B=2)
    delta = gc.uni("cadd/delta", (2 * N, 4), -1, 1)
    c1 = co.clone().to(dev)
    lib.coords_add(c1, delta.to(dev))
    want = co + uncl(delta[:, :2], 2, H8, W8)
    check(c1, want, 0.0, "coords_add")


def case_direct_conv(lib, dev, params):
    ui = gc.update_inputs("upd")
    # 7x7 2->128 + relu  (ODDC.encoder.convf1_A)
    w, b = params["ODDC.encoder.convf1_A.weight"], params["ODDC.encoder.convf1_A.bias"]
    want = torch.relu(torch.nn.functional.conv2d(ui["flow_a"], w, b, padding=3))
    wp = w.permute(2, 3, 1, 0).reshape(49, 2, 128).contiguous()
    x = torch.zeros(N, 4, device=dev)
    x[:, 1:3] = cl(ui["flow_a"]).to(dev)
    out = torch.zeros(N, 130, device=dev)
    lib.conv2d_direct(x, 1, 2, wp.to(dev), b.to(dev), out, 2, 128, 7, 7, True, 1, H8, W8)
    # on the GPU the 2 -> 128 stems run as 3-pass bf16-split MFMAs (pf_flow_stem.hip: 5e-5 on outputs of +-10); the CPU emulation
    # and the vector-ALU form are exact fp32
    tol7 = 1e-4 if torch.device(dev).type == "cuda" else 2e-5
    check(uncl(out[:, 2:].cpu(), 1, H8, W8), want, tol7, "direct 7x7")
    assert float(out[:, :2].abs().max()) == 0.0
    # the same stem for three inputs in ONE launch (pf_conv2d_direct_group): bit-identical to single launches
    x4 = torch.zeros(N, 4, device=dev)
    x4[:, :2] = cl(ui["flow_a"]).to(dev)
    x4[:, 2:] = cl(ui["flow_a"].flip(1) * 0.5).to(dev)
    x2 = cl(-ui["flow_a"]).to(dev).contiguous()
    wts = [(wp * s).to(dev) for s in (1.0, 0.5, -1.0)]
    outs = [torch.zeros(N, 128, device=dev) for _ in range(3)]
    probs = [(x4, 0, wts[0], b.to(dev), outs[0], 0), (x4, 2, wts[1], b.to(dev), outs[1], 0),
             (x2, 0, wts[2], b.to(dev), outs[2], 0)]
    lib.conv2d_direct_group(probs, 2, 128, 7, 7, True, 1, H8, W8)
    for i, (xi, off, wi, bi, _, _) in enumerate(probs):
        single = torch.zeros(N, 128, device=dev)
        lib.conv2d_direct(xi, off, 2, wi, bi, single, 0, 128, 7, 7, True, 1, H8, W8)
        assert torch.equal(single, outs[i]), f"direct group problem {i}"
    check(uncl(outs[0].cpu(), 1, H8, W8), want, tol7, "direct 7x7 group[0]")
    import pytest
    from prior_flow_amd._lib import PfError
    with pytest.raises(PfError):                          # overlapping outputs in one group are refused
        lib.conv2d_direct_group([probs[0], (x2, 0, wts[2], b.to(dev), outs[0], 0)], 2, 128, 7, 7, True, 1, H8, W8)
    # 3x3 8->32 relu -> 3x3 32->16 relu (confidence stem)
    w1, b1 = params["ODDC.encoder.conv_conf1.weight"], params["ODDC.encoder.conv_conf1.bias"]
    w2, b2 = params["ODDC.encoder.conv_conf2.weight"], params["ODDC.encoder.conv_conf2.bias"]
    xin = torch.cat([ui["flaw_a"], ui["flaw_ba"]], 1)
    want = torch.relu(torch.nn.functional.conv2d(
        torch.relu(torch.nn.functional.conv2d(xin, w1, b1, padding=1)), w2, b2, padding=1))
    mid = torch.empty(N, 32, device=dev)
    out = torch.empty(N, 16, device=dev)
    lib.conv2d_direct(cl(xin).to(dev), 0, 8, w1.permute(2, 3, 1, 0).reshape(9, 8, 32).contiguous().to(dev),
                      b1.to(dev), mid, 0, 32, 3, 3, True, 1, H8, W8)
    lib.conv2d_direct(mid, 0, 32, w2.permute(2, 3, 1, 0).reshape(9, 32, 16).contiguous().to(dev),
                      b2.to(dev), out, 0, 16, 3, 3, True, 1, H8, W8)
    check(uncl(out.cpu(), 1, H8, W8), want, 2e-5, "conf stem")


def case_layout(lib, dev):
    x = gc.uni("layout/x", (2, 12, H8, W8), -3, 3)
    out = torch.zeros(2 * N, 10, device=dev)
    lib.to_channel_last(x.to(dev), 4, 6, out, 3, 2)      # tanh of channels 4..9 -> cols 3..8
    check(out[:, 3:9], cl(torch.tanh(x[:, 4:10])), 2e-6, "to_channel_last tanh")
    lib.to_channel_last(x.to(dev), 0, 3, out, 0, 1)
    check(out[:, :3], cl(torch.relu(x[:, :3])), 0.0, "to_channel_last relu")
    assert float(out[:, 9].abs().max()) == 0.0
    back = torch.empty(2, 6, H8, W8, device=dev)
    lib.to_nchw(out, 3, 6, back)
    check(back, torch.tanh(x[:, 4:10]), 2e-6, "to_nchw")
    # 2x2 space-to-depth (stem input): column (py*2+px)*C + c of pixel (Y, X) = x[c][2Y+py][2X+px]
    img = gc.uni("layout/img", (2, 3, H8, W8), -1, 1)
    s2d = torch.full((2 * (H8 // 2) * (W8 // 2), 14), 7.0, device=dev)
    lib.space_to_depth2(img.to(dev), s2d)
    want = torch.stack([img[:, :, py::2, px::2] for py in range(2) for px in range(2)], 1)   # [B,4,C,h,w]
    check(s2d[:, :12], cl(want.reshape(2, 12, H8 // 2, W8 // 2)), 0.0, "space_to_depth2")
    assert float((s2d[:, 12:] - 7.0).abs().max()) == 0.0


def case_channel_stats_and_norm_act(lib, dev):
    """InstanceNorm statistics + the residual tail relu(res' + relu(norm(y))) (core/extractor.py:41-47)."""
    B, C, h, w = 2, 96, 12, 20
    Np = h * w
    y = gc.uni("enc/y", (B, C, h, w), -3, 5)
    res = gc.uni("enc/res", (B, C, h, w), -1, 2)
    d = gc.uni("enc/ds", (B, C, h, w), -4, 4)
    sc = torch.empty(B, C, device=dev)
    sh = torch.empty(B, C, device=dev)
    part = torch.zeros(B * 16 * C * 2, dtype=torch.float64, device=dev)
    lib.channel_stats(cl(y).to(dev), B, Np, C, sc, sh, part, 16)
    mean = y.mean(dim=(2, 3))
    var = y.var(dim=(2, 3), unbiased=False)
    check(sc, 1.0 / torch.sqrt(var + 1e-5), 2e-6, "scale = rstd")
    check(sh, -mean / torch.sqrt(var + 1e-5), 2e-6, "shift = -mean*rstd")
    inorm = torch.nn.functional.instance_norm
    out = torch.empty(B * Np, C, device=dev)
    lib.norm_act(cl(y).to(dev), sc, sh, out, B, Np, C)
    check(uncl(out.cpu(), B, h, w), torch.relu(inorm(y)), 3e-6, "relu(norm(y))")
    lib.norm_act(cl(y).to(dev), sc, sh, out, B, Np, C, res=cl(res).to(dev))
    check(uncl(out.cpu(), B, h, w), torch.relu(res + torch.relu(inorm(y))), 3e-6, "identity shortcut")
    sc3 = torch.empty(B, C, device=dev)
    sh3 = torch.empty(B, C, device=dev)
    lib.channel_stats(cl(d).to(dev), B, Np, C, sc3, sh3, part, 16)
    lib.norm_act(cl(y).to(dev), sc, sh, out, B, Np, C, res=cl(d).to(dev), rs=sc3, rt=sh3)
    check(uncl(out.cpu(), B, h, w), torch.relu(inorm(d) + torch.relu(inorm(y))), 5e-6, "normalised shortcut")
    # res_relu: the skip input is relu(norm(raw)) of a raw map that was never materialised (the encoder stem) -- and it is
    # the same bits as materialising it with pf_norm_act first
    lib.norm_act(cl(y).to(dev), sc, sh, out, B, Np, C, res=cl(d).to(dev), rs=sc3, rt=sh3, res_relu=True)
    check(uncl(out.cpu(), B, h, w), torch.relu(torch.relu(inorm(d)) + torch.relu(inorm(y))), 5e-6, "activated normalised shortcut")
    x0 = torch.empty(B * Np, C, device=dev)
    out2 = torch.empty(B * Np, C, device=dev)
    lib.norm_act(cl(d).to(dev), sc3, sh3, x0, B, Np, C)
    lib.norm_act(cl(y).to(dev), sc, sh, out2, B, Np, C, res=x0)
    assert torch.equal(out, out2), "res_relu differs from the materialised skip input"


def case_small_conv_stem(lib, dev):
    """7x7 stride-2 3->64 from NCHW (the encoders' stem, core/extractor.py:112,144)."""
    B, H, W = 2, 32, 128
    x = gc.uni("stem/x", (B, 3, H, W), -1, 1)
    w = gc.uni("stem/w", (64, 3, 7, 7), -0.2, 0.2)
    b = gc.uni("stem/b", (64,), -0.1, 0.1)
    want = torch.nn.functional.conv2d(x, w, b, stride=2, padding=3)
    wp = w.permute(2, 3, 1, 0).reshape(49, 3, 64).contiguous()
    out = torch.full((B * (H // 2) * (W // 2), 66), -5.0, device=dev)
    lib.conv2d_small(x.to(dev), True, 0, 3, wp.to(dev), b.to(dev), out, 1, 64, 7, 7, 2, False, B, H // 2, W // 2)
    check(uncl(out[:, 1:65].cpu(), B, H // 2, W // 2), want, 3e-5, "stem 7x7/2")
    assert float((out[:, 0] + 5).abs().max()) == 0.0 and float((out[:, 65] + 5).abs().max()) == 0.0
    # channel-last input, stride 2, ReLU, odd channel count
    x2 = gc.uni("stem/x2", (1, 5, 16, 64), -1, 1)
    w2 = gc.uni("stem/w2", (32, 5, 3, 3), -0.3, 0.3)
    b2 = gc.uni("stem/b2", (32,), -0.1, 0.1)
    want = torch.relu(torch.nn.functional.conv2d(x2, w2, b2, stride=2, padding=1))
    out = torch.empty(8 * 32, 32, device=dev)
    lib.conv2d_small(cl(x2).to(dev), False, 0, 5, w2.permute(2, 3, 1, 0).reshape(9, 5, 32).contiguous().to(dev),
                     b2.to(dev), out, 0, 32, 3, 3, 2, True, 1, 8, 32)
    check(uncl(out.cpu(), 1, 8, 32), want, 3e-5, "3x3/2 channel-last")


def case_flow_head_out(lib, dev):
    """FlowHead.conv2 (3x3, 256->2) fused with coords1 += delta_flow (core/update.py:13-14, prior_raft.py:193)."""
    x = gc.uni("fho/x", (2, 256, H8, W8), -1, 1)
    w = gc.uni("fho/w", (2, 256, 3, 3), -0.05, 0.05)
    b = gc.uni("fho/b", (2,), -0.1, 0.1)
    co = gc.nasty_coords("fho", B=2)
    want = torch.nn.functional.conv2d(x, w, b, padding=1)
    c1 = co.clone().to(dev)
    delta = torch.full((2 * N, 4), 9.0, device=dev)
    lib.flow_head_out(cl(x).to(dev), 256, w.permute(0, 2, 3, 1).reshape(2, 9, 256).contiguous().to(dev),
                      b.to(dev), c1, delta)
    check(uncl(delta[:, :2].cpu(), 2, H8, W8), want, 2e-5, "delta_flow")
    assert float((delta[:, 2:] - 9.0).abs().max()) == 0.0
    check(c1, co + uncl(delta[:, :2].cpu(), 2, H8, W8), 0.0, "coords1 += delta_flow")
    c2 = co.clone().to(dev)
    lib.flow_head_out(cl(x).to(dev), 256, w.permute(0, 2, 3, 1).reshape(2, 9, 256).contiguous().to(dev),
                      b.to(dev), c2, None)
    check(c2, c1, 0.0, "delta buffer is optional")


def case_split_bf16(lib, dev):
    """fp32 -> bf16 hi|lo rows: hi = bf16(x) (RNE), lo = bf16(x - hi); hi + lo keeps 16 mantissa bits."""
    x = gc.uni("split/x", (37, 96), -50, 50)
    x[0, :4] = torch.tensor([0.0, 1.0, -1.0, 3.0e-20])
    out = torch.zeros(37, 3, 2, 32, dtype=torch.bfloat16, device=dev)
    lib.split_bf16(x.to(dev), out)
    o = out.cpu().float()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert torch.equal(o[:, :, 0].reshape(37, 96), hi.float()), "hi halves"
    assert torch.equal(o[:, :, 1].reshape(37, 96), lo.float()), "lo halves"
    rec = o[:, :, 0].reshape(37, 96) + o[:, :, 1].reshape(37, 96)
    assert float(((rec - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -15


def case_bad_args(lib, dev):
    """Error behaviour: negative PF_ERR codes surface as PfError, nothing is written."""
    from prior_flow_amd._lib import PfError
    co = gc.nasty_coords("prep").to(dev)
    bad = torch.zeros(N, 1, device=dev)
    try:
        lib.flow_prep(co, None, bad, 0)       # 2 channels do not fit in ld=1
    except PfError:
        pass
    else:
        raise AssertionError("expected PfError for a destination slice that does not fit")
    tiny = torch.zeros(1, 2, 8, 8, device=dev)   # level 3 would be 1x1 -> division by zero
    lv = [torch.zeros(64, (8 >> i) * (8 >> i), device=dev) for i in range(4)]
    g = torch.zeros(2, 8, 8, device=dev)
    o = torch.zeros(64, 324, device=dev)
    r = torch.zeros(64, 324, device=dev)
    try:
        lib.dccl_lookup(tiny, lv, lv, g, o, r)
    except PfError:
        pass
    else:
        raise AssertionError("expected PfError for an image below the smallest legal size")


def case_flow_metrics(lib, dev):
    """EPE / SEPE per pixel and the region sums (evaluate.py:246-275) vs the oracle."""
    pre, gt = gc.flows("eval/pre", 2), gc.flows("eval/gt", 2)
    pre[:, 1, 3, :] = -9.0                        # end points clamped at the top edge
    gt[0, :, 4, :] = pre[0, :, 4, :]              # identical flows -> exactly zero distance
    B, _, h, w = pre.shape
    epe = torch.empty(B, h, w, device=dev)
    sd = torch.empty(B, h, w, device=dev)
    lib.flow_metrics(pre.to(dev), gt.to(dev), epe, sd)
    check(epe, po.epe(pre, gt), 1e-6, "epe")
    check(sd, po.great_circle_distance(pre, gt), 2e-6, "sepe")
    assert float(sd[0, 4].abs().max()) == 0.0 and float(epe[0, 4].abs().max()) == 0.0
    # method='Cosine': arccos is ill-conditioned near 0, so compare away from coincident end points (|d| > 0.05 rad: error of
    # the argument 1e-7 / sin(d)) and against the Haversine form, which is the same quantity
    cs = torch.empty(B, h, w, device=dev)
    lib.flow_metrics(pre.to(dev), gt.to(dev), None, cs, cosine=True)
    want_c = po.great_circle_distance_cosine(pre, gt)
    far = want_c > 0.05
    assert int(far.sum()) > far.numel() // 2
    assert float((cs.cpu() - want_c)[far].abs().max()) < 5e-6, "cosine form vs oracle"
    assert float((cs.cpu() - sd.cpu())[far].abs().max()) < 5e-6, "cosine form vs haversine form"
    only = torch.empty(B, h, w, device=dev)
    lib.flow_metrics(pre.to(dev), gt.to(dev), None, only)      # either output is optional
    check(only, sd, 0.0, "sd only")
    # region sums: 3 overlapping regions, weights, 5 pixel chunks (ragged last chunk)
    n = h * w
    bits = torch.zeros(n, dtype=torch.uint8)
    bits[:] = 1
    bits[: n // 3] |= 2
    bits[n // 2:] |= 4
    wts = gc.uni("eval/w", (n,), 0.0, 1.0)
    part = torch.zeros(B, 5, 3, 3, dtype=torch.float64, device=dev)
    lib.region_sums(epe, sd, wts.to(dev), bits.to(dev), 3, part)
    got = part.sum(1).cpu()
    e64, s64 = epe.cpu().double().view(B, n), sd.cpu().double().view(B, n)
    for r, mk in enumerate([(bits & 1) > 0, (bits & 2) > 0, (bits & 4) > 0]):
        want = torch.stack([e64[:, mk].sum(1), s64[:, mk].sum(1), (s64 * wts.double())[:, mk].sum(1)], 1)
        assert float((got[:, r] - want).abs().max()) < 1e-9, ("region_sums", r)


def case_dccl_backward(lib, dev):
    """pf_dccl_combine_bwd + pf_dccl_lookup_bwd vs autograd through the oracle's dccl_lookup."""
    import math
    h, w = H8, W8
    n = h * w
    f1, f2 = gc.fmaps("dbw/a", 1, h, w)
    f3, f4 = gc.fmaps("dbw/b", 1, h, w)
    pyr_a = [p.clone().requires_grad_(True) for p in po.build_pyramid(po.corr_volume(f1, f2))]
    pyr_b = [p.clone().requires_grad_(True) for p in po.build_pyramid(po.corr_volume(f3, f4))]
    coords = gc.nasty_coords("dbw/co", 1, h, w)
    g_w2c = po.sample_grid(h, w, po.rotation_x(math.pi / 2))
    g_back = po.sample_grid(h, w, po.rotation_x(-math.pi / 2))
    own, cross = po.dccl_lookup(coords, pyr_a, pyr_b, g_w2c, g_back)
    G = gc.uni("dbw/G", (1, 324, h, w), -1, 1)
    ((own + cross) * G).sum().backward()
    d_corr = cl(G).to(dev).contiguous()
    d_raw = torch.zeros(n, 324, device=dev)
    lib.dccl_combine_bwd(d_corr, g_back.to(dev).contiguous(), d_raw, 1, h, w)
    g_own = [torch.zeros(n, (h >> i) * (w >> i), device=dev) for i in range(4)]
    g_oth = [torch.zeros(n, (h >> i) * (w >> i), device=dev) for i in range(4)]
    lib.dccl_lookup_bwd(coords.to(dev), g_w2c.to(dev).contiguous(), d_corr, d_raw, g_own, g_oth)
    for i in range(4):
        check(g_own[i], pyr_a[i].grad.reshape(n, -1), 2e-5, f"own pyramid gradient level {i}")
        check(g_oth[i], pyr_b[i].grad.reshape(n, -1), 5e-5, f"other pyramid gradient level {i}")
    # pyramid -> dense volume gradient -> feature gradients (odd map: floor pooling), vs autograd
    for hh, ww in ((16, 32), (17, 27)):
        nn_ = hh * ww
        a1, a2 = gc.fmaps(f"dbw/p{hh}", 1, hh, ww)
        a1 = a1[:, :32].clone().requires_grad_(True)
        a2 = a2[:, :32].clone().requires_grad_(True)
        pyr = po.build_pyramid(po.corr_volume(a1, a2))
        gl = [gc.uni(f"dbw/g{hh}/{i}", tuple(p.shape), -1, 1) for i, p in enumerate(pyr)]
        sum((p * g).sum() for p, g in zip(pyr, gl)).backward()
        dv = [g.reshape(nn_, -1).contiguous().to(dev) for g in gl]
        lib.pyramid_bwd(dv, 1, hh, ww)
        r1, r2 = cl(a1.detach()).to(dev), cl(a2.detach()).to(dev)          # [N, C]
        scale = 1.0 / math.sqrt(32.0)
        check(cl(a1.grad), (dv[0] @ r2) * scale, 2e-4, f"f1 gradient {hh}x{ww}")
        check(cl(a2.grad), (dv[0].t() @ r1) * scale, 2e-4, f"f2 gradient {hh}x{ww}")


def case_training_pieces(lib, dev):
    """pf_seq_loss / pf_sum_squares / pf_adamw_step vs the oracle's restatement of train_flow.py."""
    from gen_golden_train import adam_case, loss_case
    preds, gt, valid = loss_case(16, 32, 2, 2)
    B, _, h, w = gt.shape
    uni = po.spherical_mask(h, w).contiguous()
    want_loss, want_metrics, want_grads = po.uniform_loss(preds, gt, valid, gamma=0.8)
    total = 0.0
    for i, p in enumerate(preds):
        part = torch.zeros(B, 3, 6, dtype=torch.float64, device=dev)
        g = torch.empty(B, 2, h, w, device=dev)
        wgt = 0.8 ** (len(preds) - i - 1)
        lib.seq_loss(p.to(dev), gt.to(dev), valid.to(dev), uni.view(-1).to(dev), wgt, 400.0, g, part)
        check(g, want_grads[i], 1e-9, f"loss gradient {i}")
        tot = part.sum((0, 1)).cpu()
        total += wgt * float(tot[0])
    assert abs(total - want_loss) < 1e-9 * abs(want_loss)
    assert abs(float(tot[1] / tot[2]) - want_metrics["epe"]) < 1e-7
    for j, key in ((3, "1px"), (4, "3px"), (5, "5px")):
        assert abs(float(tot[j] / tot[2]) - want_metrics[key]) < 1e-12
    p0, grads = adam_case(1031, 3)
    part = torch.zeros(7, dtype=torch.float64, device=dev)
    lib.sum_squares(grads[0].to(dev), part)
    assert abs(float(part.sum()) - float((grads[0].double() ** 2).sum())) < 1e-9 * float((grads[0].double() ** 2).sum())
    p, m, v = p0.clone().to(dev), torch.zeros(1031, device=dev), torch.zeros(1031, device=dev)
    wp, wm, wv = p0.clone(), torch.zeros(1031), torch.zeros(1031)
    for k, gk in enumerate(grads):
        lr = po.one_cycle_lr(k, 1e-4, 60000)
        c = po.clip_coef(float((gk.double() ** 2).sum().sqrt()), 1.0)
        lib.adamw_step(p, gk.to(dev), m, v, lr, 0.9, 0.999, 1e-8, 5e-5, k + 1, c)
        wp, wm, wv = po.adamw_step(wp, gk * np.float32(c), wm, wv, lr, k + 1, 5e-5)
    check(p, wp, 2e-8, "adamw params")
    check(m, wm, 1e-7, "adamw exp_avg")
    check(v, wv, 1e-7, "adamw exp_avg_sq")


def case_gru_gate_backward(lib, dev):
    """pf_gru_q_bwd / pf_gru_zr_bwd against torch autograd of the SepConvGRU gate math (core/update.py:46-60);
    the convolutions are replaced by free tensors, so only the gate arithmetic is differentiated."""
    rows, Cc = 200, 128
    g = lambda name, lo=-2.0, hi=2.0: gc.uni(f"grubwd/{name}", (rows, Cc), lo, hi)    # noqa: E731
    az, ar, aq_lin, h, dhn = g("az"), g("ar"), g("aq"), g("h", -1, 1), g("dhn", -1, 1)
    az, ar, aq_lin, h = (t.clone().requires_grad_(True) for t in (az, ar, aq_lin, h))
    z, r = torch.sigmoid(az), torch.sigmoid(ar)
    rh = r * h
    rh.retain_grad()
    q = torch.tanh(aq_lin + 0.5 * rh)                      # stand-in for convq(cat[r*h, x]): d_rh = 0.5 * dq_pre
    hn = (1 - z) * h + z * q
    hn.backward(dhn)
    # HIP / emu: stage Q, the "data gradient of convq" (here 0.5 * dq_pre), stage ZR
    d = lambda t: t.detach().to(dev).contiguous()          # noqa: E731
    zz, rr, qq, hh = d(z), d(r), d(q), d(h)
    wide = torch.zeros(rows, 3 * Cc, device=dev)           # column slices exercise the leading dimensions
    dq_pre, dz, dh = wide[:, :Cc], wide[:, Cc:2 * Cc], wide[:, 2 * Cc:]
    lib.gru_q_bwd(d(dhn), zz, qq, hh, dq_pre, dz, dh)
    check(dq_pre.cpu(), aq_lin.grad, 2e-6, "gru dq_pre")
    d_rh = (0.5 * dq_pre).contiguous()
    check(d_rh.cpu(), rh.grad, 2e-6, "gru d_rh (stand-in conv)")
    dzr = torch.zeros(rows, 2 * Cc + 8, device=dev)
    lib.gru_zr_bwd(dz, d_rh, zz, rr, hh, dzr, dh)
    check(dzr[:, :Cc].cpu(), az.grad, 2e-6, "gru dz_pre")
    check(dzr[:, Cc:2 * Cc].cpu(), ar.grad, 2e-6, "gru dr_pre")
    check(dh.cpu(), h.grad, 2e-6, "gru dh")
    assert float(dzr[:, 2 * Cc:].abs().max()) == 0.0


def case_norm_backward(lib, dev):
    """pf_norm_bwd against autograd of relu(instance_norm(x)) (fnet, core/extractor.py:112-113) and of the folded
    BatchNorm(eval) affine + relu (cnet)."""
    B, Cc, H, W = 2, 64, 12, 20
    Np = H * W
    x = gc.uni("normbwd/x", (B, Cc, H, W), -2, 2).requires_grad_(True)
    dy = gc.uni("normbwd/dy", (B, Cc, H, W), -1, 1)
    torch.relu(torch.nn.functional.instance_norm(x, eps=1e-5)).backward(dy)
    xr = cl(x.detach()).to(dev).contiguous()
    scale = torch.empty(B, Cc, device=dev); shift = torch.empty(B, Cc, device=dev)
    part = torch.empty(B * 4 * Cc * 2, dtype=torch.float64, device=dev)
    lib.channel_stats(xr, B, Np, Cc, scale, shift, part, 4)
    dx = torch.empty_like(xr)
    lib.norm_bwd(cl(dy).to(dev).contiguous(), xr, scale, shift, True, True, dx, B, Np, Cc, nblk=7)
    check(uncl(dx.cpu(), B, H, W), x.grad, 2e-5, "instance norm + relu backward")
    # fixed statistics
    s = gc.uni("normbwd/s", (Cc,), 0.5, 1.5); t = gc.uni("normbwd/t", (Cc,), -0.5, 0.5)
    x2 = x.detach().clone().requires_grad_(True)
    torch.relu(x2 * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)).backward(dy)
    sc = s.view(1, -1).expand(B, Cc).contiguous().to(dev); sh = t.view(1, -1).expand(B, Cc).contiguous().to(dev)
    lib.norm_bwd(cl(dy).to(dev).contiguous(), xr, sc, sh, True, False, dx, B, Np, Cc)
    check(uncl(dx.cpu(), B, H, W), x2.grad, 1e-6, "affine + relu backward")


def case_pack_conv_weights(lib, dev):
    """pf_pack_conv_weights against the torch packing it replaces (engine.pack_mfma + split_bf16, Conv.dgrad_of): forward and
    data-gradient operands, one tensor and two concatenated on Cout (fused z|r), odd channel counts, a rotated channel order."""
    from prior_flow_amd.engine import Conv, pack_mfma, split_bf16
    g = torch.Generator().manual_seed(5)
    for cout0, cout1, cin, kh, kw in ((124, 0, 272, 3, 3), (128, 128, 384, 1, 5), (2, 0, 256, 3, 3), (576, 0, 256, 1, 1), (32, 0, 8, 3, 3)):
        w0 = (torch.rand(cout0, cin, kh, kw, generator=g) * 2 - 1).to(dev)
        b0 = (torch.rand(cout0, generator=g) * 2 - 1).to(dev)
        w1 = (torch.rand(cout1, cin, kh, kw, generator=g) * 2 - 1).to(dev) if cout1 else None
        b1 = (torch.rand(cout1, generator=g) * 2 - 1).to(dev) if cout1 else None
        w = w0 if w1 is None else torch.cat([w0, w1], 0)
        b = b0 if b1 is None else torch.cat([b0, b1], 0)
        wp, bp = pack_mfma(w, b)
        got_w, got_b = lib.pack_conv_weights(w0, b0, w1, b1, mode=0)
        assert torch.equal(got_w.cpu().view(torch.int16), split_bf16(wp).cpu().view(torch.int16)), ("fwd", cout0, cout1, cin)
        assert torch.equal(got_b.cpu(), bp.cpu())
        for rot in (0, 128 if cin > 128 else 0):
            cp = (cout0 + cout1 + 3) // 4 * 4
            wpad = torch.zeros(cp, cin, kh, kw, device=dev)
            wpad[:cout0 + cout1] = torch.cat([w[:, rot:], w[:, :rot]], 1)
            ref = Conv.dgrad_of(wpad, 1)            # PREC_BF16X3
            got_w, got_b = lib.pack_conv_weights(w0, None, w1, None, mode=1, cin_rot=rot)
            assert tuple(got_w.shape) == tuple(ref.w.shape), (got_w.shape, ref.w.shape)
            assert torch.equal(got_w.cpu().view(torch.int16), ref.w.cpu().view(torch.int16)), ("dgrad", cout0, cout1, cin, rot)
            assert float(got_b.abs().max()) == 0.0


def case_unpack_wgrads(lib, dev):
    """pf_unpack_wgrads (training: packed dW / db of many convolutions -> the parameters' .grad tensors, 16 jobs per launch) against
    the torch expression it replaces (slice, permute, scale, +=): 19 jobs so the second launch is exercised, with a fused two-module
    job (o_off) and a bias-less one.  One multiply-add per element on both sides: 1 ulp of the result allowed (fma or not)."""
    from prior_flow_amd._lib import PfError
    gen = torch.Generator().manual_seed(3)
    jobs, want = [], []
    shapes = [(64, 3, 7, 7), (128, 384, 1, 5), (2, 256, 3, 3), (576, 256, 1, 1), (128, 272, 3, 3)] * 4
    for i, (cout, cin, kh, kw) in enumerate(shapes[:19]):
        o_off = 128 if i % 5 == 1 else 0
        op = (o_off + cout + 127) // 128 * 128
        cin_pad = (cin + 31) // 32 * 32
        dw = torch.randn(op, kh * kw, cin_pad, generator=gen).to(dev)
        db = torch.randn(op, generator=gen).to(dev)
        gw = torch.randn(cout, cin, kh, kw, generator=gen).to(dev)
        gb = None if i == 3 else torch.randn(cout, generator=gen).to(dev)
        scale = 0.25 if i % 4 == 0 else 1.0
        ref_w = gw + scale * dw[o_off:o_off + cout, :, :cin].reshape(cout, kh, kw, cin).permute(0, 3, 1, 2)
        ref_b = None if gb is None else gb + scale * db[o_off:o_off + cout]
        jobs.append((dw, db if gb is not None else None, gw, gb, cout, cin, kh * kw, cin_pad, o_off, scale))
        want.append((ref_w, ref_b))
    lib.unpack_wgrads(jobs)
    for (dw, db, gw, gb, *_), (ref_w, ref_b) in zip(jobs, want):
        assert torch.allclose(gw, ref_w, rtol=2e-7, atol=1e-7)
        if gb is not None:
            assert torch.allclose(gb, ref_b, rtol=2e-7, atol=1e-7)
    try:
        lib.unpack_wgrads([(jobs[0][0], jobs[0][1], jobs[1][2], jobs[0][3], 64, 3, 49, 32, 0, 1.0)])      # gw of another shape
    except PfError:
        pass
    else:
        raise AssertionError("a gradient tensor of the wrong shape was accepted")


def case_frozen_batchnorm(lib, dev):
    """pf_bn_frozen_fwd / pf_bn_frozen_bwd (the context encoder's BatchNorm with frozen statistics [+ ReLU], core/extractor.py:
    114-115, train_flow.py:107-108) against torch's F.batch_norm(training=False) [+ relu] and its autograd: output, dx, d gamma,
    d beta to 1e-5 of their scale; the three encoder widths, an odd row count; accumulate on and off."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(17)
    for relu in (True, False):
        for Cc, (Bn, Hh, Ww) in ((64, (2, 9, 21)), (96, (1, 12, 16)), (128, (3, 6, 8))):
            x = torch.randn(Bn, Cc, Hh, Ww, generator=gen).to(dev).requires_grad_()
            gamma = (torch.rand(Cc, generator=gen) + 0.5).to(dev).requires_grad_()
            beta = (torch.randn(Cc, generator=gen) * 0.3).to(dev).requires_grad_()
            mean = (torch.randn(Cc, generator=gen) * 0.2).to(dev)
            var = (torch.rand(Cc, generator=gen) + 0.3).to(dev)
            g = torch.randn(Bn, Cc, Hh, Ww, generator=gen).to(dev)
            ref = F.batch_norm(x, mean, var, gamma, beta, False, 0.1, 1e-5)
            ref = torch.relu(ref) if relu else ref
            rdx, rdg, rdb = torch.autograd.grad(ref, (x, gamma, beta), g)
            rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, Cc).contiguous()      # noqa: E731
            xr, gr = rows(x), rows(g)
            out = lib.bn_frozen_fwd(xr, gamma.detach(), beta.detach(), mean, var, 1e-5, relu, torch.empty_like(xr))
            dx, dg, db = torch.empty_like(xr), torch.full((Cc,), 2.0, device=dev), torch.full((Cc,), -1.0, device=dev)
            lib.bn_frozen_bwd(gr, xr, gamma.detach(), beta.detach(), mean, var, 1e-5, relu, dx, dg, db, True)
            for a, b in ((out, rows(ref)), (dx, rows(rdx)), (dg - 2.0, rdg), (db + 1.0, rdb)):
                assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), (Cc, relu)
            lib.bn_frozen_bwd(gr, xr, gamma.detach(), beta.detach(), mean, var, 1e-5, relu, dx, dg, db, False)
            assert float((dg - rdg).abs().max()) <= 1e-5 * max(1.0, float(rdg.abs().max()))


def case_add_relu(lib, dev):
    """pf_add_relu / pf_relu_mask (the ResidualBlock tail relu(x + y) of the training tape, core/extractor.py:47, and the backward
    of a ReLU from its output): exact against torch, a length that is not a multiple of 4."""
    gen = torch.Generator().manual_seed(9)
    for n in (4096, 1027):
        x, y, g = (torch.randn(n, generator=gen).to(dev) for _ in range(3))
        out = lib.add_relu(x, y, torch.empty_like(x))
        assert torch.equal(out, torch.relu(x + y))
        dx = lib.relu_mask(g, out, torch.empty_like(g))
        assert torch.equal(dx, torch.where(out > 0, g, torch.zeros_like(g)))


ELEMENTWISE_CASES = [case_sample_grid, case_img_rotate, case_normalise_images, case_prepare_images, case_flow_prep, case_flo_rotate, case_dccl,
                     case_warp_gcorr, case_motion_prep, case_conf_stem, case_upsample, case_coords_add, case_layout,
                     case_channel_stats_and_norm_act, case_small_conv_stem, case_flow_head_out, case_split_bf16, case_pack_conv_weights,
                     case_flow_metrics, case_training_pieces, case_dccl_backward, case_upsample_backward,
                     case_warp_gcorr_backward, case_gru_gate_backward, case_norm_backward, case_unpack_wgrads,
                     case_frozen_batchnorm, case_add_relu, case_bad_args]
