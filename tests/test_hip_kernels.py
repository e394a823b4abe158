"""GPU parity tests proper: every C-ABI entry point of libpriorflow_hip.so against the CPU
oracle and the reference-generated golden vectors.  Run with ``-m gpu`` on an MI355X."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import kernel_cases as kc
import priorflow_oracle as po

pytestmark = pytest.mark.gpu
T = torch.from_numpy
H8, W8, N = gc.H8, gc.W8, gc.H8 * gc.W8


@pytest.fixture(scope="module")
def lib():
    from prior_flow_amd import _lib
    return _lib.load()          # raises if the HIP library was not built: no fallback


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def params():
    from prior_flow_amd.modules import state_dict_shapes
    return gc.det_state_dict(state_dict_shapes())


def test_native_library_is_loaded(lib):
    assert "gfx950" in lib.version()
    loaded = open("/proc/self/maps").read()
    assert "libpriorflow_hip.so" in loaded


@pytest.mark.parametrize("case", kc.ELEMENTWISE_CASES, ids=lambda c: c.__name__)
def test_elementwise_case(lib, dev, case):
    case(lib, dev)
    torch.cuda.synchronize()


def test_direct_conv(lib, dev, params):
    kc.case_direct_conv(lib, dev, params)


# ---- MFMA implicit-GEMM convolution --------------------------------------------------------
def _conv_ref(x, w, b, pad):
    return torch.nn.functional.conv2d(x, w, b, padding=pad)


PRECISIONS = ["fp32", "bf16x3"]
# exact-fp32 MFMA reproduces CPU conv2d to accumulation-order noise; the 3-pass bf16 split keeps
# 16 mantissa bits per operand (2^-17 relative per product)
TOL = {"fp32": 2e-5, "bf16x3": 1.5e-4}


def _prec(name):
    from prior_flow_amd._lib import PREC_BF16X3, PREC_F32
    return {"fp32": PREC_F32, "bf16x3": PREC_BF16X3}[name]


def _run_conv(lib, dev, x_nchw, w, b, epilogue, scale=1.0, off_out=0, extra_cols=0, prec="fp32"):
    from prior_flow_amd.engine import Conv, pack_mfma
    Bn, cin, h, wd = x_nchw.shape
    cout = w.shape[0]
    wp, bp = pack_mfma(w.to(dev), b.to(dev))
    cv = Conv(wp, bp, w.shape[2], w.shape[3], cin, cout, _prec(prec))
    xin = kc.cl(x_nchw).to(dev)
    out = torch.full((Bn * h * wd, cout + off_out + extra_cols), -777.0, device=dev)
    d = cv.desc(xin, 0, cin, out, off_out, epilogue, scale=scale)
    lib.conv2d([d], Bn, h, wd, out)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 16, 32), (1, 24, 40), (1, 8, 8)],
                         ids=lambda s: "B%dx%dx%d" % s)
@pytest.mark.parametrize("conv", [
    ("1x1_324_256", 324, 256, 1, 1), ("3x3_272_124", 272, 124, 3, 3), ("3x3_128_64", 128, 64, 3, 3),
    ("3x3_256_2", 256, 2, 3, 3), ("1x5_384_128", 384, 128, 1, 5), ("5x1_384_256", 384, 256, 5, 1),
    ("1x1_256_576", 256, 576, 1, 1), ("3x3_256_192", 256, 192, 3, 3)], ids=lambda c: c[0])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_conv_mfma_vs_cpu_conv2d(lib, dev, shape, conv, prec):
    from prior_flow_amd._lib import EPI_LINEAR, EPI_RELU
    B, h, w_ = shape
    name, cin, cout, kh, kw = conv
    x = gc.uni(f"conv/{name}/x{B}{h}{w_}", (B, cin, h, w_), -1, 1)
    bound = (3.0 / (cin * kh * kw)) ** 0.5
    w = gc.uni(f"conv/{name}/w", (cout, cin, kh, kw), -bound, bound)
    b = gc.uni(f"conv/{name}/b", (cout,), -0.1, 0.1)
    want = _conv_ref(x, w, b, (kh // 2, kw // 2))
    out = _run_conv(lib, dev, x, w, b, EPI_RELU, off_out=4, extra_cols=3, prec=prec)
    got = kc.uncl(out[:, 4:4 + cout].cpu(), B, h, w_)
    kc.check(got, torch.relu(want), TOL[prec], f"{name} relu")
    assert float((out[:, :4] + 777.0).abs().max()) == 0.0, "columns left of the slice were touched"
    assert float((out[:, 4 + cout:] + 777.0).abs().max()) == 0.0, "columns right of the slice were touched"
    out = _run_conv(lib, dev, x, w, b, EPI_LINEAR, scale=0.25, prec=prec)
    kc.check(kc.uncl(out.cpu(), B, h, w_), 0.25 * want, TOL[prec], f"{name} linear*0.25")


@pytest.mark.parametrize("prec", PRECISIONS)
def test_conv_mfma_identity_asymmetric(lib, dev, prec):
    """A = delta kernel with an ASYMMETRIC weight map: catches row/col swaps in the MFMA
    operand / accumulator layout (cdna_hip_programming.md §3)."""
    from prior_flow_amd._lib import EPI_LINEAR
    cin, cout = 64, 96
    x = gc.uni("conv/id/x", (1, cin, 16, 32), -1, 1)
    w = torch.zeros(cout, cin, 3, 3)
    for o in range(cout):
        w[o, (o * 7 + 3) % cin, 1, 1] = 1.0 + o      # out[o] = (1+o) * x[(7o+3) % 64]
    b = torch.arange(cout, dtype=torch.float32) * 0.5
    out = _run_conv(lib, dev, x, w, b, EPI_LINEAR, prec=prec)
    want = torch.stack([(1.0 + o) * x[0, (o * 7 + 3) % cin] + 0.5 * o for o in range(cout)])[None]
    # bf16x3: integer weights <= 96 are exact in bf16; x keeps 16 bits -> 2^-17 * 97
    kc.check(kc.uncl(out.cpu(), 1, 16, 32), want, 1e-5 if prec == "fp32" else 1e-3, "identity/asymmetric")


@pytest.mark.parametrize("prec", PRECISIONS)
def test_conv_mfma_groups_and_gru(lib, dev, params, prec):
    """Grouped launch (branch A | branch B) with two input segments and the fused SepConvGRU
    epilogues, against the oracle's GRU (core/update.py:46-60) and the reference golden."""
    from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR
    from prior_flow_amd.engine import Conv
    import argparse
    from prior_flow_amd.prior_raft import PriOr_RAFT
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(params)
    model = model.to(dev)
    ui = gc.update_inputs("upd")
    mf = T(gc.load("update_A")["motion"])
    x = torch.cat([ui["inp"], mf], 1)
    h0 = ui["net"]
    net = [kc.cl(h0).to(dev), torch.empty(N, 128, device=dev)]
    net_b = [kc.cl(h0).to(dev), torch.empty(N, 128, device=dev)]
    xr = kc.cl(x).to(dev)
    z = [torch.empty(N, 128, device=dev) for _ in range(2)]
    rh = [torch.empty(N, 128, device=dev) for _ in range(2)]
    c = 0
    for tag in ("1", "2"):
        descs = []
        for gi, (blk, nn_) in enumerate(((model.ODDC, net), (model.update_block, net_b))):
            cv = Conv.fused(getattr(blk.gru, "convz" + tag), getattr(blk.gru, "convr" + tag), _prec(prec))
            descs.append(cv.desc(nn_[c], 0, 128, z[gi], 0, EPI_GRU_ZR, in1=xr, off1=0, c1=256, h=nn_[c], aux=rh[gi]))
        lib.conv2d(descs, 1, H8, W8, xr)
        descs = []
        for gi, (blk, nn_) in enumerate(((model.ODDC, net), (model.update_block, net_b))):
            cv = Conv.of(getattr(blk.gru, "convq" + tag), _prec(prec))
            descs.append(cv.desc(rh[gi], 0, 128, nn_[c ^ 1], 0, EPI_GRU_Q, in1=xr, off1=0, c1=256, h=nn_[c], z=z[gi]))
        lib.conv2d(descs, 1, H8, W8, xr)
        c ^= 1
    torch.cuda.synchronize()
    kc.check(kc.uncl(net[c].cpu(), 1, H8, W8), gc.load("gru")["out"], TOL[prec], "ODDC.gru vs reference")
    want_b = po.sepconv_gru(params, "update_block.gru.", h0, x)
    kc.check(kc.uncl(net_b[c].cpu(), 1, H8, W8), want_b, TOL[prec], "update_block.gru vs oracle")


def test_motion_prep_equals_the_five_launches_it_replaces(lib, dev):
    """pf_motion_prep (flows of both branches, flo_rotate(flow_B), both feature warps + groupwise correlations;
    core/prior_raft.py:171-182) against pf_flow_prep x2 + pf_flo_rotate + pf_warp_gcorr x2 on nasty coordinates
    (seam crossers, out-of-range y, multi-wrap x): every output bit for bit, untouched columns untouched."""
    import math
    from prior_flow_amd.engine import rotation_x
    B, h, w = 2, H8, W8
    n = h * w
    c1a, c1b = gc.nasty_coords("mprep/a", B).to(dev), gc.nasty_coords("mprep/b", B).to(dev)
    f1, f2 = (kc.cl(t).to(dev) for t in gc.fmaps("mprep", B))
    g_a2b, g_b2a = torch.empty(2, h, w, device=dev), torch.empty(2, h, w, device=dev)
    lib.sample_grid(g_a2b, rotation_x(-math.pi / 2))
    lib.sample_grid(g_b2a, rotation_x(math.pi / 2))

    def bufs():
        z = lambda *s: torch.full(s, 7.25, device=dev)        # sentinel: columns no kernel owns must keep it
        return dict(flow4=z(B * n, 4), flow2=z(B * n, 2), xa=z(B * n, 256), xb=z(B * n, 256), conf=z(B * n, 8))
    want, got = bufs(), bufs()
    flow_b, flow_ba = torch.empty(B, 2, h, w, device=dev), torch.empty(B, 2, h, w, device=dev)
    lib.flow_prep(c1a, None, want["flow4"], 0, want["xa"], 252)
    lib.flow_prep(c1b, flow_b, want["flow2"], 0, want["xb"], 254)
    lib.flo_rotate(flow_b, g_a2b, g_b2a, flow_ba, want["flow4"], 2, want["xa"], 254)
    lib.warp_gcorr(f1, f2, c1a, False, want["conf"], 0)
    lib.warp_gcorr(f1, f2, flow_ba, True, want["conf"], 4)
    lib.motion_prep(c1a, c1b, g_a2b, g_b2a, f1, f2, got["flow4"], got["flow2"], got["conf"], got["xa"], 252, got["xb"], 254)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(want[k], got[k]), k
    assert float(got["xa"][:, :252].min()) == 7.25 and float(got["xb"][:, :254].max()) == 7.25
    assert float(got["flow4"].abs().max()) > 1.0 and float(got["conf"].abs().max()) > 1e-3


def test_combine_conv1x1_equals_combine_then_conv(lib, dev, params):
    """pf_dccl_combine_conv1x1 (rotate-back + add + convc1 + ReLU in one launch, the 324-channel tensor never written)
    against pf_dccl_combine followed by pf_conv2d: bit for bit, both branches in one launch, on a map whose pixel count
    is not a multiple of the 64-pixel tile, with sentinels in the columns it does not own."""
    import argparse
    import math
    from prior_flow_amd._lib import EPI_RELU, PREC_BF16X3
    from prior_flow_amd.engine import Conv, rotation_x
    from prior_flow_amd.prior_raft import PriOr_RAFT
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(params)
    model = model.to(dev)
    convs = [Conv.of(model.ODDC.encoder.convc1_A, PREC_BF16X3), Conv.of(model.update_block.encoder.convc1, PREC_BF16X3)]
    B, h, w = 2, 17, 27                       # 918 rows: 14 full tiles + 22 pixels
    n = h * w
    grids = [torch.empty(2, h, w, device=dev) for _ in range(2)]
    lib.sample_grid(grids[0], rotation_x(math.pi / 2))
    lib.sample_grid(grids[1], rotation_x(-math.pi / 2))
    own = [gc.uni(f"cc/own{i}", (B * n, 324), -3, 3).to(dev) for i in range(2)]
    raw = [gc.uni(f"cc/raw{i}", (B * n, 324), -3, 3).to(dev) for i in range(2)]
    want = [torch.full((B * n, 264), 7.25, device=dev) for _ in range(2)]
    got = [torch.full((B * n, 264), 7.25, device=dev) for _ in range(2)]
    corr = [torch.empty(B * n, 324, device=dev) for _ in range(2)]
    for i in range(2):
        lib.dccl_combine(own[i], raw[i], grids[i], corr[i], B, h, w)
    lib.conv2d([convs[i].desc(corr[i], 0, 324, want[i], 4, EPI_RELU) for i in range(2)], B, h, w, corr[0])
    lib.dccl_combine_conv1x1([(own[i], raw[i], grids[i], convs[i], got[i], 4) for i in range(2)], B, h, w)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(want[i], got[i]), f"branch {i}: max diff {float((want[i] - got[i]).abs().max()):.3e}"
        assert float(got[i][:, :4].min()) == 7.25 and float(got[i][:, 260:].max()) == 7.25
        assert float((got[i][:, 4:260] > 0).float().mean()) > 0.2
    single = torch.full((B * n, 264), 7.25, device=dev)
    lib.dccl_combine_conv1x1([(own[1], raw[1], grids[1], convs[1], single, 4)], B, h, w)       # one group
    assert torch.equal(single, want[1])


# ---- corr volume + pyramid -------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 16, 64), (1, 24, 40), (1, 16, 24)],
                         ids=lambda s: "B%dx%dx%d" % s)
@pytest.mark.parametrize("prec", PRECISIONS)
def test_corr_pyramid(lib, dev, shape, prec):
    B, h, w = shape
    f1 = gc.uni(f"corrk/f1/{h}x{w}", (B, 256, h, w), -1.7, 1.7)
    f2 = gc.uni(f"corrk/f2/{h}x{w}", (B, 256, h, w), -1.7, 1.7)
    n = h * w
    lv = [torch.full((B * n, (h >> i) * (w >> i)), float("nan"), device=dev) for i in range(4)]
    if prec == "fp32":
        lib.corr_pyramid(kc.cl(f1).to(dev), kc.cl(f2).to(dev), lv, B, h, w)
    else:
        s1 = torch.empty(B * n, 8, 2, 32, dtype=torch.bfloat16, device=dev)
        s2 = torch.empty_like(s1)
        lib.split_bf16(kc.cl(f1).to(dev), s1)
        lib.split_bf16(kc.cl(f2).to(dev), s2)
        lib.corr_pyramid_bf16x3(s1, s2, lv, B, h, w, 256)
    torch.cuda.synchronize()
    pyr = po.build_pyramid(po.corr_volume(f1, f2))
    for i in range(4):
        # bf16x3: 256 products of magnitude <= 2.9, 2^-17 relative each, /16
        kc.check(lv[i].cpu(), pyr[i].reshape(B * n, -1), 3e-5 if prec == "fp32" else 2e-4, f"level {i}")


@pytest.mark.parametrize("shape", [(1, 64, 128), (2, 24, 128), (1, 48, 64), (3, 16, 64), (1, 32, 256), (2, 40, 64)],
                         ids=lambda s: "B%dx%dx%d" % s)
def test_corr_role_split_kernel_matches_tile_kernel(lib, dev, shape, monkeypatch):
    """The role-split kernel (round 5: a transposed GEMM -- query fragments in registers, target tiles of 2 map rows x 64 columns
    streamed by LDS-DMA -- on four MFMA waves, scaling / pooling / every global store on four store waves fed through an LDS
    staging image; items of 16 or 8 map rows x 128 -- or 64 when W/8 is not a multiple of 128 -- columns) against the tile kernel
    (PRIORFLOW_CORR_RS=0; maps whose width is not a multiple of 64, e.g. 640x1280, stay on it).  Round 6 put
    v_mfma_f32_16x16x32_bf16 on the MFMA waves: the three split passes of a 32-channel chunk add in another order than the tile
    kernel's 32x32x16 pairs, so level 0 agrees to fp32 rounding (the oracle pins the values: test_corr_pyramid), while
      * every pooled level is avg_pool2d of the level below IN THE KERNEL'S OWN BITS (tl + tr + bl + br, then * 0.25),
      * repeated launches are bit-identical (a stale ring tile -- a missed DMA wait -- would show up as a rare 128 x 32 patch),
      * nothing outside the four levels is written (one guard row of NaNs behind every level)."""
    B, h, w = shape
    n = h * w
    gen = torch.Generator().manual_seed(h * w + B)
    f = [(torch.rand(B * n, 256, generator=gen) * 3.4 - 1.7).to(dev) for _ in range(2)]
    sp = [lib.split_bf16(x, torch.empty(B * n, 8, 2, 32, dtype=torch.bfloat16, device=dev)) for x in f]
    out = {}
    for rep, mode in enumerate(("0",) + ("1",) * 5):
        monkeypatch.setenv("PRIORFLOW_CORR_RS", mode)
        lv = [torch.full((B * n + 1, (h >> i) * (w >> i)), float("nan"), device=dev) for i in range(4)]
        lib.corr_pyramid_bf16x3(sp[0], sp[1], [x[:B * n] for x in lv], B, h, w, 256)
        torch.cuda.synchronize()
        if rep >= 2:
            for i in range(4):
                assert bool(torch.isnan(lv[i][B * n]).all()) and torch.equal(out["1"][i][:B * n], lv[i][:B * n]), \
                    f"level {i}: run {rep} of the role-split kernel differs from its first run"
        else:
            out[mode] = lv
    rs, tile = out["1"], out["0"]
    for i in range(4):
        assert torch.isnan(rs[i][B * n]).all(), f"level {i}: wrote past the end"
        assert torch.isfinite(rs[i][:B * n]).all(), f"level {i}: unwritten elements"
        d = (tile[i][:B * n] - rs[i][:B * n]).abs()
        assert float(d.max()) <= 4e-6, (f"level {i}: max |role-split - tile| {float(d.max()):.3e} (values up to {float(tile[i][:B * n].abs().max()):.2f}), "
                                        f"first at (row, col) {(d > 4e-6).nonzero()[0].tolist()}")
    for i in range(3):          # level i + 1 = 2x2 mean of level i, avg_pool2d's order of operations, on the kernel's own level i
        hi, wi = h >> i, w >> i
        v = rs[i][:B * n].view(B * n, hi // 2, 2, wi // 2, 2)
        q = v[:, :, 0, :, 0] + v[:, :, 0, :, 1]
        q = q + v[:, :, 1, :, 0]
        q = (q + v[:, :, 1, :, 1]) * 0.25
        assert torch.equal(q.reshape(B * n, -1), rs[i + 1][:B * n]), f"level {i + 1} is not the 2x2 mean of the kernel's level {i}"


@pytest.mark.parametrize("shape", [(8, 64, 128), (3, 150, 146), (16, 66, 70)], ids=lambda s: "B%dx%dx%d" % s)
def test_flow_head_tile_form_matches_strip_form_bitwise(lib, dev, shape):
    """Round 6: from 65 536 pixels per launch on (8 pairs of 512x1024), pf_flow_head_out runs one wave per 4 x 4 tile instead of per
    strip of 4 pixels (2.25 instead of 4.5 KB of input rows per pixel: at batch 32 the map streams from HBM).  Same products, same
    order of additions inside a lane, same cross-lane tree: the batched launch must equal, bit for bit, the per-image launches (which
    are below the threshold and take the strip form) -- delta_flow and the updated coords1, on maps whose sides are and are not
    multiples of 4 -- and torch's conv2d to rounding."""
    B, h, w = shape
    assert B * h * w >= 65536 and h * w < 65536
    gen = torch.Generator().manual_seed(B * h + w)
    x = (torch.rand(B * h * w, 256, generator=gen) * 2 - 1).to(dev)
    wt = ((torch.rand(2, 9, 256, generator=gen) * 2 - 1) * 0.05).to(dev)
    bias = torch.tensor([0.03, -0.07], device=dev)
    c0 = (torch.rand(B, 2, h, w, generator=gen) * 50).to(dev)
    c_all, d_all = c0.clone(), torch.full((B * h * w, 4), 9.0, device=dev)
    lib.flow_head_out(x, 256, wt, bias, c_all, d_all)
    n = h * w
    for b in range(B):
        cb, db = c0[b:b + 1].clone(), torch.full((n, 4), 9.0, device=dev)
        lib.flow_head_out(x[b * n:(b + 1) * n], 256, wt, bias, cb, db)
        assert torch.equal(db, d_all[b * n:(b + 1) * n]), f"image {b}: delta_flow differs between the two forms"
        assert torch.equal(cb[0], c_all[b]), f"image {b}: coords1 differs between the two forms"
    want = torch.nn.functional.conv2d(x.view(B, h, w, 256).permute(0, 3, 1, 2), wt.view(2, 3, 3, 256).permute(0, 3, 1, 2), bias, padding=1)
    got = d_all[:, :2].view(B, h, w, 2).permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) < 2e-5 and float((d_all[:, 2:] - 9.0).abs().max()) == 0.0


def test_corr_pyramid_vs_reference_golden(lib, dev):
    g = gc.load("corr_pyramid")
    f1, f2 = gc.fmaps("corr", B=2)
    lv = [torch.empty(2 * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
    lib.corr_pyramid(kc.cl(f1).to(dev), kc.cl(f2).to(dev), lv, 2, H8, W8)
    torch.cuda.synchronize()
    rows = g["rows"]
    for i in range(4):
        kc.check(lv[i].cpu()[rows], T(g[f"l{i}"]).reshape(len(rows), -1), 3e-5, f"level {i} vs reference")
        assert float(lv[i].double().sum()) == pytest.approx(float(g["checksum"][i]), abs=2e-2 * (1 + i))
    # known answers: diagonal of corr(f,f) = |f|^2 / 16
    lib.corr_pyramid(kc.cl(f1).to(dev), kc.cl(f1).to(dev), lv, 2, H8, W8)
    diag = torch.diagonal(lv[0][:N].cpu())
    kc.check(diag, (f1[0].reshape(256, -1) ** 2).sum(0) / 16.0, 1e-4, "diag")


# ---- update blocks through the engine -----------------------------------------------------------
@pytest.mark.parametrize("presplit", ["1", "0"])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_update_blocks_vs_reference_golden(lib, dev, params, prec, presplit, monkeypatch):
    """presplit=1 (default, bf16x3): every MFMA conv of the blocks reads split twins through the all-DMA kernel and the
    GRU input / motion features exist as twins only; presplit=0: the fp32-staged kernels."""
    import argparse
    from prior_flow_amd.engine import Engine, Workspace, pack_update_blocks, unsplit
    from prior_flow_amd.prior_raft import PriOr_RAFT
    monkeypatch.setenv("PRIORFLOW_PRESPLIT", presplit)
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(params)
    model = model.to(dev)
    P = pack_update_blocks(model.ODDC, model.update_block, _prec(prec))
    ws = Workspace(lib, 1, 128, 256, dev)
    eng = Engine(lib)
    ui = gc.update_inputs("upd")
    up = lambda t: kc.cl(t).to(dev)
    ws.net_a[0].copy_(up(ui["net"])); ws.net_b[0].copy_(up(ui["net"]))
    ws.x_a[:, :128] = up(ui["inp"]); ws.x_b[:, :128] = up(ui["inp"])
    ws.x_a[:, 252:254] = up(ui["flow_a"]); ws.x_a[:, 254:256] = up(ui["flow_ba"])
    ws.x_b[:, 254:256] = up(ui["flow_a"])
    ws.flow4_a[:, 0:2] = up(ui["flow_a"]); ws.flow4_a[:, 2:4] = up(ui["flow_ba"])
    ws.flow2_b.copy_(up(ui["flow_a"]))
    ws.corr_a.copy_(up(ui["corr"])); ws.corr_b.copy_(up(ui["corr"]))
    ws.conf_in[:, :4] = up(ui["flaw_a"]); ws.conf_in[:, 4:] = up(ui["flaw_ba"])
    ps = eng.presplit(P)
    assert ps == (presplit == "1" and prec != "fp32")
    if ps:
        ws.sync_twins(lib)
    cur = eng.update_blocks(ws, P, 0, need_b=True, mask_a=True, mask_b=True)
    torch.cuda.synchronize()
    ga, gb = gc.load("update_A"), gc.load("update_B")
    back = lambda rows: kc.uncl(rows.cpu(), 1, H8, W8)
    tol = 3e-5 if prec == "fp32" else 2e-4
    xa, xb = (unsplit(ws.x_a_s, 256), unsplit(ws.x_b_s, 256)) if ps else (ws.x_a, ws.x_b)
    kc.check(back(xa[:, 128:]), ga["motion"], tol, "motion features A")
    kc.check(back(xb[:, 128:]), gb["motion"], tol, "motion features B")
    if ps:      # the hidden state exists in both forms: the twin is the split of the fp32 rows, bit for bit
        from prior_flow_amd.engine import split_twin
        for net, tw in ((ws.net_a[cur], ws.net_a_s[cur]), (ws.net_b[cur], ws.net_b_s[cur])):
            assert torch.equal(lib.split_bf16(net.contiguous(), split_twin(net.shape[0], 128, dev)), tw)
    kc.check(back(ws.net_a[cur]), ga["net"], tol, "net A")
    kc.check(back(ws.net_b[cur]), gb["net"], tol, "net B")
    kc.check(back(ws.delta_a[:, :2]), ga["delta"], tol, "delta A")
    kc.check(back(ws.delta_b[:, :2]), gb["delta"], tol, "delta B")
    # coords1 (zero here) += delta_flow is fused into the flow-head kernel
    kc.check(ws.c1a, back(ws.delta_a[:, :2]), 0.0, "coords1_A += delta")
    kc.check(ws.c1b, back(ws.delta_b[:, :2]), 0.0, "coords1_B += delta")
    kc.check(back(ws.mask_a)[:, 3::8], ga["mask"], tol, "mask A")
    kc.check(back(ws.mask_b)[:, 3::8], gb["mask"], tol, "mask B")


# ---- encoders on the HIP kernels ----------------------------------------------------------------
def test_conv_stride2_and_input_affine(lib, dev):
    """Generic kernel with stride 2 (3x3/2, 1x1/2) and the halo kernel's folded input norm+ReLU."""
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3
    from prior_flow_amd.engine import Conv, pack_mfma
    x = gc.uni("s2/x", (2, 64, 16, 64), -1, 1)
    for name, cout, k in (("3x3s2", 96, 3), ("1x1s2", 96, 1)):
        w = gc.uni(f"s2/{name}/w", (cout, 64, k, k), -0.1, 0.1)
        b = gc.uni(f"s2/{name}/b", (cout,), -0.1, 0.1)
        want = torch.nn.functional.conv2d(x, w, b, stride=2, padding=k // 2)
        wp, bp = pack_mfma(w.to(dev), b.to(dev))
        cv = Conv(wp, bp, k, k, 64, cout, PREC_BF16X3)
        out = torch.empty(2 * 8 * 32, cout, device=dev)
        xin = kc.cl(x).to(dev)
        lib.conv2d([cv.desc(xin, 0, 64, out, 0, EPI_LINEAR, stride=2)], 2, 8, 32, xin)
        kc.check(kc.uncl(out.cpu(), 2, 8, 32), want, 1.5e-4, name)
    # input affine + relu folded into the halo load (per image, per channel)
    w = gc.uni("aff/w", (64, 64, 3, 3), -0.1, 0.1)
    b = gc.uni("aff/b", (64,), -0.1, 0.1)
    sc = gc.uni("aff/sc", (2, 64), 0.5, 1.5)
    sh = gc.uni("aff/sh", (2, 64), -0.5, 0.5)
    xn = torch.relu(x * sc[:, :, None, None] + sh[:, :, None, None])
    want = torch.nn.functional.conv2d(xn, w, b, padding=1)
    wp, bp = pack_mfma(w.to(dev), b.to(dev))
    cv = Conv(wp, bp, 3, 3, 64, 64, PREC_BF16X3)
    out = torch.empty(2 * 16 * 64, 64, device=dev)
    xin = kc.cl(x).to(dev)
    lib.conv2d([cv.desc(xin, 0, 64, out, 0, EPI_LINEAR, in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True)],
               2, 16, 64, xin)
    kc.check(kc.uncl(out.cpu(), 2, 16, 64), want, 1.5e-4, "halo conv with folded norm+relu")
    _check_fused_stats(lib, dev, cv, xin, 64, 64, 2, 16, 64, expect_tile=3)
    # Cout = 96 on the 128-channel tile (ragged channel tail), no input affine
    w = gc.uni("st/w", (96, 64, 3, 3), -0.1, 0.1)
    wp, bp = pack_mfma(w.to(dev), gc.uni("st/b", (96,), -0.1, 0.1).to(dev))
    _check_fused_stats(lib, dev, Conv(wp, bp, 3, 3, 64, 96, PREC_BF16X3), xin, 64, 96, 2, 16, 64, expect_tile=3)


def test_conv_stride2_into_96_channels_takes_the_128x96_tile(lib, dev):
    """Round 6: the generic kernel's 128 px x 96 channel tile (pf_conv2d_tile 7) for the stride-2 convolutions into the encoders'
    layer 2 (core/extractor.py:29-38: 3x3 / 2 and the 1x1 / 2 downsample, 64 -> 96) on a map that fills the chip: against
    torch's fp32 convolution, bit for bit against the same image run alone (a launch of one image is too small for the tile and
    takes the 64-pixel one: per output the K order is the same), and its fused InstanceNorm partials (one per 128 pixels)."""
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3
    from prior_flow_amd.engine import Conv, pack_mfma
    B, h, w = 2, 64, 256                                   # output map; input 128 x 512
    x = gc.uni("s2t/x", (B, 64, 2 * h, 2 * w), -1, 1)
    xin = kc.cl(x).to(dev)
    for name, k in (("3x3s2", 3), ("1x1s2", 1)):
        wt = gc.uni(f"s2t/{name}/w", (96, 64, k, k), -0.1, 0.1)
        bs = gc.uni(f"s2t/{name}/b", (96,), -0.1, 0.1)
        want = torch.nn.functional.conv2d(x, wt, bs, stride=2, padding=k // 2)
        wp, bp = pack_mfma(wt.to(dev), bs.to(dev))
        cv = Conv(wp, bp, k, k, 64, 96, PREC_BF16X3)
        out = torch.full((B * h * w, 98), 5.0, device=dev)
        d = cv.desc(xin, 0, 64, out, 1, EPI_LINEAR, stride=2)
        assert lib.conv2d_tile([d], B, h, w) == 7
        nblk = lib.conv2d_stats_blocks([d], B, h, w)
        assert nblk == h * w // 128
        part = torch.full((B, nblk, 96, 2), float("nan"), dtype=torch.float64, device=dev)
        d.stats_out = part.data_ptr()
        lib.conv2d([d], B, h, w, xin)
        kc.check(kc.uncl(out[:, 1:97].cpu(), B, h, w), want, 1.5e-4, name)
        assert float((out[:, 0] - 5.0).abs().max()) == 0.0 and float((out[:, 97] - 5.0).abs().max()) == 0.0, "wrote outside its columns"
        y = out[:, 1:97].reshape(B, h * w, 96).double()
        assert float((part.sum(1)[..., 0] - y.sum(1)).abs().max()) < 1e-9 * h * w
        assert float((part.sum(1)[..., 1] - (y * y).sum(1)).abs().max()) < 1e-9 * h * w
        # one image alone: another tile, the same bits
        one = torch.empty(h * w, 96, device=dev)
        x1 = xin[:4 * h * w].contiguous()
        d1 = cv.desc(x1, 0, 64, one, 0, EPI_LINEAR, stride=2)
        assert lib.conv2d_tile([d1], 1, h, w) in (1, 2)
        lib.conv2d([d1], 1, h, w, x1)
        assert torch.equal(one, out[:h * w, 1:97]), f"{name}: tile 7 differs from the 64-pixel tile"


def test_conv_96_channels_takes_the_256x96_halo_tile(lib, dev):
    """Round 6: pf_conv_halo_kernel<3,3,3,.,8> (pf_conv2d_tile 8: 256 px x 96 channels) for the 3x3 96 -> 96 convolutions of the
    encoders' layer 2 (core/extractor.py:16-17 at 1/4 resolution) on a map that fills the chip: against torch's fp32 convolution
    with and without the folded input norm + ReLU, bit for bit against one image run alone (too small for the tile: the
    128 px x 128 channel one, a quarter of it padding -- per output the K order is the same), and its fused statistics."""
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3
    from prior_flow_amd.engine import Conv, pack_mfma
    B, h, w = 2, 128, 256
    x = gc.uni("t8/x", (B, 96, h, w), -1, 1)
    wt = gc.uni("t8/w", (96, 96, 3, 3), -0.1, 0.1)
    bs = gc.uni("t8/b", (96,), -0.1, 0.1)
    sc = gc.uni("t8/sc", (B, 96), 0.5, 1.5)
    sh = gc.uni("t8/sh", (B, 96), -0.5, 0.5)
    wp, bp = pack_mfma(wt.to(dev), bs.to(dev))
    cv = Conv(wp, bp, 3, 3, 96, 96, PREC_BF16X3)
    xin = kc.cl(x).to(dev)
    for affine in (False, True):
        kw = dict(in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True) if affine else {}
        xr = torch.relu(x * sc[:, :, None, None] + sh[:, :, None, None]) if affine else x
        want = torch.nn.functional.conv2d(xr, wt, bs, padding=1)
        out = torch.full((B * h * w, 98), 5.0, device=dev)
        d = cv.desc(xin, 0, 96, out, 1, EPI_LINEAR, **kw)
        assert lib.conv2d_tile([d], B, h, w) == 8
        lib.conv2d([d], B, h, w, xin)
        kc.check(kc.uncl(out[:, 1:97].cpu(), B, h, w), want, 2e-4, f"tile 8, affine={affine}")
        assert float((out[:, 0] - 5.0).abs().max()) == 0.0 and float((out[:, 97] - 5.0).abs().max()) == 0.0, "wrote outside its columns"
        one = torch.empty(h * w, 96, device=dev)
        x1 = xin[h * w:].contiguous()                                     # the SECOND image alone
        kw1 = dict(in_scale=sc[1:].contiguous().to(dev), in_shift=sh[1:].contiguous().to(dev), in_relu=True) if affine else {}
        d1 = cv.desc(x1, 0, 96, one, 0, EPI_LINEAR, **kw1)
        assert lib.conv2d_tile([d1], 1, h, w) == 4
        lib.conv2d([d1], 1, h, w, x1)
        assert torch.equal(one, out[h * w:, 1:97]), f"affine={affine}: tile 8 differs from tile 4"
    _check_fused_stats(lib, dev, cv, xin, 96, 96, B, h, w, expect_tile=8)


def _check_fused_stats(lib, dev, cv, xin, cin, cout, B, h, w, expect_tile):
    """InstanceNorm statistics fused into the conv epilogue == pf_channel_stats of the stored output."""
    from prior_flow_amd._lib import EPI_LINEAR
    out = torch.empty(B * h * w, cout, device=dev)
    d = cv.desc(xin, 0, cin, out, 0, EPI_LINEAR)
    tile = lib.conv2d_tile([d], B, h, w)
    assert tile == expect_tile
    nblk = lib.conv2d_stats_blocks([d], B, h, w)
    if tile in (3, 4, 5, 8):
        th = 8 if tile in (5, 8) else 4
        assert nblk == ((h + th - 1) // th) * ((w + 31) // 32)
    part = torch.full((B, nblk, cout, 2), float("nan"), dtype=torch.float64, device=dev)
    d.stats_out = part.data_ptr()
    lib.conv2d([d], B, h, w, xin)
    sc, sh = torch.empty(B, cout, device=dev), torch.empty(B, cout, device=dev)
    lib.channel_stats_final(part, B, h * w, cout, nblk, sc, sh)
    sc2, sh2 = torch.empty(B, cout, device=dev), torch.empty(B, cout, device=dev)
    lib.channel_stats(out, B, h * w, cout, sc2, sh2, torch.empty(B * 16 * cout * 2, dtype=torch.float64, device=dev), 16)
    y = out.view(B, h * w, cout).double()
    assert float((part.sum(1)[..., 0] - y.sum(1)).abs().max()) < 1e-9 * h * w
    kc.check(sc, sc2, 1e-6 * float(sc2.abs().max()), f"fused stats scale (tile {tile})")
    kc.check(sh, sh2, 1e-6 * float(sh2.abs().max()) + 1e-7, f"fused stats shift (tile {tile})")


def test_small_conv_partial_segments(lib, dev):
    """The small-Cin MFMA kernel works on 32-pixel row segments: a width that is not a multiple of 32
    (W8 = 27, 45, 120) ends in a partial segment (core/update.py:171-178 at sizes like 480x960)."""
    for H8, W8 in ((17, 27), (9, 45), (6, 120)):
        x = gc.uni(f"rag/x{W8}", (2, 2, H8, W8), -3, 3)
        w = gc.uni("rag/w", (128, 2, 7, 7), -0.2, 0.2)
        b = gc.uni("rag/b", (128,), -0.1, 0.1)
        want = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=3))
        out = torch.full((2 * H8 * W8, 130), 5.0, device=dev)
        lib.conv2d_direct(kc.cl(x).to(dev), 0, 2, w.permute(2, 3, 1, 0).reshape(49, 2, 128).contiguous().to(dev),
                          b.to(dev), out, 1, 128, 7, 7, True, 2, H8, W8)
        kc.check(kc.uncl(out[:, 1:129].cpu(), 2, H8, W8), want, 2e-5, f"7x7 small conv W8={W8}")
        assert float((out[:, 0] - 5.0).abs().max()) == 0.0 and float((out[:, 129] - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("cout", [128, 64])
def test_flow_stem_kernels_partial_tiles(lib, dev, cout):
    """The 7x7 2 -> Cout flow stems with 16-byte aligned outputs: Cout = 128 (the motion encoders' shape, core/update.py:87,173,175)
    takes pf_flow_stem_kernel (MFMA, 4-row x 32-column tiles, 3-pass bf16 split: 1e-4 on outputs of +-10), any other multiple of 64
    pf_stem7x7c2_valu (64-pixel row segments, lane = pixel, exact fp32: 2e-5).  Ragged widths and heights, batch 2, an input column
    offset and a padded input row stride, against torch conv2d; the columns beside the output slice stay untouched; with a split
    twin as the second output (what the DMA-fed 3x3 behind the stem reads) the twin decodes to the fp32 rows exactly."""
    from prior_flow_amd.engine import split_twin
    for H8, W8 in ((17, 27), (9, 45), (6, 120), (5, 200), (64, 128)):
        x = gc.uni(f"ragv/x{W8}", (2, 2, H8, W8), -3, 3)
        w = gc.uni("ragv/w", (128, 2, 7, 7), -0.2, 0.2)[:cout].contiguous()
        b = gc.uni("ragv/b", (128,), -0.1, 0.1)[:cout].contiguous()
        want = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=3))
        xin = torch.full((2 * H8 * W8, 4), 7.0, device=dev)
        xin[:, 2:4] = kc.cl(x).to(dev)
        out = torch.full((2 * H8 * W8, cout + 8), 5.0, device=dev)
        lib.conv2d_direct(xin, 2, 2, w.permute(2, 3, 1, 0).reshape(49, 2, cout).contiguous().to(dev),
                          b.to(dev), out, 4, cout, 7, 7, True, 2, H8, W8)
        kc.check(kc.uncl(out[:, 4:4 + cout].cpu(), 2, H8, W8), want, 1e-4 if cout == 128 else 2e-5, f"7x7 stem Cout={cout} W8={W8}")
        assert float((out[:, :4] - 5.0).abs().max()) == 0.0 and float((out[:, 4 + cout:] - 5.0).abs().max()) == 0.0
        # rows + twin in one launch: the same rows, and the twin is the bf16 hi | lo split of exactly those values
        rows2 = torch.empty(2 * H8 * W8, cout, device=dev)
        twin = split_twin(2 * H8 * W8, cout, dev)
        wt = w.permute(2, 3, 1, 0).reshape(49, 2, cout).contiguous().to(dev)
        lib.conv2d_direct_group([(xin, 2, wt, b.to(dev), rows2, 0, twin)], 2, cout, 7, 7, True, 2, H8, W8)
        assert torch.equal(rows2, out[:, 4:4 + cout])
        hi = rows2.to(torch.bfloat16)
        lo = (rows2 - hi.float()).to(torch.bfloat16)
        ref = torch.stack([hi.view(-1, cout // 32, 32), lo.view(-1, cout // 32, 32)], 2)
        assert torch.equal(twin.view(torch.int16), ref.contiguous().view(torch.int16))


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_stem_as_space_to_depth_conv(lib, dev, prec):
    """The 7x7 stride-2 stem (core/extractor.py:122) as space-to-depth + 4x4 stride-1 conv (even
    kernel: window [-2, +1]) on both conv kernels, vs torch conv2d on the original weights."""
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3, PREC_F32
    from prior_flow_amd.engine import Conv, pack_mfma, stem_s2d_weight
    precision = PREC_F32 if prec == "fp32" else PREC_BF16X3
    B, H, W = 2, 40, 128                     # output 20 x 64: halo kernel in bf16x3 mode
    img = gc.uni("stem2/img", (B, 3, H, W), -1, 1)
    w = gc.uni("stem2/w", (64, 3, 7, 7), -0.15, 0.15)
    b = gc.uni("stem2/b", (64,), -0.1, 0.1)
    want = torch.nn.functional.conv2d(img, w, b, stride=2, padding=3)
    wp, bp = pack_mfma(stem_s2d_weight(w).to(dev), b.to(dev))
    cv = Conv(wp, bp, 4, 4, 12, 64, precision)
    s2d = torch.empty(B * (H // 2) * (W // 2), 12, device=dev)
    lib.space_to_depth2(img.to(dev), s2d)
    out = torch.empty(B * (H // 2) * (W // 2), 64, device=dev)
    d = cv.desc(s2d, 0, 12, out, 0, EPI_LINEAR)
    assert lib.conv2d_tile([d], B, H // 2, W // 2) == (3 if prec == "bf16x3" else 1)
    lib.conv2d([d], B, H // 2, W // 2, s2d)
    kc.check(kc.uncl(out.cpu(), B, H // 2, W // 2), want, 3e-6 if prec == "fp32" else 1e-4, f"s2d stem {prec}")


def test_conv_8row_tile_layer1_size(lib, dev):
    """Tile 5 (256 px x 64 ch, 8-row halo tile) is only chosen once the map can fill the chip:
    run the encoder layer-1 shapes at 256x512 (core/extractor.py:122-127) against torch conv2d.  Round 5: the 3x3 64 -> 64
    convolutions of that size take the weights-stationary kernel (pf_conv2d_tile code 6, csrc/pf_enc_conv.hip); the 4x4 stem
    form stays on tile 5."""
    from prior_flow_amd._lib import EPI_LINEAR, EPI_RELU, PREC_BF16X3
    from prior_flow_amd.engine import Conv, pack_mfma, stem_s2d_weight
    B, h, w = 2, 256, 512
    # 3x3 64 -> 64 with the folded input norm + ReLU
    x = gc.uni("t5/x", (B, 64, h, w), -1, 1)
    wt = gc.uni("t5/w", (64, 64, 3, 3), -0.1, 0.1)
    b = gc.uni("t5/b", (64,), -0.1, 0.1)
    sc = gc.uni("t5/sc", (B, 64), 0.5, 1.5)
    sh = gc.uni("t5/sh", (B, 64), -0.5, 0.5)
    want = torch.nn.functional.conv2d(torch.relu(x * sc[:, :, None, None] + sh[:, :, None, None]), wt, b, padding=1)
    wp, bp = pack_mfma(wt.to(dev), b.to(dev))
    cv = Conv(wp, bp, 3, 3, 64, 64, PREC_BF16X3)
    xin = kc.cl(x).to(dev)
    out = torch.empty(B * h * w, 64, device=dev)
    d = cv.desc(xin, 0, 64, out, 0, EPI_LINEAR, in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True)
    assert lib.conv2d_tile([d], B, h, w) == 6
    lib.conv2d([d], B, h, w, xin)
    kc.check(kc.uncl(out.cpu(), B, h, w), want, 1.5e-4, "8-row tile 3x3 affine")
    _check_fused_stats(lib, dev, cv, xin, 64, 64, B, h, w, expect_tile=6)
    d = cv.desc(xin, 0, 64, out, 0, EPI_RELU)
    lib.conv2d([d], B, h, w, xin)
    kc.check(kc.uncl(out.cpu(), B, h, w), torch.relu(torch.nn.functional.conv2d(x, wt, b, padding=1)), 1.5e-4,
             "8-row tile 3x3 relu")
    # 4x4 space-to-depth stem
    img = gc.uni("t5/img", (B, 3, 2 * h, 2 * w), -1, 1)
    w7 = gc.uni("t5/w7", (64, 3, 7, 7), -0.15, 0.15)
    want = torch.nn.functional.conv2d(img, w7, b, stride=2, padding=3)
    wp, bp = pack_mfma(stem_s2d_weight(w7).to(dev), b.to(dev))
    cv = Conv(wp, bp, 4, 4, 12, 64, PREC_BF16X3)
    s2d = torch.empty(B * h * w, 12, device=dev)
    lib.space_to_depth2(img.to(dev), s2d)
    d = cv.desc(s2d, 0, 12, out, 0, EPI_LINEAR)
    assert lib.conv2d_tile([d], B, h, w) == 5
    lib.conv2d([d], B, h, w, s2d)
    kc.check(kc.uncl(out.cpu(), B, h, w), want, 1e-4, "8-row tile 4x4 stem")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["1x256x512", "4x64x128", "2x40x96", "3x72x160"])
def test_weights_stationary_conv_matches_halo_kernel_bitwise(shape, tmp_path):
    """pf_enc_conv64_kernel (round 5: the encoders' 3x3 64 -> 64 convolutions with W_hi in registers, W_lo in LDS and the input in
    a 10-row ring walked down a 32-column strip) against the halo kernel it replaces (PRIORFLOW_ENC_CONV64=0; both read once per
    process -> child processes, tests/run_conv_l1_check.py): outputs bit-identical with and without the folded input norm + ReLU
    with the ReLU epilogue and with the residual tail (PF_EPI_RELU_RES: cnet's folded-BatchNorm blocks), and the InstanceNorm scale / shift that pf_channel_stats_final makes of the fused partials (per
    segment, row phase and strip instead of per 8-row tile) equal to 1e-6 (bit-identical in practice).  PRIORFLOW_ENC_CONV64=2 forces the kernel
    onto maps too small to fill the chip (segments of 8 rows, several images, widths of 3 and 5 strips)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for mode in ("0", "2"):
        path = str(tmp_path / f"l1_{mode}.pt")
        subprocess.run([sys.executable, os.path.join(here, "run_conv_l1_check.py"), shape, path], check=True,
                       env=dict(os.environ, PRIORFLOW_ENC_CONV64=mode), timeout=600)
        outs[mode] = torch.load(path)
    old, new = outs["0"], outs["2"]
    assert [int(t) for t in new[3::4]] == [6, 6, 6, 6] and all(int(t) in (3, 5) for t in old[3::4])
    for k in range(4):
        o0, sc0, sh0 = old[4 * k: 4 * k + 3]
        o1, sc1, sh1 = new[4 * k: 4 * k + 3]
        assert torch.isfinite(o1).all() and torch.equal(o0, o1), (k, float((o0 - o1).abs().max()))
        assert float((sc0 - sc1).abs().max()) <= 1e-6 * float(sc0.abs().max() + 1e-30)
        assert float((sh0 - sh1).abs().max()) <= 1e-6 * float(sh0.abs().max() + 1e-30) + 1e-7


@pytest.mark.parametrize("which", ["fnet", "cnet"])
def test_encoder_plan_vs_reference_golden(lib, dev, params, which):
    """BasicEncoder (core/extractor.py:98-158) through EncoderPlan vs the reference's own output."""
    import argparse
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3
    from prior_flow_amd.engine import EncoderPlan
    from prior_flow_amd.prior_raft import PriOr_RAFT
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(params)
    model = model.to(dev).eval()
    im = gc.uni("enc/img", (2, 3, 128, 256), -1, 1)
    plan = EncoderPlan(lib, getattr(model, which), PREC_BF16X3)
    out = torch.empty(2 * 16 * 32, 256, device=dev)
    plan.run(im.to(dev).contiguous(), out, EPI_LINEAR)
    torch.cuda.synchronize()
    got = kc.uncl(out.cpu(), 2, 16, 32)
    g = gc.load("encoders")
    sel = slice(0, None, 4) if which == "fnet" else slice(1, None, 4)
    kc.check(got[:, sel], g[which], 3e-4, f"{which} vs reference")
    want = po.encoder(params, which + ".", im, "instance" if which == "fnet" else "batch")
    kc.check(got, want, 3e-4, f"{which} vs oracle (all channels)")


@pytest.mark.gpu
def test_conv_roles_kernel_matches_symmetric_kernel_bitwise(tmp_path):
    """pf_conv_ws_kernel (four MFMA waves + four loader waves, either tile) accumulates every output in the symmetric
    halo kernel's order: same bits.  PRIORFLOW_CONV_WS is read once per process, so each form runs in a child."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for ws in ("0", "1", "2"):
        path = str(tmp_path / f"ws{ws}.pt")
        env = dict(os.environ, PRIORFLOW_CONV_WS=ws)
        subprocess.run([sys.executable, os.path.join(here, "run_conv_case.py"), path], check=True, env=env, timeout=600)
        res[ws] = torch.load(path)
    roles = {ws: {k: v for k, v in res[ws]["roles"].items() if not k.startswith("relu3x3_64")} for ws in res}   # (tile 5: symmetric kernel only)
    assert set(roles["0"].values()) == {0}
    assert set(roles["1"].values()) == {1}
    assert 2 in roles["2"].values() and 1 in roles["2"].values()       # both tiles of the default are exercised
    for key, ref in res["0"]["out"].items():
        assert torch.isfinite(ref).all() and ref.abs().max() > 0.05, key
        for ws in ("1", "2"):
            assert torch.equal(res[ws]["out"][key], ref), (key, ws, (res[ws]["out"][key] - ref).abs().max().item())


@pytest.mark.gpu
def test_conv_dma_kernel_matches_symmetric_kernel_bitwise(tmp_path):
    """pf_conv_dma_kernel (operands handed over as split twins, both by LDS-DMA) against the symmetric halo kernel on the
    same fp32 inputs: the twins hold the bits the fp32 kernels make while staging and the MFMA order per accumulator is the
    same, so every fp32 output is equal bit for bit; and every output twin equals pf_split_bf16 of the fp32 output."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, env in (("ref", dict(PRIORFLOW_CONV_WS="0")), ("dma", dict(PF_CASE_SPLIT="1"))):
        path = str(tmp_path / f"{tag}.pt")
        subprocess.run([sys.executable, os.path.join(here, "run_conv_case.py"), path], check=True,
                       env=dict(os.environ, **env), timeout=600)
        res[tag] = torch.load(path)
    assert set(res["ref"]["roles"].values()) == {0}
    assert set(res["dma"]["roles"].values()) == {17, 18}, res["dma"]["roles"]         # both tiles of the DMA kernel ran
    assert all(res["dma"]["twin_ok"].values()), res["dma"]["twin_ok"]
    for key, ref in res["ref"]["out"].items():
        got = res["dma"]["out"][key]
        assert torch.equal(got, ref), (key, (got - ref).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 64, 128), (2, 22, 40), (8, 64, 128)])
def test_gru_half_step_with_hoisted_context_equals_full_convolutions(lib, dev, shape):
    """pf_conv_desc.pre (accumulators start from a stored map) and a second operand segment at a channel offset: one SepConvGRU
    half-step (core/update.py:46-60) as the engine runs it with the context hoisted -- pre = conv_inp(inp) + bias once, then
    z|r and q over [h | motion] only -- against the same half-step with the full 384-channel convolutions.  Same products,
    another summation order: equal to fp32 rounding of the pre-activations (bound 2e-5 on gate outputs of size <= 1)."""
    from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR, EPI_LINEAR, PREC_BF16X3, PfError
    from prior_flow_amd.engine import Conv, split_twin
    B, H8, W8 = shape
    N = B * H8 * W8
    torch.manual_seed(11)
    for kh, kw in ((1, 5), (5, 1)):
        cz, cr, cq = [torch.nn.Conv2d(384, 128, (kh, kw), padding=(kh // 2, kw // 2)).to(dev) for _ in range(3)]
        h = (torch.rand(N, 128, device=dev) * 2 - 1)
        x = (torch.rand(N, 256, device=dev) * 2 - 1)

        def twin(t):
            return lib.split_bf16(t, split_twin(t.shape[0], t.shape[1], dev))
        hs, xs = twin(h), twin(x)

        def half_step(hoisted):
            z = torch.zeros(N, 128, device=dev)
            hn = torch.zeros(N, 128, device=dev)
            rhs, hns = split_twin(N, 128, dev), split_twin(N, 128, dev)
            if hoisted:
                pre = torch.zeros(N, 384, device=dev)
                cp = Conv.of_slices((cz, cr, cq), [(128, 256)], PREC_BF16X3, with_bias=True)
                czr = Conv.of_slices((cz, cr), [(0, 128), (256, 384)], PREC_BF16X3, with_bias=False)
                cqq = Conv.of_slices((cq,), [(0, 128), (256, 384)], PREC_BF16X3, with_bias=False)
                lib.conv2d([cp.desc(None, 0, 128, pre, 0, EPI_LINEAR, in0s=xs)], B, H8, W8, x)
                kz = dict(off1=128, c1=128, pre=pre, off_pre=0)
                kq = dict(off1=128, c1=128, pre=pre, off_pre=256)
            else:
                czr, cqq = Conv.fused(cz, cr, PREC_BF16X3), Conv.of(cq, PREC_BF16X3)
                kz = kq = dict(off1=0, c1=256)
            dz = czr.desc(None, 0, 128, z, 0, EPI_GRU_ZR, h=h, in0s=hs, in1s=xs, auxs=rhs, **kz)
            assert lib.conv2d_roles([dz], B, H8, W8) >= 16            # the all-DMA kernel
            lib.conv2d([dz], B, H8, W8, x)
            lib.conv2d([cqq.desc(None, 0, 128, hn, 0, EPI_GRU_Q, h=h, z=z, in0s=rhs, in1s=xs, outs=hns, **kq)], B, H8, W8, x)
            torch.cuda.synchronize()
            return z, hn, hns
        z0, h0, hs0 = half_step(False)
        z1, h1, hs1 = half_step(True)
        assert float((z1 - z0).abs().max()) < 2e-5 and float((h1 - h0).abs().max()) < 2e-5, \
            (kh, kw, float((z1 - z0).abs().max()), float((h1 - h0).abs().max()))
        assert torch.equal(hs1, twin(h1))                               # the output twin is the split of the fp32 output
        assert float(h0.abs().mean()) > 0.05
    # start values are implemented by the all-DMA kernel only: fp32 operands + pre is refused, not silently ignored
    c = Conv.of(cq, PREC_BF16X3)
    with pytest.raises(PfError):
        lib.conv2d([c.desc(h, 0, 128, torch.zeros(N, 128, device=dev), 0, EPI_LINEAR, in1=x, off1=0, c1=256,
                           pre=torch.zeros(N, 384, device=dev), off_pre=0)], B, H8, W8, x)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H8,W8", [(1, 48, 64), (2, 24, 40), (1, 16, 16)])
def test_lookup_backward_per_query_kernel_matches_the_per_channel_scatter(dev, B, H8, W8, tmp_path):
    """pf_dccl_lookup_bwd: the wave-per-query kernel (LDS windows, one global atomic per touched cell; default) against the
    per-(query, channel) scatter pf_lookup_bwd_elem (PRIORFLOW_LOOKUP_BWD=elem, read once per process -> child processes): the
    same eight pyramid gradients up to the order of the additions.  (16, 16): level 3 is a 2 x 2 map -- every window folds onto
    itself; coordinates run outside the map on every side; a random sampling grid tears the other view's windows apart (the
    direct-atomic fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import math, sys, torch
sys.path.insert(0, {root!r})
from prior_flow_amd import _lib
from prior_flow_amd.engine import rotation_x
lib = _lib.load(); dev = torch.device('cuda:0')
B, H8, W8 = {B}, {H8}, {W8}; N = H8 * W8
g = torch.Generator().manual_seed(11)
xs = torch.arange(W8).view(1, 1, 1, W8).expand(B, 1, H8, W8).float()
ys = torch.arange(H8).view(1, 1, H8, 1).expand(B, 1, H8, W8).float()
coords = (torch.cat([xs, ys], 1) + (torch.rand(B, 2, H8, W8, generator=g) * 30 - 15)).contiguous().to(dev)
g8 = torch.empty(2, H8, W8, device=dev); lib.sample_grid(g8, rotation_x(math.pi / 2))
g_rand = torch.stack([torch.rand(H8, W8, generator=g) * (W8 + 4) - 2, torch.rand(H8, W8, generator=g) * (H8 + 4) - 2]).to(dev).contiguous()
d_own = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
d_raw = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
res = []
for grid in (g8, g_rand):
    own = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
    oth = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
    for _ in range(2):                      # accumulated into: two launches
        lib.dccl_lookup_bwd(coords, grid, d_own, d_raw, own, oth)
    res += [t.cpu() for t in own + oth]
torch.save(res, sys.argv[1])
"""
    outs = {}
    for mode in ("rows", "elem"):
        path = str(tmp_path / f"lkb_{mode}.pt")
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, PRIORFLOW_LOOKUP_BWD=mode), timeout=600)
        outs[mode] = torch.load(path)
    assert len(outs["rows"]) == len(outs["elem"]) == 16
    for i, (r, e) in enumerate(zip(outs["rows"], outs["elem"])):
        assert float(e.abs().max()) > 0.1, i
        assert float((r - e).abs().max()) < 3e-5 * max(1.0, float(e.abs().max())), ("view / level", i, float((r - e).abs().max()))


@pytest.mark.gpu
def test_lookup_backward_clears_its_raw_gradient_and_coords_add_to(lib, dev):
    """Round 5, two launches per iteration less in the training loop: pf_dccl_lookup_bwd(clear_raw=1) leaves d_raw all zero (the
    next pf_dccl_combine_bwd scatters into it) and gives the same gradients; pf_coords_add_to writes src + delta to another buffer
    (the in-place form behind a copy, bit for bit)."""
    import math
    from prior_flow_amd.engine import rotation_x
    B, H8, W8 = 2, 24, 40
    N = H8 * W8
    g = torch.Generator().manual_seed(3)
    xs = torch.arange(W8).view(1, 1, 1, W8).expand(B, 1, H8, W8).float()
    ys = torch.arange(H8).view(1, 1, H8, 1).expand(B, 1, H8, W8).float()
    coords = (torch.cat([xs, ys], 1) + (torch.rand(B, 2, H8, W8, generator=g) * 10 - 5)).contiguous().to(dev)
    g8 = torch.empty(2, H8, W8, device=dev)
    lib.sample_grid(g8, rotation_x(math.pi / 2))
    d_own = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
    d_raw = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
    outs = []
    for clear in (False, True):
        own = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
        oth = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
        raw = d_raw.clone()
        lib.dccl_lookup_bwd(coords, g8, d_own, raw, own, oth, clear_raw=clear)
        assert (float(raw.abs().max()) == 0.0) if clear else torch.equal(raw, d_raw)
        outs.append(own + oth)
    for a, b in zip(*outs):
        assert float(a.abs().max()) > 0.1 and float((a - b).abs().max()) < 1e-5
    delta = torch.zeros(B * N, 4, device=dev)
    delta[:, :2] = (torch.rand(B * N, 2, generator=g) - 0.5).to(dev)
    inplace = coords.clone()
    lib.coords_add(inplace, delta)
    dst = torch.full_like(coords, float("nan"))
    lib.coords_add(dst, delta, src=coords)
    assert torch.equal(dst, inplace) and not torch.equal(dst, coords)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 64, 128), (2, 24, 40), (1, 17, 27)])
def test_lookup_window_kernel_matches_per_thread_kernel_bitwise(dev, shape, tmp_path):
    """pf_lookup_win_kernel (a wave per pixel: shared x / y tap geometry, cooperative window loads through LDS;
    PRIORFLOW_LOOKUP_WIN=1, read once per process -> child processes) against the per-thread statement pf_lookup_elem
    (PRIORFLOW_LOOKUP_WIN=0): same bits for both outputs, planar and interleaved grid, on coordinates that exercise every edge
    rule (negative and multi-wrap x, x in (W-1, W), y far outside, exact integers, flows of +-W/2) and on maps with odd pyramid
    levels.  (Round 5: the per-thread results used to come from pf_dccl_lookup_pair in the same process; that entry point lost its
    A/B twice and was removed.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    outs = {}
    for win in ("0", "1"):
        path = str(tmp_path / f"lookup_win{win}.pt")
        code = "import os, sys, torch; sys.path[:0] = [%r, %r, %r]; import test_hip_kernels as t; from prior_flow_amd import _lib; " \
               "t._lookup_window_case(_lib.load(), torch.device('cuda:0'), %r, %r)" % (here, root, os.path.join(root, "oracle"), tuple(shape), path)
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PRIORFLOW_LOOKUP_WIN=win), timeout=600)
        outs[win] = torch.load(path)
    assert len(outs["0"]) == len(outs["1"]) == 8
    for k, (a, b) in enumerate(zip(outs["0"], outs["1"])):
        assert torch.isfinite(a).all() and torch.equal(a, b), (k, float((a - b).abs().max()))


def _lookup_window_case(lib, dev, shape, out_path):
    B, H8, W8 = shape
    N = H8 * W8
    g = torch.Generator().manual_seed(5)
    base = torch.stack(torch.meshgrid(torch.arange(H8, dtype=torch.float32), torch.arange(W8, dtype=torch.float32), indexing="ij")[::-1])
    flow = (torch.rand(B, 2, H8, W8, generator=g) - 0.5) * 24.0
    flow[:, :, ::3, ::5] = torch.round(flow[:, :, ::3, ::5])                   # exact integers
    flow[:, 0, 1::4] += 2.5 * W8                                              # multi-wrap
    flow[:, 0, 2::4] -= 1.75 * W8                                             # negative
    flow[:, 1, :, 3::7] += 3.0 * H8                                           # far below
    flow[:, 1, :, 5::7] -= 2.0 * H8                                           # far above
    coords = (base[None] + flow).contiguous()
    coords[:, 0, 0, :4] = torch.tensor([W8 - 0.5, W8 - 1.0, -0.25, W8 + 0.0])
    pyr = [[torch.randn(B * N, (H8 >> l) * (W8 >> l), generator=g).to(dev) for l in range(4)] for _ in range(4)]
    grid = torch.stack([torch.rand(H8, W8, generator=g) * (W8 + 4) - 2, torch.rand(H8, W8, generator=g) * (H8 + 4) - 2]).to(dev).contiguous()
    g_il = grid.reshape(2, -1).t().contiguous()
    co = coords.to(dev)
    saved = []
    for il in (None, g_il):
        out = [torch.full((B * N, 324), float("nan"), device=dev) for _ in range(4)]
        lib.dccl_lookup(co, pyr[0], pyr[1], grid, out[0], out[1], il)
        lib.dccl_lookup(co, pyr[2], pyr[3], grid, out[2], out[3], il)
        torch.cuda.synchronize()
        saved += [t.cpu() for t in out]
    torch.save(saved, out_path)


@pytest.mark.gpu
def test_conv_twin_slice_must_end_at_the_row_end_or_on_a_chunk(lib, dev):
    """the all-DMA kernel copies whole 32-channel chunks of a split twin, so an operand slice whose width is not a
    multiple of 32 is only legal when it ends at the end of the twin's row (where the columns past the logical width are zero by
    contract).  A 16-channel slice in the middle of a wider row would multiply its neighbour's live columns by zero weights --
    wrong the moment they hold Inf / NaN -- and is rejected; the same slice at the row end runs and ignores poisoned fp32 data
    outside the twin."""
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3, PfError
    from prior_flow_amd.engine import Conv, pack_mfma, split_twin
    B, H8, W8 = 1, 16, 32
    N = B * H8 * W8
    torch.manual_seed(3)
    x = torch.rand(N, 80, device=dev) * 2 - 1                    # 2.5 chunks: [64 live | 16 live | 16 zero padding]
    xs = lib.split_bf16(torch.cat([x, torch.zeros(N, 16, device=dev)], 1), split_twin(N, 96, dev))
    w = (torch.rand(64, 16, 3, 3, device=dev) * 2 - 1) * 0.1
    cv = Conv(*pack_mfma(w, torch.zeros(64, device=dev)), 3, 3, 16, 64, PREC_BF16X3)
    out = torch.zeros(N, 64, device=dev)
    # 16 channels at offset 64 of a 96-column twin: ends at the row end (64 + 32 == 96) -> legal
    lib.conv2d([cv.desc(None, 64, 16, out, 0, EPI_LINEAR, in0s=xs)], B, H8, W8, out)
    ref = torch.nn.functional.conv2d(x[:, 64:80].reshape(B, H8, W8, 16).permute(0, 3, 1, 2), w, padding=1)
    assert float((out.reshape(B, H8, W8, 64).permute(0, 3, 1, 2) - ref).abs().max()) < 1e-3
    # the same 16 channels at offset 32 (columns 32..47 of the row, live columns 48..63 behind them) -> rejected
    with pytest.raises(PfError):
        lib.conv2d([cv.desc(None, 32, 16, out, 0, EPI_LINEAR, in0s=xs)], B, H8, W8, out)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 128, 256), (1, 136, 216), (3, 64, 80)])
def test_enc_stem_matches_conv2d(lib, dev, shape):
    """pf_enc_stem (round 4: the encoders' 7x7 / 2 stem from the NCHW image, K = the 7x7x3 patch) against torch's fp32 conv2d
    (core/extractor.py:122,144) on full and ragged tile grids: values to the bf16x3 class (1e-4 of the largest output), the ReLU
    form, the split twin bit-equal to pf_split_bf16 of the fp32 output, and the fused InstanceNorm statistics against the
    mean / variance of the stored output.  Every launch runs behind pf_debug_dirty_lds (all of LDS = NaN patterns)."""
    from prior_flow_amd.engine import pack_stem7x7, split_twin
    Bn, H, W = shape
    torch.manual_seed(17)
    img = (torch.rand(Bn, 3, H, W, device=dev) * 2 - 1)
    w = (torch.rand(64, 3, 7, 7, device=dev) * 2 - 1) * 0.2
    b = (torch.rand(64, device=dev) * 2 - 1) * 0.3
    ref = torch.nn.functional.conv2d(img, w, b, stride=2, padding=3)                  # [Bn,64,H/2,W/2]
    h, w2 = H // 2, W // 2
    rows = Bn * h * w2
    wp = pack_stem7x7(w)
    nblk = ((h + 7) // 8) * ((w2 + 31) // 32)
    for relu in (False, True):
        out = torch.full((rows, 64), float("nan"), device=dev)
        tw = split_twin(rows, 64, dev)
        part = torch.zeros(Bn * nblk * 64 * 2, dtype=torch.float64, device=dev)
        # the fragment reads touch padding floats of the LDS patch (x zero weights); with NaNs left in LDS by an
        # earlier kernel an uninitialised pad would poison column x0 + 31 of every tile and the fused statistics
        lib.debug_dirty_lds(0x7fc00000, like=img)
        lib.enc_stem(img, wp, b, out=out, out_split=tw, relu=relu, stats=part)
        want = ref.relu() if relu else ref
        got = out.view(Bn, h, w2, 64).permute(0, 3, 1, 2)
        assert torch.isfinite(out).all()
        assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max()), float((got - want).abs().max())
        assert torch.equal(lib.split_bf16(out, split_twin(rows, 64, dev)), tw)
        sc, sh = torch.empty(Bn, 64, device=dev), torch.empty(Bn, 64, device=dev)
        lib.channel_stats_final(part, Bn, h * w2, 64, nblk, sc, sh)
        o3 = out.view(Bn, h * w2, 64).double()
        mean, var = o3.mean(1), o3.var(1, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        assert float((sc.double() - rstd).abs().max()) < 1e-5 * float(rstd.abs().max())
        assert float((sh.double() + mean * rstd).abs().max()) < 1e-5 * float((mean * rstd).abs().max() + 1.0)
    # twin only (cnet's folded form): no fp32 rows
    tw2 = split_twin(rows, 64, dev)
    lib.debug_dirty_lds(0x7fc00000, like=img)
    lib.enc_stem(img, wp, b, out=None, out_split=tw2, relu=True)
    assert torch.equal(tw2, tw)


@pytest.mark.parametrize("relu", [True, False])
def test_frozen_batchnorm_act_matches_torch_autograd(relu):
    """autograd.HipFrozenBnAct (pf_bn_frozen_fwd / pf_bn_frozen_bwd: the context encoder's BatchNorm with frozen statistics + ReLU,
    core/extractor.py:114-115, train_flow.py:107-108) against torch's F.batch_norm(training=False) [+ relu] and its autograd: output,
    dx, d gamma, d beta to 1e-5 of their scale; channel counts of the three encoder stages, odd row count."""
    import torch.nn.functional as F
    from prior_flow_amd.autograd import HipFrozenBnAct
    gen = torch.Generator().manual_seed(17)
    for Cc, (Bn, Hh, Ww) in ((64, (2, 9, 21)), (96, (1, 12, 16)), (128, (3, 6, 8))):
        x = torch.randn(Bn, Cc, Hh, Ww, generator=gen).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        gamma = (torch.rand(Cc, generator=gen) + 0.5).cuda().requires_grad_()
        beta = (torch.randn(Cc, generator=gen) * 0.3).cuda().requires_grad_()
        mean = (torch.randn(Cc, generator=gen) * 0.2).cuda()
        var = (torch.rand(Cc, generator=gen) + 0.3).cuda()
        g = torch.randn(Bn, Cc, Hh, Ww, generator=gen).cuda()
        ref = F.batch_norm(x, mean, var, gamma, beta, False, 0.1, 1e-5)
        ref = torch.relu(ref) if relu else ref
        rdx, rdg, rdb = torch.autograd.grad(ref, (x, gamma, beta), g)
        out = HipFrozenBnAct.apply(x, gamma, beta, mean, var, 1e-5, relu)
        dx, dg, db = torch.autograd.grad(out, (x, gamma, beta), g)
        for a, b in ((out.detach(), ref.detach()), (dx, rdx), (dg, rdg), (db, rdb)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), (Cc, relu)
