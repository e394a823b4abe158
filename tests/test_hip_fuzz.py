"""Seeded random-shape sweep of the conv / corr / lookup kernels through the C-ABI against torch conv2d and
the CPU oracle: ragged maps, channel counts that are not tile multiples, two-segment inputs, grouped
launches, both precisions.  Complements the hand-picked cases of test_hip_kernels.py."""
import random

import pytest
import torch

import golden_cases as gc
import kernel_cases as kc
import priorflow_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from prior_flow_amd import _lib
    return _lib.load()


def _conv_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        k = rng.choice([(3, 3), (1, 5), (5, 1), (1, 1), (3, 3), (7, 7)])
        c0 = 4 * rng.randint(1, 40)
        c1 = rng.choice([0, 0, 32 * rng.randint(1, 4)])
        if c1:
            c0 = 32 * rng.randint(1, 4)                 # a second segment needs c0 % 32 == 0
        cout = rng.choice([2, 30, 64, 96, 126, 128, 192, 256])
        B = rng.choice([1, 2, 3])
        H = rng.randint(3, 21)
        W = rng.choice([rng.randint(5, 70), 32, 64, 96])
        relu = rng.random() < 0.5
        groups = rng.choice([1, 1, 2, 3])
        out.append((i, k, c0, c1, cout, B, H, W, relu, groups))
    return out


@pytest.mark.parametrize("prec", ["bf16x3", "fp32"])
def test_random_convs(lib, prec):
    from prior_flow_amd._lib import EPI_LINEAR, EPI_RELU, PREC_BF16X3, PREC_F32
    from prior_flow_amd.engine import Conv, pack_mfma
    dev = torch.device("cuda")
    precision = PREC_BF16X3 if prec == "bf16x3" else PREC_F32
    for i, (kh, kw), c0, c1, cout, B, H, W, relu, groups in _conv_cases(24, 2024):
        descs, wants, outs, keep = [], [], [], []        # `keep`: descriptors hold raw pointers only
        for g in range(groups):
            x0 = gc.uni(f"fz/{i}/{g}/x0", (B, c0, H, W), -1, 1)
            x1 = gc.uni(f"fz/{i}/{g}/x1", (B, c1, H, W), -1, 1) if c1 else None
            cin = c0 + c1
            s = (1.0 / (cin * kh * kw)) ** 0.5 * 1.5
            w = gc.uni(f"fz/{i}/{g}/w", (cout, cin, kh, kw), -s, s)
            b = gc.uni(f"fz/{i}/{g}/b", (cout,), -0.2, 0.2)
            xin = x0 if x1 is None else torch.cat([x0, x1], 1)
            want = torch.nn.functional.conv2d(xin, w, b, padding=(kh // 2, kw // 2))
            wants.append(torch.relu(want) if relu else want)
            wp, bp = pack_mfma(w.to(dev), b.to(dev))
            cv = Conv(wp, bp, kh, kw, cin, cout, precision)
            # inputs live in wider row buffers at a column offset (the engine's virtual concat)
            ld0, off0 = c0 + 8, 4
            buf0 = torch.full((B * H * W, ld0), 9.0, device=dev)
            buf0[:, off0:off0 + c0] = kc.cl(x0).to(dev)
            kw_args = {}
            if x1 is not None:
                buf1 = torch.full((B * H * W, c1 + 4), -7.0, device=dev)
                buf1[:, :c1] = kc.cl(x1).to(dev)
                kw_args = dict(in1=buf1, off1=0, c1=c1)
            out = torch.full((B * H * W, cout + 6), 3.0, device=dev)
            outs.append(out)
            keep.append((cv, buf0, kw_args))
            descs.append(cv.desc(buf0, off0, c0, out, 2, EPI_RELU if relu else EPI_LINEAR, **kw_args))
        lib.conv2d(descs, B, H, W, outs[0])
        tol = 3e-5 if prec == "fp32" else 2e-4
        for g in range(groups):
            got = kc.uncl(outs[g][:, 2:2 + cout].cpu(), B, H, W)
            kc.check(got, wants[g], tol * max(1.0, float(wants[g].abs().max())),
                     f"conv case {i} group {g}: k={kh}x{kw} cin={c0}+{c1} cout={cout} B={B} {H}x{W} relu={relu} {prec}")
            assert float((outs[g][:, :2] - 3.0).abs().max()) == 0.0 and float((outs[g][:, 2 + cout:] - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("size", [(16, 32), (17, 27), (24, 64), (20, 45), (32, 96)])
def test_corr_pyramid_and_lookup_sizes(lib, size):
    """corr + pyramid (fused and generic paths, both precisions) and the DCCL lookup at several map sizes."""
    import math
    from prior_flow_amd.engine import rotation_x
    dev = torch.device("cuda")
    h, w = size
    n = h * w
    f1, f2 = gc.fmaps(f"fz/corr{h}x{w}", 1, h, w)
    want = po.build_pyramid(po.corr_volume(f1, f2))
    rows = lambda f: kc.cl(f).to(dev).contiguous()                                    # noqa: E731
    for mode in ("fp32", "bf16x3"):
        lv = [torch.full((n, (h >> i) * (w >> i)), 5.0, device=dev) for i in range(4)]
        if mode == "fp32":
            lib.corr_pyramid(rows(f1), rows(f2), lv, 1, h, w)
        else:
            sp = [lib.split_bf16(rows(f), torch.empty(n, 8, 2, 32, dtype=torch.bfloat16, device=dev)) for f in (f1, f2)]
            lib.corr_pyramid_bf16x3(sp[0], sp[1], lv, 1, h, w, 256)
        for i in range(4):
            kc.check(lv[i], want[i].reshape(n, -1), 2e-5 if mode == "fp32" else 3e-4, f"pyramid level {i} {h}x{w} {mode}")
    # lookup + combine against the oracle with the exact-fp32 pyramid
    coords = gc.nasty_coords(f"fz/co{h}x{w}", 1, h, w)
    g_w2c = po.sample_grid(h, w, po.rotation_x(math.pi / 2))
    g_back = po.sample_grid(h, w, po.rotation_x(math.pi / 2))
    own_w, cross_w = po.dccl_lookup(coords, want, want, g_w2c, g_back)
    pyr = [p.reshape(n, -1).contiguous().to(dev) for p in want]
    own, raw, out = (torch.empty(n, 324, device=dev) for _ in range(3))
    lib.dccl_lookup(coords.to(dev), pyr, pyr, g_w2c.to(dev).contiguous(), own, raw)
    lib.dccl_combine(own, raw, g_back.to(dev).contiguous(), out, 1, h, w)
    kc.check(kc.uncl(own.cpu(), 1, h, w), own_w, 2e-5, f"own lookup {h}x{w}")
    kc.check(kc.uncl(out.cpu(), 1, h, w), own_w + cross_w, 5e-5, f"own + cross lookup {h}x{w}")
