"""CPU oracle for the PriOr-RAFT hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch fp32 restatement (explicit floor / index / weight arithmetic; the
only library arithmetic used is ``conv2d``/``matmul``) of the reference's inner
loop.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product path
(``prior-flow_amd/``) never does and fails loudly without its HIP library.

Parity pinning: the reference has no tests / golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference itself,
imported in the build container by ``oracle/gen_golden.py`` and committed as
fixtures under ``tests/golden/`` (``tests/test_oracle_golden.py`` checks every one
of them; the generators ``oracle/gen_golden*.py`` are the only code that imports the reference).

Every function cites the reference lines (relative to /root/reference/PriOr-RAFT)
it restates.  Tensors are torch CPU fp32, NCHW unless noted.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

CORR_LEVELS = 4      # core/prior_raft.py:34
CORR_RADIUS = 4      # core/prior_raft.py:35
HIDDEN = 128         # core/prior_raft.py:32
CONTEXT = 128        # core/prior_raft.py:33


# --------------------------------------------------------------------------------------
# samplers
# --------------------------------------------------------------------------------------
def pymod(x: Tensor, w: float) -> Tensor:
    """Python-style float remainder in [0, w)  (``xgrid % W``, core/utils/utils.py:83)."""
    return torch.remainder(x, w)


def _roundtrip(p: Tensor, size: int) -> Tensor:
    """pixel -> [-1,1] -> pixel, exactly as the callers' normalisation followed by
    grid_sample(align_corners=True) does it in fp32 (core/utils/utils.py:85-89)."""
    pn = 2 * p / (size - 1) - 1
    return (pn + 1) * ((size - 1) / 2.0)


def bilin0(img: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """Zero-padded bilinear sample.  img [B,C,H,W]; x,y [B,*] pixel coords -> [B,C,*].

    Restates ``F.grid_sample(bilinear, zeros, align_corners=True)`` after the
    ``2x/(W-1)-1`` normalisation of core/utils/utils.py:61-75 / :78-95.
    """
    B, C, H, W = img.shape
    shp = x.shape[1:]
    x = _roundtrip(x.reshape(B, -1), W)
    y = _roundtrip(y.reshape(B, -1), H)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    wx = x - x0
    wy = y - y0
    ex = 1 - wx
    ey = 1 - wy
    flat = img.reshape(B, C, H * W)
    out = None
    for dy, dx, wgt in ((0, 0, ey * ex), (0, 1, ey * wx), (1, 0, wy * ex), (1, 1, wy * wx)):
        xi = x0 + dx
        yi = y0 + dy
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()
        val = torch.gather(flat, 2, idx[:, None, :].expand(B, C, -1))
        term = val * (wgt * ok)[:, None, :]
        out = term if out is None else out + term
    return out.reshape(B, C, *shp)


def cycle_bilinear_sampler(img: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """x wrapped mod W, then zero-padded bilinear (core/utils/utils.py:78-95 and the
    private copy core/utils/projection_prim_ortho.py:119-135)."""
    return bilin0(img, pymod(x, img.shape[-1]), y)


def wrapgather(t: Tensor, gx: Tensor, gy: Tensor, is_grid: bool) -> Tensor:
    """True-wrap (x) / clamp (y) bilinear gather with optional seam un-wrapping of
    channel 0 (core/utils/my_cycle_sample.py:6-97).  t [B,C,H,W]; gx,gy [B,h,w]."""
    B, C, H, W = t.shape
    shp = gx.shape[1:]
    gx = pymod(gx.reshape(B, -1), W)
    gy = gy.reshape(B, -1)
    fx = torch.floor(gx)
    fy = torch.floor(gy)
    wx = gx - fx
    wy = gy - fy
    x0 = torch.remainder(fx.long(), W)
    x1 = torch.remainder(fx.long() + 1, W)
    y0 = fy.long().clamp(0, H - 1)
    y1 = (fy.long() + 1).clamp(0, H - 1)
    flat = t.reshape(B, C, H * W)

    def take(yy, xx):
        return torch.gather(flat, 2, (yy * W + xx)[:, None, :].expand(B, C, -1)).clone()

    ia, ib, ic, idd = take(y0, x0), take(y1, x0), take(y0, x1), take(y1, x1)
    if is_grid:
        a0 = ia[:, 0]
        for other in (ib, ic, idd):
            other[:, 0] = a0 + pymod((other[:, 0] - a0) + W / 2, W) - W / 2
    wa = ((1 - wx) * (1 - wy))[:, None]
    wb = ((1 - wx) * wy)[:, None]
    wc = (wx * (1 - wy))[:, None]
    wd = (wx * wy)[:, None]
    out = wa * ia + wb * ib + wc * ic + wd * idd
    return out.reshape(B, C, *shp)


# --------------------------------------------------------------------------------------
# ERP geometry
# --------------------------------------------------------------------------------------
def rotation_x(theta: float) -> Tensor:
    """R = Rz(0) Ry(0) Rx(theta) in fp32 (core/utils/projection_prim_ortho.py:23-48)."""
    c = torch.cos(torch.tensor(theta)).float()
    s = torch.sin(torch.tensor(theta)).float()
    one, zero = torch.tensor(1.0), torch.tensor(0.0)
    rx = torch.stack([torch.stack([one, zero, zero]),
                      torch.stack([zero, c, -s]),
                      torch.stack([zero, s, c])])
    return torch.eye(3) @ torch.eye(3) @ torch.eye(3) @ rx


def _nudge(t: Tensor, eps: float = 1e-6) -> Tensor:
    """core/utils/projection_prim_ortho.py:69-74."""
    return t + torch.sign(t) * (t.abs() < eps) * eps


def sample_grid(H: int, W: int, R: Tensor) -> Tensor:
    """ERP pixel -> sphere -> rotate -> ERP pixel; returns [2,H,W] = (m', n')
    (core/utils/projection_prim_ortho.py:432-443, :10-20, :397-411, :77-89, :247-261,
    :51-66, :413-429)."""
    m = torch.arange(W).view(1, W).repeat(H, 1).float()
    n = torch.arange(H).view(H, 1).repeat(1, W).float()
    theta = ((m + 0.5) / W - 0.5) * 2 * math.pi
    phi = (0.5 - (n + 0.5) / H) * math.pi
    v = torch.stack([torch.cos(phi) * torch.cos(theta),
                     torch.cos(phi) * torch.sin(theta),
                     torch.sin(phi)], dim=-1)                        # [H,W,3]
    vr = torch.matmul(R.view(1, 1, 3, 3), v.unsqueeze(-1)).squeeze(-1)
    phi2 = torch.arcsin(vr[..., 2])
    theta2 = torch.atan2(_nudge(vr[..., 1]), _nudge(vr[..., 0]))
    m2 = (theta2 / (2 * math.pi) + 0.5) * W - 0.5
    n2 = (0.5 - phi2 / math.pi) * H - 0.5
    return torch.stack([m2, n2], dim=0)


def img_rotate(img: Tensor, grid: Tensor) -> Tensor:
    """core/utils/projection_prim_ortho.py:507-514.  grid [2,H,W] shared by the batch."""
    B = img.shape[0]
    gx = grid[0][None].expand(B, -1, -1)
    gy = grid[1][None].expand(B, -1, -1)
    return cycle_bilinear_sampler(img, gx, gy)


def flo_rotate(flow: Tensor, g_w2c: Tensor, g_c2w: Tensor) -> Tensor:
    """Flow field of view X expressed in the other view
    (core/utils/projection_prim_ortho.py:531-546, :200-218, :234-244)."""
    B, _, H, W = flow.shape
    xs = torch.arange(W).view(1, 1, W).expand(B, H, W).float()
    ys = torch.arange(H).view(1, H, 1).expand(B, H, W).float()
    ex = pymod(xs + flow[:, 0] + 0.5, W) - 0.5
    ey = torch.clamp(ys + flow[:, 1], min=-0.5, max=H - 0.5)
    gw = g_w2c[None].expand(B, -1, -1, -1)
    end_c = wrapgather(gw, ex, ey, True)
    fc = end_c - gw
    fc = torch.stack([pymod(fc[:, 0] + W / 2, W) - W / 2, fc[:, 1]], dim=1)
    gc = g_c2w[None].expand(B, -1, -1, -1)
    return wrapgather(fc, gc[:, 0], gc[:, 1], False)


def coords_grid(B: int, H: int, W: int) -> Tensor:
    """core/utils/utils.py:98-101."""
    xs = torch.arange(W).view(1, 1, W).expand(B, H, W).float()
    ys = torch.arange(H).view(1, H, 1).expand(B, H, W).float()
    return torch.stack([xs, ys], dim=1)


# --------------------------------------------------------------------------------------
# correlation volume, pyramid, lookups
# --------------------------------------------------------------------------------------
def corr_volume(f1: Tensor, f2: Tensor) -> Tensor:
    """V[b,n1,n2] = <f1[b,:,n1], f2[b,:,n2]> / sqrt(C)   (core/prior_raft.py:69-75)."""
    B, C, H, W = f1.shape
    v = torch.matmul(f1.reshape(B, C, H * W).transpose(1, 2), f2.reshape(B, C, H * W))
    return v.reshape(B, H, W, H, W) / torch.sqrt(torch.tensor(float(C)))


def build_pyramid(vol: Tensor, levels: int = CORR_LEVELS) -> List[Tensor]:
    """[B*N,1,H>>i,W>>i] by successive 2x2 means (core/corr.py:99-111)."""
    B, H, W = vol.shape[:3]
    cur = vol.reshape(B * H * W, 1, H, W)
    out = [cur]
    for _ in range(levels - 1):
        h, w = cur.shape[-2] // 2, cur.shape[-1] // 2
        c = cur[..., : 2 * h, : 2 * w]
        cur = (c[..., 0::2, 0::2] + c[..., 0::2, 1::2] + c[..., 1::2, 0::2] + c[..., 1::2, 1::2]) * 0.25
        out.append(cur)
    return out


def dccl_lookup(coords: Tensor, pyr_own: Sequence[Tensor], pyr_other: Sequence[Tensor],
                g_w2c: Tensor, g_back: Tensor, radius: int = CORR_RADIUS) -> Tuple[Tensor, Tensor]:
    """Own-view + cross-view (2r+1)^2 x levels lookup (core/corr.py:113-144).

    Channel = level*81 + a*9 + b with x += d[a], y += d[b]  (the *slow* window axis
    offsets x: ``meshgrid(dy, dx)`` stacked as (x,y), core/corr.py:120-126).
    Returns (own [B,324,H,W], cross [B,324,H,W]).
    """
    B, _, H, W = coords.shape
    N = H * W
    k = 2 * radius + 1
    d = torch.linspace(-radius, radius, k)
    off_x = d.view(k, 1).expand(k, k).reshape(1, k * k)   # slow axis a -> x
    off_y = d.view(1, k).expand(k, k).reshape(1, k * k)   # fast axis b -> y
    cx0 = coords[:, 0].reshape(B * N, 1)
    cy0 = coords[:, 1].reshape(B * N, 1)
    gw = g_w2c[None].expand(B, -1, -1, -1)
    own_l, cross_l = [], []
    for i, (po, px) in enumerate(zip(pyr_own, pyr_other)):
        cx = cx0 / 2 ** i + off_x                           # [B*N,81]
        cy = cy0 / 2 ** i + off_y
        own = cycle_bilinear_sampler(po, cx, cy)            # [B*N,1,81]
        own_l.append(own.reshape(B, H, W, k * k))
        # level-i coordinates index the LEVEL-0 sample grid (core/corr.py:132-133)
        g = cycle_bilinear_sampler(gw, cx.reshape(B, N * k * k), cy.reshape(B, N * k * k))
        gx = g[:, 0].reshape(B * N, k * k)
        gy = g[:, 1].reshape(B * N, k * k)
        raw = cycle_bilinear_sampler(px, gx, gy)            # row n of the OTHER volume
        raw = raw.reshape(B, H, W, k * k).permute(0, 3, 1, 2)
        cross = img_rotate(raw, g_back)                     # core/corr.py:138
        cross_l.append(cross.permute(0, 2, 3, 1))
    own = torch.cat(own_l, dim=-1).permute(0, 3, 1, 2).contiguous()
    cross = torch.cat(cross_l, dim=-1).permute(0, 3, 1, 2).contiguous()
    return own, cross


def warp_groupwise_corr(f1: Tensor, f2: Tensor, coords: Tensor, groups: int = 4) -> Tensor:
    """mean over channel groups of f1 * warp(f2, coords)
    (core/prior_raft.py:173-174, :77-83)."""
    B, C, H, W = f1.shape
    warped = cycle_bilinear_sampler(f2, coords[:, 0], coords[:, 1])
    return (f1 * warped).view(B, groups, C // groups, H, W).mean(dim=2)


# --------------------------------------------------------------------------------------
# update blocks
# --------------------------------------------------------------------------------------
def _conv(p: Mapping[str, Tensor], name: str, x: Tensor, pad) -> Tensor:
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], padding=pad)


def motion_encoder_A(p, pre, flow_a, corr_a, flaw_a, flow_ba, flaw_ba) -> Tensor:
    """BasicMultiMotionEncoder (core/update.py:162-201)."""
    cor = F.relu(_conv(p, pre + "convc1_A", corr_a, 0))
    cor = F.relu(_conv(p, pre + "convc2_A", cor, 1))
    fa = F.relu(_conv(p, pre + "convf1_A", flow_a, 3))
    fa = F.relu(_conv(p, pre + "convf2_A", fa, 1))
    fb = F.relu(_conv(p, pre + "convf1_B", flow_ba, 3))
    fb = F.relu(_conv(p, pre + "convf2_B", fb, 1))
    conf = F.relu(_conv(p, pre + "conv_conf1", torch.cat([flaw_a, flaw_ba], 1), 1))
    conf = F.relu(_conv(p, pre + "conv_conf2", conf, 1))
    out = F.relu(_conv(p, pre + "conv_A", torch.cat([cor, fa, fb, conf], 1), 1))
    return torch.cat([out, flow_a, flow_ba], 1)


def motion_encoder_B(p, pre, flow, corr) -> Tensor:
    """BasicMotionEncoder (core/update.py:81-99)."""
    cor = F.relu(_conv(p, pre + "convc1", corr, 0))
    cor = F.relu(_conv(p, pre + "convc2", cor, 1))
    fl = F.relu(_conv(p, pre + "convf1", flow, 3))
    fl = F.relu(_conv(p, pre + "convf2", fl, 1))
    out = F.relu(_conv(p, pre + "conv", torch.cat([cor, fl], 1), 1))
    return torch.cat([out, flow], 1)


def sepconv_gru(p, pre, h: Tensor, x: Tensor) -> Tensor:
    """SepConvGRU (core/update.py:35-60): (1x5) pass then (5x1) pass."""
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(p, pre + "convz" + tag, hx, pad))
        r = torch.sigmoid(_conv(p, pre + "convr" + tag, hx, pad))
        q = torch.tanh(_conv(p, pre + "convq" + tag, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    return h


def heads(p, pre, net: Tensor) -> Tuple[Tensor, Tensor]:
    """FlowHead + mask head (core/update.py:6-14, :124-127, :133-135)."""
    delta = _conv(p, pre + "flow_head.conv2", F.relu(_conv(p, pre + "flow_head.conv1", net, 1)), 1)
    mask = 0.25 * _conv(p, pre + "mask.2", F.relu(_conv(p, pre + "mask.0", net, 1)), 0)
    return mask, delta


def update_A(p, net, inp, flow_a, corr_a, flaw_a, flow_ba, flaw_ba):
    """BasicMultiUpdateBlock = ODDC (core/update.py:139-159)."""
    mf = motion_encoder_A(p, "ODDC.encoder.", flow_a, corr_a, flaw_a, flow_ba, flaw_ba)
    net = sepconv_gru(p, "ODDC.gru.", net, torch.cat([inp, mf], 1))
    mask, delta = heads(p, "ODDC.", net)
    return net, mask, delta


def update_B(p, net, inp, corr, flow):
    """BasicUpdateBlock (core/update.py:117-136)."""
    mf = motion_encoder_B(p, "update_block.encoder.", flow, corr)
    net = sepconv_gru(p, "update_block.gru.", net, torch.cat([inp, mf], 1))
    mask, delta = heads(p, "update_block.", net)
    return net, mask, delta


def upsample_flow(flow: Tensor, mask: Tensor) -> Tensor:
    """Convex 8x upsampling (core/prior_raft.py:58-67).  mask channel = 64k+8i+j,
    k = 3ky+kx; neighbours are ZERO padded (F.unfold), not cyclic."""
    B, _, H, W = flow.shape
    m = torch.softmax(mask.view(B, 1, 9, 8, 8, H, W), dim=2)
    fp = F.pad(8 * flow, (1, 1, 1, 1))
    nb = torch.stack([fp[:, :, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], dim=2)
    up = (m * nb.view(B, 2, 9, 1, 1, H, W)).sum(dim=2)          # [B,2,8,8,H,W]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H, 8 * W)


# --------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------
def _norm(p, name: str, x: Tensor, kind: str) -> Tensor:
    if kind == "instance":   # nn.InstanceNorm2d: eps 1e-5, biased var, no affine/stats
        mu = x.mean(dim=(2, 3), keepdim=True)
        var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
        return (x - mu) / torch.sqrt(var + 1e-5)
    # nn.BatchNorm2d, always eval (core/prior_raft.py:43-48, train_flow.py:107-108)
    w, b = p[name + ".weight"], p[name + ".bias"]
    rm, rv = p[name + ".running_mean"], p[name + ".running_var"]
    scale = w / torch.sqrt(rv + 1e-5)
    return (x - rm.view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def _resblock(p, pre: str, x: Tensor, kind: str, stride: int) -> Tensor:
    """ResidualBlock (core/extractor.py:8-47)."""
    y = F.conv2d(x, p[pre + "conv1.weight"], p[pre + "conv1.bias"], stride=stride, padding=1)
    y = F.relu(_norm(p, pre + "norm1", y, kind))
    y = F.conv2d(y, p[pre + "conv2.weight"], p[pre + "conv2.bias"], padding=1)
    y = F.relu(_norm(p, pre + "norm2", y, kind))
    if stride != 1:
        x = F.conv2d(x, p[pre + "downsample.0.weight"], p[pre + "downsample.0.bias"], stride=stride)
        x = _norm(p, pre + "downsample.1", x, kind)   # == norm3 (same module object)
    return F.relu(x + y)


def encoder(p, pre: str, x: Tensor, kind: str) -> Tensor:
    """BasicEncoder (core/extractor.py:98-158); eval mode (dropout off)."""
    x = F.conv2d(x, p[pre + "conv1.weight"], p[pre + "conv1.bias"], stride=2, padding=3)
    x = F.relu(_norm(p, pre + "norm1", x, kind))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _resblock(p, f"{pre}{layer}.0.", x, kind, stride)
        x = _resblock(p, f"{pre}{layer}.1.", x, kind, 1)
    return F.conv2d(x, p[pre + "conv2.weight"], p[pre + "conv2.bias"])


# --------------------------------------------------------------------------------------
# full forward
# --------------------------------------------------------------------------------------
def grids_for(H: int, W: int) -> Dict[str, Tensor]:
    """The 4 distinct sample grids of core/prior_raft.py:115-125
    (grid(R_A2B^T) == grid(R_B2A) and vice versa, bit-exactly)."""
    r_a2b = rotation_x(-math.pi / 2)
    r_b2a = rotation_x(math.pi / 2)
    return {
        "a2b": sample_grid(H, W, r_a2b), "a2b_8": sample_grid(H // 8, W // 8, r_a2b),
        "b2a": sample_grid(H, W, r_b2a), "b2a_8": sample_grid(H // 8, W // 8, r_b2a),
        "a2b_w2c_8": sample_grid(H // 8, W // 8, r_a2b.T),
        "b2a_w2c_8": sample_grid(H // 8, W // 8, r_b2a.T),
    }


def forward_with_grad(p: Mapping[str, Tensor], image1: Tensor, image2: Tensor, iters: int = 12,
                      init_flow: Optional[Tensor] = None, test_mode: bool = False,
                      trace: Optional[dict] = None):
    """PriOr_RAFT.forward (core/prior_raft.py:107-215).  ``trace`` (if given) collects
    per-stage tensors for per-kernel parity tests."""
    image1 = 2 * (image1 / 255.0) - 1.0
    image2 = 2 * (image2 / 255.0) - 1.0
    B, _, H, W = image1.shape
    g = grids_for(H, W)
    rot = img_rotate(torch.cat([image1, image2], 1), g["a2b"])
    image1_b, image2_b = rot[:, :3].contiguous(), rot[:, 3:].contiguous()

    cn = encoder(p, "cnet.", torch.cat([image1, image1_b], 0), "batch")
    net_a, inp_a = torch.tanh(cn[:B, :HIDDEN]), torch.relu(cn[:B, HIDDEN:])
    net_b, inp_b = torch.tanh(cn[B:, :HIDDEN]), torch.relu(cn[B:, HIDDEN:])
    fm = encoder(p, "fnet.", torch.cat([image1, image2, image1_b, image2_b], 0), "instance")
    f1a, f2a, f1b, f2b = fm[:B], fm[B:2 * B], fm[2 * B:3 * B], fm[3 * B:]

    pyr_a = build_pyramid(corr_volume(f1a, f2a))
    pyr_b = build_pyramid(corr_volume(f1b, f2b))

    H8, W8 = H // 8, W // 8
    c0 = coords_grid(B, H8, W8)
    c1a, c1b = c0.clone(), c0.clone()
    if init_flow is not None:
        c1a = c1a + init_flow
        c1b = c1b + flo_rotate(init_flow, g["a2b_w2c_8"], g["a2b_8"])
    if trace is not None:
        trace.update(image1_b=image1_b, image2_b=image2_b, net_a=net_a, inp_a=inp_a, net_b=net_b,
                     inp_b=inp_b, f1a=f1a, f2a=f2a, f1b=f1b, f2b=f2b, grids=g, iters=[])

    preds_a, preds_b = [], []
    for _ in range(iters):
        c1a, c1b = c1a.detach(), c1b.detach()      # core/prior_raft.py:171,176 (no gradient through the coordinates)
        flow_a = c1a - c0
        flaw_a = warp_groupwise_corr(f1a, f2a, c1a)
        flow_b = c1b - c0
        flow_ba = flo_rotate(flow_b, g["b2a_w2c_8"], g["b2a_8"])
        flaw_ba = warp_groupwise_corr(f1a, f2a, c0 + flow_ba)
        own_a, cross_a = dccl_lookup(c1a, pyr_a, pyr_b, g["a2b_w2c_8"], g["b2a_8"])
        own_b, cross_b = dccl_lookup(c1b, pyr_b, pyr_a, g["b2a_w2c_8"], g["a2b_8"])
        corr_a = own_a + cross_a
        corr_b = own_b + cross_b
        net_a, mask_a, d_a = update_A(p, net_a, inp_a, flow_a, corr_a, flaw_a, flow_ba, flaw_ba)
        net_b, mask_b, d_b = update_B(p, net_b, inp_b, corr_b, flow_b)
        c1a = c1a + d_a
        c1b = c1b + d_b
        up_a = upsample_flow(c1a - c0, mask_a)
        up_b = upsample_flow(c1b - c0, mask_b)
        preds_a.append(up_a)
        preds_b.append(up_b)
        if trace is not None:
            trace["iters"].append(dict(flow_a=flow_a, flaw_a=flaw_a, flow_b=flow_b, flow_ba=flow_ba,
                                       flaw_ba=flaw_ba, corr_a=corr_a, corr_b=corr_b, net_a=net_a,
                                       net_b=net_b, mask_a=mask_a, mask_b=mask_b, d_a=d_a, d_b=d_b,
                                       up_a=up_a, up_b=up_b))
    if test_mode:
        return preds_a[-1]
    return preds_a, preds_b


@torch.no_grad()
def forward(p: Mapping[str, Tensor], image1: Tensor, image2: Tensor, iters: int = 12,
            init_flow: Optional[Tensor] = None, test_mode: bool = False, trace: Optional[dict] = None):
    """Inference form of `forward_with_grad` (no autograd graph).  `forward_with_grad` on parameters with
    requires_grad reproduces the reference's training-step gradients (tests/golden/train_step.npz)."""
    return forward_with_grad(p, image1, image2, iters, init_flow, test_mode, trace)


def epe(a: Tensor, b: Tensor) -> Tensor:
    """Per-pixel end-point error between two flow fields [B,2,H,W]."""
    return torch.sqrt(((a - b) ** 2).sum(dim=1))


# --------------------------------------------------------------------------------------
# evaluation counterpart: SEPE, region masks, region metrics
# --------------------------------------------------------------------------------------
def _endpoint_sph(flow: Tensor) -> Tuple[Tensor, Tensor]:
    """(theta, phi) [B,H,W] of the end points of `flow` [B,2,H,W]: x wraps, y clamps
    (core/utils/projection_prim_ortho.py:200-218 `flow2endpoint`, :397-411 `plane2spherical`)."""
    B, _, H, W = flow.shape
    g = coords_grid(B, H, W)
    e0 = pymod(g[:, 0] + flow[:, 0] + 0.5, W) - 0.5
    e1 = torch.clamp(g[:, 1] + flow[:, 1], min=-0.5, max=H - 0.5)
    theta = ((e0 + 0.5) / W - 0.5) * 2 * math.pi
    phi = (0.5 - (e1 + 0.5) / H) * math.pi
    return theta, phi


def _haversine(x: Tensor) -> Tensor:
    """core/utils/spherical.py:73-77."""
    return torch.square(torch.sin(x / 2))


def great_circle_distance(pre: Tensor, gt: Tensor) -> Tensor:
    """SEPE: haversine great-circle distance [B,H,W] between the end points of two flow fields on
    the unit sphere (core/utils/spherical.py:20-53, method='Haversine', R=1; :80-84 inverse)."""
    tp, pp = _endpoint_sph(pre)
    tg, pg = _endpoint_sph(gt)
    hav = _haversine(pg - pp) + torch.cos(pp) * torch.cos(pg) * _haversine(tg - tp)
    return 2 * torch.arcsin(torch.sqrt(hav))


def great_circle_distance_cosine(pre: Tensor, gt: Tensor) -> Tensor:
    """The method='Cosine' form of the same distance (core/utils/spherical.py:40-46): arccos of the spherical law of cosines."""
    tp, pp = _endpoint_sph(pre)
    tg, pg = _endpoint_sph(gt)
    cos_alpha = torch.sin(pp) * torch.sin(pg) + torch.cos(pp) * torch.cos(pg) * torch.cos(tg - tp)
    return torch.arccos(cos_alpha)


def spherical_mask(H: int, W: int) -> Tensor:
    """cos(latitude) weights normalised to sum 1 (core/utils/spherical.py:11-17)."""
    n = torch.arange(0, H).view(-1, 1).repeat(1, W)
    phi = (0.5 - (n + 0.5) / H) * math.pi
    m = torch.cos(phi)
    return m / m.sum()


def generate_polemask(H: int, W: int, delta_phi: float = math.pi / 2) -> Tuple[Tensor, Tensor]:
    """(pole_mask_A, pole_mask_B) as long [1,H,W] (core/utils/polemask.py:7-26): A = rows outside
    the +-delta_phi/2 band; B = A seen from view B (img_A2B = img_rotate with Rx(-pi/2),
    core/utils/projection_prim_ortho.py:517-519), binarised at 0.5."""
    import numpy as np
    phi2n = lambda phi: (0.5 - phi / np.pi) * H - 0.5                       # noqa: E731  (:317-327)
    min_n = int(np.round(phi2n(delta_phi / 2)))
    max_n = int(np.round(phi2n(-delta_phi / 2)))
    center = torch.zeros((1, H, W))
    center[:, min_n:max_n, :] = 1
    pole_a = 1 - center
    pole_b = img_rotate(pole_a.unsqueeze(1), sample_grid(H, W, rotation_x(-math.pi / 2))).squeeze(1)
    pole_b = torch.where(pole_b < 0.5, torch.zeros_like(pole_b), pole_b)
    pole_b = torch.where(pole_b > 0, torch.ones_like(pole_b), pole_b)
    return pole_a.long(), pole_b.long()


def region_masks(H: int, W: int) -> Dict[str, Tensor]:
    """Boolean [H*W] masks of evaluate.py:246-254: All / Equator / Poles / Center."""
    pole, center = generate_polemask(H, W)
    return {"All": torch.ones(H * W, dtype=torch.bool),
            "Equator": (1 - pole).view(-1) >= 0.5,
            "Poles": pole.view(-1) >= 0.5,
            "Center": center.view(-1) >= 0.5}


def region_metrics(preds: Sequence[Tensor], gts: Sequence[Tensor]) -> Dict[str, Dict[str, float]]:
    """evaluate.py:234-282 on a list of (flow [2,H,W], flow_gt [2,H,W]) pairs: per region the mean EPE
    and mean SEPE over every masked pixel of every sample, plus the cos-latitude weighted SEPE
    `sd_uni` of :208-213 (per-image weighted mean, then mean over images)."""
    H, W = preds[0].shape[-2:]
    masks = region_masks(H, W)
    uni = spherical_mask(H, W).view(-1)
    out = {}
    for name, mk in masks.items():
        e_all, s_all, u_all = [], [], []
        for p, g in zip(preds, gts):
            e = epe(p[None], g[None])[0].view(-1)
            s = great_circle_distance(p[None], g[None])[0].view(-1)
            e_all.append(e[mk]); s_all.append(s[mk])
            u_all.append(float(((s * uni)[mk] / uni[mk].sum()).sum()))
        out[name] = {"epe": float(torch.cat(e_all).double().mean()),
                     "sd": float(torch.cat(s_all).double().mean()),
                     "sd_uni": float(sum(u_all) / len(u_all))}
    return out


# --------------------------------------------------------------------------------------
# training-step counterpart: loss, schedule, optimiser, clipping
# --------------------------------------------------------------------------------------
MAX_FLOW = 400.0     # train_flow.py:46


def uniform_loss(flow_preds: Sequence[Tensor], flow_gt: Tensor, valid: Tensor, gamma: float = 0.8,
                 max_flow: float = MAX_FLOW):
    """train_flow.py:55-79: (loss, metrics, gradients of the loss w.r.t. every prediction).
    loss = sum_i gamma^(n-i-1) * sum(valid * cos-latitude mask * |pred_i - gt|_1); metrics over the
    valid pixels of the last prediction.  Gradients are written out by hand (abs' = sign)."""
    B, _, H, W = flow_gt.shape
    uni = spherical_mask(H, W)[None]
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    ok = (valid >= 0.5) & (mag < max_flow)
    n = len(flow_preds)
    loss = torch.zeros((), dtype=torch.float64)
    grads = []
    for i, p in enumerate(flow_preds):
        wgt = gamma ** (n - i - 1)
        m = ok * uni
        loss = loss + wgt * torch.sum((m * torch.sum((p - flow_gt).abs(), dim=1)).double())
        grads.append(wgt * m[:, None] * torch.sign(p - flow_gt))
    e = torch.sum((flow_preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[ok.view(-1)]
    metrics = {"epe": float(e.double().mean()), "1px": float((e < 1).double().mean()),
               "3px": float((e < 3).double().mean()), "5px": float((e < 5).double().mean())}
    return float(loss), metrics, grads


def one_cycle_lr(step: int, max_lr: float, num_steps: int, pct_start: float = 0.05, div_factor: float = 25.0,
                 final_div_factor: float = 1e4) -> float:
    """Learning rate after `step` scheduler steps of OneCycleLR(max_lr, num_steps + 100, pct_start=0.05,
    cycle_momentum=False, anneal_strategy='linear') (train_flow.py:89-90)."""
    total = num_steps + 100
    initial, min_lr = max_lr / div_factor, max_lr / div_factor / final_div_factor
    end1 = float(pct_start * total) - 1
    if step <= end1:
        return (max_lr - initial) * (step / end1) + initial
    pct = (step - end1) / ((total - 1) - end1)
    return (min_lr - max_lr) * pct + max_lr


def clip_coef(total_norm: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (train_flow.py:137): grads *= min(1, max_norm / (norm + 1e-6))."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, step: int, wd: float, eps: float = 1e-8,
               b1: float = 0.9, b2: float = 0.999):
    """One torch.optim.AdamW update (single-tensor form), fp32 tensors, python-float scalars."""
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - b1)
    v = v * b2 + (g * g) * (1 - b2)
    denom = v.sqrt() / math.sqrt(1 - b2 ** step) + eps
    p = p + (m / denom) * (-(lr / (1 - b1 ** step)))
    return p, m, v
