"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

Run only in the build container:  ``python oracle/gen_golden.py``.
The fixtures hold inputs/outputs only (data, not reference source).  Inputs come
from ``oracle/golden_cases.py`` so the tests can rebuild them bit-identically.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import load_reference, reference_model  # noqa: E402


def save(name, **arrays):
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
           for k, v in arrays.items()}
    path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
    np.savez(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KB")


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    ref = load_reference()
    proj, utils, cyc = ref.proj, ref.utils, ref.cyc
    h, w = gc.H8, gc.W8

    # ---- a8/a9: rotation matrices + sample grids ---------------------------------
    r_a2b = proj.generate_rotation_metrix(theta_list=[0., 0., -np.pi / 2])
    r_b2a = proj.generate_rotation_metrix(theta_list=[0., 0., np.pi / 2])
    grids = {}
    for tag, (gh, gw) in (("16x32", (16, 32)), ("64x128", (64, 128)), ("80x160", (80, 160))):
        grids[f"a2b_{tag}"] = proj.generate_samplegrid([1, 3, gh, gw], r_a2b)[0]
        grids[f"b2a_{tag}"] = proj.generate_samplegrid([1, 3, gh, gw], r_b2a)[0]
        grids[f"a2bT_{tag}"] = proj.generate_samplegrid([1, 3, gh, gw], r_a2b.T)[0]
        grids[f"b2aT_{tag}"] = proj.generate_samplegrid([1, 3, gh, gw], r_b2a.T)[0]
        # the transposed-rotation grids are bit-identical to the opposite direction's
        assert torch.equal(grids[f"a2bT_{tag}"], grids[f"b2a_{tag}"])
        assert torch.equal(grids[f"b2aT_{tag}"], grids[f"a2b_{tag}"])
    save("grids", r_a2b=r_a2b, r_b2a=r_b2a,
         **{k: v for k, v in grids.items() if "T_" not in k or k.endswith("16x32")})
    g_a2b = grids["a2b_16x32"][None]
    g_b2a = grids["b2a_16x32"][None]
    g_a2bT = grids["a2bT_16x32"][None]
    g_b2aT = grids["b2aT_16x32"][None]

    # ---- a5/a6: wrap-x zero-pad sampler -------------------------------------------
    img = gc.uni("sampler/img", (2, 3, h, w), -2, 2)
    co = gc.nasty_coords("sampler", B=2)
    out = utils.cycle_bilinear_sampler(img, co.permute(0, 2, 3, 1).clone())
    ones = utils.cycle_bilinear_sampler(torch.ones(1, 1, h, w),
                                        torch.tensor([[[[w - 0.5, 3.0], [5.0, -0.5], [w - 1.0, 2.0],
                                                        [-0.25, 2.0], [3.0, h - 0.5]]]]))
    save("sampler", out=out, ones=ones)

    # ---- a10: img_rotate (64x128, 6 channels) -------------------------------------
    im6 = gc.uni("img_rotate/img", (1, 6, 64, 128), -1, 1)
    save("img_rotate", out=proj.img_rotate(im6, sample_grid=grids["a2b_64x128"][None].clone()))

    # ---- a11/a12: flo_rotate --------------------------------------------------------
    fl = gc.flows("flo_rotate", B=2)
    out_b2a = proj.flo_rotate(fl.clone(), sample_grid_W2C=g_b2aT.repeat(2, 1, 1, 1).clone(),
                              sample_grid_C2W=g_b2a.repeat(2, 1, 1, 1).clone())
    out_a2b = proj.flo_rotate(fl.clone(), sample_grid_W2C=g_a2bT.repeat(2, 1, 1, 1).clone(),
                              sample_grid_C2W=g_a2b.repeat(2, 1, 1, 1).clone())
    save("flo_rotate", b2a=out_b2a, a2b=out_a2b)

    # ---- a2/a3: corr volume + pyramid ----------------------------------------------
    model = ref.prior_raft.PriOr_RAFT(ref.args())
    f1, f2 = gc.fmaps("corr", B=2)
    vol = model.corr(f1, f2)
    dccl = ref.corr.DCCL(radius=4)
    pyr = dccl.build_pyramid(vol)
    rows = np.array([0, 1, 31, 32, 100, 255, 256, 300, 511, 512 + 7, 512 + 480, 1023])
    save("corr_pyramid", rows=rows, **{f"l{i}": p[rows] for i, p in enumerate(pyr)},
         checksum=np.array([float(p.double().sum()) for p in pyr]),
         abssum=np.array([float(p.double().abs().sum()) for p in pyr]))

    # ---- a4: DCCL lookups (white-noise volumes; nasty coords) ----------------------
    va, vb = gc.volumes("dccl")
    pa, pb = dccl.build_pyramid(va), dccl.build_pyramid(vb)
    co = gc.nasty_coords("dccl")
    own, cross = dccl(co.clone(), pa, pb, g_a2bT.clone(), g_b2a.clone())
    own2, cross2 = dccl(co.clone(), pb, pa, g_b2aT.clone(), g_a2b.clone())
    # rows 0..7 hold all the structured special cases of nasty_coords (+2 random rows)
    save("dccl", own_a=own[:, :, :8], cross_a=cross[:, :, :8], corr_b=(own2 + cross2)[:, :, :8],
         cross_b_tail=cross2[:, :, 8:, ::4])

    # ---- K5: warp + groupwise corr --------------------------------------------------
    f1, f2 = gc.fmaps("gwc")
    co = gc.nasty_coords("gwc")
    warped = utils.cycle_bilinear_sampler(f2, co.permute(0, 2, 3, 1).clone())
    save("warp_gcorr", flaw=model.groupwise_corr(f1, warped, num_groups=4))

    # ---- a13-a16 update blocks, a17 upsample, a19 encoders, a1 full forward -------
    with reference_model(gc.det_state_dict) as m:
        ui = gc.update_inputs("upd")
        net, mask, delta = m.ODDC(ui["net"], ui["inp"], ui["flow_a"], ui["corr"], ui["flaw_a"],
                                  ui["flow_ba"], ui["flaw_ba"])
        mf = m.ODDC.encoder(ui["flow_a"], ui["corr"], ui["flaw_a"], ui["flow_ba"], ui["flaw_ba"])
        save("update_A", net=net, mask=mask[:, 3::8], delta=delta, motion=mf)
        save("gru", out=m.ODDC.gru(ui["net"], torch.cat([ui["inp"], mf], 1)))
        net, mask, delta = m.update_block(ui["net"], ui["inp"], ui["corr"], ui["flow_a"])
        mf = m.update_block.encoder(ui["flow_a"], ui["corr"])
        save("update_B", net=net, mask=mask[:, 3::8], delta=delta, motion=mf)

        fl8 = gc.uni("up/flow", (1, 2, h, w), -6, 6)
        mk = gc.uni("up/mask", (1, 576, h, w), -2, 2)
        save("upsample", out=m.upsample_flow(fl8, mk))

        im = gc.uni("enc/img", (2, 3, 128, 256), -1, 1)
        save("encoders", fnet=m.fnet(im)[:, ::4], cnet=m.cnet(im)[:, 1::4])

        # full forward, 128x256
        i1, i2 = gc.synthetic_pair(1, 128, 256)
        pa12, pb12 = m(i1, i2, iters=12)
        tm = m(i1, i2, iters=12, test_mode=True)
        assert torch.equal(tm, pa12[11])
        sub = lambda t: t[:, :, ::2, ::2]   # intermediate predictions: every other pixel
        save("forward_128x256_it12", a11=pa12[11], b11=pb12[11],
             **{f"a{i}": sub(pa12[i]) for i in (0, 2, 6)}, **{f"b{i}": sub(pb12[i]) for i in (0, 2, 6)})
        pa3, pb3 = m(i1, i2, iters=3)
        save("forward_128x256_it3", a2=sub(pa3[2]), b2=sub(pb3[2]))
        pa1, pb1 = m(i1, i2, iters=1)
        save("forward_128x256_it1", a0=sub(pa1[0]), b0=sub(pb1[0]))
        init = gc.uni("fwd/init_flow", (1, 2, 16, 32), -3, 3)
        save("forward_128x256_init", out=m(i1, i2, iters=3, init_flow=init, test_mode=True))
        # batch of 2 (different pairs) -> batch independence
        j1, j2 = gc.synthetic_pair(2, 128, 256, seed=77)
        save("forward_128x256_b2", out=m(j1, j2, iters=2, test_mode=True)[:, :, ::2, ::2])
        # BASELINE.json configs[0] (demo.py) and configs[4] (640x1280, iters=32): oracle/gen_golden_configs.py


if __name__ == "__main__":
    main()
