"""GPU tests of the evaluation counterpart: masks, SEPE, region metrics and the
validate loop through libpriorflow_hip.so, against the reference-generated goldens and the oracle."""
import argparse

import numpy as np
import pytest
import torch

import golden_cases as gc
import priorflow_oracle as po

pytestmark = pytest.mark.gpu


def test_polemask_and_sepe_vs_reference():
    from prior_flow_amd import evaluate as ev
    g = gc.load("eval")
    for h, w in ((16, 32), (64, 128)):
        a, b = ev.generate_polemask(h, w)
        assert a.dtype == torch.long and a.shape == (1, h, w) and a.is_cuda
        assert np.array_equal(a.cpu().numpy().astype(np.uint8), g[f"pole_a_{h}x{w}"])
        assert np.array_equal(b.cpu().numpy().astype(np.uint8), g[f"pole_b_{h}x{w}"])
    pre, gt = gc.flows("eval/pre", 2), gc.flows("eval/gt", 2)
    sd = ev.calculate_great_circle_distance(pre.cuda(), gt.cuda())
    assert float((sd.cpu() - torch.from_numpy(g["sd_rand"])).abs().max()) < 2e-6
    kat = torch.zeros(1, 2, 64, 128, device="cuda")
    kat[:, 0] = 4.0
    sd = ev.calculate_great_circle_distance(kat, torch.zeros_like(kat))[0, :, 0].cpu()
    assert float((sd - torch.from_numpy(g["sd_kat"])).abs().max()) < 1e-6 and abs(float(sd[31]) - 0.19629) < 1e-5
    assert np.allclose(ev.spherical_mask(64, 128)[:, 0], g["uni_col"], rtol=0, atol=1e-9)
    # method='Cosine' (core/utils/spherical.py:40-46): the same distance where arccos is well conditioned
    sc = ev.calculate_great_circle_distance(kat, torch.zeros_like(kat), method="Cosine")[0, :, 0].cpu()
    ref_kat = torch.from_numpy(g["sd_kat"])
    far = ref_kat > 0.05                      # arccos loses 1e-7 / sin(d): rows close to the poles are compared loosely
    assert int(far.sum()) >= 32 and float((sc - ref_kat)[far].abs().max()) < 5e-6
    assert float((sc - ref_kat).abs().max()) < 1e-3


def test_region_evaluator_vs_reference_numbers():
    from gen_golden_eval import eval_samples
    from prior_flow_amd.evaluate import RegionEvaluator
    g = gc.load("eval")
    ev = RegionEvaluator(64, 128)
    for pr, gt in eval_samples():
        ev.update(pr.cuda(), gt.cuda())
    res = ev.results()
    for i, name in enumerate(("All", "Equator", "Poles", "Center")):
        for j, key in enumerate(("epe", "sd", "sd_uni")):
            want = g["regions"][i, j]
            assert abs(res[name][key] - want) <= 5e-6 * abs(want), (name, key, res[name][key], want)
    # batched update == one by one
    ev2 = RegionEvaluator(64, 128)
    s = eval_samples()
    ev2.update(torch.stack([p for p, _ in s]).cuda(), torch.stack([q for _, q in s]).cuda())
    for name in res:
        for key in res[name]:
            assert abs(ev2.results()[name][key] - res[name][key]) <= 1e-12 * max(1.0, abs(res[name][key]))


def test_validate_regions_with_the_model(capsys):
    """validate_MPF_regions' loop body with the drop-in model on synthetic ERP pairs whose size needs
    padding (126 x 250 -> 128 x 256); the metrics must equal the oracle's on the model's own flows."""
    from prior_flow_amd import evaluate as ev
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(gc.det_state_dict(state_dict_shapes()), strict=True)
    model = model.cuda().eval()
    data = []
    for i in range(2):
        i1, i2 = gc.synthetic_pair(1, 128, 256, seed=77 + i)
        gt = torch.stack([gc.uni(f"val/u{i}", (126, 250), -6, 6), gc.uni(f"val/v{i}", (126, 250), -3, 3)])
        data.append((i1[0, :, 1:127, 3:253].contiguous(), i2[0, :, 1:127, 3:253].contiguous(), gt, None))
    res = ev.validate_MPF_regions(model, iters=2, scene="EFT", dataset=data)
    out = capsys.readouterr().out
    assert "All-EFT: epe" in out and "Center-EFT" in out
    flows = []
    for i1, i2, gt, _ in data:
        pad = ev.InputPadder(i1[None].shape)
        a, b = pad.pad(i1[None].cuda(), i2[None].cuda())
        flows.append(pad.unpad(model(a.contiguous(), b.contiguous(), iters=2, test_mode=True)[0]).cpu())
    want = po.region_metrics(flows, [d[2] for d in data])
    for name in want:
        for key in ("epe", "sd", "sd_uni"):
            assert abs(res[name][key] - want[name][key]) <= 1e-5 * abs(want[name][key]) + 1e-7, (name, key)
    with pytest.raises(FileNotFoundError):
        ev.validate_MPF_regions(model)


def test_plain_validate_loops_equal_the_references_arithmetic(capsys):
    """validate_MPF / validate_FlowScape (evaluate.py:337-397: the in-training validation call of train_flow.py:187-194): EPE over
    all pixels of all samples and SEPE as the mean of per-sample means, on the model's own flows, against the reference's
    arithmetic spelled out in torch (sqrt of the summed squares; the golden-pinned great-circle distance); the model comes back in
    the mode it was given in."""
    from prior_flow_amd import evaluate as ev
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(gc.det_state_dict(state_dict_shapes()), strict=True)
    model = model.cuda().train()
    model.freeze_bn()                           # train_flow.py:107-108: every BatchNorm layer in eval mode inside a training model
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert bns and not any(m.training for m in bns)
    data = []
    for i in range(3):
        i1, i2 = gc.synthetic_pair(1, 128, 256, seed=177 + i)
        gt = torch.stack([gc.uni(f"val2/u{i}", (126, 250), -6, 6), gc.uni(f"val2/v{i}", (126, 250), -3, 3)])
        data.append((i1[0, :, 1:127, 3:253].contiguous(), i2[0, :, 1:127, 3:253].contiguous(), gt, None))
    res = ev.validate_MPF(model, iters=2, scene="EFT", dataset=data)
    assert model.training and model.fnet.training and model.update_block.training
    assert not any(m.training for m in bns), "validate() put frozen BatchNorm layers back into training mode (ADVICE r5)"
    assert "Validation (EFT) EPE:" in capsys.readouterr().out
    model.eval()
    epes, sds = [], []
    with torch.no_grad():
        for i1, i2, gt, _ in data:
            pad = ev.InputPadder(i1[None].shape)
            a, b = pad.pad(i1[None].cuda(), i2[None].cuda())
            flow = pad.unpad(model(a.contiguous(), b.contiguous(), iters=2, test_mode=True)[0]).cpu()
            epes.append(torch.sum((flow - gt) ** 2, dim=0).sqrt().view(-1).numpy())
            sds.append(float(ev.calculate_great_circle_distance(flow[None].cuda(), gt[None].cuda())[0].mean()))
    want_epe, want_sd = float(np.mean(np.concatenate(epes))), float(np.mean(np.array(sds)))
    assert abs(res["EFT-epe"] - want_epe) <= 1e-5 * want_epe and abs(res["EFT-SEPE"] - want_sd) <= 1e-5 * want_sd, (res, want_epe, want_sd)
    res2 = ev.validate_FlowScape(model, iters=2, scene="sunny", dataset=data[:1])
    assert set(res2) == {"FlowScape-sunny-epe", "FlowScape-sunny-SEPE"}
    with pytest.raises(FileNotFoundError):
        ev.validate_MPF(model)
