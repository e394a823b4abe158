"""On-disk formats: Middlebury .flo reader / writer against hand-built bytes and
the bytes the reference's own writeFlow produced (tests/golden/flo_io.npz), and InputPadder."""
import os
import struct

import numpy as np
import pytest
import torch

import golden_cases as gc
from prior_flow_amd import frame_utils as fu


def test_flo_known_bytes(tmp_path):
    uv = np.array([[[1.5, -2.0], [0.0, 3.25], [7.0, 8.0]], [[-1.0, -1.0], [100.5, 0.125], [2.0, 4.0]]], np.float32)
    want = struct.pack("<f", 202021.25) + struct.pack("<ii", 3, 2) + uv.astype("<f4").tobytes()
    assert want[:4] == b"PIEH"                                   # the Middlebury tag
    fn = str(tmp_path / "k.flo")
    fu.writeFlow(fn, uv)
    assert open(fn, "rb").read() == want
    fu.writeFlow(fn, uv[:, :, 0], uv[:, :, 1])                   # separate u, v planes
    assert open(fn, "rb").read() == want
    back = fu.readFlow(fn)
    assert back.dtype == np.float32 and back.shape == (2, 3, 2) and np.array_equal(back, uv)


def test_flo_reference_bytes(tmp_path):
    g = gc.load("flo_io")
    fn = str(tmp_path / "r.flo")
    g["bytes"].tofile(fn)
    assert np.array_equal(fu.readFlow(fn), g["read"])
    fu.writeFlow(fn, g["read"])
    assert np.array_equal(np.fromfile(fn, dtype=np.uint8), g["bytes"])


def test_flo_bad_magic_and_shapes(tmp_path, capsys):
    fn = str(tmp_path / "bad.flo")
    open(fn, "wb").write(struct.pack("<f", 1.0) + struct.pack("<ii", 1, 1) + b"\0" * 8)
    assert fu.readFlow(fn) is None
    assert "Magic number incorrect" in capsys.readouterr().out
    with pytest.raises(AssertionError):
        fu.writeFlow(fn, np.zeros((2, 3, 3), np.float32))
    with pytest.raises(AssertionError):
        fu.writeFlow(fn, np.zeros((2, 3), np.float32), np.zeros((3, 2), np.float32))
    fu.writeFlow(fn, np.zeros((0, 4, 2), np.float32))            # empty flow: header only
    assert os.path.getsize(fn) == 12 and fu.readFlow(fn).shape == (0, 4, 2)


def test_input_padder():
    from prior_flow_amd.evaluate import InputPadder
    x = torch.arange(2 * 3 * 13 * 21, dtype=torch.float32).view(2, 3, 13, 21)
    p = InputPadder(x.shape)
    assert p._pad == [1, 2, 1, 2]                                # 21 -> 24, 13 -> 16, split symmetric ("sintel")
    y, = p.pad(x)
    assert y.shape == (2, 3, 16, 24)
    assert torch.equal(y[:, :, 0, 1:22], x[:, :, 0])             # replicate padding
    assert torch.equal(p.unpad(y), x)
    assert InputPadder((1, 3, 13, 21), mode="kitti")._pad == [1, 2, 0, 3]
    assert InputPadder((1, 3, 512, 1024))._pad == [0, 0, 0, 0]   # the benchmark sizes are not padded


def test_load_things_ckpt_matches_the_reference(tmp_path, capsys):
    """load_things_ckpt (core/prior_raft.py:85-104): `module.` prefix stripped, same-name / same-shape entries
    loaded, ODDC.{gru,flow_head,mask}.* filled from update_block.*, everything else skipped with the reference's
    message.  tests/golden/things_ckpt.npz records what the reference did with the synthetic RAFT-things
    checkpoint of oracle/gen_golden_ckpt.py (key -> checkpoint entry it ended up holding, skip lines)."""
    import argparse
    from gen_golden_ckpt import things_checkpoint          # oracle/ is on sys.path (tests/conftest.py)
    from prior_flow_amd.prior_raft import PriOr_RAFT
    g = gc.load("things_ckpt")
    torch.manual_seed(7)
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    assert list(before) == [str(k) for k in g["keys"]]
    ckpt = things_checkpoint({k: tuple(v.shape) for k, v in before.items()})
    path = str(tmp_path / "raft-things.pth")
    torch.save(ckpt, path)
    capsys.readouterr()
    model.load_things_ckpt(path)
    printed = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert printed == [str(s) for s in g["skipped"]]
    after = model.state_dict()
    n_alias = 0
    for k, src in zip(g["keys"], g["source"]):
        k, src = str(k), str(src)
        want = ckpt[src] if src else before[k]
        assert torch.equal(after[k], want), (k, src)
        n_alias += bool(src) and src != "module." + k
    assert n_alias >= 20                                         # the ODDC <- update_block remap really happened
