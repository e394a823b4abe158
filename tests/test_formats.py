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
