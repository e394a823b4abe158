"""Middlebury ``.flo`` reader / writer with the reference's names and behaviour
(PriOr-RAFT/core/utils/frame_utils.py:12-31 ``readFlow``, :70-99 ``writeFlow``).

Format (little endian): float32 tag 202021.25 | int32 width | int32 height | float32 [height][width][2]
(u, v interleaved).  ``readFlow`` returns an ``[H, W, 2]`` float32 array, or ``None`` (after printing the
reference's message) when the tag is wrong; ``writeFlow(filename, uv, v=None)`` accepts either a
stacked ``[H, W, 2]`` array or separate u, v planes.
"""
from __future__ import annotations

import numpy as np

TAG_FLOAT = 202021.25
TAG_CHAR = np.array([TAG_FLOAT], np.float32)


def readFlow(fn):
    with open(fn, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or np.frombuffer(head[:4], "<f4")[0] != np.float32(TAG_FLOAT):
            print("Magic number incorrect. Invalid .flo file")
            return None
        w, h = (int(v) for v in np.frombuffer(head[4:], "<i4"))
        data = np.frombuffer(f.read(8 * w * h), "<f4")
    # np.resize semantics of the reference (:31): a short file repeats its data, it does not raise
    return np.resize(data.astype(np.float32, copy=False), (h, w, 2))


def writeFlow(filename, uv, v=None):
    uv = np.asarray(uv)
    if v is None:
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u, v = uv, np.asarray(v)
    assert u.shape == v.shape
    h, w = u.shape
    out = np.empty((h, w, 2), dtype="<f4")
    out[:, :, 0] = u
    out[:, :, 1] = v
    with open(filename, "wb") as f:
        f.write(np.array([TAG_FLOAT], "<f4").tobytes())
        f.write(np.array([w, h], "<i4").tobytes())
        f.write(out.tobytes())
