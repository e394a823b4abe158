"""Parameter tree of PriOr-RAFT with the reference's exact ``state_dict`` contract.

Callers of the reference do ``load_state_dict(strict=True)`` on (optionally
``module.``-prefixed) checkpoints (evaluate.py:410-411, demo_image.py:30-31,
train_flow.py:101), so sub-module attribute names and parameter shapes below are
part of the drop-in boundary: 217 entries, top-level prefixes
``fnet. cnet. ODDC. update_block.``.

Every module here is a *parameter container only* (round 6 removed the two plain-torch encoder ``forward``s that survived as
CPU shape checks): the arithmetic runs in the HIP library (``csrc/``) -- inference through ``engine.EncoderPlan`` / ``engine.Engine``,
training through ``autograd.encoder_forward`` / ``train_loop`` -- never through ``nn.Conv2d.forward``.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _conv(cin, cout, k, pad=0, stride=1):
    return nn.Conv2d(cin, cout, k, padding=pad, stride=stride)


# ---- encoders (core/extractor.py:8-47, :98-158) -------------------------------------------
def _norm(kind: str, ch: int) -> nn.Module:
    if kind == "instance":
        return nn.InstanceNorm2d(ch)          # no affine, no running stats
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    def __init__(self, cin, ch, kind, stride):
        super().__init__()
        self.conv1 = _conv(cin, ch, 3, 1, stride)
        self.conv2 = _conv(ch, ch, 3, 1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(kind, ch)
        self.norm2 = _norm(kind, ch)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(kind, ch)
            self.downsample = nn.Sequential(_conv(cin, ch, 1, 0, stride), self.norm3)


class BasicEncoder(nn.Module):
    def __init__(self, output_dim, norm_fn, dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = _conv(3, 64, 7, 3, 2)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, norm_fn, 1), ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, norm_fn, 2), ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, norm_fn, 2), ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = _conv(128, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():              # same init family as core/extractor.py:123-129
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


# ---- update blocks (core/update.py) ----------------------------------------------------------
class FlowHead(nn.Module):
    def __init__(self, cin=128, hidden=256):
        super().__init__()
        self.conv1 = _conv(cin, hidden, 3, 1)
        self.conv2 = _conv(hidden, 2, 3, 1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden=128, cin=256):
        super().__init__()
        for tag, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{tag}", nn.Conv2d(hidden + cin, hidden, k, padding=p))


def _mask_head(hidden=128):
    return nn.Sequential(_conv(hidden, 256, 3, 1), nn.ReLU(inplace=True), _conv(256, 64 * 9, 1))


class BasicMotionEncoder(nn.Module):
    def __init__(self, cor_planes=324):
        super().__init__()
        self.convc1 = _conv(cor_planes, 256, 1)
        self.convc2 = _conv(256, 192, 3, 1)
        self.convf1 = _conv(2, 128, 7, 3)
        self.convf2 = _conv(128, 64, 3, 1)
        self.conv = _conv(64 + 192, 128 - 2, 3, 1)


class BasicMultiMotionEncoder(nn.Module):
    def __init__(self, cor_planes=324):
        super().__init__()
        self.convc1_A = _conv(cor_planes, 256, 1)
        self.convc2_A = _conv(256, 128, 3, 1)
        self.convf1_A = _conv(2, 128, 7, 3)
        self.convf2_A = _conv(128, 64, 3, 1)
        self.convf1_B = _conv(2, 128, 7, 3)
        self.convf2_B = _conv(128, 64, 3, 1)
        self.conv_conf1 = _conv(8, 32, 3, 1)
        self.conv_conf2 = _conv(32, 16, 3, 1)
        self.conv_A = _conv(128 + 64 + 64 + 16, 128 - 4, 3, 1)


class BasicUpdateBlock(nn.Module):
    def __init__(self, hidden=128):
        super().__init__()
        self.encoder = BasicMotionEncoder()
        self.gru = SepConvGRU(hidden, 128 + hidden)
        self.flow_head = FlowHead(hidden, 256)
        self.mask = _mask_head(hidden)


class BasicMultiUpdateBlock(nn.Module):
    def __init__(self, hidden=128):
        super().__init__()
        self.encoder = BasicMultiMotionEncoder()
        self.gru = SepConvGRU(hidden, 128 + 128)
        self.flow_head = FlowHead(hidden, 256)
        self.mask = _mask_head(hidden)


def build_tree(dropout: float = 0.0):
    """Returns (fnet, cnet, ODDC, update_block)  (core/prior_raft.py:37-41)."""
    return (BasicEncoder(256, "instance", dropout), BasicEncoder(256, "batch", dropout),
            BasicMultiUpdateBlock(128), BasicUpdateBlock(128))


def state_dict_shapes():
    """name -> shape for the full 217-entry contract, without allocating weights."""
    with torch.device("meta"):
        fnet, cnet, oddc, upd = build_tree()
    out = {}
    for prefix, mod in (("fnet.", fnet), ("cnet.", cnet), ("ODDC.", oddc), ("update_block.", upd)):
        for k, v in mod.state_dict().items():
            out[prefix + k] = tuple(v.shape)
    return out
