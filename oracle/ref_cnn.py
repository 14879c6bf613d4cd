"""ORACLE (test infrastructure, CPU only) -- build-defined conv encoder / decoder in plain torch.

NOT part of the product path (see ``oracle/README.md``).

The reference instantiates ``cnn.Encoder`` / ``cnn.Decoder`` from the un-vendored package ``cnn``
3.1.1 @ git c669849 (``uv.lock:442-444``); only their YAML configs are in the reference
(``mrssm/mopoe_mrssm/configs/default.yaml:31-92``).  The architecture below is therefore
*build-defined*, constrained by those YAML field names.  PARITY UNPINNED upstream: parity for these
two networks is HIP-vs-this-file, and stated as such wherever it is tested.

Encoder (fields: linear_sizes, activation_name, out_activation_name, channels, kernel_sizes,
strides, paddings, num_residual_blocks, residual_intermediate_size, residual_output_size,
coord_conv):
    x[*, C, H, W] -> (+2 coordinate channels yy, xx in [-1, 1] if coord_conv)
    -> for i: Conv2d(channels[i], k_i, s_i, p_i) -> act
    -> Conv2d(channels[-1] -> residual_output_size, 3, 1, 1)                       (``res_in``)
    -> n x { x + Conv1x1(act(Conv3x3(act(x), -> residual_intermediate_size)), -> residual_output_size) }
    -> act -> flatten -> Linear(linear_sizes[0]) [-> act -> Linear(linear_sizes[i])...] -> out_act

Decoder (fields: linear_sizes, conv_in_shape, activation_name, out_activation_name, channels,
kernel_sizes, strides, paddings, output_paddings, num_residual_blocks,
residual_intermediate_size, residual_input_size):
    f[*, F] -> Linear(linear_sizes[0]) -> act -> ... -> Linear(linear_sizes[-1]) -> reshape conv_in_shape
    -> n x { x + Conv1x1(act(Conv3x3(act(x), -> residual_intermediate_size)), -> residual_input_size) }
    -> act -> for i: ConvTranspose2d(channels[i], k_i, s_i, p_i, op_i) -> (act | out_act after the last)

Both accept arbitrary leading batch dims (the reference passes ``[B,T,C,H,W]`` and ``[B,C,H,W]``,
``mrssm/mopoe_mrssm/core.py:179-180,215-216,272-273``).
"""

from __future__ import annotations

from typing import Any

import torch
from torch import Tensor, nn


def _act(name: str) -> nn.Module:
    return getattr(nn, name)()


def coord_channels(height: int, width: int, like: Tensor) -> Tensor:
    """[2, H, W]: row coordinate then column coordinate, each linspace(-1, 1)."""
    yy = torch.linspace(-1.0, 1.0, height, dtype=like.dtype, device=like.device)
    xx = torch.linspace(-1.0, 1.0, width, dtype=like.dtype, device=like.device)
    return torch.stack([yy[:, None].expand(height, width), xx[None, :].expand(height, width)], dim=0)


class ResidualBlock(nn.Module):
    def __init__(self, channels: int, intermediate: int, activation_name: str) -> None:
        super().__init__()
        self.conv3 = nn.Conv2d(channels, intermediate, 3, 1, 1)
        self.conv1 = nn.Conv2d(intermediate, channels, 1, 1, 0)
        self.act = _act(activation_name)

    def forward(self, x: Tensor) -> Tensor:
        return x + self.conv1(self.act(self.conv3(self.act(x))))


class Encoder(nn.Module):
    def __init__(self, config: dict[str, Any] | Any) -> None:
        super().__init__()
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        self.cfg = cfg
        self.coord_conv = bool(cfg.get("coord_conv", False))
        self.act = _act(cfg["activation_name"])
        self.out_act = _act(cfg.get("out_activation_name", "Identity"))
        in_shape = cfg.get("input_shape")
        self.input_shape = tuple(in_shape) if in_shape is not None else None
        self.convs = nn.ModuleList()
        self.res_in: nn.Module | None = None
        self.res = nn.ModuleList()
        self.linears = nn.ModuleList()
        if self.input_shape is not None:
            self.materialize(self.input_shape)

    def materialize(self, input_shape: tuple[int, ...]) -> None:
        cfg = self.cfg
        c, h, w = input_shape
        self.input_shape = (c, h, w)
        cin = c + (2 if self.coord_conv else 0)
        for ch, k, s, p in zip(cfg["channels"], cfg["kernel_sizes"], cfg["strides"], cfg["paddings"], strict=True):
            self.convs.append(nn.Conv2d(cin, ch, k, s, p))
            h = (h + 2 * p - k) // s + 1
            w = (w + 2 * p - k) // s + 1
            cin = ch
        if cfg.get("num_residual_blocks", 0) > 0:
            rout = cfg["residual_output_size"]
            self.res_in = nn.Conv2d(cin, rout, 3, 1, 1)
            for _ in range(cfg["num_residual_blocks"]):
                self.res.append(ResidualBlock(rout, cfg["residual_intermediate_size"], cfg["activation_name"]))
            cin = rout
        width = cin * h * w
        for out in cfg["linear_sizes"]:
            self.linears.append(nn.Linear(width, out))
            width = out
        self.feature_hw = (h, w)

    def forward(self, x: Tensor) -> Tensor:
        if len(self.linears) == 0:
            self.materialize(tuple(x.shape[-3:]))
            self.to(x.device)
        lead = x.shape[:-3]
        x = x.reshape(-1, *x.shape[-3:])
        if self.coord_conv:
            cc = coord_channels(x.shape[-2], x.shape[-1], x)
            x = torch.cat([x, cc.unsqueeze(0).expand(x.shape[0], -1, -1, -1)], dim=1)
        for conv in self.convs:
            x = self.act(conv(x))
        if self.res_in is not None:
            x = self.res_in(x)
            for blk in self.res:
                x = blk(x)
            x = self.act(x)
        x = x.flatten(start_dim=1)
        for i, lin in enumerate(self.linears):
            x = lin(x)
            if i + 1 < len(self.linears):
                x = self.act(x)
        x = self.out_act(x)
        return x.reshape(*lead, x.shape[-1])


class Decoder(nn.Module):
    def __init__(self, config: dict[str, Any] | Any) -> None:
        super().__init__()
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        self.cfg = cfg
        self.act = _act(cfg["activation_name"])
        self.out_act = _act(cfg.get("out_activation_name", "Identity"))
        self.conv_in_shape = tuple(cfg["conv_in_shape"])
        self.linears = nn.ModuleList()
        self.res = nn.ModuleList()
        self.deconvs = nn.ModuleList()
        cin = self.conv_in_shape[0]
        for _ in range(cfg.get("num_residual_blocks", 0)):
            self.res.append(ResidualBlock(cin, cfg["residual_intermediate_size"], cfg["activation_name"]))
        ops = cfg.get("output_paddings", [0] * len(cfg["channels"]))
        for ch, k, s, p, op in zip(cfg["channels"], cfg["kernel_sizes"], cfg["strides"], cfg["paddings"], ops, strict=True):
            self.deconvs.append(nn.ConvTranspose2d(cin, ch, k, s, p, op))
            cin = ch
        if cfg.get("in_features") is not None:
            self.materialize(int(cfg["in_features"]))

    def materialize(self, in_features: int) -> None:
        width = in_features
        for out in self.cfg["linear_sizes"]:
            self.linears.append(nn.Linear(width, out))
            width = out
        self.in_features = in_features

    def forward(self, f: Tensor) -> Tensor:
        if len(self.linears) == 0:
            self.materialize(f.shape[-1])
            self.to(f.device)
        lead = f.shape[:-1]
        x = f.reshape(-1, f.shape[-1])
        for i, lin in enumerate(self.linears):
            x = lin(x)
            if i + 1 < len(self.linears):
                x = self.act(x)
        x = x.reshape(-1, *self.conv_in_shape)
        if len(self.res) > 0:
            for blk in self.res:
                x = blk(x)
            x = self.act(x)
        for i, dc in enumerate(self.deconvs):
            x = dc(x)
            x = self.act(x) if i + 1 < len(self.deconvs) else self.out_act(x)
        return x.reshape(*lead, *x.shape[-3:])
