"""Conv encoder / decoder (the ``cnn.Encoder`` / ``cnn.Decoder`` the reference YAMLs instantiate).

The reference takes these from the un-vendored ``cnn`` 3.1.1 package; only the YAML field names are
in the reference tree (``mrssm/mopoe_mrssm/configs/default.yaml:31-92``), so the architecture is
build-defined ("parity unpinned") and identical to ``oracle/ref_cnn.py``:

Encoder: [+2 coordinate channels] -> (Conv k,s,p -> act) x n -> Conv3x3 to ``residual_output_size``
         -> n_res x {x + Conv1x1(act(Conv3x3(act x)))} -> act -> flatten -> Linear(s) -> out_act
Decoder: Linear -> act -> Linear -> reshape ``conv_in_shape`` -> n_res x {residual block} -> act
         -> (ConvTranspose k,s,p -> act) x (n-1) -> ConvTranspose -> out_act

Both take arbitrary leading batch dims (``[B,T,C,H,W]`` and ``[B,C,H,W]`` are both used upstream,
``mrssm/mopoe_mrssm/core.py:179-180,215-216,272-273``).  Frames are folded to one ``[B*T, C, H, W]`` batch so
every layer is a single dense launch over all B*T frames.

Every convolution runs on the hand-written MFMA implicit-GEMM kernels of ``csrc/conv.hip`` / ``conv_split.h`` /
``conv_resident.h`` (``conv.py``): because each stack is "activation -> conv" throughout, the activation is fused into the
consumer's operand staging (``pre_act``) and never materialised; bias is fused into the epilogue; the backward kernels fuse the
multiplication by act'(x).  The Linear layers run on ``csrc/gemm.hip`` (``linear.linear``: fp32 MFMA with the activation of
the operand, bias, and the gradients' epilogues fused).  No rocBLAS / MIOpen call is made anywhere (MIOpen's first-run kernel
JIT costs minutes on a fresh box).
The nn.Conv2d / nn.ConvTranspose2d / nn.Linear objects below are parameter containers only.
"""

from __future__ import annotations

from typing import Any

import torch
from torch import Tensor, nn

from multimodal_mtrssm_amd import _lib
from multimodal_mtrssm_amd.linear import linear
from multimodal_mtrssm_amd.conv import (
    gemm_pieces,
    conv2d,
    conv2d_pair,
    conv_transpose2d,
    conv_transpose2d_pair,
    residual_block,
    residual_block_pair,
)


def _act(name: str) -> nn.Module:
    return getattr(nn, name)()


def _cfg(config: Any) -> dict[str, Any]:  # noqa: ANN401
    return dict(config) if isinstance(config, dict) else dict(vars(config))


def _conv(x: Tensor, m: nn.Conv2d, *, pre_act: bool, act: int, coords: Tensor | None = None) -> Tensor:
    return conv2d(x, m.weight, m.bias, stride=m.stride[0], padding=m.padding[0], pre_act=pre_act, act=act, coords=coords)


def _deconv(x: Tensor, m: nn.ConvTranspose2d, *, pre_act: bool, act: int) -> Tensor:
    return conv_transpose2d(x, m.weight, m.bias, stride=m.stride[0], padding=m.padding[0],
                            output_padding=m.output_padding[0], pre_act=pre_act, act=act)


class ResidualBlock(nn.Module):
    """``x + Conv1x1(act(Conv3x3(act(x))))`` -- both activations fused into the convs' operand staging."""

    def __init__(self, channels: int, intermediate: int, activation_name: str) -> None:
        super().__init__()
        self.conv3 = nn.Conv2d(channels, intermediate, 3, 1, 1)
        self.conv1 = nn.Conv2d(intermediate, channels, 1, 1, 0)
        self.act_id = _lib.ACT_IDS[activation_name]

    def forward(self, x: Tensor) -> Tensor:
        return residual_block(x, self.conv3.weight, self.conv3.bias, self.conv1.weight, self.conv1.bias, act=self.act_id)

    def params(self) -> tuple[Tensor, Tensor, Tensor, Tensor]:
        return self.conv3.weight, self.conv3.bias, self.conv1.weight, self.conv1.bias


def _res_pair(res_a: nn.ModuleList, res_b: nn.ModuleList, xa: Tensor, xb: Tensor) -> tuple[Tensor, Tensor]:
    """Two residual stacks walked in lockstep: blocks of equal shape go out as paired launches (``conv.paired``)."""
    n = min(len(res_a), len(res_b))
    for ba, bb in zip(res_a[:n], res_b[:n], strict=True):
        if ba.act_id == bb.act_id and ba.conv3.weight.shape == bb.conv3.weight.shape and ba.conv1.weight.shape == bb.conv1.weight.shape:
            xa, xb = residual_block_pair(xa, ba.params(), xb, bb.params(), act=ba.act_id)
        else:
            xa, xb = ba(xa), bb(xb)
    for ba in res_a[n:]:
        xa = ba(xa)
    for bb in res_b[n:]:
        xb = bb(xb)
    return xa, xb


def _same_layer(ma: nn.Module, mb: nn.Module) -> bool:
    return (ma.weight.shape[0] == mb.weight.shape[0] and ma.weight.shape[2:] == mb.weight.shape[2:] and ma.stride == mb.stride
            and ma.padding == mb.padding and getattr(ma, "output_padding", 0) == getattr(mb, "output_padding", 0))


def _conv_pair(xa: Tensor, ma: nn.Conv2d, ca: Tensor | None, xb: Tensor, mb: nn.Conv2d, cb: Tensor | None, *, pre_act: bool,
               act_a: int, act_b: int) -> tuple[Tensor, Tensor]:
    if _same_layer(ma, mb) and act_a == act_b:
        return conv2d_pair((xa, ma.weight, ma.bias, ma.stride[0], ma.padding[0], pre_act, act_a, ca),
                           (xb, mb.weight, mb.bias, mb.stride[0], mb.padding[0], pre_act, act_b, cb))
    return _conv(xa, ma, pre_act=pre_act, act=act_a, coords=ca), _conv(xb, mb, pre_act=pre_act, act=act_b, coords=cb)


def encode_pair(enc_a: "Encoder", enc_b: "Encoder", xa: Tensor, xb: Tensor) -> tuple[Tensor, Tensor]:
    """``(enc_a(xa), enc_b(xb))`` with the layers of the two stacks that have the same shape sharing their launches."""
    lead_a, lead_b = xa.shape[:-3], xb.shape[:-3]
    xa, ca = enc_a.prepare(xa)
    xb, cb = enc_b.prepare(xb)
    if len(enc_a.convs) == len(enc_b.convs) and (enc_a.res_in is None) == (enc_b.res_in is None):
        for i, (ma, mb) in enumerate(zip(enc_a.convs, enc_b.convs, strict=True)):
            xa, xb = _conv_pair(xa, ma, ca if i == 0 else None, xb, mb, cb if i == 0 else None, pre_act=i > 0, act_a=enc_a.act_id,
                                act_b=enc_b.act_id)
        if enc_a.res_in is not None:
            xa, xb = _conv_pair(xa, enc_a.res_in, None, xb, enc_b.res_in, None, pre_act=True, act_a=enc_a.act_id, act_b=enc_b.act_id)
    else:
        xa, xb = enc_a.stem_convs(xa, ca), enc_b.stem_convs(xb, cb)
    xa, xb = _res_pair(enc_a.res, enc_b.res, xa, xb)
    return enc_a.head(xa, lead_a), enc_b.head(xb, lead_b)


def decode_pair(dec_a: "Decoder", dec_b: "Decoder", fa: Tensor, fb: Tensor, *, raw: bool = False) -> tuple[Tensor, Tensor]:
    """``(dec_a(fa), dec_b(fb))`` with the layers of the two stacks that have the same shape sharing their launches.
    ``raw``: without the final ``out_activation`` (the fused Gaussian NLL applies it while it reads the prediction)."""
    lead_a, lead_b = fa.shape[:-1], fb.shape[:-1]
    xa, xb = dec_a.stem(fa), dec_b.stem(fb)
    xa, xb = _res_pair(dec_a.res, dec_b.res, xa, xb)
    if len(dec_a.deconvs) != len(dec_b.deconvs) or (len(dec_a.res) > 0) != (len(dec_b.res) > 0):
        return dec_a.tail(xa, lead_a, raw=raw), dec_b.tail(xb, lead_b, raw=raw)
    for i, (ma, mb) in enumerate(zip(dec_a.deconvs, dec_b.deconvs, strict=True)):
        pre = i > 0 or len(dec_a.res) > 0
        if _same_layer(ma, mb) and dec_a.act_id == dec_b.act_id:
            xa, xb = conv_transpose2d_pair(
                (xa, ma.weight, ma.bias, ma.stride[0], ma.padding[0], ma.output_padding[0], pre, dec_a.act_id),
                (xb, mb.weight, mb.bias, mb.stride[0], mb.padding[0], mb.output_padding[0], pre, dec_b.act_id))
        else:
            xa, xb = _deconv(xa, ma, pre_act=pre, act=dec_a.act_id), _deconv(xb, mb, pre_act=pre, act=dec_b.act_id)
    return dec_a.finish(xa, lead_a, raw=raw), dec_b.finish(xb, lead_b, raw=raw)


class Encoder(nn.Module):
    """``Encoder(config)``; optional extra key ``input_shape: [C, H, W]`` builds the Linear eagerly."""

    def __init__(self, config: Any) -> None:  # noqa: ANN401
        super().__init__()
        self.cfg = _cfg(config)
        self.coord_conv = bool(self.cfg.get("coord_conv", False))
        self.act = _act(self.cfg["activation_name"])
        self.act_id = _lib.ACT_IDS[self.cfg["activation_name"]]
        self.out_act = _act(self.cfg.get("out_activation_name", "Identity"))
        self.convs = nn.ModuleList()
        self.res_in: nn.Module | None = None
        self.res = nn.ModuleList()
        self.linears = nn.ModuleList()
        self.input_shape: tuple[int, int, int] | None = None
        self._coords: Tensor | None = None
        if self.cfg.get("input_shape") is not None:
            self.materialize(tuple(self.cfg["input_shape"]))

    def materialize(self, input_shape: tuple[int, ...]) -> None:
        cfg = self.cfg
        c, h, w = (int(v) for v in input_shape)
        self.input_shape = (c, h, w)
        cin = c + (2 if self.coord_conv else 0)
        for ch, k, s, p in zip(cfg["channels"], cfg["kernel_sizes"], cfg["strides"], cfg["paddings"], strict=True):
            self.convs.append(nn.Conv2d(cin, ch, k, s, p))
            h, w, cin = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1, ch
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

    def _coord_channels(self, x: Tensor) -> Tensor:
        h, w = x.shape[-2:]
        if self._coords is None or self._coords.shape[-2:] != (h, w) or self._coords.device != x.device:
            yy = torch.linspace(-1.0, 1.0, h, dtype=x.dtype, device=x.device)
            xx = torch.linspace(-1.0, 1.0, w, dtype=x.dtype, device=x.device)
            self._coords = torch.stack([yy[:, None].expand(h, w), xx[None, :].expand(h, w)], dim=0)
        return self._coords

    def forward(self, x: Tensor) -> Tensor:
        lead = x.shape[:-3]
        x = self.stem(x)
        for blk in self.res:
            x = blk(x)
        return self.head(x, lead)

    def prepare(self, x: Tensor) -> tuple[Tensor, Tensor | None]:
        """Flattened fp32 frames and the coordinate planes of the first conv."""
        if len(self.linears) == 0:
            self.materialize(tuple(x.shape[-3:]))
            self.to(x.device)
        x = x.reshape(-1, *x.shape[-3:]).float()
        # the coordinate channels are frame-independent: the first conv gathers them from one [2,H,W] plane
        return x, (self._coord_channels(x) if self.coord_conv else None)

    def stem_convs(self, x: Tensor, coords: Tensor | None) -> Tensor:
        for i, conv in enumerate(self.convs):
            x = _conv(x, conv, pre_act=i > 0, act=self.act_id, coords=coords if i == 0 else None)
        if self.res_in is not None:
            x = _conv(x, self.res_in, pre_act=True, act=self.act_id)
        return x

    def stem(self, x: Tensor) -> Tensor:
        """Strided convs (+ the conv into the residual stack) over the flattened frames."""
        return self.stem_convs(*self.prepare(x))

    def head(self, x: Tensor, lead: torch.Size) -> Tensor:
        # "act -> flatten -> Linear [-> act -> Linear ...]": every activation rides in the following GEMM's operand staging
        x = x.flatten(start_dim=1)
        for lin in self.linears:
            x = linear(x, lin.weight, lin.bias, pre_act=self.act_id, pieces=gemm_pieces())
        return self.out_act(x).reshape(*lead, -1)


class Decoder(nn.Module):
    """``Decoder(config)``; optional extra key ``in_features`` builds the first Linear eagerly."""

    def __init__(self, config: Any) -> None:  # noqa: ANN401
        super().__init__()
        cfg = self.cfg = _cfg(config)
        self.act = _act(cfg["activation_name"])
        self.act_id = _lib.ACT_IDS[cfg["activation_name"]]
        self.out_act = _act(cfg.get("out_activation_name", "Identity"))
        self.conv_in_shape = tuple(int(v) for v in cfg["conv_in_shape"])
        self.linears = nn.ModuleList()
        self.res = nn.ModuleList()
        self.deconvs = nn.ModuleList()
        cin = self.conv_in_shape[0]
        for _ in range(cfg.get("num_residual_blocks", 0)):
            self.res.append(ResidualBlock(cin, cfg["residual_intermediate_size"], cfg["activation_name"]))
        n = len(cfg["channels"])
        ops = cfg.get("output_paddings", [0] * n)
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

    def forward(self, f: Tensor, *, raw: bool = False) -> Tensor:
        lead = f.shape[:-1]
        x = self.stem(f)
        for blk in self.res:
            x = blk(x)
        return self.tail(x, lead, raw=raw)

    @property
    def out_act_id(self) -> int | None:
        """Activation id of ``out_activation`` if the fused NLL kernel can apply it (Identity / Tanh), else None."""
        name = type(self.out_act).__name__
        return {"Identity": 0, "Tanh": 3}.get(name)

    def stem(self, f: Tensor) -> Tensor:
        if len(self.linears) == 0:
            self.materialize(f.shape[-1])
            self.to(f.device)
        x = f.reshape(-1, f.shape[-1])
        for i, lin in enumerate(self.linears):
            x = linear(x, lin.weight, lin.bias, pre_act=self.act_id if i > 0 else 0, pieces=gemm_pieces())
        return x.reshape(-1, *self.conv_in_shape)

    def tail(self, x: Tensor, lead: torch.Size, *, raw: bool = False) -> Tensor:
        for i, dc in enumerate(self.deconvs):
            # "act -> deconv" everywhere except a first deconv fed straight by the Linear (no residual stack)
            x = _deconv(x, dc, pre_act=i > 0 or len(self.res) > 0, act=self.act_id)
        return self.finish(x, lead, raw=raw)

    def finish(self, x: Tensor, lead: torch.Size, *, raw: bool = False) -> Tensor:
        if not raw:
            x = self.out_act(x)
        return x.reshape(*lead, *x.shape[-3:])
