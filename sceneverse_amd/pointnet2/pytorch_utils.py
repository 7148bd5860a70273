"""Shared point-wise MLP building blocks with the reference's module tree, so that checkpoints
written by modules/third_party/pointnet2/pytorch_utils.py load unchanged:

    SharedMLP                ref :11-36    children  layer0, layer1, ...
      Conv2d (1x1)           ref :157-188  children  conv [, bn.bn] [, activation]
    Conv1d / FC / BatchNorm* ref :45-64, :123-154, :233-278

(conv has no bias whenever a batch-norm follows, ref :87; conv weights kaiming-normal, BN 1/0.)
"""
from __future__ import annotations

from typing import List, Tuple

import torch.nn as nn

_BN = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}
_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}


class _BNBase(nn.Sequential):
    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, batch_norm=_BN[1], name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=_BN[2], name=name)


class BatchNorm3d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=_BN[3], name=name)


class _ConvBase(nn.Sequential):
    """[bn, act,] conv [, bn, act] depending on `preact` -- a point-wise linear map."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bias and not bn
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=use_bias)
        init(conv_unit.weight)
        if use_bias:
            nn.init.constant_(conv_unit.bias, 0)
        norm = batch_norm(in_size if preact else out_size) if bn else None

        def _norm_act():
            if norm is not None:
                self.add_module(name + "bn", norm)
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            _norm_act()
        self.add_module(name + "conv", conv_unit)
        if not preact:
            _norm_act()


def _conv_cls(dim: int, bn_cls, default_k, default_s, default_p):
    class _Conv(_ConvBase):
        def __init__(self, in_size: int, out_size: int, *, kernel_size=default_k, stride=default_s,
                     padding=default_p, activation=nn.ReLU(inplace=True), bn: bool = False,
                     init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                     name: str = ""):
            super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                             conv=_CONV[dim], batch_norm=bn_cls, bias=bias, preact=preact, name=name)
    return _Conv


Conv1d = _conv_cls(1, BatchNorm1d, 1, 1, 0)
Conv1d.__name__ = Conv1d.__qualname__ = "Conv1d"
Conv2d = _conv_cls(2, BatchNorm2d, (1, 1), (1, 1), (0, 0))
Conv2d.__name__ = Conv2d.__qualname__ = "Conv2d"
Conv3d = _conv_cls(3, BatchNorm3d, (1, 1, 1), (1, 1, 1), (0, 0, 0))
Conv3d.__name__ = Conv3d.__qualname__ = "Conv3d"


class SharedMLP(nn.Sequential):
    """Chain of 1x1 Conv2d(+BN+ReLU) applied independently to every (point, sample) position."""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True),
                 preact: bool = False, first: bool = False, name: str = ""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # very first pre-activated layer: no bn/act
            self.add_module(
                name + f"layer{i}",
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))


class FC(nn.Sequential):
    def __init__(self, in_size: int, out_size: int, *, activation=nn.ReLU(inplace=True),
                 bn: bool = False, init=None, preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)

        def _norm_act(width):
            if bn:
                self.add_module(name + "bn", BatchNorm1d(width))
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            _norm_act(in_size)
        self.add_module(name + "fc", fc)
        if not preact:
            _norm_act(out_size)
