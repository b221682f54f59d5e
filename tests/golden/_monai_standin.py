"""Minimal stand-in for the handful of MONAI 0.4.0 symbols that the reference's L3 files import.

TEST INFRASTRUCTURE ONLY (golden generation in the build container).  MONAI is pinned by the reference
(`ref:requirements.txt:7`) but is neither vendored under /root/reference nor installed here, so the
reference's model/loss files cannot be imported without it.  This module registers just enough of the
`monai.*` namespace in `sys.modules` for

    ref:params/networks/blocks/convolutions.py:18-19
    ref:params/networks/blocks/attentionblock.py:3
    ref:params/networks/nets/unet2d5_spvPA.py:16-19
    ref:params/losses/dice_spvPA.py:20-21

to import and run unmodified from /root/reference.  Nothing here is shipped to the GPU box as product.
"""
import enum
import sys
import types

import torch
import torch.nn as nn


def _same_padding(kernel_size, dilation=1):
    ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size,)
    dl = dilation if isinstance(dilation, (tuple, list)) else (dilation,) * len(ks)
    out = []
    for k, d in zip(ks, dl):
        if (k - 1) * d % 2 == 1:
            raise NotImplementedError("same padding not available for this kernel/dilation")
        out.append((k - 1) * d // 2)
    return tuple(out) if len(out) > 1 else out[0]


class _Factory:
    def __init__(self, table):
        self._table = table
        for name in table:
            setattr(self, name.upper(), name)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            name, dim = key
            return self._table[name.lower()](dim)
        return self._table[key.lower()](None)


Act = _Factory({"prelu": lambda _: nn.PReLU, "relu": lambda _: nn.ReLU, "sigmoid": lambda _: nn.Sigmoid})
Norm = _Factory(
    {
        "batch": lambda d: (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)[d - 1],
        "instance": lambda d: (nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)[d - 1],
    }
)
Conv = _Factory(
    {
        "conv": lambda d: (nn.Conv1d, nn.Conv2d, nn.Conv3d)[d - 1],
        "convtrans": lambda d: (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)[d - 1],
    }
)
Dropout = _Factory({"dropout": lambda d: (nn.Dropout, nn.Dropout2d, nn.Dropout3d)[d - 1]})


def split_args(args):
    if isinstance(args, str):
        return args, {}
    return args[0], args[1]


class SkipConnection(nn.Module):
    def __init__(self, submodule, cat_dim=1):
        super().__init__()
        self.submodule = submodule
        self.cat_dim = cat_dim

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], self.cat_dim)


def export(_modname):
    return lambda obj: obj


def alias(*_names):
    return lambda obj: obj


class LossReduction(enum.Enum):
    NONE = "none"
    MEAN = "mean"
    SUM = "sum"


class Weight(enum.Enum):
    SQUARE = "square"
    SIMPLE = "simple"
    UNIFORM = "uniform"


def one_hot(labels, num_classes, dtype=torch.float, dim=1):
    shape = list(labels.shape)
    shape[dim] = num_classes
    out = torch.zeros(shape, dtype=dtype, device=labels.device)
    return out.scatter_(dim, labels.long(), 1)


def install():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("monai")
    mod("monai.networks", one_hot=one_hot)
    mod("monai.networks.nets")
    mod("monai.networks.blocks")
    mod("monai.networks.utils", one_hot=one_hot)
    mod("monai.networks.layers")
    mod("monai.networks.layers.convutils", same_padding=_same_padding)
    mod("monai.networks.layers.factories", Act=Act, Norm=Norm, Conv=Conv, Dropout=Dropout, split_args=split_args)
    mod("monai.networks.layers.simplelayers", SkipConnection=SkipConnection)
    mod("monai.utils", export=export, alias=alias, LossReduction=LossReduction, Weight=Weight)
    mod("monai.utils.aliases", alias=alias)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
