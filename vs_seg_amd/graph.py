"""Static description of the 2.5D attention U-Net as a flat op list (no GPU, no torch modules).

Structure follows ref:params/networks/nets/unet2d5_spvPA.py:56-93 (`_create_block` recursion) with the hyper-parameters
hard-coded at ref:params/VSparams.py:343-374.  State-dict key prefixes are the reference's, so checkpoints interchange.

The op list is what `vs_seg_amd.engine` lowers to HIP launches (forward in order, backward in reverse).  The
skip-connection concat (MONAI SkipConnection, `cat([x, sub(x)], 1)`) is a *two-part* tensor: the pair (encoder output,
upsample output) of two dense buffers — nothing is copied, and neither operand becomes a strided channel slice (slices
of a twice-as-wide buffer were measured at less than half the HBM efficiency of dense tensors on the 16/32-channel levels).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

HP = dict(
    in_channels=1,
    out_channels=2,
    channels=(16, 32, 48, 64, 80, 96),
    strides=((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)),
    kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
    sample_kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
    num_res_units=2,
    dropout=0.1,
)


def level_dims(patch, hp=HP):
    dims = [tuple(patch)]
    for s in hp["strides"]:
        d = dims[-1]
        assert all(a % b == 0 for a, b in zip(d, s)), f"spatial size {d} is not divisible by the stride {s}"
        dims.append(tuple(a // b for a, b in zip(d, s)))
    return dims


@dataclass
class Layer:
    """One Conv3d / ConvTranspose3d with its state-dict keys."""

    prefix: str  # e.g. 'model.0.conv.unit0'  (keys: prefix + '.conv.weight' ...) or 'model.0.residual' (prefix + '.weight')
    cin: int
    cout: int
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    transposed: bool
    level: int  # resolution level of the INPUT
    plain: bool = False  # True: keys are prefix.weight / prefix.bias (ResidualUnit.residual)
    has_bn: bool = False

    @property
    def wkey(self):
        return self.prefix + (".weight" if self.plain else ".conv.weight")

    @property
    def bkey(self):
        return self.prefix + (".bias" if self.plain else ".conv.bias")

    @property
    def wshape(self):
        return (self.cin, self.cout, *self.kernel) if self.transposed else (self.cout, self.cin, *self.kernel)

    def in_dims(self, patch, hp=HP):
        return level_dims(patch, hp)[self.level]

    @property
    def out_level(self):
        if all(s == 1 for s in self.stride):
            return self.level
        return self.level - 1 if self.transposed else self.level + 1


@dataclass
class TensorSpec:
    name: str
    level: int
    c: int  # channels of the underlying buffer
    kind: str = "act"  # 'act' (compute dtype), 'f32'
    base: Optional["TensorSpec"] = None  # channel slice of another tensor
    c0: int = 0
    creal: int = 0  # meaningful channels when the buffer is zero-extended (network input: 1 of 8)
    parts: Optional[Tuple["TensorSpec", "TensorSpec"]] = None  # channel concatenation of two dense tensors (no buffer of its own)

    @property
    def real(self):
        return self.creal or self.c

    def slice(self, c0, c):
        root = self.base or self
        return TensorSpec(f"{root.name}[{self.c0 + c0}:{self.c0 + c0 + c}]", self.level, c, self.kind, root, self.c0 + c0)

    @property
    def root(self):
        return self.base or self


@dataclass
class ConvBnAct:  # Convolution: conv -> BatchNorm -> Dropout -> PReLU  (+ residual add of a ResidualUnit)
    layer: Layer
    x: TensorSpec
    out: TensorSpec
    res: Optional[TensorSpec] = None


@dataclass
class ConvPlain:  # conv + bias + optional activation (+ residual): attention convs, 1x1x1 residual convs, the final conv
    layer: Layer
    x: TensorSpec
    out: TensorSpec
    act: str = "none"  # none | relu | sigmoid
    res: Optional[TensorSpec] = None


@dataclass
class AttGate:  # AttentionBlock2: out = x * (1 + att)
    x: TensorSpec
    att: TensorSpec
    out: TensorSpec


@dataclass
class Program:
    ops: list
    tensors: List[TensorSpec]
    layers: List[Layer]
    input: TensorSpec
    logits: TensorSpec
    att_maps: List[TensorSpec]  # coarsest -> finest (hook order, ref:.../unet2d5_spvPA.py:101-104)


def build_program(attention: bool = True, hp: dict = HP) -> Program:
    ch, st, ks, sks = hp["channels"], hp["strides"], hp["kernel_sizes"], hp["sample_kernel_sizes"]
    ops, tensors, layers, atts = [], [], [], []

    def new(name, level, c, kind="act"):
        t = TensorSpec(name, level, c, kind)
        tensors.append(t)
        return t

    def layer(**kw):
        L = Layer(**kw)
        layers.append(L)
        return L

    def residual_unit(x, p, k, lvl, subunits, out, last_conv_only=False):
        cin, cout = x.real, out.c
        if cin != cout:  # 1x1x1 conv residual (ref:.../convolutions.py:241-250)
            r = new(p + ":res", lvl, cout, out.kind)
            ops.append(ConvPlain(layer(prefix=p + ".residual", cin=cin, cout=cout, kernel=(1, 1, 1), stride=(1, 1, 1), transposed=False, level=lvl, plain=True), x, r))
        else:
            r = x
        cur = x
        for su in range(subunits):
            last = su == subunits - 1
            dst = out if last else new(f"{p}:u{su}", lvl, cout)
            L = layer(prefix=f"{p}.conv.unit{su}", cin=cur.real, cout=cout, kernel=k, stride=(1, 1, 1), transposed=False, level=lvl, has_bn=not (last_conv_only and last))
            if last_conv_only and last:
                ops.append(ConvPlain(L, cur, dst, "none", r))
            else:
                ops.append(ConvBnAct(L, cur, dst, r if last else None))
            cur = dst

    def attention_block(x, p, k, lvl):
        c = x.c
        h = new(p + ":h", lvl, c // 2)
        a = new(p + ":att", lvl, 1, "f32")
        g = new(p + ":gated", lvl, c)
        ops.append(ConvPlain(layer(prefix=p + ".conv1", cin=c, cout=c // 2, kernel=k, stride=(1, 1, 1), transposed=False, level=lvl), x, h, "relu"))
        ops.append(ConvPlain(layer(prefix=p + ".conv2", cin=c // 2, cout=1, kernel=k, stride=(1, 1, 1), transposed=False, level=lvl), h, a, "sigmoid"))
        ops.append(AttGate(x, a, g))
        atts.append(a)
        return g

    def block(x, p, lvl, outc, is_top):
        c, k, sk, s = ch[lvl], ks[lvl], sks[lvl], st[lvl]
        d, up = new(f"skip{lvl}", lvl, c), new(f"upcat{lvl}", lvl, c)
        cat = TensorSpec(f"cat{lvl}", lvl, 2 * c, parts=(d, up))
        residual_unit(x, p + ".0", k, lvl, hp["num_res_units"], d)
        sub = p + ".1.submodule"
        ad = new(f"down{lvl}", lvl + 1, c)
        ops.append(ConvBnAct(layer(prefix=sub + ".0", cin=c, cout=c, kernel=sk, stride=s, transposed=False, level=lvl, has_bn=True), d, ad))
        if lvl + 2 < len(ch):
            u = block(ad, sub + ".1", lvl + 1, ch[lvl + 1], False)
        else:
            kb = ks[lvl + 1]
            u = new("bottom", lvl + 1, ch[lvl + 1])
            if attention:
                g = attention_block(ad, sub + ".1.0.0", kb, lvl + 1)
                residual_unit(g, sub + ".1.1", kb, lvl + 1, hp["num_res_units"], u)
            else:
                residual_unit(ad, sub + ".1", kb, lvl + 1, hp["num_res_units"], u)
        ops.append(ConvBnAct(layer(prefix=sub + ".2", cin=ch[lvl + 1], cout=c, kernel=sk, stride=s, transposed=True, level=lvl + 1, has_bn=True), u, up))
        out = new("logits" if is_top else f"up{lvl}", lvl, outc, "f32" if is_top else "act")
        if attention:
            g = attention_block(cat, p + ".2.0.0", k, lvl)
            residual_unit(g, p + ".2.1", k, lvl, 1, out, last_conv_only=is_top)
        else:
            residual_unit(cat, p + ".2", k, lvl, 1, out, last_conv_only=is_top)
        return out

    x0 = new("input", 0, 8)
    x0.creal = hp["in_channels"]  # 1 real channel, zero-extended to one 8-channel K-group for the MFMA path
    logits = block(x0, "model", 0, hp["out_channels"], True)
    return Program(ops, tensors, layers, x0, logits, atts)


def conv_layers(attention: bool = True, hp: dict = HP) -> List[Layer]:
    return build_program(attention, hp).layers


def state_manifest(attention: bool = True, hp: dict = HP):
    """(key, shape) of every state_dict entry, in the reference model's order (conv.weight/bias, norm.*, act.weight per
    Convolution; `conv.unit{i}` then `residual` per ResidualUnit; `0` / `1.submodule.{0,1,2}` / `2` per level)."""
    ch, ks, sks = hp["channels"], hp["kernel_sizes"], hp["sample_kernel_sizes"]

    def convolution(p, cin, cout, k, transposed=False, bare=False):
        out = [(p + ".conv.weight", (cin, cout, *k) if transposed else (cout, cin, *k)), (p + ".conv.bias", (cout,))]
        if not bare:
            out += [(p + ".norm.weight", (cout,)), (p + ".norm.bias", (cout,)), (p + ".norm.running_mean", (cout,)), (p + ".norm.running_var", (cout,)),
                    (p + ".norm.num_batches_tracked", ()), (p + ".act.weight", (1,))]
        return out

    def residual_unit(p, cin, cout, k, subunits, last_conv_only=False):
        out, c = [], cin
        for su in range(subunits):
            out += convolution(f"{p}.conv.unit{su}", c, cout, k, bare=last_conv_only and su == subunits - 1)
            c = cout
        if cin != cout:
            out += [(p + ".residual.weight", (cout, cin, 1, 1, 1)), (p + ".residual.bias", (cout,))]
        return out

    def att(p, c, k):
        return convolution(p + ".conv1", c, c // 2, k, bare=True) + convolution(p + ".conv2", c // 2, 1, k, bare=True)

    def block(p, inc, outc, lvl, is_top):
        c, k, sk = ch[lvl], ks[lvl], sks[lvl]
        out = residual_unit(p + ".0", inc, c, k, hp["num_res_units"])
        sub = p + ".1.submodule"
        out += convolution(sub + ".0", c, c, sk)
        if lvl + 2 < len(ch):
            out += block(sub + ".1", c, ch[lvl + 1], lvl + 1, False)
        elif attention:
            out += att(sub + ".1.0.0", c, ks[lvl + 1]) + residual_unit(sub + ".1.1", c, ch[lvl + 1], ks[lvl + 1], hp["num_res_units"])
        else:
            out += residual_unit(sub + ".1", c, ch[lvl + 1], ks[lvl + 1], hp["num_res_units"])
        out += convolution(sub + ".2", ch[lvl + 1], c, sk, transposed=True)
        if attention:
            out += att(p + ".2.0.0", 2 * c, k) + residual_unit(p + ".2.1", 2 * c, outc, k, 1, last_conv_only=is_top)
        else:
            out += residual_unit(p + ".2", 2 * c, outc, k, 1, last_conv_only=is_top)
        return out

    return block("model", hp["in_channels"], hp["out_channels"], 0, True)
