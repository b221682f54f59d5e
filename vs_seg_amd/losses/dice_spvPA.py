"""`Dice_spvPA` — drop-in for ref:params/losses/dice_spvPA.py:170-297 on fused gfx950 kernels.

loss = sum_l (1/L) Dice(att_{L-1-l}, G_l) + Dice_softmax_onehot(logits, target; weight w),  G_{l+1} = MaxPool3d(G_l),
w = 0.6 |softmax(x) - onehot(target)| + 0.4 (differentiable, ref :279-283).

Forward = label pyramid + per-sample fp64 reductions (logits read once); backward = one elementwise kernel for the
logits and one per supervised attention map.  Returns a 0-dim tensor supporting `.backward()` / `.item()`
(ref call sites: params/VSparams.py:460-463, 488-492).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.nn.modules.loss import _Loss

from .. import _lib as L


def _cl_logits(x: torch.Tensor) -> torch.Tensor:
    """[B,2,X,Y,Z] -> contiguous channels-last [B,X,Y,Z,2] fp32 (zero-copy when the tensor came from vs_seg_amd's network)."""
    v = x.detach().permute(0, 2, 3, 4, 1)
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.to(torch.float32).contiguous()
    return v


def _c1(x: torch.Tensor) -> torch.Tensor:
    v = x.detach()
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.to(torch.float32).contiguous()
    return v


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, supervised, hardness, *atts):
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        dev = logits.device
        B, Cc, X, Y, Z = logits.shape
        if Cc != 2:
            raise NotImplementedError("Dice_spvPA HIP path: 2-class logits (the configuration VSparams uses)")
        if target.shape != (B, 1, X, Y, Z):
            raise AssertionError(f"ground truth has differing shape ({tuple(target.shape)}) from input ({tuple(logits.shape)})")
        lg, lab = _cl_logits(logits), _c1(target)
        nvox = X * Y * Z
        nl = len(atts) if supervised else 0
        sums = torch.empty(B * 6 + max(nl, 1) * B * 3, dtype=torch.float64, device=dev)  # [pred | att levels], zeroed through the C ABI (no ATen fill kernels on the step)
        L.check(lib.vsseg_memset_zero(sums.data_ptr(), sums.numel() * 8, stream), "memset_zero")
        pred_sums, att_sums = sums[: B * 6], sums[B * 6 :]
        L.check(lib.vsseg_dice_pred_sums(lg.data_ptr(), 2, lab.data_ptr(), B, nvox, int(hardness), pred_sums.data_ptr(), stream), "dice_pred_sums")
        labels: List[torch.Tensor] = []
        amaps = []
        if nl:
            g, gdims = lab, (X, Y, Z)
            for level in range(nl):  # finest attention map first (ref :256-277)
                a = atts[nl - level - 1]
                adims = tuple(a.shape[2:])
                if adims != gdims:
                    assert all(x % y == 0 for x, y in zip(gdims, adims)), "attention-map pyramid must divide (ref dice_spvPA.py:273)"
                    ratio = tuple(x // y for x, y in zip(gdims, adims))
                    g2 = torch.empty((B, 1, *adims), dtype=torch.float32, device=dev)
                    L.check(lib.vsseg_maxpool_label(g.data_ptr(), B, L.i3(gdims), L.i3(ratio), g2.data_ptr(), stream), "maxpool_label")
                    g, gdims = g2, adims
                ac = _c1(a)
                if tuple(ac.shape) != (B, 1, *adims):
                    raise AssertionError(f"ground truth has differing shape ({(B, 1, *gdims)}) from input ({tuple(ac.shape)})")
                L.check(lib.vsseg_dice_att_sums(ac.data_ptr(), g.data_ptr(), B, adims[0] * adims[1] * adims[2], att_sums.data_ptr() + 8 * level * B * 3, stream), "dice_att_sums")
                labels.append(g)
                amaps.append(ac)
        loss = torch.empty((), dtype=torch.float32, device=dev)  # vsseg_dice_finalize assigns the loss and every coefficient it is asked for
        coef = torch.empty(B * 4 + max(nl, 1) * B * 2, dtype=torch.float32, device=dev)
        L.check(lib.vsseg_dice_finalize(pred_sums.data_ptr(), att_sums.data_ptr(), B, nl, loss.data_ptr(), coef.data_ptr(), stream), "dice_finalize")
        ctx.lg, ctx.lab, ctx.labels, ctx.coef, ctx.hardness, ctx.nl, ctx.shape = lg, lab, labels, coef, int(hardness), nl, (B, X, Y, Z)
        ctx.att_shapes = [tuple(a.shape) for a in atts]
        return loss

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        B, X, Y, Z = ctx.shape
        dev = ctx.lg.device
        gs = gout.detach().to(torch.float32).contiguous()
        dlog = torch.empty((B, X, Y, Z, 2), dtype=torch.float32, device=dev)
        L.check(lib.vsseg_dice_pred_bwd(ctx.lg.data_ptr(), 2, ctx.lab.data_ptr(), B, X * Y * Z, ctx.hardness, ctx.coef.data_ptr(), gs.data_ptr(), dlog.data_ptr(), stream), "dice_pred_bwd")
        datts: List[Optional[torch.Tensor]] = [None] * len(ctx.att_shapes)
        for level in range(ctx.nl):
            i = ctx.nl - level - 1
            shp = ctx.att_shapes[i]
            d = torch.empty(shp, dtype=torch.float32, device=dev)
            nv = shp[2] * shp[3] * shp[4]
            L.check(lib.vsseg_dice_att_bwd(ctx.labels[level].data_ptr(), B, nv, ctx.coef.data_ptr() + 4 * (B * 4 + level * B * 2), 1.0 / ctx.nl, gs.data_ptr(), d.data_ptr(), stream), "dice_att_bwd")
            datts[i] = d
        return (dlog.permute(0, 4, 1, 2, 3), None, None, None, *datts)


class Dice_spvPA(_Loss):
    def __init__(self, include_background: bool = True, to_onehot_y: bool = False, sigmoid: bool = False, softmax: bool = False, other_act=None, squared_pred: bool = False,
                 jaccard: bool = False, reduction="mean", supervised_attention=True, hardness_weighting=True) -> None:
        super().__init__(reduction=getattr(reduction, "value", reduction))
        if other_act is not None and not callable(other_act):
            raise TypeError(f"other_act must be None or callable but is {type(other_act).__name__}.")
        if int(sigmoid) + int(softmax) + int(other_act is not None) > 1:
            raise ValueError("Incompatible values: more than 1 of [sigmoid=True, softmax=True, other_act is not None].")
        # the reference's Dice_spvPA.forward ignores every option except the two below (it builds its inner Dice objects with fixed options)
        self.include_background, self.to_onehot_y, self.sigmoid, self.softmax = include_background, to_onehot_y, sigmoid, softmax
        self.other_act, self.squared_pred, self.jaccard = other_act, squared_pred, jaccard
        self.supervised_attention = supervised_attention
        self.hardness_weighting = hardness_weighting

    def forward(self, input, target: torch.Tensor, smooth: float = 1e-5) -> torch.Tensor:
        x, att_maps = input
        if not x.is_cuda:
            raise RuntimeError("vs_seg_amd.Dice_spvPA runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
        if smooth != 1e-5:
            raise NotImplementedError("smooth is fixed at 1e-5 (the value the reference always uses)")
        atts: Sequence[torch.Tensor] = list(att_maps) if self.supervised_attention else []
        return _DiceFn.apply(x, target, bool(self.supervised_attention), bool(self.hardness_weighting), *atts)
