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


class _State:
    """What the forward leaves for the backward: channels-last logits, labels (full resolution + the pyramid), the coefficients of vsseg_dice_finalize."""
    __slots__ = ("lg", "lab", "labels", "coef", "hardness", "nl", "shape", "att_shapes", "loss")


def _dice_forward(logits, target, supervised, hardness, atts) -> _State:
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
    labels: List[torch.Tensor] = []
    plan = _fused_plan(B, (X, Y, Z), [tuple(a.shape) for a in atts], lg, lab, [_c1(a) for a in atts]) if nl else None
    if plan is not None:
        # fewer passes and launches (csrc/loss.hip): a level whose next level pools (2, 2, 1) computes its sums, the logits' sums (finest level) and the next level's label
        # in one pass; the remaining coarse levels are one launch.  Same sums (fixed-point, order-independent up to the fp32 partial sums of a thread).
        g, gdims = lab, (X, Y, Z)
        for level in range(plan):
            ac = _c1(atts[nl - level - 1])
            ndims = tuple(atts[nl - level - 2].shape[2:])
            g2 = torch.empty((B, 1, *ndims), dtype=torch.float32, device=dev)
            L.check(lib.vsseg_dice_level_sums(lg.data_ptr() if level == 0 else None, ac.data_ptr(), g.data_ptr(), B, L.i3(gdims), int(hardness), pred_sums.data_ptr(), att_sums.data_ptr() + 8 * level * B * 3,
                                              g2.data_ptr(), stream), "dice_level_sums")
            labels.append(g)
            g, gdims = g2, ndims
        td = L.DiceTailDesc()
        td.n, td.nlevels, td.src, td.sdims, td.sums = B, nl - plan, g.data_ptr(), L.i3(gdims), att_sums.data_ptr() + 8 * plan * B * 3
        for j, level in enumerate(range(plan, nl)):
            a = atts[nl - level - 1]
            adims = tuple(a.shape[2:])
            assert all(x % y == 0 for x, y in zip(gdims, adims)), "attention-map pyramid must divide (ref dice_spvPA.py:273)"
            gl = g if adims == gdims else torch.empty((B, 1, *adims), dtype=torch.float32, device=dev)
            td.att[j], td.label[j] = _c1(a).data_ptr(), (None if gl is g else gl.data_ptr())
            td.dims[j][0], td.dims[j][1], td.dims[j][2] = adims
            labels.append(gl)
        L.check(lib.vsseg_dice_tail_sums(td, stream), "dice_tail_sums")
    else:
        L.check(lib.vsseg_dice_pred_sums(lg.data_ptr(), 2, lab.data_ptr(), B, nvox, int(hardness), pred_sums.data_ptr(), stream), "dice_pred_sums")
    if nl and plan is None:
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
    loss = torch.empty((), dtype=torch.float32, device=dev)  # vsseg_dice_finalize assigns the loss and every coefficient it is asked for
    coef = torch.empty(B * 4 + max(nl, 1) * B * 2, dtype=torch.float32, device=dev)
    L.check(lib.vsseg_dice_finalize(pred_sums.data_ptr(), att_sums.data_ptr(), B, nl, loss.data_ptr(), coef.data_ptr(), stream), "dice_finalize")
    st = _State()
    st.lg, st.lab, st.labels, st.coef, st.hardness, st.nl, st.shape = lg, lab, labels, coef, int(hardness), nl, (B, X, Y, Z)
    st.att_shapes = [tuple(a.shape) for a in atts]
    st.loss = loss
    return st


def _fused_plan(B, dims, att_shapes, lg, lab, acs) -> Optional[int]:
    """How many of the finest levels take the fused pass of csrc/loss.hip (vsseg_dice_level_sums: the level is made of 2 x 2 x 4 blocks and the next one pools (2, 2, 1)),
    the rest going to the one tail launch; None when not even the finest level qualifies or a level is too many for the tail (the generic sequence is used)."""
    nl = len(att_shapes)
    if tuple(att_shapes[nl - 1][2:]) != tuple(dims) or any(t.data_ptr() % 16 for t in (lg, lab, *acs)):
        return None
    k, g = 0, tuple(dims)
    while k + 1 < nl:
        nxt = tuple(att_shapes[nl - k - 2][2:])
        if tuple(att_shapes[nl - k - 1][2:]) != g or g[0] % 2 or g[1] % 2 or g[2] % 4 or nxt != (g[0] // 2, g[1] // 2, g[2]) or (B * nxt[0] * nxt[1] * nxt[2]) % 4:
            break
        k, g = k + 1, nxt
    if k == 0 or nl - k > L.DICE_MAX_LEVELS:
        return None
    if any(s[2] * s[3] * s[4] >= (1 << 22) for s in att_shapes[: nl - k]):  # the tail launch takes levels of < 2^22 voxels per sample
        return None
    return k


def _att_backward(st: _State, level: int, gs_ptr, dst: torch.Tensor):
    B = st.shape[0]
    shp = st.att_shapes[st.nl - level - 1]
    L.check(L.lib().vsseg_dice_att_bwd(st.labels[level].data_ptr(), B, shp[2] * shp[3] * shp[4], st.coef.data_ptr() + 4 * (B * 4 + level * B * 2), 1.0 / st.nl, gs_ptr, dst.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "dice_att_bwd")


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, supervised, hardness, *atts):
        ctx.st = st = _dice_forward(logits, target, supervised, hardness, atts)
        return st.loss

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        st = ctx.st
        stream = torch.cuda.current_stream().cuda_stream
        B, X, Y, Z = st.shape
        dev = st.lg.device
        gs = gout.detach().to(torch.float32).contiguous()
        dlog = torch.empty((B, X, Y, Z, 2), dtype=torch.float32, device=dev)
        L.check(lib.vsseg_dice_pred_bwd(st.lg.data_ptr(), 2, st.lab.data_ptr(), B, X * Y * Z, st.hardness, st.coef.data_ptr(), gs.data_ptr(), dlog.data_ptr(), stream), "dice_pred_bwd")
        datts: List[Optional[torch.Tensor]] = [None] * len(st.att_shapes)
        for level in range(st.nl):
            i = st.nl - level - 1
            d = torch.empty(st.att_shapes[i], dtype=torch.float32, device=dev)
            _att_backward(st, level, gs.data_ptr(), d)
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

    def forward_backward_into(self, input, target: torch.Tensor, landing):
        """The loss AND its gradients in one call, outside autograd (the fused train step of vs_seg_amd.parallel.DataParallelTrainer): d(loss)/d(logits) and
        d(loss)/d(att_i) are written where `landing` says — `(descriptor of the staged gradient of the logits, [fp32 buffer per attention map or None])`, as
        UNet2d5_spvPA.train_forward_landing hands it out — in the layout and dtype the network's backward reads, instead of fp32 tensors that the backward
        would copy and cast (7 passes per step).  The values are those of `loss.backward()` (upstream gradient 1).  Returns (loss, indices of the attention
        maps whose buffer was written)."""
        x, att_maps = input
        if not x.is_cuda:
            raise RuntimeError("vs_seg_amd.Dice_spvPA runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
        atts: Sequence[torch.Tensor] = list(att_maps) if self.supervised_attention else []
        glogits_dst, gatt_bufs = landing
        with torch.no_grad():
            st = _dice_forward(x, target, bool(self.supervised_attention), bool(self.hardness_weighting), atts)
            B, X, Y, Z = st.shape
            stream = torch.cuda.current_stream().cuda_stream
            L.check(L.lib().vsseg_dice_pred_bwd_to(st.lg.data_ptr(), 2, st.lab.data_ptr(), B, X * Y * Z, st.hardness, st.coef.data_ptr(), None, glogits_dst, stream), "dice_pred_bwd_to")
            written, coarse = [], []
            for level in range(st.nl):
                i = st.nl - level - 1
                buf = gatt_bufs[i]
                if buf is None:
                    continue  # this map has no gradient path in the network
                nv = st.att_shapes[i][2] * st.att_shapes[i][3] * st.att_shapes[i][4]
                if buf.numel() != B * nv or buf.dtype != torch.float32:
                    raise AssertionError(f"gradient buffer of attention map {i} does not match its shape {st.att_shapes[i]}")
                if nv * B >= (1 << 22) or len(coarse) == L.DICE_MAX_LEVELS:
                    _att_backward(st, level, None, buf)  # a level large enough to fill the GPU: its own launch
                else:
                    coarse.append((level, buf, nv))
                written.append(i)
            if coarse:  # the coarse levels: one launch
                bd = L.DiceBwdLevelsDesc()
                bd.n, bd.nlevels, bd.gscale = B, len(coarse), None
                for j, (level, buf, nv) in enumerate(coarse):
                    bd.label[j], bd.datt[j], bd.nvox[j] = st.labels[level].data_ptr(), buf.data_ptr(), nv
                    bd.coef[j] = st.coef.data_ptr() + 4 * (B * 4 + level * B * 2)
                L.check(L.lib().vsseg_dice_att_bwd_levels(bd, stream), "dice_att_bwd_levels")
        return st.loss, written
