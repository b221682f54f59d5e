"""CPU oracle for the VS_Seg hot path (2.5D attention U-Net fwd/bwd, Dice_spvPA loss, Adam, sliding window).

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE ***
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.  The product
path (`vs_seg_amd`) never imports it and fails loudly when its HIP extension is missing.

What it is: a *functional* fp32 restatement (plain `torch.nn.functional` calls on CPU tensors, explicit state-dict
keys, no `nn.Module`s) of the algorithm implemented by the reference's L3 files, written from their behaviour:

  * network      ref:params/networks/nets/unet2d5_spvPA.py:24-206, ref:params/networks/blocks/convolutions.py:22-255,
                 ref:params/networks/blocks/attentionblock.py:6-47, hyper-parameters ref:params/VSparams.py:343-374
  * loss         ref:params/losses/dice_spvPA.py:90-167 (Dice), :238-297 (Dice_spvPA)
  * optimiser    torch.optim.Adam(lr, weight_decay) as constructed at ref:params/VSparams.py:388-391
  * hard Dice    ref:params/VSparams.py:393-408
  * sliding window  call site ref:params/VSparams.py:568-574; the algorithm lives in MONAI 0.4.0
                 (`monai/inferers/utils.py`, pinned by ref:requirements.txt:7, NOT vendored) and is restated from its
                 published behaviour.  **Parity unpinned** for that one function: no reference
                 test or importable implementation exists in this image; it is pinned by known-answer window tables.

Pinning: `tests/test_oracle_golden.py` checks every function here against golden vectors produced by importing the
reference's own model/loss files in the build container (`tests/golden/make_goldens.py`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# Hyper-parameters hard-coded by the reference at ref:params/VSparams.py:343-374
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
BN_EPS = 1e-5  # torch BatchNorm3d default (ref:params/networks/blocks/convolutions.py:152 passes no args)
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------------------------------
# state-dict manifest (key names/shapes/order of the reference model; ref:params/networks/nets/unet2d5_spvPA.py:56-93)
# ----------------------------------------------------------------------------------------------------------------
def _convolution_keys(p: str, cin: int, cout: int, k, transposed=False, conv_only=False, plain=False):
    wshape = (cin, cout, *k) if transposed else (cout, cin, *k)
    out = [(p + ".conv.weight", wshape), (p + ".conv.bias", (cout,))]
    if not conv_only and not plain:
        out += [
            (p + ".norm.weight", (cout,)),
            (p + ".norm.bias", (cout,)),
            (p + ".norm.running_mean", (cout,)),
            (p + ".norm.running_var", (cout,)),
            (p + ".norm.num_batches_tracked", ()),
            (p + ".act.weight", (1,)),
        ]
    return out


def _residual_unit_keys(p: str, cin: int, cout: int, k, subunits: int, last_conv_only=False):
    out = []
    c = cin
    for su in range(subunits):
        out += _convolution_keys(f"{p}.conv.unit{su}", c, cout, k, conv_only=last_conv_only and su == subunits - 1)
        c = cout
    if cin != cout:
        out += [(p + ".residual.weight", (cout, cin, 1, 1, 1)), (p + ".residual.bias", (cout,))]
    return out


def _att_keys(p: str, c: int, k):
    return _convolution_keys(p + ".conv1", c, c // 2, k, plain=True) + _convolution_keys(p + ".conv2", c // 2, 1, k, plain=True)


def manifest(attention: bool = True, hp: dict = HP) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in the reference's state_dict order."""
    ch, st, ks, sks = hp["channels"], hp["strides"], hp["kernel_sizes"], hp["sample_kernel_sizes"]

    def block(p, inc, outc, lvl, is_top):
        c, k, sk = ch[lvl], ks[lvl], sks[lvl]
        out = _residual_unit_keys(p + ".0", inc, c, k, hp["num_res_units"])
        sub = p + ".1.submodule"
        out += _convolution_keys(sub + ".0", c, c, sk)
        if lvl + 2 < len(ch):
            out += block(sub + ".1", c, ch[lvl + 1], lvl + 1, False)
        else:  # bottom layer
            if attention:
                out += _att_keys(sub + ".1.0.0", c, ks[lvl + 1])
                out += _residual_unit_keys(sub + ".1.1", c, ch[lvl + 1], ks[lvl + 1], hp["num_res_units"])
            else:
                out += _residual_unit_keys(sub + ".1", c, ch[lvl + 1], ks[lvl + 1], hp["num_res_units"])
        out += _convolution_keys(sub + ".2", ch[lvl + 1], c, sk, transposed=True)
        if attention:
            out += _att_keys(p + ".2.0.0", 2 * c, k)
            out += _residual_unit_keys(p + ".2.1", 2 * c, outc, k, 1, last_conv_only=is_top)
        else:
            out += _residual_unit_keys(p + ".2", 2 * c, outc, k, 1, last_conv_only=is_top)
        return out

    return block("model", hp["in_channels"], hp["out_channels"], 0, True)


def seeded_state_dict(attention: bool = True, seed: int = 0, hp: dict = HP) -> Dict[str, torch.Tensor]:
    """Deterministic, well-conditioned weights drawn from one numpy stream in state_dict order (goldens store no weights)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in manifest(attention, hp):
        leaf = key.rsplit(".", 2)[-2:]
        if key.endswith("num_batches_tracked"):
            v = np.zeros((), np.int64)
        elif leaf == ["norm", "weight"] or key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == ["norm", "bias"] or key.endswith("running_mean"):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == ["act", "weight"]:
            v = rng.uniform(0.1, 0.4, shape).astype(np.float32)
        elif key.endswith("bias"):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:  # conv / convT / residual weight: unit-gain fan-in scaling
            is_t = key.endswith(".2.conv.weight") and ".submodule.2." in key
            fan_in = (shape[0] if is_t else shape[1]) * int(np.prod(shape[2:]))
            if is_t:
                fan_in = max(fan_in // 4, 1)  # stride-2 transposed conv: ~1/4 (or 1/8) of the taps hit each output
            v = (rng.standard_normal(shape) / math.sqrt(fan_in)).astype(np.float32)
        sd[key] = torch.from_numpy(np.asarray(v))
    return sd


# ----------------------------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------------------------
def _same_pad(k):
    return tuple((kk - 1) // 2 for kk in k)  # MONAI same_padding, dilation 1 (ref:.../convolutions.py:85)


class Ctx:
    """Per-call state: mode, dropout masks (explicit, so parity tests can inject the HIP path's masks), BN updates."""

    def __init__(self, train: bool, dropout_p: float, masks: Optional[Dict[str, torch.Tensor]] = None, rng: Optional[torch.Generator] = None):
        self.train, self.p, self.masks, self.rng = train, dropout_p, masks, rng
        self.bn_updates: Dict[str, torch.Tensor] = {}
        self.used_masks: Dict[str, torch.Tensor] = {}


def convolution(x, sd, p, ctx: Ctx, *, stride=(1, 1, 1), kernel=(3, 3, 3), transposed=False, conv_only=False, plain_act=None):
    """ref:params/networks/blocks/convolutions.py:22-156 — (Conv|ConvT) -> BatchNorm -> Dropout -> PReLU.

    `plain_act` ('relu'/'sigmoid') selects the attention-block flavour: conv + activation, no norm, no dropout
    (ref:params/networks/blocks/attentionblock.py:10-29).
    """
    pad = _same_pad(kernel)
    w, b = sd[p + ".conv.weight"], sd[p + ".conv.bias"]
    if transposed:
        # output_padding = stride + 2*pad - (k-1) - 1  (ref:.../convolutions.py:117-123)
        opad = tuple(s + 2 * pp - (k - 1) - 1 for s, pp, k in zip(stride, pad, kernel))
        y = F.conv_transpose3d(x, w, b, stride=stride, padding=pad, output_padding=opad)
    else:
        y = F.conv3d(x, w, b, stride=stride, padding=pad)
    if plain_act is not None:
        return torch.relu(y) if plain_act == "relu" else torch.sigmoid(y)
    if conv_only:
        return y
    g, beta = sd[p + ".norm.weight"], sd[p + ".norm.bias"]
    rm, rv = sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"]
    if ctx.train:
        red = (0, 2, 3, 4)
        n = y.numel() // y.shape[1]
        mean = y.mean(red)
        var = y.var(red, unbiased=False)
        with torch.no_grad():  # running stats use the unbiased variance (torch BatchNorm semantics)
            ctx.bn_updates[p + ".norm.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            ctx.bn_updates[p + ".norm.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
            ctx.bn_updates[p + ".norm.num_batches_tracked"] = sd[p + ".norm.num_batches_tracked"] + 1
        sh = (1, -1, 1, 1, 1)
        y = (y - mean.view(sh)) / torch.sqrt(var.view(sh) + BN_EPS) * g.view(sh) + beta.view(sh)
    else:
        y = F.batch_norm(y, rm, rv, g, beta, False, BN_MOMENTUM, BN_EPS)
    if ctx.train and ctx.p > 0:
        if ctx.masks is not None:
            m = ctx.masks[p]
        else:
            m = (torch.rand(y.shape, generator=ctx.rng) >= ctx.p).to(y.dtype)
        ctx.used_masks[p] = m
        y = y * m / (1.0 - ctx.p)
    return F.prelu(y, sd[p + ".act.weight"])


def residual_unit(x, sd, p, ctx, *, kernel, subunits, last_conv_only=False):
    """ref:params/networks/blocks/convolutions.py:159-255 (strides are always 1 in this network)."""
    cin = x.shape[1]
    cx = x
    for su in range(subunits):
        cx = convolution(cx, sd, f"{p}.conv.unit{su}", ctx, kernel=kernel, conv_only=last_conv_only and su == subunits - 1)
    if (p + ".residual.weight") in sd:
        res = F.conv3d(x, sd[p + ".residual.weight"], sd[p + ".residual.bias"])
    else:
        assert cin == cx.shape[1]
        res = x
    return cx + res


def attention(x, sd, p, ctx, *, kernel):
    """AttentionBlock1 + AttentionBlock2 (ref:params/networks/blocks/attentionblock.py:6-47): returns (gated x, att)."""
    a = convolution(x, sd, p + ".conv1", ctx, kernel=kernel, plain_act="relu")
    a = convolution(a, sd, p + ".conv2", ctx, kernel=kernel, plain_act="sigmoid")
    return a * x + x, a


def unet_forward(sd, x, *, train=False, attention_module=True, dropout_p=None, masks=None, rng=None, hp=HP):
    """ref:params/networks/nets/unet2d5_spvPA.py:56-93,204-206.  Returns (logits, [att coarsest..finest], ctx)."""
    ch, st, ks, sks = hp["channels"], hp["strides"], hp["kernel_sizes"], hp["sample_kernel_sizes"]
    ctx = Ctx(train, hp["dropout"] if dropout_p is None else dropout_p, masks, rng)
    atts: List[torch.Tensor] = []

    def block(x, p, lvl, is_top):
        k, sk, s = ks[lvl], sks[lvl], st[lvl]
        d = residual_unit(x, sd, p + ".0", ctx, kernel=k, subunits=hp["num_res_units"])
        sub = p + ".1.submodule"
        y = convolution(d, sd, sub + ".0", ctx, stride=s, kernel=sk)
        if lvl + 2 < len(ch):
            y = block(y, sub + ".1", lvl + 1, False)
        else:
            kb = ks[lvl + 1]
            if attention_module:
                y, a = attention(y, sd, sub + ".1.0.0", ctx, kernel=kb)
                atts.append(a)  # hook order: bottleneck first (ref:.../unet2d5_spvPA.py:101-104)
                y = residual_unit(y, sd, sub + ".1.1", ctx, kernel=kb, subunits=hp["num_res_units"])
            else:
                y = residual_unit(y, sd, sub + ".1", ctx, kernel=kb, subunits=hp["num_res_units"])
        y = convolution(y, sd, sub + ".2", ctx, stride=s, kernel=sk, transposed=True)
        c = torch.cat([d, y], 1)  # MONAI SkipConnection (cat on dim 1)
        if attention_module:
            c, a = attention(c, sd, p + ".2.0.0", ctx, kernel=k)
            atts.append(a)
            return residual_unit(c, sd, p + ".2.1", ctx, kernel=k, subunits=1, last_conv_only=is_top)
        return residual_unit(c, sd, p + ".2", ctx, kernel=k, subunits=1, last_conv_only=is_top)

    return block(x, "model", 0, True), atts, ctx


# ----------------------------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------------------------
def one_hot(labels, num_classes):
    shape = list(labels.shape)
    shape[1] = num_classes
    dt = labels.dtype if labels.is_floating_point() else torch.float32
    return torch.zeros(shape, dtype=dt).scatter_(1, labels.long(), 1)


def dice(inp, target, *, softmax=False, to_onehot_y=False, hardness_weight=None, include_background=True, smooth=1e-5):
    """ref:params/losses/dice_spvPA.py:90-167 with the options the hot path uses (mean reduction)."""
    n_ch = inp.shape[1]
    if softmax and n_ch > 1:
        inp = torch.softmax(inp, 1)
    if to_onehot_y and n_ch > 1:
        target = one_hot(target, n_ch)
    if not include_background and n_ch > 1:
        target, inp = target[:, 1:], inp[:, 1:]
    assert target.shape == inp.shape, f"ground truth has differing shape ({target.shape}) from input ({inp.shape})"
    ax = list(range(2, inp.dim()))
    w = hardness_weight
    inter = torch.sum(target * inp if w is None else w * target * inp, ax)
    ground = torch.sum(target if w is None else w * target, ax)
    pred = torch.sum(inp if w is None else w * inp, ax)
    f = 1.0 - (2.0 * inter + smooth) / (ground + pred + smooth)
    return f.mean()


def dice_spvpa(logits, att_maps, target, *, supervised_attention=True, hardness_weighting=True):
    """ref:params/losses/dice_spvPA.py:238-297."""
    total = torch.zeros(())
    if supervised_attention:
        L = len(att_maps)
        g = target
        for level in range(L):
            total = total + dice(att_maps[L - level - 1], g) / L
            if level < L - 1:
                cur, nxt = att_maps[L - level - 1].shape, att_maps[L - level - 2].shape
                assert all(a % b == 0 for a, b in zip(cur, nxt))
                ratio = [a // b for a, b in zip(cur, nxt)][2:5]
                g = F.max_pool3d(g, kernel_size=ratio, stride=ratio)
    w = None
    if hardness_weighting:  # not detached: gradients flow through the weight (ref :279-283)
        w = 0.6 * torch.abs(torch.softmax(logits, 1) - one_hot(target, logits.shape[1])) + 0.4
    return total + dice(logits, target, softmax=True, to_onehot_y=True, hardness_weight=w)


def compute_dice_score(pred, label):
    """ref:params/VSparams.py:393-408 — hard Dice of argmax vs label on the foreground channel, shape [1,1]."""
    n = pred.shape[1]
    y = one_hot(torch.argmax(pred, 1, keepdim=True), n)
    return (1 - dice(y, label, to_onehot_y=True, include_background=False)).reshape(1, 1)


# ----------------------------------------------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, *, lr=1e-4, wd=1e-7, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (coupled L2 weight decay, bias-corrected), one tensor; `step` is 1-based.  Returns (p, m, v)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1**step, 1 - b2**step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# ----------------------------------------------------------------------------------------------------------------
# sliding window inference (MONAI 0.4.0 behaviour; parity unpinned)
# ----------------------------------------------------------------------------------------------------------------
def swi_geometry(image_size: Sequence[int], roi_size: Sequence[int], overlap: float):
    """Integer window arithmetic.  Returns (padded_size, pad_before, scan_interval, window starts in reference order)."""
    roi = [r if r > 0 else s for r, s in zip(roi_size, image_size)]  # fall_back_tuple
    padded = [max(s, r) for s, r in zip(image_size, roi)]
    pad_before = [(p - s) // 2 for p, s in zip(padded, image_size)]  # half, diff-half
    interval = [r if r == p else max(int(r * (1 - overlap)), 1) for r, p in zip(roi, padded)]
    starts_per_dim = []
    for size, r, iv in zip(padded, roi, interval):
        num = int(math.ceil(float(size) / iv))
        scan = next(d for d in range(num) if d * iv + r >= size) + 1
        starts_per_dim.append([d * iv - max(d * iv + r - size, 0) for d in range(scan)])
    starts = [(a, b, c) for a in starts_per_dim[0] for b in starts_per_dim[1] for c in starts_per_dim[2]]
    return roi, padded, pad_before, interval, starts


def gaussian_importance_map(roi: Sequence[int], sigma_scale=0.125) -> torch.Tensor:
    """Separable erf-approximated Gaussian centred at roi//2, /max, zeros -> min non-zero."""
    m = torch.zeros(tuple(roi), dtype=torch.float32)
    m[tuple(r // 2 for r in roi)] = 1.0
    for d, r in enumerate(roi):
        sigma = r * sigma_scale
        tail = int(max(sigma * 4.0, 0.5) + 0.5)
        xs = torch.arange(-tail, tail + 1, dtype=torch.float32)
        t = 0.70710678 / abs(sigma)
        taps = (0.5 * ((t * (xs + 0.5)).erf() - (t * (xs - 0.5)).erf())).clamp(min=0)
        shape = [1, 1, 1, 1, 1]
        shape[2 + d] = taps.numel()
        pad = [0, 0, 0]
        pad[d] = tail
        m = F.conv3d(m[None, None], taps.view(shape), padding=pad)[0, 0]
    m = m / m.max()
    m[m == 0] = m[m != 0].min()
    return m


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant", return_windows=False):
    """MONAI 0.4.0 `sliding_window_inference` (call site ref:params/VSparams.py:568-574)."""
    B = inputs.shape[0]
    img = list(inputs.shape[2:])
    roi, padded, pad_before, interval, starts = swi_geometry(img, roi_size, overlap)
    pad = []
    for k in range(2, -1, -1):
        diff = max(roi[k] - img[k], 0)
        pad += [diff // 2, diff - diff // 2]
    x = F.pad(inputs, pad, mode="constant", value=0.0)
    imap = gaussian_importance_map(roi) if mode == "gaussian" else torch.ones(tuple(roi))
    slices = [(b, s) for b in range(B) for s in starts]
    out = cnt = None
    for g in range(0, len(slices), sw_batch_size):
        grp = slices[g : g + sw_batch_size]
        win = torch.cat([x[b : b + 1, :, s[0] : s[0] + roi[0], s[1] : s[1] + roi[1], s[2] : s[2] + roi[2]] for b, s in grp])
        seg = predictor(win)
        if out is None:
            out = torch.zeros((B, seg.shape[1], *padded), dtype=torch.float32)
            cnt = torch.zeros_like(out)
        for i, (b, s) in enumerate(grp):
            sl = (b, slice(None), slice(s[0], s[0] + roi[0]), slice(s[1], s[1] + roi[1]), slice(s[2], s[2] + roi[2]))
            out[sl] += imap * seg[i]
            cnt[sl] += imap
    out = out / cnt
    out = out[:, :, pad_before[0] : pad_before[0] + img[0], pad_before[1] : pad_before[1] + img[1], pad_before[2] : pad_before[2] + img[2]]
    return (out, starts) if return_windows else out
