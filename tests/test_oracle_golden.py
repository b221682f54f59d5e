"""Pin the CPU oracle against golden vectors captured from the reference's own code (tests/golden/make_goldens.py)."""
import json

import numpy as np
import pytest
import torch

from oracle import vsseg_oracle as O
from tests.helpers import check_summary, load, synth_input, synth_label

TOL = 2e-5


def _close(got, want, rel):
    """|got-want| <= rel * max|want| (gradients span orders of magnitude; fp32 summation noise scales with the max)."""
    np.testing.assert_allclose(got, want, atol=rel * float(np.abs(want).max()) + 1e-30, rtol=rel)
  # fp32 CPU restatement vs fp32 CPU reference: summation-order noise only


def test_manifest_matches_reference_state_dict(golden_dir):
    man = json.load(open(f"{golden_dir}/manifest.json"))
    for att, key in ((True, "attention"), (False, "no_attention")):
        assert [(k, tuple(s)) for k, s in man[key]] == [(k, tuple(s)) for k, s in O.manifest(att)]
    assert len(man["attention"]) == 256
    n_param = sum(int(np.prod(s)) for k, s in man["attention"] if "running" not in k and "num_batches" not in k)
    assert n_param == 3453012


@pytest.mark.parametrize("name", ["b2_32x32x8", "b1_64x64x16", "b1_32x32x8_noatt", "b1_128x128x32", "b1_64x32x24", "b1_384x128x128"])
def test_net_eval(name):
    g = load(f"net_eval_{name}.npz")
    att, seed, shape = bool(g["attention"]), int(g["seed"]), tuple(int(v) for v in g["shape"])
    sd = O.seeded_state_dict(att, seed)
    with torch.no_grad():
        logits, atts, _ = O.unet_forward(sd, synth_input(seed, shape), train=False, attention_module=att)
    assert len(atts) == int(g["n_att"])
    if "logits" in g:
        np.testing.assert_allclose(logits.numpy(), g["logits"], atol=TOL * 5, rtol=1e-5)
    check_summary(logits, g["logits_meta"], g["logits_sub"], atol=TOL * 5, rtol=1e-5)
    for i, a in enumerate(atts):
        if f"att{i}" in g:
            np.testing.assert_allclose(a.numpy(), g[f"att{i}"], atol=TOL)
        check_summary(a, g[f"att{i}_meta"], g[f"att{i}_sub"], atol=TOL, rtol=1e-6)


@pytest.mark.parametrize("name", ["b2_32x32x8", "b2_32x32x8_noatt_nohard", "b1_64x64x16"])
def test_net_train_fwd_bwd(name):
    g = load(f"net_train_{name}.npz")
    att, hard, seed, shape = bool(g["attention"]), bool(g["hardness"]), int(g["seed"]), tuple(int(v) for v in g["shape"])
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in O.seeded_state_dict(att, seed).items()}
    x, y = synth_input(seed, shape), synth_label(seed, shape)
    logits, atts, ctx = O.unet_forward(sd, x, train=True, attention_module=att, dropout_p=0.0)
    logits.retain_grad()
    [a.retain_grad() for a in atts]
    loss = O.dice_spvpa(logits, atts, y, supervised_attention=att, hardness_weighting=hard)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=1e-4, rtol=1e-5)
    _close(logits.grad.numpy(), g["dlogits"], 2e-3)
    for i, a in enumerate(atts):
        np.testing.assert_allclose(a.detach().numpy(), g[f"att{i}"], atol=TOL)
        _close(a.grad.numpy(), g[f"datt{i}"], 2e-3)
    sums = json.loads(str(g["grad_sums"]))
    for k, (s, asum, sq) in sums.items():
        gk = sd[k].grad.double().flatten()
        if k.endswith("conv.bias") and k.replace("conv.bias", "norm.weight") in sd:
            # a bias in front of a training-mode BatchNorm has an analytically zero gradient: both sides hold only noise
            assert float(gk.abs().max()) < 1e-6 and sq < 1e-10, k
            continue
        sub = gk[:: max(1, gk.numel() // 64)][:64].float().numpy()
        scale = max(np.sqrt(sq / gk.numel()), 1e-12)
        # 32x32x8 inputs leave 2 values per channel at the bottleneck BatchNorm: ill-conditioned, fp32 noise is amplified
        rel = 3e-2 if (k.endswith("act.weight") or "32x32x8" in name) else 2e-3  # PReLU slope grad = one heavily cancelling full-tensor fp32 sum
        np.testing.assert_allclose(sub, g["gsub:" + k], atol=5 * rel * scale, rtol=rel, err_msg=k)
        if ("gabs:" + k) in g.files:  # PReLU slopes, BatchNorm gamma / beta: the error against the SUM OF |TERMS| the reference summed (round 5) — fp32 rounding of a cancelling sum
            err = np.abs(sub.astype(np.float64) - g["gsub:" + k].astype(np.float64)) / (g["gabs:" + k] + 1e-30)
            assert float(err.max()) < (5e-2 if "32x32x8" in name else 1e-4), (k, float(err.max()))  # (32x32x8: 2 values per bottleneck channel, see above)
        if k.endswith("act.weight"):
            continue
        assert abs(float((gk * gk).sum()) - sq) <= 2e-2 * sq + 1e-20, k
    for k in g.files:
        if k.startswith("bn:"):
            np.testing.assert_allclose(ctx.bn_updates[k[3:]].numpy(), g[k], atol=1e-5, rtol=1e-5, err_msg=k)


def _blk_sd(g, tag, prefix):
    return {prefix + "." + k.split(":p:")[1]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(tag + ":p:")}


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("i,name,k,s,tr,cin", [(0, "c331_s1", (3, 3, 1), (1, 1, 1), False, 8), (1, "c333_s1", (3, 3, 3), (1, 1, 1), False, 8), (2, "c331_s221", (3, 3, 1), (2, 2, 1), False, 16), (3, "c333_s222", (3, 3, 3), (2, 2, 2), False, 16), (4, "t331_s221", (3, 3, 1), (2, 2, 1), True, 16), (5, "t333_s222", (3, 3, 3), (2, 2, 2), True, 16)])
def test_convolution_block(i, name, k, s, tr, cin, mode):
    g = load("blocks.npz")
    tag = f"{name}_{mode}"
    sd = _blk_sd(g, tag, "b")
    for kk in g.files:
        if kk.startswith(tag + ":b:"):
            sd["b." + kk.split(":b:")[1]] = torch.from_numpy(g[kk])
    if mode == "train":  # goldens hold post-step buffers; invert the momentum update is not needed: stats come from the batch
        pass
    x = synth_input(200 + i, (2, cin, 8, 8, 4)).requires_grad_(True)
    ctx = O.Ctx(mode == "train", 0.0)
    if mode == "eval":
        y = O.convolution(x, sd, "b", ctx, stride=s, kernel=k, transposed=tr)
    else:
        y = O.convolution(x, sd, "b", ctx, stride=s, kernel=k, transposed=tr)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + ":y"], atol=TOL)
    y.backward(synth_input(300 + i, tuple(y.shape)))
    np.testing.assert_allclose(x.grad.numpy(), g[tag + ":dx"], atol=5e-5, rtol=1e-4)
    for kk in g.files:
        if kk.startswith(tag + ":g:"):
            np.testing.assert_allclose(sd["b." + kk.split(":g:")[1]].grad.numpy(), g[kk], atol=2e-4, rtol=1e-4, err_msg=kk)


@pytest.mark.parametrize("i,name,cin,sub,last", [(0, "ru_2", 8, 2, False), (1, "ru_1", 16, 1, False), (2, "ru_1_last", 16, 1, True), (3, "ru_2_same", 16, 2, False)])
def test_residual_unit(i, name, cin, sub, last):
    g = load("blocks.npz")
    sd = _blk_sd(g, name, "b")
    for kk in list(sd):  # buffers are not needed in train mode but the keys must exist
        if kk.endswith("norm.weight"):
            c = sd[kk].shape[0]
            sd[kk.replace("weight", "running_mean")] = torch.zeros(c)
            sd[kk.replace("weight", "running_var")] = torch.ones(c)
            sd[kk.replace("weight", "num_batches_tracked")] = torch.zeros((), dtype=torch.long)
    x = synth_input(500 + i, (2, cin, 8, 8, 4)).requires_grad_(True)
    y = O.residual_unit(x, sd, "b", O.Ctx(True, 0.0), kernel=(3, 3, 3), subunits=sub, last_conv_only=last)
    np.testing.assert_allclose(y.detach().numpy(), g[name + ":y"], atol=TOL)
    y.backward(synth_input(600 + i, tuple(y.shape)))
    np.testing.assert_allclose(x.grad.numpy(), g[name + ":dx"], atol=5e-5, rtol=1e-4)
    for kk in g.files:
        if kk.startswith(name + ":g:"):
            np.testing.assert_allclose(sd["b." + kk.split(":g:")[1]].grad.numpy(), g[kk], atol=3e-4, rtol=1e-4, err_msg=kk)


def test_attention_block():
    g = load("blocks.npz")
    sd = _blk_sd(g, "att", "b")
    x = synth_input(701, (2, 16, 8, 8, 4)).requires_grad_(True)
    y, att = O.attention(x, sd, "b", O.Ctx(True, 0.1), kernel=(3, 3, 3))
    att.retain_grad()
    np.testing.assert_allclose(y.detach().numpy(), g["att:y"], atol=TOL)
    np.testing.assert_allclose(att.detach().numpy(), g["att:att"], atol=TOL)
    ((y * synth_input(702, tuple(y.shape))).sum() + (att * synth_input(703, tuple(att.shape))).sum()).backward()
    np.testing.assert_allclose(x.grad.numpy(), g["att:dx"], atol=5e-5, rtol=1e-4)
    for kk in g.files:
        if kk.startswith("att:g:"):
            np.testing.assert_allclose(sd["b." + kk.split(":g:")[1]].grad.numpy(), g[kk], atol=2e-4, rtol=1e-4, err_msg=kk)


@pytest.mark.parametrize("att", [True, False])
@pytest.mark.parametrize("hard", [True, False])
def test_loss(att, hard):
    g = load("loss.npz")
    shape = (2, 1, 32, 32, 8)
    att_shapes = [(2, 1, 1, 1, 1), (2, 1, 2, 2, 2), (2, 1, 4, 4, 4), (2, 1, 8, 8, 8), (2, 1, 16, 16, 8), (2, 1, 32, 32, 8)]
    y = synth_label(31, shape)
    logits = (2.0 * synth_input(32, (2, 2, 32, 32, 8))).requires_grad_(True)
    atts = [torch.sigmoid(synth_input(40 + i, s)).requires_grad_(True) for i, s in enumerate(att_shapes)] if att else []
    loss = O.dice_spvpa(logits, atts, y, supervised_attention=att, hardness_weighting=hard)
    loss.backward()
    tag = f"att{int(att)}_hard{int(hard)}"
    assert abs(float(loss) - float(g[tag + ":loss"])) < 2e-6
    np.testing.assert_allclose(logits.grad.numpy(), g[tag + ":dlogits"], atol=1e-9, rtol=1e-4)
    for i, a in enumerate(atts):
        np.testing.assert_allclose(a.grad.numpy(), g[f"{tag}:datt{i}"], atol=1e-9, rtol=1e-4)


def test_adam():
    g = load("adam.npz")
    p = torch.from_numpy(g["p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step, gr in enumerate(g["grads"], 1):
        p, m, v = O.adam_step(p, torch.from_numpy(gr), m, v, step)
        np.testing.assert_allclose(p.numpy(), g["after"][step - 1], atol=1e-7, rtol=1e-6)


def test_hard_dice_matches_reference_golden():
    """`compute_dice_score` (ref:params/VSparams.py:393-408) pinned by the reference's own DiceLoss class (tests/golden/make_c3_golden.py)."""
    g = load("hard_dice.npz")
    names = sorted({k.split(":")[0] for k in g.files})
    assert {"perfect", "empty_label", "empty_pred", "both_empty"} <= set(names)
    for n in names:
        got = O.compute_dice_score(torch.from_numpy(g[n + ":p"]), torch.from_numpy(g[n + ":label"]))
        assert tuple(got.shape) == (1, 1)
        assert abs(float(got) - float(g[n + ":dice"])) < 1e-6, n
    assert abs(float(g["perfect:dice"]) - 1.0) < 1e-6 and float(g["both_empty:dice"]) == 1.0


def test_three_step_trajectory_matches_reference_golden():
    """Three steps of the reference's training loop (ref:params/VSparams.py:454-467) restated with the oracle: forward + Dice_spvPA + backward,
    `adam_step` on every parameter, BatchNorm running statistics / counters carried from step to step (tests/golden/make_goldens.py::golden_trajectory)."""
    g = load("trajectory_b1_128x128x32.npz")
    seed, shape, steps = int(g["seed"]), tuple(int(v) for v in g["shape"]), int(g["steps"])
    sd0 = O.seeded_state_dict(True, seed)
    sd = {k: v.clone() for k, v in sd0.items()}
    x, y = synth_input(seed, shape), synth_label(seed, shape)
    pkeys = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    m = {k: torch.zeros_like(sd[k]) for k in pkeys}
    v = {k: torch.zeros_like(sd[k]) for k in pkeys}
    for step in range(1, steps + 1):
        leaf = {k: (t.clone().requires_grad_(True) if k in m else t) for k, t in sd.items()}
        logits, atts, ctx = O.unet_forward(leaf, x, train=True, attention_module=True, dropout_p=0.0)
        loss = O.dice_spvpa(logits, atts, y)
        loss.backward()
        assert abs(float(loss) - float(g["losses"][step - 1])) < (2e-6 if step == 1 else 2e-5), (step, float(loss))
        # step 1 is the pinned single step; afterwards elements whose gradient sits at the fp32 noise floor may have moved by +-lr instead of -+lr
        np.testing.assert_allclose(logits.detach().flatten()[::97][:512].numpy(), g["logits_sub"][step - 1], atol=2e-5 if step == 1 else 1e-3)
        for k in pkeys:
            gk = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(sd[k])  # conv biases in front of a training-mode BatchNorm
            sd[k], m[k], v[k] = O.adam_step(sd[k], gk, m[k], v[k], step)
        for k, t in ctx.bn_updates.items():
            sd[k] = t.detach()
    for k in g.files:
        if k.startswith("bn:"):
            # (running means follow the conv biases in front of them, which the reference's Adam moves by +-lr on rounding noise: 1e-4, not 1e-6)
            np.testing.assert_allclose(sd[k[3:]].numpy(), g[k], atol=1e-4, rtol=1e-4, err_msg=k)
        elif k.startswith("cnt:"):
            assert int(sd[k[4:]]) == int(g[k]) == steps
        elif k.startswith("dp:"):
            if k.endswith("conv.bias") and ("dp:" + k[3:].replace("conv.bias", "norm.weight")) in g.files:
                continue  # a bias in front of a training-mode BatchNorm has zero gradient: Adam turns the reference's rounding noise into +-lr moves
            dp = (sd[k[3:]].double() - sd0[k[3:]].double()).flatten()
            sub = dp[:: max(1, dp.numel() // 256)][:256].float().numpy()
            # Adam's first steps move every element by ~lr whatever the gradient's size: an element whose gradient is at the fp32 noise floor can
            # take a different sign — barred in aggregate (relative L2), not element by element
            err = np.linalg.norm(sub - g[k]) / max(np.linalg.norm(g[k]), 1e-12)
            assert err < 2e-2, (k, err)
