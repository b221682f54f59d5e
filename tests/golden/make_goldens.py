"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own model/loss code on CPU.

Run in the build container only (needs /root/reference; the GPU box never sees the reference):

    python tests/golden/make_goldens.py

The reference files are imported unmodified from /root/reference via the MONAI stand-in in `_monai_standin.py`.
Fixtures hold only data: input seeds, outputs (full or strided sub-samples + checksums) — never source text.
Weights are regenerated from `oracle.vsseg_oracle.seeded_state_dict(attention, seed)` so they are not stored.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _monai_standin  # noqa: E402

_monai_standin.install()
from params.losses.dice_spvPA import Dice_spvPA  # noqa: E402  (reference)
from params.networks.blocks.attentionblock import AttentionBlock1, AttentionBlock2  # noqa: E402
from params.networks.blocks.convolutions import Convolution, ResidualUnit  # noqa: E402
from params.networks.nets.unet2d5_spvPA import UNet2d5_spvPA  # noqa: E402

from oracle import vsseg_oracle as O  # noqa: E402

torch.set_num_threads(8)


def build_reference_model(attention=True, dropout=0.1):
    hp = O.HP
    return UNet2d5_spvPA(
        dimensions=3,
        in_channels=1,
        out_channels=2,
        channels=hp["channels"],
        strides=hp["strides"],
        kernel_sizes=hp["kernel_sizes"],
        sample_kernel_sizes=hp["sample_kernel_sizes"],
        num_res_units=2,
        norm="batch",
        dropout=dropout,
        attention_module=attention,
    )


def synth_input(seed, shape):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def synth_label(seed, shape):
    """Sparse blob mask (a tumour is a small fraction of the voxels); sample 0 of a batch >1 is left empty."""
    rng = np.random.default_rng(seed + 7919)
    B, _, X, Y, Z = shape
    lab = np.zeros(shape, np.float32)
    for b in range(B):
        if B > 1 and b == 0:
            continue
        c = [rng.integers(s // 4, max(s // 4 + 1, 3 * s // 4)) for s in (X, Y, Z)]
        r = [max(1, s // 6) for s in (X, Y, Z)]
        gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
        lab[b, 0] = (((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0).astype(np.float32)
    return torch.from_numpy(lab)


def summarize(t, stride=None):
    """Compact description of a big tensor: checksums + a strided sub-sample of the flattened data."""
    a = t.detach().double().flatten()
    n = a.numel()
    if stride is None:
        stride = max(1, n // 4096)
    return dict(sum=float(a.sum()), abssum=float(a.abs().sum()), sqsum=float((a * a).sum()), n=int(n), stride=int(stride)), a[::stride].float().numpy()


def golden_manifest():
    out = {}
    for att in (True, False):
        m = build_reference_model(att)
        sd = m.state_dict()
        out["attention" if att else "no_attention"] = [[k, list(v.shape)] for k, v in sd.items()]
        assert [(k, tuple(s)) for k, s in out["attention" if att else "no_attention"]] == [(k, tuple(s)) for k, s in O.manifest(att)]
    json.dump(out, open(os.path.join(HERE, "manifest.json"), "w"), indent=0)
    print("manifest: %d / %d keys" % (len(out["attention"]), len(out["no_attention"])))


def golden_net_eval(only=None):
    cases = [("b2_32x32x8", True, 11, (2, 1, 32, 32, 8)), ("b1_64x64x16", True, 12, (1, 1, 64, 64, 16)), ("b1_32x32x8_noatt", False, 13, (1, 1, 32, 32, 8)), ("b1_128x128x32", True, 14, (1, 1, 128, 128, 32)), ("b1_64x32x24", True, 15, (1, 1, 64, 32, 24)),
             # BASELINE.json's full sizes: the benchmark patch / sliding-window roi and the reference-native roi; sub-sample + checksums only
             ("b1_384x128x128", True, 16, (1, 1, 384, 128, 128)), ("b1_384x384x64", True, 17, (1, 1, 384, 384, 64))]
    for name, att, seed, shape in cases:
        if only and name not in only:
            continue
        model = build_reference_model(att)
        model.load_state_dict(O.seeded_state_dict(att, seed))
        model.eval()
        x = synth_input(seed, shape)
        with torch.no_grad():
            logits, atts = model(x)
        d = dict(seed=seed, shape=np.array(shape), attention=att)
        if logits.numel() <= 300000:
            d["logits"] = logits.numpy()
        meta, sub = summarize(logits)
        d["logits_meta"] = json.dumps(meta)
        d["logits_sub"] = sub
        for i, a in enumerate(atts):
            if a.numel() <= 70000:
                d[f"att{i}"] = a.numpy()
            meta, sub = summarize(a)
            d[f"att{i}_meta"] = json.dumps(meta)
            d[f"att{i}_sub"] = sub
        d["n_att"] = len(atts)
        np.savez_compressed(os.path.join(HERE, f"net_eval_{name}.npz"), **d)
        print("net_eval", name, float(logits.abs().mean()))


def golden_net_train(only=None):
    """One training-mode fwd + loss + bwd with dropout p=0 (torch's dropout stream cannot be reproduced elsewhere)."""
    for name, att, hard, seed, shape in [("b2_32x32x8", True, True, 21, (2, 1, 32, 32, 8)), ("b2_32x32x8_noatt_nohard", False, False, 22, (2, 1, 32, 32, 8)), ("b1_64x64x16", True, True, 23, (1, 1, 64, 64, 16)),
                                         ("b1_384x128x128", True, True, 24, (1, 1, 384, 128, 128)),  # the benchmark patch: sub-samples + checksums only
                                         ("b1_384x384x64", True, True, 25, (1, 1, 384, 384, 64))]:  # the reference's own default patch (ref:params/VSparams.py:76, train_batch_size 1)
        if only and name not in only:
            continue
        big = int(np.prod(shape)) > 1_000_000
        model = build_reference_model(att, dropout=0.0)
        model.load_state_dict(O.seeded_state_dict(att, seed))
        model.train()
        x, y = synth_input(seed, shape), synth_label(seed, shape)
        loss_fn = Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=att, hardness_weighting=hard)
        # Sum of the ABSOLUTE terms of every scalar-like gradient that is one cancelling sum over a whole activation tensor (round 5): the PReLU slope
        # d(alpha) = sum over the negative branch of dA * x, and BatchNorm's d(gamma)[c] = sum dz * xhat, d(beta)[c] = sum dz.  tests/parity_check.py bars
        # |got - want| / sum|terms|: an error measured against what was summed, which tells "the sum cancels to ~0" from "wrong" (a sign count cannot).
        gabs = {}

        def hook_act(name):
            def fwd(mod, inp, out):
                xin = inp[0].detach()
                out.register_hook(lambda g: gabs.__setitem__(name + ".weight", np.array([float((g.double() * xin.double())[xin < 0].abs().sum())])))
            return fwd

        def hook_norm(name):
            def fwd(mod, inp, out):
                yin = inp[0].detach().double()
                red = (0, 2, 3, 4)
                xhat = (yin - yin.mean(red, keepdim=True)) / torch.sqrt(yin.var(red, unbiased=False, keepdim=True) + mod.eps)

                def bwd(g):
                    g = g.double()
                    gabs[name + ".weight"] = (g * xhat).abs().sum(red).numpy()
                    gabs[name + ".bias"] = g.abs().sum(red).numpy()
                out.register_hook(bwd)
            return fwd

        for mname, mod in model.named_modules():
            if isinstance(mod, torch.nn.PReLU):
                mod.register_forward_hook(hook_act(mname))
            elif isinstance(mod, torch.nn.BatchNorm3d):
                mod.register_forward_hook(hook_norm(mname))
        logits, atts = model(x)
        logits.retain_grad()
        for a in atts:
            a.retain_grad()
        loss = loss_fn((logits, atts), y)
        loss.backward()
        d = dict(seed=seed, shape=np.array(shape), attention=att, hardness=hard, loss=float(loss), n_att=len(atts))
        if big:
            for key, t in [("logits", logits.detach()), ("dlogits", logits.grad)] + [(f"att{i}", a.detach()) for i, a in enumerate(atts)] + [(f"datt{i}", a.grad) for i, a in enumerate(atts)]:
                meta, sub = summarize(t)
                d[key + "_meta"], d[key + "_sub"] = json.dumps(meta), sub
        else:
            d["logits"], d["dlogits"] = logits.detach().numpy(), logits.grad.numpy()
            for i, a in enumerate(atts):
                d[f"att{i}"] = a.detach().numpy()
                d[f"datt{i}"] = a.grad.numpy()
        gs, gsub = {}, {}
        for k, p in model.named_parameters():
            g = p.grad.double().flatten()
            gs[k] = [float(g.sum()), float(g.abs().sum()), float((g * g).sum())]
            gsub[k] = g[:: max(1, g.numel() // 64)][:64].float().numpy()
        d["grad_sums"] = json.dumps(gs)
        for k, v in gsub.items():
            d["gsub:" + k] = v
        for k, v in gabs.items():  # same sub-sampling as gsub: element i of gabs:<key> is the sum of |terms| of element i of gsub:<key>
            assert k in gsub, k
            d["gabs:" + k] = v[:: max(1, v.size // 64)][:64].astype(np.float64)
        sd = model.state_dict()
        for k, v in sd.items():
            if "running_" in k:
                d["bn:" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, f"net_train_{name}.npz"), **d)
        print("net_train", name, float(loss))


def golden_trajectory():
    """THREE steps of the reference's training loop (ref:params/VSparams.py:454-467: zero_grad, forward, Dice_spvPA, backward, Adam step with
    lr 1e-4 / weight_decay 1e-7, ref:388-391) on one fixed batch, dropout p = 0: loss per step, every parameter and every BatchNorm buffer
    (running_mean / running_var with momentum 0.1, num_batches_tracked) after the third step.  Pins what a single step cannot: the BatchNorm
    running-statistics recursion, Adam's bias correction over steps on the real network, and that step k+1 starts from step k's parameters."""
    seed, shape, steps = 27, (1, 1, 128, 128, 32), 3  # 64 values per bottleneck channel: below that training-mode BatchNorm amplifies fp32 summation-order noise to percents
    model = build_reference_model(True, dropout=0.0)
    model.load_state_dict(O.seeded_state_dict(True, seed))
    model.train()
    x, y = synth_input(seed, shape), synth_label(seed, shape)
    loss_fn = Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    opt = torch.optim.Adam(model.parameters(), 1e-4, weight_decay=1e-7)
    losses, logit_sub = [], []
    for _ in range(steps):
        opt.zero_grad()
        outputs = model(x)
        loss = loss_fn(outputs, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        logit_sub.append(outputs[0].detach().flatten()[::97][:512].numpy().copy())
    d = dict(seed=seed, shape=np.array(shape), steps=steps, lr=1e-4, weight_decay=1e-7, losses=np.array(losses, np.float64), logits_sub=np.stack(logit_sub))
    sd0 = O.seeded_state_dict(True, seed)
    for k, v in model.state_dict().items():
        v = v.detach()
        if k.endswith("num_batches_tracked"):
            d["cnt:" + k] = np.array(int(v))
        elif "running_" in k:
            d["bn:" + k] = v.numpy()
        else:  # the parameter's displacement over the three steps (values are O(1), displacements O(3e-4)): a strided sub-sample + checksums
            dp = (v.double() - sd0[k].double()).flatten()
            d["dp:" + k] = dp[:: max(1, dp.numel() // 256)][:256].float().numpy()
            d["dpsum:" + k] = np.array([float(dp.sum()), float(dp.abs().sum()), float((dp * dp).sum())])
    np.savez_compressed(os.path.join(HERE, "trajectory_b1_128x128x32.npz"), **d)
    print("trajectory", losses)


def golden_blocks():
    """Per-block goldens: every conv flavour of the network (kernel/stride/transposed), ResidualUnit and attention."""
    d = {}
    specs = [("c331_s1", (3, 3, 1), (1, 1, 1), False, 8, 16), ("c333_s1", (3, 3, 3), (1, 1, 1), False, 8, 16), ("c331_s221", (3, 3, 1), (2, 2, 1), False, 16, 16), ("c333_s222", (3, 3, 3), (2, 2, 2), False, 16, 16), ("t331_s221", (3, 3, 1), (2, 2, 1), True, 16, 8), ("t333_s222", (3, 3, 3), (2, 2, 2), True, 16, 8)]
    for i, (name, k, s, tr, cin, cout) in enumerate(specs):
        for train in (True, False):
            torch.manual_seed(100 + i)
            blk = Convolution(3, cin, cout, strides=s, kernel_size=k, norm="batch", dropout=0.0, is_transposed=tr)
            with torch.no_grad():
                blk.norm.running_mean.normal_(0, 0.1)
                blk.norm.running_var.uniform_(0.5, 1.5)
                blk.norm.weight.uniform_(0.5, 1.5)
                blk.norm.bias.normal_(0, 0.1)
            blk.train(train)
            x = synth_input(200 + i, (2, cin, 8, 8, 4)).requires_grad_(True)
            y = blk(x)
            gy = synth_input(300 + i, tuple(y.shape))
            y.backward(gy)
            tag = f"{name}_{'train' if train else 'eval'}"
            d[tag + ":y"] = y.detach().numpy()
            d[tag + ":dx"] = x.grad.numpy()
            for kk, p in blk.named_parameters():
                d[f"{tag}:p:{kk}"] = p.detach().numpy()
                d[f"{tag}:g:{kk}"] = p.grad.numpy()
            for kk, b in blk.named_buffers():
                d[f"{tag}:b:{kk}"] = b.numpy()
    # ResidualUnit flavours
    for i, (name, cin, cout, sub, last) in enumerate([("ru_2", 8, 16, 2, False), ("ru_1", 16, 8, 1, False), ("ru_1_last", 16, 2, 1, True), ("ru_2_same", 16, 16, 2, False)]):
        torch.manual_seed(400 + i)
        blk = ResidualUnit(3, cin, cout, strides=1, kernel_size=(3, 3, 3), subunits=sub, norm="batch", dropout=0.0, last_conv_only=last)
        blk.train(True)
        x = synth_input(500 + i, (2, cin, 8, 8, 4)).requires_grad_(True)
        y = blk(x)
        gy = synth_input(600 + i, tuple(y.shape))
        y.backward(gy)
        d[name + ":y"] = y.detach().numpy()
        d[name + ":dx"] = x.grad.numpy()
        for kk, p in blk.named_parameters():
            d[f"{name}:p:{kk}"] = p.detach().numpy()
            d[f"{name}:g:{kk}"] = p.grad.numpy()
    # attention
    torch.manual_seed(700)
    a1, a2 = AttentionBlock1(3, 16, 16, (3, 3, 3), norm=None, dropout=0.1), AttentionBlock2(3, 16, 16, (3, 3, 3), norm=None, dropout=0.1)
    x = synth_input(701, (2, 16, 8, 8, 4)).requires_grad_(True)
    att, xx = a1(x)
    att.retain_grad()
    y = a2((att, xx))
    gy, ga = synth_input(702, tuple(y.shape)), synth_input(703, tuple(att.shape))
    (y * gy).sum().add((att * ga).sum()).backward()
    d["att:y"], d["att:att"], d["att:dx"] = y.detach().numpy(), att.detach().numpy(), x.grad.numpy()
    for kk, p in a1.named_parameters():
        d[f"att:p:{kk}"] = p.detach().numpy()
        d[f"att:g:{kk}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **d)
    print("blocks", len(d))


def golden_loss():
    d = {}
    shape = (2, 1, 32, 32, 8)
    att_shapes = [(2, 1, 1, 1, 1), (2, 1, 2, 2, 2), (2, 1, 4, 4, 4), (2, 1, 8, 8, 8), (2, 1, 16, 16, 8), (2, 1, 32, 32, 8)]
    y = synth_label(31, shape)
    for att in (True, False):
        for hard in (True, False):
            logits = (2.0 * synth_input(32, (2, 2, 32, 32, 8))).requires_grad_(True)
            atts = [torch.sigmoid(synth_input(40 + i, s)).requires_grad_(True) for i, s in enumerate(att_shapes)] if att else []
            loss = Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=att, hardness_weighting=hard)((logits, atts), y)
            loss.backward()
            tag = f"att{int(att)}_hard{int(hard)}"
            d[tag + ":loss"] = np.float64(loss.item())
            d[tag + ":dlogits"] = logits.grad.numpy()
            for i, a in enumerate(atts):
                d[f"{tag}:datt{i}"] = a.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **d)
    print("loss", {k: float(v) for k, v in d.items() if k.endswith(":loss")})


def golden_adam():
    """torch.optim.Adam exactly as constructed at ref:params/VSparams.py:388-391, 3 steps, plus the LR halving rule."""
    rng = np.random.default_rng(51)
    p0 = rng.standard_normal(1000).astype(np.float32)
    grads = [rng.standard_normal(1000).astype(np.float32) * s for s in (1.0, 0.1, 3.0)]
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p], lr=1e-4, weight_decay=1e-7)
    outs = []
    for g in grads:
        opt.zero_grad()
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        outs.append(p.detach().numpy().copy())
    np.savez_compressed(os.path.join(HERE, "adam.npz"), p0=p0, grads=np.stack(grads), after=np.stack(outs))
    print("adam ok")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--only-net-eval":  # regenerate selected eval cases only, e.g. the full-size ones
        golden_net_eval(set(sys.argv[2:]))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--only-net-train":
        golden_net_train(set(sys.argv[2:]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--only-trajectory":
        golden_trajectory()
        sys.exit(0)
    golden_manifest()
    golden_trajectory()
    golden_blocks()
    golden_loss()
    golden_adam()
    golden_net_eval()
    golden_net_train()
