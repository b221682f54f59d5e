"""Per-layer self-check of the HIP training step on the GPU box: for every Convolution (conv -> BN -> dropout -> PReLU)
of the lowered program, recompute that single op and its backward with torch float64 on the device from the plan's OWN
input buffers, and report the relative error of every tensor the HIP path produced (y, out, dy, dW, dgamma, dbeta, dalpha).
Isolates a wrong layer without needing the whole-network oracle.  Debug tool — not part of the product path.
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import vs_seg_amd as V  # noqa: E402
from oracle import vsseg_oracle as O  # noqa: E402
from tests.helpers import synth_input, synth_label  # noqa: E402
from vs_seg_amd.graph import ConvBnAct, ConvPlain  # noqa: E402


def cl2ncdhw(t, c0=0, c=None):
    c = c or t.shape[-1] - c0
    return t[..., c0 : c0 + c].permute(0, 4, 1, 2, 3).double()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main(att=True, hard=False, seed=24, shape=(1, 1, 128, 128, 32), dtype="fp32"):
    hp = O.HP
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, channels=hp["channels"], strides=hp["strides"], kernel_sizes=hp["kernel_sizes"], sample_kernel_sizes=hp["sample_kernel_sizes"],
                        num_res_units=2, norm="batch", dropout=0.0, attention_module=att, compute_dtype=dtype)
    m.load_state_dict(O.seeded_state_dict(att, seed))
    m = m.cuda().train()
    x, y = synth_input(seed, shape).cuda(), synth_label(seed, shape).cuda()
    logits, atts = m(x)
    loss = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=att, hardness_weighting=hard)((logits, atts), y)
    loss.backward()
    torch.cuda.synchronize()
    eng = m._engine
    plan = next(p for k, p in eng.plans.items() if k[2])
    sd = {k: v.detach().double() for k, v in m.state_dict().items()}
    grads = {k: p.grad.detach().double() for k, p in m.named_parameters()}

    def tensor(spec, store):
        buf = store[spec.root.name]
        return cl2ncdhw(buf, spec.c0, spec.c)

    print(f"{'layer':62s} {'y':>9s} {'out':>9s} {'dy':>9s} {'dW':>9s} {'dgam':>9s} {'dbet':>9s} {'dalp':>9s} {'dx':>9s}")
    for op in eng.prog.ops:
        if not isinstance(op, ConvBnAct):
            continue
        Lr, pre = op.layer, op.layer.prefix
        xin = tensor(op.x, plan.bufs)[:, : Lr.cin].clone().requires_grad_(True)
        w = sd[Lr.wkey].clone().requires_grad_(True)
        b = sd[Lr.bkey]
        gam, bet, alp = (sd[pre + s].clone().requires_grad_(True) for s in (".norm.weight", ".norm.bias", ".act.weight"))
        pad = tuple((k - 1) // 2 for k in Lr.kernel)
        if Lr.transposed:
            opad = tuple(s + 2 * p - (k - 1) - 1 for s, p, k in zip(Lr.stride, pad, Lr.kernel))
            yy = F.conv_transpose3d(xin, w, b, stride=Lr.stride, padding=pad, output_padding=opad)
        else:
            yy = F.conv3d(xin, w, b, stride=Lr.stride, padding=pad)
        mean, var = yy.mean((0, 2, 3, 4)), yy.var((0, 2, 3, 4), unbiased=False)
        sh = (1, -1, 1, 1, 1)
        z = (yy - mean.view(sh)) / torch.sqrt(var.view(sh) + 1e-5) * gam.view(sh) + bet.view(sh)
        out = F.prelu(z, alp)
        if op.res is not None:
            out = out + tensor(op.res, plan.bufs)
        y_hip = cl2ncdhw(plan.bufs["y:" + pre])
        out_hip = tensor(op.out, plan.bufs)
        dA = tensor(op.out, plan.grads)
        yy.retain_grad()
        out.backward(dA)
        dy_hip = cl2ncdhw(plan.bufs["dy:" + pre])
        # dx only comparable when this op is the sole producer of g[x]; print anyway for reference
        dx_hip = tensor(op.x, plan.grads)[:, : Lr.cin] if op.x.root.name in plan.grads else None
        print(f"{pre:62s} {rel(y_hip, yy.detach()):9.2e} {rel(out_hip, out.detach()):9.2e} {rel(dy_hip, yy.grad):9.2e} {rel(grads[Lr.wkey], w.grad):9.2e} "
              f"{rel(grads[pre + '.norm.weight'], gam.grad):9.2e} {rel(grads[pre + '.norm.bias'], bet.grad):9.2e} {rel(grads[pre + '.act.weight'], alp.grad):9.2e} "
              f"{(rel(dx_hip, xin.grad) if dx_hip is not None else float('nan')):9.2e}")
    print("plain convolutions (dW, dbias given the HIP path's own dy):")
    for op in eng.prog.ops:
        if not isinstance(op, ConvPlain):
            continue
        Lr = op.layer
        xin = tensor(op.x, plan.bufs)[:, : Lr.cin].clone().requires_grad_(True)
        w = sd[Lr.wkey].clone().requires_grad_(True)
        b = sd[Lr.bkey].clone().requires_grad_(True)
        pad = tuple((k - 1) // 2 for k in Lr.kernel)
        yy = F.conv3d(xin, w, b, stride=Lr.stride, padding=pad)
        if op.act == "sigmoid":
            dy = cl2ncdhw(plan.bufs["dpre:" + op.out.name], 0, 1)
        elif op.out.name.endswith(":res") or op.out.name == eng.prog.logits.name:
            tgt = op.out
            # residual convs share the gradient of the tensor they are added into
            for o2 in eng.prog.ops:
                if getattr(o2, "res", None) is not None and o2.res.name == op.out.name:
                    tgt = o2.out
            dy = cl2ncdhw(plan.bufs["g:logits8"], 0, 2) if tgt.kind == "f32" else tensor(tgt, plan.grads)
        else:
            dy = tensor(op.out, plan.grads)
        yy.backward(dy)
        print(f"{Lr.prefix:62s} dW {rel(grads[Lr.wkey], w.grad):9.2e}  db {rel(grads[Lr.bkey], b.grad):9.2e}")


if __name__ == "__main__":
    main()
