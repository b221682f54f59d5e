"""Per-tensor gradient error of the HIP path (fp32 and bf16 compute modes) against the reference golden at 384x128x128 (diagnostic)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vs_seg_amd as V  # noqa: E402
from tests import parity_check as PC  # noqa: E402
from tests.helpers import seeded_weights_for, synth_input, synth_label  # noqa: E402

HP = dict(channels=(16, 32, 48, 64, 80, 96), strides=((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)), kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
          sample_kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3)))
g, seed, shape = PC.golden_train_case()
sums = json.loads(str(g["grad_sums"]))
res = {}
for dtype in ("fp32", "bf16"):
    torch.manual_seed(1000 + seed)
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, num_res_units=2, norm="batch", dropout=0.0, attention_module=True, compute_dtype=dtype, **HP)
    m.load_state_dict(seeded_weights_for(m.state_dict(), seed))
    m = m.to("cuda").train()
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True)
    logits, atts = m(synth_input(seed, shape).cuda())
    loss = loss_fn((logits, atts), synth_label(seed, shape).cuda())
    loss.backward()
    for k, p in m.named_parameters():
        gk = p.grad.double().flatten().cpu()
        sub = gk[:: max(1, gk.numel() // 64)][:64].numpy()
        want = g["gsub:" + k].astype(np.float64)
        rel = float(np.linalg.norm(sub - want) / (np.linalg.norm(want) + 1e-30))
        sq = float((gk * gk).sum())
        res.setdefault(k, {})[dtype] = (rel, sq, sums[k][2], gk.numel())
    del m
print(f"{'tensor':95s} {'n':>8s} {'|g|ref':>10s} {'rel fp32':>9s} {'rel bf16':>9s} {'sq bf16/ref':>11s}")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["bf16"][0]):
    print(f"{k[-95:]:95s} {v['bf16'][3]:8d} {v['bf16'][2] ** 0.5:10.3e} {v['fp32'][0]:9.2e} {v['bf16'][0]:9.2e} {v['bf16'][1] / (v['bf16'][2] + 1e-300):11.3f}")
