"""HIP-event time of the phases of one training step as the trainer runs it (zero_grad, forward, loss, backward, Adam): python tools/time_phases.py [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import vs_seg_amd as V  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = B.build_model("bf16", dev).train()
model.reuse_output_buffers = True
loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
opt = V.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
img, lab = B.synth_batch(4, B.PATCH, 0, dev)
names = ["zero_grad", "forward", "loss", "backward", "adam"]
acc = {k: 0.0 for k in names}
for it in range(steps + 6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record(); opt.zero_grad()
    ev[1].record(); out = model(img)
    ev[2].record(); loss = loss_fn(out, lab)
    ev[3].record(); loss.backward()
    ev[4].record(); opt.step()
    ev[5].record()
    torch.cuda.synchronize()
    if it >= 6:
        for i, k in enumerate(names):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
tot = sum(acc.values()) / steps
for k in names:
    print(f"{k:10s} {acc[k] / steps:7.3f} ms")
print(f"{'total':10s} {tot:7.3f} ms")
