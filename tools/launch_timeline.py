"""Every launch of one training step (and of one eval window) in execution order with its HIP-event time, alone on the GPU:
python tools/launch_timeline.py [train|eval|both] > timeline.txt.  Used to see what the deep levels (3-5) cost launch by launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import vs_seg_amd as V  # noqa: E402
from vs_seg_amd import parallel as DP  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")


def dump(title, events, reps):
    print(f"=== {title}")
    acc = {}
    order = []
    for i, (name, meta, e0, e1) in enumerate(events):
        k = i % (len(events) // reps)
        if k not in acc:
            acc[k] = [name, meta, []]
            order.append(k)
        acc[k][2].append(e0.elapsed_time(e1))
    tot = 0.0
    for k in order:
        name, meta, ts = acc[k]
        ms = min(ts)
        tot += ms
        tag = (meta or {}).get("tag", "")
        print(f"{k:4d} {ms * 1000:9.1f} us  {name:28s} {tag}")
    print(f"=== sum {tot:.3f} ms over {len(order)} launches")


if what in ("train", "both"):
    model = B.build_model("bf16", dev)
    model.reuse_output_buffers = True
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    opt = V.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    trainer = DP.DataParallelTrainer(model.train(), loss_fn, opt)
    img, lab = B.synth_batch(4, B.PATCH, 1000, dev)
    for _ in range(3):
        trainer.step(img, lab)
    plan = next(p for k, p in model._engine.plans.items() if k[2])
    plan.timer = dict(only=None, events=[])
    reps = 3
    for _ in range(reps):
        trainer.step(img, lab)
    torch.cuda.synchronize()
    ev = plan.timer["events"]
    plan.timer = None
    dump("training step, batch 4, 384x128x128 bf16", ev, reps)
    del trainer, model, opt
    torch.cuda.empty_cache()

if what in ("eval", "both"):
    model = B.build_model("bf16", dev).eval()
    x = torch.randn(1, 1, *B.PATCH, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        plan = next(p for k, p in model._engine.plans.items() if not k[2])
        plan.timer = dict(only=None, events=[])
        reps = 3
        for _ in range(reps):
            model(x)
        torch.cuda.synchronize()
    ev = plan.timer["events"]
    plan.timer = None
    dump("eval forward, one window 384x128x128 bf16", ev, reps)
