"""Wall time of the training step exactly as a user runs it (hipGraph replay, no per-launch events): python tools/time_step.py [steps]
Environment switches under test are read by the engine (VSSEG_OVERLAP, VSSEG_GRAPHS, ...)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import vs_seg_amd as V  # noqa: E402
from vs_seg_amd.parallel import DataParallelTrainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    model = B.build_model("bf16", dev).train()
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    opt = V.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    tr = DataParallelTrainer(model, loss_fn, opt)
    img, lab = B.synth_batch(4, B.PATCH, 0, dev)
    for _ in range(6):
        tr.step(img, lab)
    live = os.environ.get("VSSEG_TIME_LIVE")  # e.g. "mconv<bf16,2>": two HIP events around every launch of that kernel group, as bench.py's timed region has them
    if live:
        plan = next(p for k, p in model._engine.plans.items() if k[2])
        plan.timer = dict(only={live}, events=[])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(img, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{dt * 1e3:.3f} ms/step  ({4 / dt:.1f} patches/s)  OVERLAP={os.environ.get('VSSEG_OVERLAP', '0')} GRAPHS={os.environ.get('VSSEG_GRAPHS', '1')}")


if __name__ == "__main__":
    main()
