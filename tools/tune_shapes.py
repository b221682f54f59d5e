"""Measure the launch plans of further shapes into $VSSEG_TUNE_CACHE (merged into vs_seg_amd/tuned_gfx950.json by hand) and print the step / forward times:
    VSSEG_TUNE_CACHE=gpurun_out/tune_extra.json python tools/tune_shapes.py 1x384x384x64 2x384x384x64
Each argument is batch x X x Y x Z: three training steps (fwd + Dice_spvPA + bwd + Adam) and the eval forward at batch 1 of that patch, on two streams
(the sliding-window inferer's lanes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import vs_seg_amd as V  # noqa: E402
from vs_seg_amd.parallel import DataParallelTrainer  # noqa: E402

dev = torch.device("cuda:0")
for arg in sys.argv[1:]:
    n, *patch = (int(v) for v in arg.split("x"))
    model = B.build_model("bf16", dev).train()
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    tr = DataParallelTrainer(model, loss_fn, V.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7))
    img, lab = B.synth_batch(n, tuple(patch), 0, dev)
    t0 = time.perf_counter()
    for _ in range(4):
        tr.step(img, lab)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(img, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    model.eval()
    x = img[:1]
    with torch.no_grad():
        vol = torch.randn(1, 1, patch[0] + 64, patch[1] + 64, patch[2] + 16, device=dev)
        for _ in range(3):
            V.sliding_window_inference(vol, tuple(patch), 1, lambda w: model(w)[0], overlap=0.5, mode="gaussian")
        pred = model.segmentation_predictor()
        for swb in (2, 4):  # the eval plans of 2 / 4 windows per predictor call (bench.py: sliding_window.sw_batch_size_2 / _4), on both lanes
            for _ in range(3):
                V.sliding_window_inference(vol, tuple(patch), swb, pred, overlap=0.5, mode="gaussian")
        for _ in range(4):  # the caller's stream has its own eval plan: lowered and captured before the timed forwards
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(x)
        torch.cuda.synchronize()
        de = (time.perf_counter() - t0) / 10
    print(f"{arg}: first 4 steps {t_first:.1f} s (plans measured / loaded), then {dt * 1e3:.2f} ms per training step = {n / dt:.1f} patches/s; eval forward {de * 1e3:.2f} ms", flush=True)
    del tr, model
    torch.cuda.empty_cache()
