"""Sliding-window throughput on one 512x512x120 volume (BASELINE config 3 shape) for several (sw_batch_size, concurrent_groups) settings.
usage: python tools/time_swi.py [volumes]      (GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vs_seg_amd as V
from bench import build_model, PATCH

dev = torch.device("cuda:0")
model = build_model("bf16", dev, dropout=0.1).eval()
preds = {"segmentation_predictor()": model.segmentation_predictor(), "lambda w: model(w)[0]": (lambda w: model(w)[0])}
vol = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 1, 512, 512, 120), dtype=np.float32)).to(dev)
nvol = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with torch.no_grad():
    for (pname, pred), (swb, lanes) in ((pp, sl) for sl in ((1, 2), (1, 1), (2, 2), (4, 2)) for pp in preds.items()):
        for _ in range(3):
            V.sliding_window_inference(vol, PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nvol):
            V.sliding_window_inference(vol, PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nvol
        print(f"{pname:26s} sw_batch_size {swb} concurrent_groups {lanes}: {1e3 * dt:.2f} ms/volume  {1 / dt:.2f} volumes/s", flush=True)
