"""Per-kernel HIP-event breakdown of one eval forward on a 384x128x128 window (the sliding-window predictor), then N whole sliding-window volumes
(512x512x120, roi 384x128x128, overlap 0.5, sw_batch_size 1: 14 windows each) so that a rocprofv3 run of this script traces the sliding-window path:
    python tools/profile_eval.py [volumes]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m = bench.build_model("bf16", torch.device("cuda")).eval()
NB = int(os.environ.get("EVAL_BATCH", "1"))  # windows per forward (sw_batch_size)
x = torch.randn(NB, 1, 384, 128, 128, device="cuda")
with torch.no_grad():
    for _ in range(3): m(x)
    plan = next(p for k, p in m._engine.plans.items() if not k[2])
    plan.timer = dict(only=None, events=[])
    m(x)
    agg = bench.summarize_events(plan.timer["events"])
    rows = sorted(((e0.elapsed_time(e1), name, (meta or {}).get("tag", "")) for name, meta, e0, e1 in plan.timer["events"]), key=lambda r: -r[0])
    plan.timer = None
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    print(f"eval forward: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms wall; kernel sum {sum(a['ms'] for a in agg.values()):.3f} ms over {sum(a['n'] for a in agg.values())} launches")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:28s} n={a['n']:3d} {a['ms']:7.3f} ms")
for ms, name, tag in rows[:int(os.environ.get("EVAL_ROWS", "25"))]:
    print(f"{ms:7.3f} {name:18s} {tag[:170]}")

nvol = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if nvol:
    import vs_seg_amd as V
    vol = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 1, 512, 512, 120), dtype=np.float32)).cuda()
    pred = m.segmentation_predictor()  # what vs_seg_amd/params.py and bench.py run
    with torch.no_grad():
        V.sliding_window_inference(vol, bench.PATCH, 1, pred, overlap=0.5, mode="gaussian")
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(nvol):
            V.sliding_window_inference(vol, bench.PATCH, 1, pred, overlap=0.5, mode="gaussian")
        torch.cuda.synchronize()
    print(f"sliding window: {(time.perf_counter() - t) / nvol * 1e3:.2f} ms per volume ({nvol} volumes, 14 windows each)")
