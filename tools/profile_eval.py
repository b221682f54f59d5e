"""Per-kernel HIP-event breakdown of one eval forward on a 384x128x128 window (the sliding-window predictor)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m = bench.build_model("bf16", torch.device("cuda")).eval()
x = torch.randn(1, 1, 384, 128, 128, device="cuda")
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
for ms, name, tag in rows[:25]:
    print(f"{ms:7.3f} {name:18s} {tag[:170]}")
