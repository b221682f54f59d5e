import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench, vs_seg_amd as V
m = bench.build_model("bf16", torch.device("cuda")).eval()
vol = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 1, 512, 512, 120), dtype=np.float32)).cuda()
pred = lambda w: m(w)[0]
with torch.no_grad():
    for swb in (1, 2, 4):
        for lanes in (1, 2, 3, 4):
            for _ in range(2): V.sliding_window_inference(vol, bench.PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(8): V.sliding_window_inference(vol, bench.PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 8
            print(f"swb={swb} lanes={lanes}: {dt*1e3:.2f} ms/volume  {1/dt:.1f} vol/s", flush=True)
