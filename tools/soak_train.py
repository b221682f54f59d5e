import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench, vs_seg_amd as V
from vs_seg_amd import parallel as DP
dev = torch.device("cuda")
m = bench.build_model("bf16", dev); m.reuse_output_buffers = True
loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
opt = V.Adam(m.parameters(), lr=1e-3, weight_decay=1e-7)
tr = DP.DataParallelTrainer(m.train(), loss_fn, opt)
img, lab = bench.synth_batch(4, bench.PATCH, 1000, dev)
losses = []
t = time.perf_counter()
for i in range(120):
    l = tr.step(img, lab)
    if i % 20 == 0 or i == 119: losses.append((i, float(l)))
torch.cuda.synchronize()
print("losses", [(i, round(v, 4)) for i, v in losses], f"{(time.perf_counter()-t)/120*1e3:.1f} ms/step")
assert all(v == v for _, v in losses) and losses[-1][1] < losses[0][1] - 0.1, "loss did not decrease"
m.eval()
with torch.no_grad():
    lg = m(img[:1])[0]
print("eval logits finite:", bool(torch.isfinite(lg).all()), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
