"""The four stride-(2,2,1) 3x3x1 launches that read the fine level and write the coarse one, at benchmark size: the gathering marching kernel (csrc/gconv.hip, depth -9) against
every other candidate plan.   python tools/bench_gconv.py [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K, ST = (3, 3, 1), (2, 2, 1)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


for kind, wshape, coarse, mode in (("conv_fwd", (16, 16, *K), (192, 64, 128), "stats"), ("conv_fwd", (32, 32, *K), (96, 32, 128), "stats"),
                                   ("convT_dgrad", (32, 16, *K), (192, 64, 128), "plain"), ("convT_dgrad", (48, 32, *K), (96, 32, 128), "plain"),
                                   ("conv_fwd", (32, 32, *K), (96, 32, 128), "eval")):
    w = torch.randn(*wshape) * 0.05
    kreal, nreal = P.gemm_dims(kind, wshape)
    fine = (2 * coarse[0], 2 * coarse[1], coarse[2])
    x = torch.randn(n, *fine, kreal, device="cuda").to(torch.bfloat16)
    out = torch.zeros(n, *coarse, nreal, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(L.STAT_SHARDS * 2 * P.round_up(nreal, 16), dtype=torch.float64, device="cuda")
    sc, sh, al = torch.ones(nreal, device="cuda"), torch.zeros(nreal, device="cuda"), torch.tensor([0.2], device="cuda")
    kw = dict(stats=stats.data_ptr(), stats_stride=P.round_up(nreal, 16)) if mode == "stats" else (dict(scale=sc.data_ptr(), shift=sh.data_ptr(), act=L.ACT_PRELU, alpha=al.data_ptr()) if mode == "eval" else {})
    gb = (x.numel() + out.numel()) * 2 / 1e9
    print(f"== {kind} {wshape} coarse {coarse} x {n}, {mode}: {gb:.3f} GB")
    cls = P.lattice_classes(kind, K, ST)[0]
    rows = []
    for pl in P.candidate_plans(kind, wshape, cls, coarse, 2, kc_pad=kreal, aux_es=0, n=n):
        d = H.igemm_desc(pl, H.pack(pl, w, torch.bfloat16), H.tdesc(x), H.tdesc(out), **kw)
        if lib.vsseg_igemm(C.byref(d), H.stream()):
            continue
        rows.append((timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream())), pl))
    for us, pl in sorted(rows, key=lambda r: r[0])[:6] + [r for r in rows if r[1].depth == -9][:8]:
        print(f"   D={pl.depth:2d} {us:8.1f} us  {gb / us * 1e3:5.2f} TB/s   tile={pl.tile} mtw={pl.mtw} nt={pl.nt} ck={pl.ck} lds={pl.lds}")
