"""The two stride-(2,2,2) launches between levels 2 and 3 that the transition kernel (csrc/tconv.hip, depth -8) covers, at benchmark size: the transition plan against the
class-split plans of the general kernel, the deep-level kernel's class plans and the eight per-class launches.   python tools/bench_tconv.py [batch]"""
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
K3, ST, COARSE = (3, 3, 3), (2, 2, 2), (48, 16, 64)
FINE = tuple(2 * c for c in COARSE)


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


for kind, wshape, cin, nout, mode in (("convT_fwd", (64, 48, *K3), 64, 48, "stats"), ("convT_fwd", (64, 48, *K3), 64, 48, "plain"), ("conv_dgrad", (48, 48, *K3), 48, 48, "accumulate")):
    w = torch.randn(*wshape) * 0.05
    kreal, nreal = P.gemm_dims(kind, wshape)
    x = torch.randn(n, *COARSE, cin, device="cuda").to(torch.bfloat16)
    out = torch.zeros(n, *FINE, nout, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(L.STAT_SHARDS * 2 * 48, dtype=torch.float64, device="cuda")
    kw = dict(stats=stats.data_ptr(), stats_stride=48) if mode == "stats" else (dict(accumulate=1) if mode == "accumulate" else {})
    gflop = 2.0 * n * COARSE[0] * COARSE[1] * COARSE[2] * 27 * cin * nout / 1e9
    gb = (x.numel() + out.numel() * (2 if mode == "accumulate" else 1)) * 2 / 1e9
    print(f"== {kind} {cin} -> {nout} k3 s2, coarse {COARSE} x {n}, {mode}: {gflop:.1f} GFLOP, {gb:.3f} GB")
    rows = []
    aux_es = 2 if mode == "accumulate" else 0
    for label, pls in (("transition", P.transition_plans(kind, wshape, K3, ST, COARSE, 2, cin, nreal, kreal)),
                       ("class split", P.class_split_plans(kind, wshape, K3, ST, COARSE, 2, cin, nreal, kreal, aux_es=aux_es) or []),
                       ("deep classes", P.deep_class_plans(kind, wshape, K3, ST, COARSE, 2, cin, nreal, kreal, n))):
        for pl in pls:
            d = H.igemm_desc(pl, H.pack(pl, w, torch.bfloat16), H.tdesc(x), H.tdesc(out), **kw)
            if lib.vsseg_igemm(C.byref(d), H.stream()):
                print(f"   {label:13s} rejected: {lib.vsseg_last_error().decode()}")
                continue
            us = timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream()))
            print(f"   {label:13s} {us:8.1f} us  {gflop / us * 1e3:7.0f} TFLOP/s  {gb / us * 1e3:5.2f} TB/s   tile={pl.tile} mtw={pl.mtw} nt={pl.nt} ck={pl.ck} lds={pl.lds}")
    # the eight per-class launches, each with its best candidate plan
    total = 0.0
    for cls in P.lattice_classes(kind, K3, ST):
        best = 1e9
        for pl in P.candidate_plans(kind, wshape, cls, COARSE, 2, kc_pad=cin, aux_es=aux_es, n=n):
            d = H.igemm_desc(pl, H.pack(pl, w, torch.bfloat16), H.tdesc(x), H.tdesc(out), **kw)
            if lib.vsseg_igemm(C.byref(d), H.stream()):
                continue
            best = min(best, timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream()), 5))
        total += best
    print(f"   {'per class':13s} {total:8.1f} us  (eight launches, best plan each)")
