"""Times the compute-bound kernel (depth -3) against every plan of the general kernel on the level-2/3 3x3x3 layer shapes (HIP events, best of 5).
Usage on the GPU box: python tools/bench_cconv.py [--compute-only] [--batch N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

K3 = (3, 3, 3)
L2, L3 = (96, 32, 128), (48, 16, 64)
CASES = [("conv_fwd", 96, 48, L2, "plain"), ("conv_fwd", 96, 48, L2, "stats"), ("conv_fwd", 48, 48, L2, "stats"), ("conv_fwd", 32, 48, L2, "stats"),
         ("conv_dgrad", 96, 48, L2, "accumulate"), ("conv_dgrad", 48, 48, L2, "plain"), ("conv_dgrad", 32, 48, L2, "accumulate"),
         ("conv_fwd", 128, 64, L3, "stats"), ("conv_fwd", 64, 64, L3, "stats"), ("conv_dgrad", 64, 64, L3, "plain"), ("conv_dgrad", 128, 64, L3, "plain")]


def main():
    lib = L.lib()
    n = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
    only = "--compute-only" in sys.argv  # time the depth -3 plans only
    for kind, cin, cout, dims, mode in CASES:
        w = torch.randn(cout, cin, *K3) / (cin * 27) ** 0.5
        kreal, nreal = P.gemm_dims(kind, tuple(w.shape))
        kc = P.round_up(kreal, 16)
        cls = P.lattice_classes(kind, K3, (1, 1, 1))[0]
        x = torch.randn(n, *dims, kc, device="cuda").to(torch.bfloat16)
        out = torch.zeros(n, *dims, nreal, dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros(L.STAT_SHARDS * 2 * P.round_up(nreal, 16), dtype=torch.float64, device="cuda")
        kw = dict(stats=stats.data_ptr(), stats_stride=P.round_up(nreal, 16)) if mode == "stats" else (dict(accumulate=1) if mode == "accumulate" else {})
        cands = P.candidate_plans(kind, tuple(w.shape), cls, dims, 2, kc_pad=kc, aux_es=2 if mode == "accumulate" else 0)
        res = []
        if only:
            cands = [pl for pl in cands if pl.depth == -3]
        for pl in cands:
            d = H.igemm_desc(pl, H.pack(pl, w, x.dtype), H.tdesc(x), H.tdesc(out), **kw)
            if lib.vsseg_igemm(C.byref(d), H.stream()):
                res.append((float("inf"), pl))
                continue
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.vsseg_igemm(C.byref(d), H.stream())
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            res.append((best, pl))
        tf = 2.0 * n * np.prod(dims) * 27 * kreal * nreal / 1e9  # GFLOP
        if only:
            print(f"{kind} K={kreal} N={nreal} {dims} x{n} {mode}: compute kernel " + ", ".join(f"{r[0] * 1e3:.1f} us ({tf / r[0]:.0f} TFLOP/s)" for r in res), flush=True)
            continue
        bg = min((r for r in res if r[1].depth != -3), key=lambda r: r[0])
        cc = [r for r in res if r[1].depth == -3]
        print(f"{kind} K={kreal} N={nreal} {dims} {mode}: general best {bg[0]:.3f} ms ({tf / bg[0]:.0f} TFLOP/s, tile={bg[1].tile} nt={bg[1].nt} ck={bg[1].ck} ns={bg[1].nsplit})"
              + (f" | compute kernel {cc[0][0]:.3f} ms ({tf / cc[0][0]:.0f} TFLOP/s)" if cc else " | compute kernel n/a"), flush=True)


if __name__ == "__main__":
    main()
