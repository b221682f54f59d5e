"""Times the streaming kernel (depth -2) against the general kernel's default plan on the level-0/1 layer shapes (HIP events, best of 5).
Usage on the GPU box: python tools/bench_sconv.py   (VSSEG_SCONV_PERCU=n caps the resident workgroups per CU)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

CASES = [("conv_dgrad", (3, 3, 1), 32, 2, (384, 128, 128), "plain"), ("conv_fwd", (3, 3, 1), 32, 2, (384, 128, 128), "plain"), ("conv_fwd", (3, 3, 1), 64, 32, (192, 64, 128), "stats"), ("conv_fwd", (1, 1, 1), 64, 32, (192, 64, 128), "plain"), ("conv_fwd", (3, 3, 1), 16, 16, (384, 128, 128), "plain"), ("conv_fwd", (3, 3, 1), 16, 16, (384, 128, 128), "stats"), ("conv_fwd", (3, 3, 1), 32, 16, (384, 128, 128), "stats"),
         ("conv_dgrad", (3, 3, 1), 32, 16, (384, 128, 128), "accumulate"), ("conv_dgrad", (3, 3, 1), 16, 16, (384, 128, 128), "plain"), ("conv_fwd", (3, 3, 1), 32, 32, (192, 64, 128), "stats"),
         ("conv_dgrad", (3, 3, 1), 64, 32, (192, 64, 128), "plain"), ("conv_dgrad", (1, 1, 1), 64, 32, (192, 64, 128), "accumulate")]


def main():
    lib = L.lib()
    n = 4
    for kind, k, cin, cout, dims, mode in CASES:
        w = torch.randn(cout, cin, *k) / (cin * np.prod(k)) ** 0.5
        kreal, nreal = P.gemm_dims(kind, tuple(w.shape))
        kc = P.round_up(kreal, 8)
        cls = P.lattice_classes(kind, k, (1, 1, 1))[0]
        x = torch.randn(n, *dims, kc, device="cuda").to(torch.bfloat16)
        out = torch.zeros(n, *dims, nreal, dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros(L.STAT_SHARDS * 2 * P.round_up(nreal, 16), dtype=torch.float64, device="cuda")
        kw = dict(stats=stats.data_ptr(), stats_stride=P.round_up(nreal, 16)) if mode == "stats" else (dict(accumulate=1) if mode == "accumulate" else {})
        cands = P.candidate_plans(kind, tuple(w.shape), cls, dims, 2, kc_pad=kc, aux_es=2 if mode == "accumulate" else 0)
        res = []
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
        gb = n * np.prod(dims) * (kreal + nreal * (2 if mode == "accumulate" else 1)) * 2 / 1e9
        if os.environ.get("VSSEG_BENCH_ALL"):
            for ms, pl in res:
                print(f"    {ms:.3f} ms tile={pl.tile} nt={pl.nt} ck={pl.ck} ns={pl.nsplit} D={pl.depth}")
        bg = min((r for r in res if r[1].depth != -2), key=lambda r: r[0])
        st = [r for r in res if r[1].depth == -2]
        res = None
        print(f"{kind} {k} K={kreal} N={nreal} {dims} {mode}: general best {bg[0]:.3f} ms ({gb / bg[0]:.0f} GB/s, tile={bg[1].tile} ck={bg[1].ck} ns={bg[1].nsplit} D={bg[1].depth})"
              + (f" | streaming {st[0][0]:.3f} ms ({gb / st[0][0]:.0f} GB/s)" if st else " | streaming n/a"), flush=True)


if __name__ == "__main__":
    main()
