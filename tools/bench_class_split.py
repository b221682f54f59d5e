"""One class-split launch against the per-class launches it replaces (3x3x3 stride-(2,2,2) transposed convolution / data gradient of a strided convolution):
    python tools/bench_class_split.py [--kind conv_dgrad] [--cin 48 --cout 48 --fine 96 32 128] [--batch 4] [--accumulate]
Prints the HIP-event time of the eight per-class launches (heuristic plans) and of every class-split candidate."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="conv_dgrad")
    ap.add_argument("--cin", type=int, default=48)
    ap.add_argument("--cout", type=int, default=48)
    ap.add_argument("--fine", type=int, nargs=3, default=[96, 32, 128])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--accumulate", action="store_true")
    a = ap.parse_args()
    lib, dt, es = L.lib(), torch.bfloat16, 2
    k, st = (3, 3, 3), (2, 2, 2)
    fine, coarse = tuple(a.fine), tuple(f // 2 for f in a.fine)
    if a.kind == "convT_fwd":
        w = torch.randn(a.cin, a.cout, *k) / 10
        kin, nout = a.cin, a.cout
    else:
        w = torch.randn(a.cout, a.cin, *k) / 10
        kin, nout = a.cout, a.cin
    x = torch.randn(a.batch, *coarse, P.round_up(kin, 8), device="cuda").to(dt)
    out = torch.zeros(a.batch, *fine, nout, dtype=dt, device="cuda")
    kw = dict(accumulate=1) if a.accumulate else {}
    aux_es = es if a.accumulate else 0
    S = H.stream()
    per = []
    for cls in P.lattice_classes(a.kind, k, st):
        pl = P.plan_igemm(a.kind, tuple(w.shape), cls, coarse, es, kc_pad=x.shape[-1], aux_es=aux_es)
        per.append((pl, H.pack(pl, w, dt)))
    descs = [H.igemm_desc(pl, wp, H.tdesc(x), H.tdesc(out), **kw) for pl, wp in per]

    def run_per():
        for d in descs:
            L.check(lib.vsseg_igemm(C.byref(d), S), "igemm")

    print(f"{a.kind} {kin}->{nout} fine={fine} batch={a.batch} acc={int(a.accumulate)}")
    print(f"  per-class (8 launches, tile={per[0][0].tile} ck={per[0][0].ck}): {timed(run_per):.3f} ms")
    kreal, nreal = P.gemm_dims(a.kind, tuple(w.shape))
    for pl in P.class_split_plans(a.kind, tuple(w.shape), k, st, coarse, es, x.shape[-1], nreal, kreal, aux_es=aux_es, limit=12):
        wp = H.pack(pl, w, dt)
        d = H.igemm_desc(pl, wp, H.tdesc(x), H.tdesc(out), **kw)
        if lib.vsseg_igemm(C.byref(d), S):
            print(f"  class-split tile={pl.tile} ck={pl.ck}: rejected ({lib.vsseg_last_error().decode()})")
            continue
        print(f"  class-split tile={pl.tile} mtw={pl.mtw} nt={pl.nt} ck={pl.ck} lds={pl.lds}: {timed(lambda: lib.vsseg_igemm(C.byref(d), S)):.3f} ms")


if __name__ == "__main__":
    main()
