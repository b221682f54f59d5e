"""Micro-benchmark of one implicit-GEMM convolution launch (tuning tool; not part of the product path).

    python tools/bench_igemm.py --dims 384 128 128 --cin 32 --cout 16 --kernel 3 3 1 [--tile 8 8 4 --mtw 4 --ck 32] [--dtype bf16] [--reps 20]

Prints average HIP-event time, algorithmic TFLOP/s and GB/s.  Run under `rocprofv3 --pmc ...` for counters.
"""
import argparse
import ctypes as C
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[384, 128, 128])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=16)
    ap.add_argument("--kernel", type=int, nargs=3, default=[3, 3, 1])
    ap.add_argument("--stride", type=int, nargs=3, default=[1, 1, 1])
    ap.add_argument("--kind", default="conv_fwd")
    ap.add_argument("--tile", type=int, nargs=3, default=None)
    ap.add_argument("--mtw", type=int, default=None)
    ap.add_argument("--ck", type=int, default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--lds-budget", type=int, default=158 * 1024)
    ap.add_argument("--depth", type=int, default=0)
    a = ap.parse_args()
    dt = H.DT[a.dtype]
    es = 2 if a.dtype == "bf16" else 4
    lib = L.lib()
    k, s = tuple(a.kernel), tuple(a.stride)
    w = torch.randn(a.cout, a.cin, *k) / (a.cin * k[0] * k[1] * k[2]) ** 0.5
    cls = P.lattice_classes(a.kind, k, s)[0]
    odims = P.out_dims("conv_fwd", a.dims, k, s)
    plan = P.plan_igemm(a.kind, tuple(w.shape), cls, odims, es, kc_pad=P.round_up(a.cin, 8), mtw=a.mtw, lds_budget=a.lds_budget, aux_es=0)
    if a.tile or a.ck:
        import dataclasses

        tile = tuple(a.tile) if a.tile else plan.tile
        ck = a.ck or plan.ck
        ksteps = (plan.ntaps * (ck // 8) + 3) // 4
        plan = dataclasses.replace(plan, tile=tile, ck=ck, nchunks=plan.kc // ck, ksteps=ksteps, mtw=tile[0] * tile[1] * tile[2] // 64)
        plan.pack_map = P.pack_map(plan, tuple(w.shape))
        plan.lds = P.igemm_lds_bytes(tile, cls.is_, cls.taps, ck, ksteps, plan.nt, plan.mtw, es, plan.kc // ck, 0)
    if a.depth:
        plan.depth = a.depth  # -1: no prefetch
        plan.lds = P.igemm_lds_bytes(plan.tile, cls.is_, cls.taps, plan.ck, plan.ksteps, plan.nt, plan.mtw, es, plan.nchunks, 0, a.depth)
    x = torch.randn(a.batch, *a.dims, P.round_up(a.cin, 8), device="cuda").to(dt)
    out = torch.zeros(a.batch, *odims, a.cout, dtype=dt, device="cuda")
    wp = H.pack(plan, w, dt)
    bias = torch.randn(a.cout, device="cuda")
    kw = dict(bias=bias.data_ptr())
    if a.stats:
        st = torch.zeros(L.STAT_SHARDS * 2 * 16 * plan.nt * plan.nsplit, dtype=torch.float64, device="cuda")
        kw.update(stats=st.data_ptr(), stats_stride=16 * plan.nt * plan.nsplit)
    d = H.igemm_desc(plan, wp, H.tdesc(x), H.tdesc(out), **kw)
    S = H.stream()
    for _ in range(3):
        L.check(lib.vsseg_igemm(C.byref(d), S), "igemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        L.check(lib.vsseg_igemm(C.byref(d), S), "igemm")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    nvox = a.batch * odims[0] * odims[1] * odims[2]
    flops = 2.0 * nvox * plan.ntaps * a.cin * a.cout
    byts = es * (x.numel() / x.shape[-1] * a.cin + nvox * a.cout)
    print(f"tile={plan.tile} mtw={plan.mtw} nt={plan.nt} ck={plan.ck} ks={plan.ksteps} D={plan.depth} lds={plan.lds}  {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s  {byts / ms / 1e6:.0f} GB/s(alg)")


if __name__ == "__main__":
    main()
