"""Micro-benchmark of one weight-gradient launch (tuning tool): sweeps the H-chunk group, single/double buffering and the number of
persistent workgroups.

    python tools/bench_wgrad.py --dims 192 64 128 --cin 64 --cout 32 --kernel 3 3 1 [--batch 4] [--dtype bf16]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[192, 64, 128])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--kernel", type=int, nargs=3, default=[3, 3, 1])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tile", type=int, nargs=3, default=None)
    ap.add_argument("--blocks", type=int, nargs="*", default=[0])
    ap.add_argument("--compute-only", action="store_true")
    a = ap.parse_args()
    dt = H.DT[a.dtype]
    es = 2 if a.dtype == "bf16" else 4
    lib = L.lib()
    k = tuple(a.kernel)
    wshape = (a.cout, a.cin, *k)
    wp = P.plan_wgrad(False, wshape, k, (1, 1, 1), tuple(a.dims), es)
    tile = tuple(a.tile) if a.tile else wp.tile
    p_cl = torch.randn(a.batch, *a.dims, P.round_up(a.cout, 8), device="cuda").to(dt)
    h_cl = torch.randn(a.batch, *a.dims, P.round_up(a.cin, 8), device="cuda").to(dt)
    dw = torch.zeros(int(np.prod(wshape)), dtype=torch.float32, device="cuda")
    scr = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device="cuda")
    d = L.WgradDesc()
    d.p, d.h, d.cp_valid, d.ch_valid = H.tdesc(p_cl), H.tdesc(h_cl), a.cout, a.cin
    d.q, d.hs, d.ntaps = L.i3(wp.q), L.i3(wp.hs), len(wp.taps)
    for t, (off, widx) in enumerate(wp.taps):
        d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
        d.tap_widx[t] = widx
    d.tile, d.ntp = L.i3(tile), wp.ntp
    d.dw = dw.data_ptr()
    d.stride_p, d.stride_h, d.stride_tap = wp.stride_p, wp.stride_h, wp.stride_tap
    d.scratch, d.scratch_elems = scr.data_ptr(), scr.numel()
    hch = (a.cin + 15) // 16
    tiles = a.batch
    for ax in range(3):
        tiles *= -(-a.dims[ax] // tile[ax])
    nq = a.batch * a.dims[0] * a.dims[1] * a.dims[2]
    byts = es * nq * (a.cout + a.cin)
    flops = 2.0 * nq * len(wp.taps) * a.cin * a.cout
    S = H.stream()
    print(f"wgrad {a.cin}->{a.cout} k={k} dims={a.dims} tile={tile} ntp={wp.ntp} hch={hch} tiles={tiles}  alg {byts / 1e9:.2f} GB")
    if k == (3, 3, 3):  # the compute kernel (csrc/cwgrad.hip, march = 2): hgroup = H chunks per workgroup
        for cg in (1, 2):
            d.march, d.hgroup, d.persistent_blocks = 2, cg, 0
            if lib.vsseg_wgrad(C.byref(d), S):
                print(f"  compute cg={cg}: rejected ({lib.vsseg_last_error().decode()})")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                lib.vsseg_wgrad(C.byref(d), S)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print(f"  compute cg={cg}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF  {byts / ms / 1e6:6.0f} GB/s(alg)")
        d.march = 0
        if a.compute_only:
            return
    for hg in [g for g in (1, 2, 3, 4) if hch % g == 0]:
        for sb in (0, 1):
            for blocks in a.blocks:
                d.hgroup, d.single_buffer = hg, sb
                d.persistent_blocks = blocks if blocks > 0 else max(1, min(tiles // 4, 512, 1024 // max(1, hch // hg)))
                if lib.vsseg_wgrad(C.byref(d), S):
                    print(f"  hg={hg} sb={sb} blocks={d.persistent_blocks}: rejected ({lib.vsseg_last_error().decode()})")
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    lib.vsseg_wgrad(C.byref(d), S)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.reps
                print(f"  hg={hg} sb={sb} blocks={d.persistent_blocks:5d}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF  {byts / ms / 1e6:6.0f} GB/s(alg)")


if __name__ == "__main__":
    main()
