"""Marching weight gradient (march = 1, csrc/mwgrad.hip) against the tile kernel's best configuration on the level-0/1 3x3x1 layers (HIP events).
Usage on the GPU box: python tools/bench_mwgrad.py [filter]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

CASES = [(16, 16, (384, 128, 128), 0), (32, 16, (384, 128, 128), 16), (32, 2, (384, 128, 128), 0), (16, 32, (192, 64, 128), 0), (32, 32, (192, 64, 128), 0), (64, 32, (192, 64, 128), 32)]
TILES = {(16, 16): [(96, 64, 4), (192, 64, 4), (96, 32, 4), (96, 32, 8), (96, 64, 8), (192, 64, 8)], (32, 16): [(96, 64, 4), (192, 64, 4), (96, 32, 4), (192, 64, 2)], (32, 2): [(96, 64, 4), (192, 64, 4), (96, 32, 4)],
         (16, 32): [(48, 64, 4), (96, 64, 4), (48, 32, 8), (96, 32, 4)], (32, 32): [(48, 64, 4), (96, 64, 4), (96, 32, 4), (96, 64, 2)], (64, 32): [(96, 64, 2), (48, 64, 2), (96, 32, 2), (192, 32, 2), (96, 32, 4), (48, 32, 4)]}


def timed(lib, d, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.vsseg_wgrad(C.byref(d), H.stream())
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    lib = L.lib()
    n = 4
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    scr = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device="cuda")
    for cin, cout, dims, split in CASES:
        name = f"{cin}->{cout} {dims}"
        if flt not in name:
            continue
        wshape = (cout, cin, 3, 3, 1)
        wp = P.plan_wgrad(False, wshape, (3, 3, 1), (1, 1, 1), dims, 2)
        p_cl = torch.randn(n, *dims, P.round_up(cout, 8), device="cuda").to(torch.bfloat16)
        if cout < 8:
            p_cl[..., cout:] = 0
        if split:
            ha, hb = torch.randn(n, *dims, split, device="cuda").to(torch.bfloat16), torch.randn(n, *dims, cin - split, device="cuda").to(torch.bfloat16)
            hd = H.two_part(ha, hb)
        else:
            h_cl = torch.randn(n, *dims, cin, device="cuda").to(torch.bfloat16)
            hd = H.tdesc(h_cl)
        dw = torch.zeros(int(np.prod(wshape)), dtype=torch.float32, device="cuda")
        d = L.WgradDesc()
        d.p, d.h, d.cp_valid, d.ch_valid = H.tdesc(p_cl), hd, cout, cin
        d.q, d.hs, d.ntaps = L.i3(wp.q), L.i3(wp.hs), len(wp.taps)
        for t, (off, widx) in enumerate(wp.taps):
            d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
            d.tap_widx[t] = widx
        d.ntp = wp.ntp
        d.dw = dw.data_ptr()
        d.stride_p, d.stride_h, d.stride_tap = wp.stride_p, wp.stride_h, wp.stride_tap
        d.scratch, d.scratch_elems = scr.data_ptr(), scr.numel()
        nq = n * dims[0] * dims[1] * dims[2]
        gb = 2 * nq * (cout + cin) / 1e9
        tf = 2.0 * nq * 9 * cin * cout / 1e12
        print(f"== wgrad {name}: {gb:.2f} GB algorithmic, {tf * 1e3:.0f} GFLOP", flush=True)
        hch = (cin + 15) // 16
        best = None
        d.tile = L.i3(wp.tile)
        for hg in [g for g in (4, 3, 2, 1) if hch % g == 0]:
            for sb in (0, 1):
                for wpc in (2, 3, 4):
                    d.march, d.hgroup, d.single_buffer = 0, hg, sb
                    d.persistent_blocks = max(1, (256 * wpc) // max(1, hch // hg))
                    dw.zero_()
                    if lib.vsseg_wgrad(C.byref(d), H.stream()):
                        continue
                    torch.cuda.synchronize()
                    ref = dw.clone()
                    ms = timed(lib, d)
                    if best is None or ms < best[0]:
                        best = (ms, hg, sb, wpc, ref)
        print(f"   tile kernel best: {best[0]:7.3f} ms {gb / best[0]:5.2f} TB/s {tf / best[0] * 1e3:6.0f} TF  (hg={best[1]} sb={best[2]} wpc={best[3]})", flush=True)
        for tile in TILES[(cin, cout)]:
            d.march, d.tile = 1, L.i3(tile)
            dw.zero_()
            if lib.vsseg_wgrad(C.byref(d), H.stream()):
                print(f"   march tile={tile}: rejected ({lib.vsseg_last_error().decode()})")
                continue
            torch.cuda.synchronize()
            err = float((dw - best[4]).abs().max() / best[4].abs().max())
            ms = timed(lib, d)
            print(f"   march tile={str(tile):14s}: {ms:7.3f} ms {gb / ms:5.2f} TB/s {tf / ms * 1e3:6.0f} TF  rel diff vs tile kernel {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
