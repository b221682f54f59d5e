"""Micro-benchmark of the loss kernels at the benchmark shape (tuning tool): every launch of the fused loss phase alone, and the coarse-level tail launch by level subset.

    python tools/bench_loss.py [--batch 4] [--dims 384 128 128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vs_seg_amd import _lib as L  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dims", type=int, nargs=3, default=[384, 128, 128])
    a = ap.parse_args()
    lib = L.lib()
    S = torch.cuda.current_stream().cuda_stream
    B, d0 = a.batch, tuple(a.dims)
    ratios = [(2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)]
    dims = [d0]
    for r in ratios:
        dims.append(tuple(x // y for x, y in zip(dims[-1], r)))
    nv = [d[0] * d[1] * d[2] for d in dims]
    lg = torch.randn(B, *d0, 2, device="cuda")
    lab = (torch.rand(B, *d0, device="cuda") > 0.97).float()
    atts = [torch.rand(B, *d, device="cuda") for d in dims]
    labels = [lab] + [torch.zeros(B, *d, device="cuda") for d in dims[1:]]
    sums = torch.zeros(B * 6 + 6 * B * 3, dtype=torch.float64, device="cuda")
    ps, asum = sums.data_ptr(), sums.data_ptr() + 8 * B * 6
    coef = torch.full((B * 4 + 6 * B * 2,), 0.01, device="cuda")
    out32 = torch.empty(B, *d0, 2, device="cuda")
    out16 = torch.empty(B, *d0, 2, device="cuda", dtype=torch.bfloat16)
    gatt = [torch.empty(B, *d, device="cuda") for d in dims]
    print(f"batch {B}, levels {dims}")
    print(f"  pred_sums                 {timed(lambda: lib.vsseg_dice_pred_sums(lg.data_ptr(), 2, lab.data_ptr(), B, nv[0], 1, ps, S)):7.1f} us")
    for l in range(6):
        print(f"  att_sums level {l}          {timed(lambda: lib.vsseg_dice_att_sums(atts[l].data_ptr(), labels[l].data_ptr(), B, nv[l], asum, S)):7.1f} us")
    for l in range(5):
        print(f"  maxpool {l}->{l + 1}              {timed(lambda: lib.vsseg_maxpool_label(labels[l].data_ptr(), B, L.i3(dims[l]), L.i3(ratios[l]), labels[l + 1].data_ptr(), S)):7.1f} us")
    print(f"  level_sums 0 (+ logits)   {timed(lambda: lib.vsseg_dice_level_sums(lg.data_ptr(), atts[0].data_ptr(), lab.data_ptr(), B, L.i3(dims[0]), 1, ps, asum, labels[1].data_ptr(), S)):7.1f} us")
    print(f"  level_sums 1              {timed(lambda: lib.vsseg_dice_level_sums(None, atts[1].data_ptr(), labels[1].data_ptr(), B, L.i3(dims[1]), 1, ps, asum, labels[2].data_ptr(), S)):7.1f} us")
    for lv in ([2, 3, 4, 5], [2], [3], [4], [5], [3, 4, 5]):
        td = L.DiceTailDesc()
        td.n, td.nlevels, td.src, td.sdims, td.sums = B, len(lv), labels[2].data_ptr(), L.i3(dims[2]), asum
        for j, l in enumerate(lv):
            td.att[j], td.label[j] = atts[l].data_ptr(), (None if l == 2 else labels[l].data_ptr())
            td.dims[j][0], td.dims[j][1], td.dims[j][2] = dims[l]
        print(f"  tail levels {str(lv):14s} {timed(lambda: L.check(lib.vsseg_dice_tail_sums(td, S))):7.1f} us")
    print(f"  finalize                  {timed(lambda: lib.vsseg_dice_finalize(ps, asum, B, 6, coef.data_ptr(), coef.data_ptr(), S)):7.1f} us")
    print(f"  pred_bwd fp32             {timed(lambda: lib.vsseg_dice_pred_bwd(lg.data_ptr(), 2, lab.data_ptr(), B, nv[0], 1, coef.data_ptr(), None, out32.data_ptr(), S)):7.1f} us")
    print(f"  pred_bwd_to bf16 compact  {timed(lambda: lib.vsseg_dice_pred_bwd_to(lg.data_ptr(), 2, lab.data_ptr(), B, nv[0], 1, coef.data_ptr(), None, L.Tensor(out16.data_ptr(), L.BF16, 2, 2, B, *d0), S)):7.1f} us")
    for l in range(6):
        print(f"  att_bwd level {l}           {timed(lambda: lib.vsseg_dice_att_bwd(labels[l].data_ptr(), B, nv[l], coef.data_ptr(), 1.0, None, gatt[l].data_ptr(), S)):7.1f} us")
    bd = L.DiceBwdLevelsDesc()
    bd.n, bd.nlevels, bd.gscale = B, 4, None
    for j, l in enumerate([2, 3, 4, 5]):
        bd.label[j], bd.datt[j], bd.nvox[j], bd.coef[j] = labels[l].data_ptr(), gatt[l].data_ptr(), nv[l], coef.data_ptr()
    print(f"  att_bwd levels 2-5        {timed(lambda: lib.vsseg_dice_att_bwd_levels(bd, S)):7.1f} us")


if __name__ == "__main__":
    main()
