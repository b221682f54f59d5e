"""Micro-benchmark of the narrow-output convolution (csrc/nconv.hip, vsseg_conv_to1) against the launches it replaces: python tools/bench_nconv.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402

lib = L.lib()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for c, dims in ((16, (384, 128, 128)), (32, (192, 64, 128))):
    x = torch.randn(batch, *dims, c, device="cuda").to(torch.bfloat16)
    w = (torch.randn(1, c, 3, 3, 1) * 0.1).reshape(-1).cuda()
    b = torch.zeros(1, device="cuda")
    out = torch.zeros(batch, *dims, 1, device="cuda")
    od = L.Tensor(out.data_ptr(), L.F32, 1, 1, batch, *dims)
    byts = x.numel() * 2 + out.numel() * 4
    for lx in (0, 192, 96, 48, 24, 12, 6):
        if lx > dims[0]:
            continue
        L.check(lib.vsseg_conv_to1(H.tdesc(x), w.data_ptr(), b.data_ptr(), L.ACT_SIGMOID, od, lx, H.stream()), "conv_to1")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.vsseg_conv_to1(H.tdesc(x), w.data_ptr(), b.data_ptr(), L.ACT_SIGMOID, od, lx, H.stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{c} -> 1 at {dims} x {batch}: lx={lx:4d}  {ms * 1e3:7.1f} us  {byts / ms / 1e9:6.2f} TB/s")
