"""Micro-benchmark of the HBM-bound BatchNorm/PReLU/attention kernels through the C ABI (tuning tool, not product path).

    python tools/bench_eltwise.py [--dims 384 128 128] [--batch 4] [--c 16 32] [--dtype bf16] [--reps 10]

For every channel count prints the HIP-event time and the algorithmic GB/s (bytes the kernel must read + write once).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[384, 128, 128])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--c", type=int, nargs="+", default=[16, 32])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pdrop", type=float, default=0.1)
    ap.add_argument("--pitch-mult", type=int, default=1, help="tensors are channel slices of a buffer pitch-mult times wider (concat buffers)")
    a = ap.parse_args()
    dt = H.DT[a.dtype]
    es = 2 if a.dtype == "bf16" else 4
    lib = L.lib()
    st = H.stream()
    for c in a.c:
        dims = a.dims if c <= 16 else [d // 2 if i < 2 else d for i, d in enumerate(a.dims)] if c <= 32 else [d // 4 if i < 2 else d for i, d in enumerate(a.dims)]
        shape = (a.batch, *dims, c)
        nvox = a.batch * dims[0] * dims[1] * dims[2]
        tb = nvox * c * es  # bytes of one activation tensor
        wide = (a.batch, *dims, c * a.pitch_mult)
        y = torch.randn(wide, device="cuda").to(dt)
        res = torch.randn(wide, device="cuda").to(dt)
        out = torch.empty_like(y)
        dout = torch.randn(wide, device="cuda").to(dt)
        dy = torch.empty_like(y)
        vec = torch.randn(9, c, device="cuda")
        vec[1].abs_().add_(0.5)
        alpha = torch.full((1,), 0.25, device="cuda")
        sums = torch.zeros(L.STAT_SHARDS * 3 * c + L.STAT_SHARDS, dtype=torch.float64, device="cuda")
        acc = sums[L.STAT_SHARDS * 3 * c:]
        att = torch.rand(nvox, device="cuda")
        dpre = torch.empty((a.batch, *dims, 8), device="cuda").to(dt)
        dbias = torch.zeros(1, device="cuda")
        p = [v.data_ptr() for v in vec]
        keep = torch.zeros(nvox * c // 8, dtype=torch.uint8, device="cuda")  # stored dropout keep-mask (one byte per 8-channel group)
        KEEP_W = keep.data_ptr()
        T = lambda t: H.tdesc(t, c=c) if t.shape[-1] != 8 else H.tdesc(t)  # noqa: E731
        rows = [
            ("bn_act_fwd", 2 * tb, lambda: L.check(lib.vsseg_bn_act_fwd(T(y), p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, T(res), 0, T(out), KEEP_W, st))),
            ("bn_act_fwd+res", 3 * tb, lambda: L.check(lib.vsseg_bn_act_fwd(T(y), p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, T(res), 1, T(out), KEEP_W, st))),
            ("bn_act_fwd p=0", 2 * tb, lambda: L.check(lib.vsseg_bn_act_fwd(T(y), p[2], p[3], alpha.data_ptr(), 0.0, 7, 3, T(res), 0, T(out), KEEP_W, st))),
            ("bn_act_bwd_reduce", 2 * tb, lambda: L.check(lib.vsseg_bn_act_bwd_reduce(T(y), T(dout), p[0], p[1], p[4], p[5], p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, sums.data_ptr(), c, acc.data_ptr(), None, st))),
            ("bn_act_bwd_reduce keep", 2 * tb, lambda: L.check(lib.vsseg_bn_act_bwd_reduce(T(y), T(dout), p[0], p[1], p[4], p[5], p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, sums.data_ptr(), c, acc.data_ptr(), KEEP_W, st))),
            ("bn_act_bwd_apply", 3 * tb, lambda: L.check(lib.vsseg_bn_act_bwd_apply(T(y), T(dout), p[0], p[1], p[4], p[5], p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, p[6], p[7], T(dy), None, st))),
            ("bn_act_bwd_apply keep", 3 * tb, lambda: L.check(lib.vsseg_bn_act_bwd_apply(T(y), T(dout), p[0], p[1], p[4], p[5], p[2], p[3], alpha.data_ptr(), a.pdrop, 7, 3, p[6], p[7], T(dy), KEEP_W, st))),
            ("att_apply_fwd", 2 * tb + nvox * 4, lambda: L.check(lib.vsseg_att_apply_fwd(T(y), att.data_ptr(), T(out), st))),
            ("att_apply_bwd", 3 * tb + nvox * (4 + 8 * es), lambda: L.check(lib.vsseg_att_apply_bwd(T(y), att.data_ptr(), T(dout), None, T(dy), 0, T(dpre), dbias.data_ptr(), None, st))),
            ("torch copy", 2 * tb, lambda: out.copy_(y)),
        ]
        print(f"--- c={c} dims={dims} batch={a.batch} {a.dtype}: one tensor = {tb / 1e6:.0f} MB")
        for name, nbytes, fn in rows:
            ms = timed(fn, a.reps)
            print(f"  {name:24s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.0f} GB/s")


if __name__ == "__main__":
    main()
