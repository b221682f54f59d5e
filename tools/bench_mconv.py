"""Marching kernel (depth -5) against the streaming kernel (depth -2) and the general kernel on the level-0/1 3x3x1 layer shapes: bit-identity of
the outputs and HIP-event times (best of 5).  Usage on the GPU box: python tools/bench_mconv.py [filter]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

# (kind, cin, cout, dims, mode, in_split)
CASES = [("conv_fwd", 1, 16, (384, 128, 128), "stats", 0), ("conv_dgrad", 16, 1, (384, 128, 128), "accumulate", 0), ("conv_dgrad", 32, 2, (384, 128, 128), "plain", 0), ("conv_fwd", 16, 16, (384, 128, 128), "stats", 0), ("conv_dgrad", 16, 16, (384, 128, 128), "plain", 0), ("conv_fwd", 32, 16, (384, 128, 128), "plain", 16),
         ("conv_dgrad", 32, 16, (384, 128, 128), "gate", 0), ("conv_fwd", 32, 2, (384, 128, 128), "plain", 0),
         ("conv_fwd", 16, 32, (192, 64, 128), "stats", 0), ("conv_fwd", 32, 32, (192, 64, 128), "stats", 0), ("conv_dgrad", 32, 32, (192, 64, 128), "plain", 0),
         ("conv_fwd", 64, 32, (192, 64, 128), "plain", 32), ("conv_dgrad", 64, 32, (192, 64, 128), "accumulate", 0), ("conv_dgrad", 32, 16, (192, 64, 128), "plain", 0)]


def main():
    lib = L.lib()
    n = int(os.environ.get("N", "4"))
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for kind, cin, cout, dims, mode, in_split in CASES:
        name = f"{kind} {cin}->{cout} {dims} {mode}" + (f" split{in_split}" if in_split else "")
        if flt not in name:
            continue
        torch.manual_seed(0)
        w = torch.randn(cout, cin, 3, 3, 1) / (cin * 9) ** 0.5
        kreal, nreal = P.gemm_dims(kind, tuple(w.shape))
        kc = P.round_up(kreal, 8)
        cls = P.lattice_classes(kind, (3, 3, 1), (1, 1, 1))[0]
        if in_split:
            xa = torch.randn(n, *dims, in_split, device="cuda").to(torch.bfloat16)
            xb = torch.randn(n, *dims, kc - in_split, device="cuda").to(torch.bfloat16)
            xin = H.two_part(xa, xb)
        else:
            x = torch.randn(n, *dims, kc, device="cuda").to(torch.bfloat16)
            xin = H.tdesc(x)
        odt = torch.float32 if nreal < 4 else torch.bfloat16
        out0 = torch.randn(n, *dims, nreal, device="cuda").to(odt)
        stats = torch.zeros(L.STAT_SHARDS * 2 * P.round_up(nreal, 16), dtype=torch.float64, device="cuda")
        kw, aux_es, extra = {}, 0, 1
        keep = []
        if mode == "stats":
            kw = dict(stats=stats.data_ptr(), stats_stride=P.round_up(nreal, 16))
        elif mode == "accumulate":
            kw, aux_es, extra = dict(accumulate=1), 2, 2
        elif mode == "gate":
            res = torch.randn(n, *dims, nreal, device="cuda").to(torch.bfloat16)
            gate = torch.rand(n, *dims, device="cuda")
            keep += [res, gate]
            kw, aux_es, extra = dict(res=H.tdesc(res), res_mode=L.RES_GATE, gate=gate.data_ptr()), 2, 2
        bias = torch.randn(nreal, device="cuda")
        kw["bias"] = bias.data_ptr()
        cands = P.candidate_plans(kind, tuple(w.shape), cls, dims, 2, kc_pad=kc, aux_es=aux_es, in_split=in_split, n=n)
        ref, rows = None, []
        for pl in cands:
            if pl.depth not in (-2, -5) and pl is not cands[0]:
                continue
            out = out0.clone()
            stats.zero_()
            d = H.igemm_desc(pl, H.pack(pl, w, torch.bfloat16), xin, H.tdesc(out), **kw)
            if lib.vsseg_igemm(C.byref(d), H.stream()):
                rows.append((float("inf"), pl, "rejected: " + (L.lib().vsseg_last_error() or b"").decode()))
                continue
            torch.cuda.synchronize()
            res_out, res_stats = out.clone(), stats.clone()
            if ref is None:
                ref = (res_out, res_stats)
                same = "ref"
            else:
                same = "bit-identical" if torch.equal(res_out, ref[0]) else f"DIFFERS max {float((res_out.float() - ref[0].float()).abs().max()):.3e}"
                if mode == "stats":
                    same += f" stats rel {float((res_stats - ref[1]).abs().max() / ref[1].abs().max()):.1e}"
            best = 1e9
            for _ in range(5):
                if mode == "accumulate":
                    out.copy_(out0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.vsseg_igemm(C.byref(d), H.stream())
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            rows.append((best, pl, same))
        gb = n * np.prod(dims) * (kreal * 2 + nreal * out0.element_size() + (nreal * 2 if extra == 2 else 0)) / 1e9
        print(f"== {name}: {gb:.2f} GB algorithmic", flush=True)
        for ms, pl, same in rows:
            print(f"   {ms:7.3f} ms {gb / ms:6.0f} GB/s  D={pl.depth:2d} tile={pl.tile} mtw={pl.mtw} nt={pl.nt} ck={pl.ck} lds={pl.lds}  {same}", flush=True)


if __name__ == "__main__":
    main()
