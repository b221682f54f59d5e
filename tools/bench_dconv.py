"""Deep-level kernel (csrc/dconv.hip, plans with depth -7) against the best other plan of every launch of levels 3-5 and of the stride-2 transitions around them:
time (HIP events, best of 7) and agreement of the results.  Usage on the GPU box: python tools/bench_dconv.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

ROUNDS = 10
K3, K1, S1, S2 = (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 2, 2)
L2, L3, L4, L5 = (96, 32, 128), (48, 16, 64), (24, 8, 32), (12, 4, 16)
# (kind, cin, cout, kernel, stride, INPUT dims of the layer, mode)
CASES = [
    ("conv_fwd", 48, 48, K3, S2, L2, "stats"), ("conv_fwd", 48, 64, K1, S1, L3, "plain"), ("conv_fwd", 64, 1, K3, S1, L3, "sigmoid"), ("conv_fwd", 128, 64, K1, S1, L3, "plain"),
    ("conv_fwd", 64, 64, K3, S2, L3, "stats"), ("conv_fwd", 64, 80, K1, S1, L4, "plain"), ("conv_fwd", 64, 80, K3, S1, L4, "stats"), ("conv_fwd", 80, 80, K3, S1, L4, "stats"),
    ("conv_fwd", 160, 80, K3, S1, L4, "stats"), ("conv_fwd", 160, 80, K3, S1, L4, "relu"), ("conv_fwd", 80, 1, K3, S1, L4, "sigmoid"), ("conv_fwd", 160, 80, K1, S1, L4, "plain"),
    ("conv_fwd", 80, 80, K3, S2, L4, "stats"), ("conv_fwd", 80, 40, K3, S1, L5, "relu"), ("conv_fwd", 40, 1, K3, S1, L5, "sigmoid"), ("conv_fwd", 80, 96, K1, S1, L5, "plain"),
    ("conv_fwd", 80, 96, K3, S1, L5, "stats"), ("conv_fwd", 96, 96, K3, S1, L5, "stats"),
    ("convT_fwd", 96, 80, K3, S2, L5, "stats"), ("convT_fwd", 80, 64, K3, S2, L4, "stats"), ("convT_fwd", 64, 48, K3, S2, L3, "stats"),
    ("conv_dgrad", 80, 80, K3, S1, L4, "plain"), ("conv_dgrad", 160, 80, K3, S1, L4, "accumulate"), ("conv_dgrad", 96, 96, K3, S1, L5, "plain"), ("conv_dgrad", 80, 40, K3, S1, L5, "accumulate"),
    ("conv_dgrad", 160, 80, K1, S1, L4, "accumulate"), ("conv_dgrad", 64, 64, K3, S2, L3, "accumulate"), ("conv_dgrad", 48, 48, K3, S2, L2, "accumulate"), ("conv_dgrad", 80, 80, K3, S2, L4, "accumulate"),
    ("convT_dgrad", 64, 48, K3, S2, L3, "plain"), ("convT_dgrad", 80, 64, K3, S2, L4, "plain"), ("convT_dgrad", 96, 80, K3, S2, L5, "plain"),
]


def time_launches(lib, descs, reps=5):
    for d in descs:
        if lib.vsseg_igemm(C.byref(d), H.stream()):
            return float("inf")
    best = 1e9
    for _ in range(reps):  # R back-to-back rounds between one pair of events: the ~5 us an event pair adds to a single launch is amortised (what hipGraph replay sees)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _r in range(ROUNDS):
            for d in descs:
                lib.vsseg_igemm(C.byref(d), H.stream())
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / ROUNDS)
    return best


def main():
    lib = L.lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    torch.manual_seed(0)
    tot_old = tot_new = 0.0
    for kind, cin, cout, kern, stride, dims_in, mode in CASES:
        if only and only not in f"{kind}:{cin}:{cout}":
            continue
        transposed = kind.startswith("convT")
        wshape = (cin, cout, *kern) if transposed else (cout, cin, *kern)
        w = torch.randn(*wshape) / (cin * np.prod(kern)) ** 0.5
        kreal, nreal = P.gemm_dims(kind, wshape)
        kc = P.round_up(kreal, 8)
        fwd_kind = "convT_fwd" if transposed else "conv_fwd"
        dims_out = P.out_dims(fwd_kind, dims_in, kern, stride)
        src_dims, dst_dims = (dims_in, dims_out) if kind.endswith("fwd") else (dims_out, dims_in)
        classes = P.lattice_classes(kind, kern, stride)
        if kind in ("conv_fwd", "convT_dgrad"):
            q = dst_dims
        else:
            q = tuple((d + s - 1) // s for d, s in zip(dst_dims, stride)) if kind == "conv_dgrad" else src_dims
        x = (torch.randn(n, *src_dims, kc, device="cuda") * 0.5).to(torch.bfloat16)
        f32out = mode == "sigmoid"
        cpad = nreal if nreal < 4 else P.round_up(nreal, 4)

        def mk_out():
            o = torch.randn(n, *dst_dims, cpad, device="cuda") * 0.25
            return o if f32out else o.to(torch.bfloat16)

        stats = torch.zeros(L.STAT_SHARDS * 2 * P.round_up(nreal, 16), dtype=torch.float64, device="cuda")
        bias = torch.randn(nreal, device="cuda") * 0.1
        kw = dict(bias=bias.data_ptr())
        if mode == "stats":
            kw.update(stats=stats.data_ptr(), stats_stride=P.round_up(nreal, 16))
        elif mode == "accumulate":
            kw.update(accumulate=1)
            kw.pop("bias")
        elif mode == "relu":
            kw.update(act=L.ACT_RELU)
        elif mode == "sigmoid":
            kw.update(act=L.ACT_SIGMOID)
        aux_es = 2 if mode == "accumulate" else 0

        def run(plans_per_class, out0):
            """plans_per_class: list of plans, one launch each (all write into one output)."""
            out = out0.clone()
            descs = []
            keep = []
            for pl in plans_per_class:
                wp = H.pack(pl, w, x.dtype)
                keep.append(wp)
                descs.append(H.igemm_desc(pl, wp, H.tdesc(x), H.tdesc(out, c=nreal), **kw))
            stats.zero_()
            ms = time_launches(lib, descs)
            # one clean run for the result
            out.copy_(out0)
            stats.zero_()
            for d in descs:
                L.check(lib.vsseg_igemm(C.byref(d), H.stream()), "igemm") if ms != float("inf") else None
            torch.cuda.synchronize()
            return ms, out.float().cpu(), H.stat_decode(stats).view(L.STAT_SHARDS, -1).sum(0).cpu().clone()  # (summed over the shards: which workgroup adds to which shard differs between kernels)

        out0 = mk_out()
        # reference: the general kernel's heuristic plans (one launch per lattice class)
        ref_plans = [P.plan_igemm(kind, wshape, cls, q, 2, kc_pad=kc, aux_es=aux_es) for cls in classes]
        ref_ms, ref_out, ref_st = run(ref_plans, out0)
        # best non-deep: per-class tuned candidates, or the class-split launch
        best_old = ref_ms
        per_class = []
        for cls in classes:
            cands = [pl for pl in P.candidate_plans(kind, wshape, cls, q, 2, kc_pad=kc, aux_es=aux_es, n=n) if pl.depth != -7]
            tms = [run([pl], out0)[0] for pl in cands]
            per_class.append(min(tms))
        best_old = min(best_old, sum(per_class))
        if len(classes) > 1:
            for pl in P.class_split_plans(kind, wshape, kern, stride, q, 2, kc, nreal, kreal, aux_es=aux_es) or []:
                best_old = min(best_old, run([pl], out0)[0])
        # deep plans
        res = []
        if len(classes) > 1:
            for pl in P.deep_class_plans(kind, wshape, kern, stride, q, 2, kc, nreal, kreal, n):
                res.append((run([pl], out0), f"classes tile={pl.tile} mt={pl.mtw} nt={pl.nt} lds={pl.lds}"))
            dpc = [P.deep_plans(kind, wshape, cls, q, 2, kc, nreal, kreal, n) for cls in classes]
            if all(dpc):
                res.append((run([d[0] for d in dpc], out0), "per class (first plan each)"))
        else:
            for pl in P.deep_plans(kind, wshape, classes[0], q, 2, kc, nreal, kreal, n):
                res.append((run([pl], out0), f"tile={pl.tile} mt={pl.mtw} nt={pl.nt} ns={pl.nsplit} ck={pl.ck} lds={pl.lds}"))
        flops = 2.0 * n * np.prod(dims_out if not transposed else dims_in) * np.prod(kern) * cin * cout / (1 if kind in ("conv_fwd", "convT_dgrad") or True else 1)
        line = f"{kind:11s} {cin:3d}->{cout:3d} k{kern[0]} s{stride[0]} in{dims_in} {mode:10s} old best {best_old * 1e3:7.1f} us (heuristic {ref_ms * 1e3:7.1f})"
        best_new = float("inf")
        for (ms, out, st), tag in res:
            err = float((out - ref_out).abs().max())
            scale = float(ref_out.abs().max())
            serr = float((st - ref_st).abs().max() / max(1.0, float(ref_st.abs().max()))) if mode == "stats" else 0.0
            flag = "" if (err <= 2e-2 * max(scale, 1.0) and serr < 1e-3) else "  <-- MISMATCH"
            print(f"    deep {ms * 1e3:7.1f} us  maxerr {err:.3e} (scale {scale:.2f}) stat relerr {serr:.1e}  {tag}{flag}")
            best_new = min(best_new, ms)
        print(f"{line} | deep best {best_new * 1e3:7.1f} us  ({flops / max(best_new, 1e-9) / 1e9:6.0f} TFLOP/s)", flush=True)
        if best_new < float("inf"):
            tot_old += best_old
            tot_new += min(best_new, best_old)
    print(f"sum over cases: old {tot_old * 1e3:.0f} us -> with deep kernel where faster {tot_new * 1e3:.0f} us")


if __name__ == "__main__":
    main()
