"""The two convolution pairs of an eval window that vsseg_conv_chain fuses, at benchmark size (batch N, 384x128x128): best marching plan of each of the two launches against the
chained launch over its plan space (tz, mtw, lx).  python tools/bench_chain.py [N]   (10 back-to-back launches per timing)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = L.lib()
ROUNDS = 10


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ROUNDS):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / ROUNDS)
    return best * 1e3


def case(name, cin, cout, dims, split):
    k, cm = (3, 3, 1), 16
    torch.manual_seed(1)
    res1 = cin == 1
    kc = P.round_up(cin, 8)
    x_cl = torch.randn(N, *dims, kc, device="cuda").to(torch.bfloat16)
    if res1:
        x_cl[..., 1:] = 0
    wa = torch.randn(cm, cin, *k) / (cin * 9) ** 0.5
    wb = torch.randn(cout, cm, *k) / (cm * 9) ** 0.5
    vec = lambda n: torch.randn(n, device="cuda")
    ba, sca, sha, bb, scb, shb, w1, b1 = vec(cm), vec(cm), vec(cm), vec(cout), vec(cout), vec(cout), vec(cout), vec(cout)
    al = torch.tensor([0.2], device="cuda")
    compact = x_cl[..., :1].contiguous() if res1 else None
    parts = H._split_cl(x_cl, split) if split else None
    xin = H.tdesc(compact) if res1 else (H.two_part(*parts) if parts else H.tdesc(x_cl))
    cls = P.lattice_classes("conv_fwd", k, (1, 1, 1))[0]
    h_cl = torch.zeros(N, *dims, cm, dtype=torch.bfloat16, device="cuda")
    odt = torch.bfloat16 if res1 else torch.float32
    out = torch.zeros(N, *dims, cout, dtype=odt, device="cuda")
    act_b = L.ACT_PRELU if res1 else L.ACT_SIGMOID

    def plans(w, kcp, inp, outp, **kw):
        kreal, nreal = P.gemm_dims("conv_fwd", tuple(w.shape))
        best = (1e9, None, None, None)
        for pl in P.march_plans("conv_fwd", tuple(w.shape), cls, dims, 2, kcp, nreal, kreal, n=N):
            if inp.c == 1 and pl.depth != -5:
                continue
            pl.pack_map = P.pack_map(pl, tuple(w.shape))
            wp = H.pack(pl, w, torch.bfloat16)
            d = H.igemm_desc(pl, wp, inp, outp, **kw)
            if lib.vsseg_igemm(C.byref(d), H.stream()):
                continue
            t = timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream()))
            if t < best[0]:
                best = (t, pl, wp, d)
        return best

    ta, pa, wpa, _ = plans(wa, kc, xin, H.tdesc(h_cl), bias=ba.data_ptr(), act=(L.ACT_PRELU if res1 else L.ACT_RELU), alpha=al.data_ptr(), **(dict(scale=sca.data_ptr(), shift=sha.data_ptr()) if res1 else {}))
    kw = dict(bias=bb.data_ptr(), act=act_b, alpha=al.data_ptr())
    if res1:
        kw.update(scale=scb.data_ptr(), shift=shb.data_ptr(), res_mode=L.RES_IN1, in1=compact.data_ptr(), in1_w=w1.data_ptr(), in1_b=b1.data_ptr())
    tb, pb, wpb, _ = plans(wb, cm, H.tdesc(h_cl), H.tdesc(out), **kw)
    print(f"{name}: N={N} marching launches {ta:.1f} us (tile {pa.tile} D={pa.depth}) + {tb:.1f} us (tile {pb.tile} D={pb.depth}) = {ta + tb:.1f} us")
    # packed weights of the chain: nt = 1 marching packs
    def pack1(w, kcp):
        kreal, nreal = P.gemm_dims("conv_fwd", tuple(w.shape))
        pl = [p_ for p_ in P.march_plans("conv_fwd", tuple(w.shape), cls, dims, 2, kcp, nreal, kreal, n=N) if p_.depth == -5 and p_.nt == 1][0]
        pl.pack_map = P.pack_map(pl, tuple(w.shape))
        return H.pack(pl, w, torch.bfloat16)
    wpa, wpb = pack1(wa, kc), pack1(wb, cm)
    for tz, waves, lead in ((1, 8, 1), (2, 8, 1), (2, 16, 1), (4, 16, 1), (2, 8, 3), (2, 16, 3), (4, 8, 3), (4, 16, 3)):
        mtw = dims[1] * tz // (16 * waves)
        for nxs in (2, 4, 6, 8, 12):
            lx = -(-dims[0] // nxs)
            d = L.ChainDesc()
            d.inp, d.out, d.cmid = xin, H.tdesc(out), cm
            d.wpack_a, d.bias_a, d.alpha_a, d.act_a = wpa.data_ptr(), ba.data_ptr(), al.data_ptr(), (L.ACT_PRELU if res1 else L.ACT_RELU)
            if res1:
                d.scale_a, d.shift_a = sca.data_ptr(), sha.data_ptr()
            d.wpack_b, d.bias_b, d.alpha_b, d.act_b = wpb.data_ptr(), bb.data_ptr(), al.data_ptr(), act_b
            if res1:
                d.scale_b, d.shift_b, d.in1_w, d.in1_b = scb.data_ptr(), shb.data_ptr(), w1.data_ptr(), b1.data_ptr()
            d.tz, d.mtw, d.lx, d.waves, d.lead = tz, mtw, lx, waves, lead
            lds = lib.vsseg_conv_chain_lds_bytes(C.byref(d))
            if lds < 0:
                continue
            if lib.vsseg_conv_chain(C.byref(d), H.stream()):
                print("  launch failed:", lib.vsseg_last_error())
                continue
            t = timed(lambda: lib.vsseg_conv_chain(C.byref(d), H.stream()))
            wgs = N * nxs * (dims[2] // tz)
            print(f"  chain tz={tz} waves={waves} mtw={mtw} lead={lead} lx={lx:3d} ({wgs:4d} workgroups, {lds // 1024} KB LDS): {t:.1f} us")


case("encoder unit 1 -> 16 -> 16 (+ residual of the input)", 1, 16, (384, 128, 128), 0)
case("attention block 32 -> 16 -> 1", 32, 1, (384, 128, 128), 16)


def case_res(name, dims):
    """Level-1 ResidualUnit: 16 -> 32 (+ residual tiles 16 -> 32) and 32 -> 32 + residual add as two marching launches against the chained launch."""
    k, cin, cm, cout = (3, 3, 1), 16, 32, 32
    torch.manual_seed(2)
    x_cl = torch.randn(N, *dims, cin, device="cuda").to(torch.bfloat16)
    wa, wb, wr = torch.randn(cm, cin, *k) / 12, torch.randn(cout, cm, *k) / 17, torch.randn(cout, cin, 1, 1, 1) / 4
    vec = lambda n: torch.randn(n, device="cuda")
    ba, sca, sha, bb, scb, shb, br = vec(cm), vec(cm), vec(cm), vec(cout), vec(cout), vec(cout), vec(cout)
    al = torch.tensor([0.2], device="cuda")
    cls = P.lattice_classes("conv_fwd", k, (1, 1, 1))[0]
    h_cl = torch.zeros(N, *dims, cm, dtype=torch.bfloat16, device="cuda")
    r_cl = torch.zeros(N, *dims, cout, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(N, *dims, cout, dtype=torch.bfloat16, device="cuda")
    rmap = torch.from_numpy(P.residual_tile_pack_map(cin, 2, tuple(wr.shape))).cuda()
    wflat = wr.reshape(-1).to(torch.bfloat16).cuda()
    wpr = torch.where(rmap >= 0, wflat[rmap.clamp(min=0).long()], torch.zeros((), dtype=torch.bfloat16, device="cuda"))
    # A with residual tiles riding along
    best_a = 1e9
    for pl in P.march_res_plans(tuple(wa.shape), tuple(wr.shape), cls, dims, 2, cin, N):
        wp = H.pack(pl, wa, torch.bfloat16)
        d = H.igemm_desc(pl, wp, H.tdesc(x_cl), H.tdesc(h_cl), bias=ba.data_ptr(), scale=sca.data_ptr(), shift=sha.data_ptr(), act=L.ACT_PRELU, alpha=al.data_ptr(), res_tiles=2, wpack_res=wpr.data_ptr(),
                         bias_res=br.data_ptr(), res_out=H.tdesc(r_cl))
        if lib.vsseg_igemm(C.byref(d), H.stream()):
            continue
        best_a = min(best_a, timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream())))
    best_b = 1e9
    kreal, nreal = P.gemm_dims("conv_fwd", tuple(wb.shape))
    cands = P.candidate_plans("conv_fwd", tuple(wb.shape), cls, dims, 2, kc_pad=cm, aux_es=2, in_split=0, n=N)
    for pl in cands:
        if pl.pack_map is None:
            pl.pack_map = P.pack_map(pl, tuple(wb.shape))
        wp = H.pack(pl, wb, torch.bfloat16)
        d = H.igemm_desc(pl, wp, H.tdesc(h_cl), H.tdesc(out), bias=bb.data_ptr(), scale=scb.data_ptr(), shift=shb.data_ptr(), act=L.ACT_PRELU, alpha=al.data_ptr(), res=H.tdesc(r_cl), res_mode=L.RES_ADD)
        if lib.vsseg_igemm(C.byref(d), H.stream()):
            continue
        best_b = min(best_b, timed(lambda: lib.vsseg_igemm(C.byref(d), H.stream())))
    print(f"{name}: N={N} {best_a:.1f} us (16 -> 32 + residual tiles) + {best_b:.1f} us (32 -> 32 + add) = {best_a + best_b:.1f} us")

    def pack1(w, kcp, nt):
        kr_, nr_ = P.gemm_dims("conv_fwd", tuple(w.shape))
        pl = [p_ for p_ in P.march_plans("conv_fwd", tuple(w.shape), cls, dims, 2, kcp, nr_, kr_, n=N) if p_.depth == -5 and p_.nt == nt][0]
        pl.pack_map = P.pack_map(pl, tuple(w.shape))
        return H.pack(pl, w, torch.bfloat16)
    wpa, wpb = pack1(wa, cin, 2), pack1(wb, cm, 2)
    for tz, waves in ((2, 8), (4, 8)):
        mtw = dims[1] * tz // (16 * waves)
        for nxs in (2, 4, 6, 8, 12, 16):
            lx = -(-dims[0] // nxs)
            d = L.ChainDesc()
            d.inp, d.out, d.cmid = H.tdesc(x_cl), H.tdesc(out), cm
            d.wpack_a, d.bias_a, d.scale_a, d.shift_a, d.alpha_a, d.act_a = wpa.data_ptr(), ba.data_ptr(), sca.data_ptr(), sha.data_ptr(), al.data_ptr(), L.ACT_PRELU
            d.wpack_b, d.bias_b, d.scale_b, d.shift_b, d.alpha_b, d.act_b = wpb.data_ptr(), bb.data_ptr(), scb.data_ptr(), shb.data_ptr(), al.data_ptr(), L.ACT_PRELU
            d.res_tiles, d.wpack_res, d.bias_res = 2, wpr.data_ptr(), br.data_ptr()
            d.tz, d.mtw, d.lx, d.waves, d.lead = tz, mtw, lx, waves, 1
            lds = lib.vsseg_conv_chain_lds_bytes(C.byref(d))
            if lds < 0 or lib.vsseg_conv_chain(C.byref(d), H.stream()):
                continue
            t = timed(lambda: lib.vsseg_conv_chain(C.byref(d), H.stream()))
            print(f"  chain tz={tz} waves={waves} mtw={mtw} lead=1 lx={lx:3d} ({N * nxs * (dims[2] // tz):4d} workgroups, {lds // 1024} KB LDS): {t:.1f} us")


case_res("level-1 ResidualUnit 16 -> 32 -> 32 (+ residual convolution of the input)", (192, 64, 128))
