"""CPU tests of the host-side planner: lattice classes, tap tables and weight pack maps.

`planner.simulate_igemm` restates igemm_kernel's indexing literally in numpy, so a wrong tap offset, parity class or
pack-map entry fails here, without a GPU.  The expected values come from the convolution definitions themselves
(torch.nn.functional on CPU and its autograd, the same calls the oracle makes).
"""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vs_seg_amd import planner as P

CASES = [
    # kernel, stride, cin, cout, dims
    ((3, 3, 1), (1, 1, 1), 16, 16, (8, 8, 4)),
    ((3, 3, 3), (1, 1, 1), 16, 40, (6, 8, 4)),
    ((1, 1, 1), (1, 1, 1), 32, 16, (4, 4, 4)),
    ((3, 3, 1), (2, 2, 1), 16, 16, (8, 8, 4)),
    ((3, 3, 3), (2, 2, 2), 24, 16, (8, 4, 8)),
    ((3, 3, 3), (1, 1, 1), 8, 2, (4, 4, 4)),
    ((3, 3, 3), (1, 1, 1), 112, 100, (4, 4, 2)),  # nsplit=2 and several chunks
]


def _cl(x):  # NCDHW torch -> NDHWC numpy
    return x.permute(0, 2, 3, 4, 1).contiguous().numpy()


def _run(kind, w, x_cl, out_shape, lds_budget=64 * 1024):
    out = None
    wshape = tuple(w.shape)
    k = wshape[2:]
    classes = P.lattice_classes(kind, k, _run.stride)
    for cls in classes:
        if kind in ("conv_fwd", "convT_dgrad"):
            q = out_shape
        else:
            q = tuple((o + s - 1) // s for o, s in zip(out_shape, _run.stride))
        plan = P.plan_igemm(kind, wshape, cls, q, es=2, kc_pad=x_cl.shape[-1], lds_budget=lds_budget)
        assert plan.tile[0] * plan.tile[1] * plan.tile[2] == 64 * plan.mtw
        assert plan.lds <= P.LDS_LIMIT
        part = P.simulate_igemm(plan, x_cl, w.numpy().reshape(-1), out_shape)
        out = part if out is None else out + part
    return out


@pytest.mark.parametrize("k,s,cin,cout,dims", CASES)
def test_conv_fwd_and_dgrad(k, s, cin, cout, dims):
    torch.manual_seed(0)
    pad = P.same_pad(k)
    x = torch.randn(2, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, *k, dtype=torch.float64)
    y = F.conv3d(x, w, stride=s, padding=pad)
    _run.stride = s
    got = _run("conv_fwd", w, _cl(x.detach()), tuple(y.shape[2:]))
    np.testing.assert_allclose(got, _cl(y.detach()), atol=1e-9)
    gy = torch.randn_like(y)
    y.backward(gy)
    cpad = P.round_up(cout, 8)
    gy_cl = np.zeros((2, *y.shape[2:], cpad))
    gy_cl[..., :cout] = _cl(gy)
    got = _run("conv_dgrad", w, gy_cl, dims)
    np.testing.assert_allclose(got, _cl(x.grad), atol=1e-9)


@pytest.mark.parametrize("k,s,cin,cout,dims", [((3, 3, 1), (2, 2, 1), 16, 8, (4, 4, 4)), ((3, 3, 3), (2, 2, 2), 16, 24, (4, 2, 4)), ((3, 3, 3), (2, 2, 2), 96, 80, (2, 2, 2))])
def test_convT_fwd_and_dgrad(k, s, cin, cout, dims):
    torch.manual_seed(1)
    pad = P.same_pad(k)
    opad = tuple(ss + 2 * p - (kk - 1) - 1 for ss, p, kk in zip(s, pad, k))
    x = torch.randn(2, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cin, cout, *k, dtype=torch.float64)
    y = F.conv_transpose3d(x, w, stride=s, padding=pad, output_padding=opad)
    assert tuple(y.shape[2:]) == P.out_dims("convT_fwd", dims, k, s)
    _run.stride = s
    got = _run("convT_fwd", w, _cl(x.detach()), tuple(y.shape[2:]))
    np.testing.assert_allclose(got, _cl(y.detach()), atol=1e-9)
    gy = torch.randn_like(y)
    y.backward(gy)
    got = _run("convT_dgrad", w, _cl(gy), dims)
    np.testing.assert_allclose(got, _cl(x.grad), atol=1e-9)


@pytest.mark.parametrize("k,s,cin,cout,dims", [((3, 3, 3), (2, 2, 2), 16, 24, (4, 2, 4)), ((3, 3, 3), (2, 2, 2), 48, 80, (2, 4, 2)), ((3, 3, 1), (2, 2, 1), 16, 8, (4, 4, 4)), ((3, 3, 3), (2, 2, 2), 8, 16, (3, 3, 3))])
def test_class_split_plans_cover_every_parity_class_in_one_launch(k, s, cin, cout, dims):
    """planner.class_split_plans: workgroup row = output-parity class, each with its own taps / K steps (the odd-sized case leaves the
    last voxel of the classes with offset 1 outside the output)."""
    torch.manual_seed(5)
    pad = P.same_pad(k)
    opad = tuple(ss + 2 * p - (kk - 1) - 1 for ss, p, kk in zip(s, pad, k))
    x = torch.randn(2, cin, *dims, dtype=torch.float64)
    w = torch.randn(cin, cout, *k, dtype=torch.float64)
    y = F.conv_transpose3d(x, w, stride=s, padding=pad, output_padding=opad)
    kreal, nreal = P.gemm_dims("convT_fwd", tuple(w.shape))
    plans = P.class_split_plans("convT_fwd", tuple(w.shape), k, s, dims, 2, P.round_up(cin, 8), nreal, kreal)
    assert plans
    xc = np.zeros((2, *dims, P.round_up(cin, 8)))
    xc[..., :cin] = _cl(x)
    for pl in plans:
        assert pl.nsplit == len(pl.classes) == int(np.prod(s)) and pl.cls.oo == (0, 0, 0) and pl.lds <= P.LDS_LIMIT
        for c_i, c_ in enumerate(pl.classes):
            assert [pl.cls.taps[i][0] for i in pl.class_taps(c_i)] == [off for off, _ in c_.taps]
        got = P.simulate_igemm(pl, xc, w.numpy().reshape(-1), tuple(y.shape[2:]))
        np.testing.assert_allclose(got, _cl(y), atol=1e-9)
    # the data gradient of the strided convolution: same lattice, K = cout
    w2 = torch.randn(cout, cin, *k, dtype=torch.float64)
    x2 = torch.randn(1, cin, *[d * ss - (1 if d == 3 else 0) for d, ss in zip(dims, s)], dtype=torch.float64, requires_grad=True)
    y2 = F.conv3d(x2, w2, stride=s, padding=pad)
    gy = torch.randn_like(y2)
    y2.backward(gy)
    kreal, nreal = P.gemm_dims("conv_dgrad", tuple(w2.shape))
    q = tuple((d + ss - 1) // ss for d, ss in zip(x2.shape[2:], s))
    gyc = np.zeros((1, *y2.shape[2:], P.round_up(cout, 8)))
    gyc[..., :cout] = _cl(gy)
    for pl in P.class_split_plans("conv_dgrad", tuple(w2.shape), k, s, q, 2, P.round_up(cout, 8), nreal, kreal):
        np.testing.assert_allclose(P.simulate_igemm(pl, gyc, w2.numpy().reshape(-1), tuple(x2.shape[2:])), _cl(x2.grad), atol=1e-9)


@pytest.mark.parametrize("kind,cin,cout", [("convT_fwd", 64, 48), ("conv_dgrad", 48, 48)])
def test_transition_plans_compute_the_transition_and_mirror_the_kernels_lds(kind, cin, cout):
    """planner.transition_plans (csrc/tconv.hip, depth -8): one plan whose eight workgroup stages are the parity classes of the 3x3x3 stride-(2,2,2) transposed convolution /
    strided data gradient between levels 2 and 3; its packed weights simulate to torch's result (numpy restatement of the kernel's indexing), its LDS request is the kernel's,
    and nothing outside the two layers gets a plan."""
    import ctypes

    from tests import gpu_harness as H
    from vs_seg_amd import _lib as L

    k, s, coarse = (3, 3, 3), (2, 2, 2), (4, 8, 8)
    fine = tuple(2 * c for c in coarse)
    torch.manual_seed(6)
    if kind == "convT_fwd":
        x = torch.randn(1, cin, *coarse, dtype=torch.float64)
        w = torch.randn(cin, cout, *k, dtype=torch.float64)
        want = F.conv_transpose3d(x, w, stride=s, padding=1, output_padding=1)
        xin, kc = _cl(x), cin
    else:
        w = torch.randn(cout, cin, *k, dtype=torch.float64)
        xd = torch.zeros(1, cin, *fine, dtype=torch.float64, requires_grad=True)
        y = F.conv3d(xd, w, stride=s, padding=1)
        gy = torch.randn_like(y)
        y.backward(gy)
        want, xin, kc = xd.grad, _cl(gy), cout
    kreal, nreal = P.gemm_dims(kind, tuple(w.shape))
    plans = P.transition_plans(kind, tuple(w.shape), k, s, coarse, 2, kc, nreal, kreal)
    assert len(plans) == 1
    pl = plans[0]
    assert pl.depth == -8 and pl.tile == P.TRANSITION_TILE and pl.mtw == 16 and pl.nt == 3 and pl.nsplit == 8 and len(pl.classes) == 8 and pl.ck == kc and pl.nchunks == 1
    assert sorted(len(c.taps) for c in pl.classes) == [1, 2, 2, 2, 4, 4, 4, 8] and pl.lds == P.transition_lds_bytes(kc) <= P.LDS_LIMIT
    np.testing.assert_allclose(P.simulate_igemm(pl, xin, w.numpy().reshape(-1), fine), _cl(want), atol=1e-9)
    d = H.igemm_desc(pl, torch.zeros(1), L.Tensor(4096, L.BF16, kc, kc, 1, *coarse), L.Tensor(8192, L.BF16, 48, 48, 1, *fine))
    assert L.lib().vsseg_igemm_lds_bytes(ctypes.byref(d)) == pl.lds, L.lib().vsseg_last_error()
    assert P.transition_plans(kind, tuple(w.shape), k, s, (6, 8, 8), 2, kc, nreal, kreal) == []     # extent not a multiple of the 4x8x8 tile
    assert P.transition_plans(kind, tuple(w.shape), k, s, coarse, 4, kc, nreal, kreal) == []        # fp32
    assert P.transition_plans("convT_fwd", (80, 64, *k), k, s, (8, 8, 8), 2, 80, 64, 80) == []      # levels 3 -> 4: the deep-level kernel's


@pytest.mark.parametrize("kind,cin,cout,coarse", [("conv_fwd", 16, 16, (3, 32, 4)), ("conv_fwd", 32, 32, (2, 32, 2)), ("convT_dgrad", 32, 16, (2, 32, 4)), ("convT_dgrad", 48, 32, (2, 32, 2))])
def test_gather_plans_compute_the_strided_launches_and_mirror_the_kernels_lds(kind, cin, cout, coarse):
    """planner.gather_plans (csrc/gconv.hip, depth -9): marching plans of the stride-(2,2,1) 3x3x1 launches that read the fine level — their packed weights simulate to torch's
    strided convolution / transposed-convolution data gradient (numpy restatement of the kernel's indexing), their LDS request is the kernel's."""
    import ctypes

    from tests import gpu_harness as H
    from vs_seg_amd import _lib as L

    k, s = (3, 3, 1), (2, 2, 1)
    fine = (2 * coarse[0], 2 * coarse[1], coarse[2])
    torch.manual_seed(8)
    if kind == "conv_fwd":
        x = torch.randn(1, cin, *fine, dtype=torch.float64)
        w = torch.randn(cout, cin, *k, dtype=torch.float64)
        want, xin = F.conv3d(x, w, stride=s, padding=P.same_pad(k)), _cl(x)
    else:
        w = torch.randn(cin, cout, *k, dtype=torch.float64)
        xd = torch.zeros(1, cin, *coarse, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose3d(xd, w, stride=s, padding=P.same_pad(k), output_padding=(1, 1, 0))
        gy = torch.randn_like(y)
        y.backward(gy)
        want, xin = xd.grad, _cl(gy)
    kreal, nreal = P.gemm_dims(kind, tuple(w.shape))
    cls = P.lattice_classes(kind, k, s)
    assert len(cls) == 1 and tuple(cls[0].is_) == (2, 2, 1)
    plans = P.gather_plans(kind, tuple(w.shape), cls[0], coarse, 2, kreal, nreal, kreal)
    assert plans and all(pl.depth == -9 and pl.nchunks == 1 and pl.ck == kreal and pl.tile[1] == 64 * pl.mtw // pl.tile[2] for pl in plans)
    for pl in plans:
        assert pl.lds == P.gather_lds_bytes(kreal, pl.nt, pl.tile[2], pl.mtw) <= P.LDS_LIMIT and (kreal, pl.nt, pl.tile[2], pl.mtw) in P.GATHER_SHAPES
        pl.pack_map = P.pack_map(pl, tuple(w.shape))
        np.testing.assert_allclose(P.simulate_igemm(pl, xin, w.numpy().reshape(-1), coarse)[..., :nreal], _cl(want), atol=1e-9)
        d = H.igemm_desc(pl, torch.zeros(1), L.Tensor(4096, L.BF16, kreal, kreal, 1, *fine), L.Tensor(8192, L.BF16, nreal, nreal, 1, *coarse))
        assert L.lib().vsseg_igemm_lds_bytes(ctypes.byref(d)) == pl.lds, L.lib().vsseg_last_error()
    assert any(pl.depth == -9 for pl in P.candidate_plans(kind, tuple(w.shape), cls[0], coarse, 2, kc_pad=kreal, aux_es=0))
    assert P.gather_plans(kind, tuple(w.shape), cls[0], coarse, 4, kreal, nreal, kreal) == []  # fp32


def test_small_lds_budget_forces_channel_chunks():
    torch.manual_seed(2)
    k, s = (3, 3, 3), (1, 1, 1)
    x = torch.randn(1, 96, 8, 8, 8, dtype=torch.float64)
    w = torch.randn(48, 96, *k, dtype=torch.float64)
    y = F.conv3d(x, w, padding=1)
    _run.stride = s
    cls = P.lattice_classes("conv_fwd", k, s)[0]
    plan = P.plan_igemm("conv_fwd", tuple(w.shape), cls, (8, 8, 8), es=2, lds_budget=40 * 1024)
    assert plan.nchunks > 1 and plan.ck * plan.nchunks == 96
    got = P.simulate_igemm(plan, _cl(x), w.numpy().reshape(-1), (8, 8, 8))
    np.testing.assert_allclose(got, _cl(y), atol=1e-9)


def test_every_network_layer_has_a_feasible_plan():
    """All 50 convolutions of the network at the benchmark patch, both dtypes, fwd + dgrad + wgrad."""
    from vs_seg_amd.graph import conv_layers

    for es in (2, 4):
        for L in conv_layers(attention=True):
            dims = L.in_dims((384, 128, 128))
            kind = "convT_fwd" if L.transposed else "conv_fwd"
            od = P.out_dims(kind, dims, L.kernel, L.stride)
            wshape = L.wshape
            for knd, q_of in ((kind, od if not L.transposed else dims), ("convT_dgrad" if L.transposed else "conv_dgrad", None)):
                for cls in P.lattice_classes(knd, L.kernel, L.stride):
                    if knd in ("conv_fwd", "convT_dgrad"):
                        q = od if knd == "conv_fwd" else dims
                    else:
                        q = dims if knd == "convT_fwd" else tuple((d + s - 1) // s for d, s in zip(dims, L.stride))
                    plan = P.plan_igemm(knd, wshape, cls, q, es)
                    assert plan.lds <= P.LDS_LIMIT and 1 <= plan.nt <= 6 and plan.mtw in (1, 2, 4)
            wp = P.plan_wgrad(L.transposed, wshape, L.kernel, L.stride, dims if L.transposed else od, es)
            assert wp.lds <= P.LDS_LIMIT and wp.ntp <= 6 and (wp.tile[0] * wp.tile[1] * wp.tile[2]) % 32 == 0


def test_merged_residual_pack_map():
    """conv_3x3x1(x; W) + conv_1x1x1(x; Wr) == one convolution whose packed weights are pack_map(W) + pack_map_centre(Wr) (fwd and dgrad)."""
    torch.manual_seed(5)
    k, s, cin, cout, dims = (3, 3, 1), (1, 1, 1), 32, 2, (8, 8, 4)
    x = torch.randn(2, cin, *dims, dtype=torch.float64, requires_grad=True)
    w, wr = torch.randn(cout, cin, *k, dtype=torch.float64), torch.randn(cout, cin, 1, 1, 1, dtype=torch.float64)
    y = F.conv3d(x, w, padding=P.same_pad(k)) + F.conv3d(x, wr)
    flat = torch.cat([w.reshape(-1), wr.reshape(-1)]).numpy()
    off_r = w.numel()

    def merged(kind, inp_cl, out_shape, kc):
        cls = P.lattice_classes(kind, k, s)[0]
        plan = P.plan_igemm(kind, tuple(w.shape), cls, out_shape, es=2, kc_pad=kc)
        m1, m2 = plan.pack_map, P.pack_map_centre(plan, tuple(wr.shape))
        wpack = np.where(m1 >= 0, flat[np.clip(m1, 0, None)], 0.0) + np.where(m2 >= 0, flat[np.clip(m2 + off_r, 0, None)], 0.0)
        ident = dataclasses.replace(plan)
        ident.pack_map = np.arange(wpack.size, dtype=np.int32)
        return P.simulate_igemm(ident, inp_cl, wpack, out_shape)

    np.testing.assert_allclose(merged("conv_fwd", _cl(x.detach()), dims, cin), _cl(y.detach()), atol=1e-9)
    gy = torch.randn_like(y)
    y.backward(gy)
    gy_cl = np.zeros((2, *dims, 8))
    gy_cl[..., :cout] = _cl(gy)
    np.testing.assert_allclose(merged("conv_dgrad", gy_cl, dims, 8), _cl(x.grad), atol=1e-9)


@pytest.mark.parametrize("cin,cout,kind", [(1, 16, "conv_fwd"), (16, 1, "conv_fwd"), (16, 1, "conv_dgrad"), (1, 16, "conv_fwd_1x1")])
def test_z_folded_plans_compute_the_same_convolution(cin, cout, kind):
    """planner.folded_candidate_plans: the 3x3x1 (or 1x1x1) convolution with one real input / output channel as a block-diagonal
    convolution on tensors whose 8 z-neighbours are reinterpreted as channels — every candidate, fwd and dgrad."""
    torch.manual_seed(9)
    k = (1, 1, 1) if kind.endswith("1x1") else (3, 3, 1)
    kind = "conv_fwd" if kind.startswith("conv_fwd") else kind
    dims = (8, 8, 16)
    assert P.foldable(k, (1, 1, 1), False, cin, cout, dims) and not P.foldable((3, 3, 3), (1, 1, 1), False, cin, cout, dims) and not P.foldable(k, (1, 1, 1), False, 16, 16, dims)
    x = torch.randn(2, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, *k, dtype=torch.float64)
    y = F.conv3d(x, w, padding=P.same_pad(k))
    cls = P.lattice_classes(kind, k, (1, 1, 1))[0]
    if kind == "conv_fwd":
        src, want, cs, cd = x.detach(), y.detach(), cin, cout
    else:
        gy = torch.randn_like(y)
        y.backward(gy)
        src, want, cs, cd = gy, x.grad, cout, cin
    # fold: [N,X,Y,Z,C] -> [N,X,Y,Z/8,8*C] is a pure reinterpretation of the channels-last memory
    src_f = _cl(src).reshape(2, dims[0], dims[1], dims[2] // 8, 8 * cs)
    cands = P.folded_candidate_plans(kind, tuple(w.shape), cls, dims, es=2)
    assert len(cands) >= (2 if cands[0].nt <= 2 else 1)  # nt >= 3 (the narrow-input fold: 128 folded output channels) always prefetches: no single-buffer twins
    for pl in cands:
        assert pl.kc == 8 * cs and pl.nc == 8 * cd and pl.q == (8, 8, 2)
        ident = dataclasses.replace(pl)  # simulate_igemm gathers the weights through pack_map itself
        got = P.simulate_igemm(ident, src_f, w.numpy().reshape(-1), (8, 8, 2))
        np.testing.assert_allclose(got.reshape(2, *dims, cd), _cl(want), atol=1e-9, err_msg=f"tile={pl.tile} ck={pl.ck} ns={pl.nsplit}")


def test_marching_kernel_lds_layout_is_bank_conflict_free():
    """csrc/mconv.hip's plane layout [row][piece'][z]: every 16-lane service group of every operand ds_read_b128 hits 16 distinct 16-byte slots."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("lds_conflicts", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lds_conflicts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for cin, tz in sorted({(c, z) for (c, _, z, _) in P.MARCH_SHAPES if c >= 16}):  # one 8-channel group (1-channel inputs): taps share a K-step, 2-way conflicts accepted
        extra, base = mod.extra_cycles(cin, tz)
        assert extra == 0, (cin, tz, extra, base)


def test_march_plans_cover_the_benchmark_layers_and_mirror_the_kernel_lds():
    cls = P.lattice_classes("conv_fwd", (3, 3, 1), (1, 1, 1))[0]
    for cin, cout, dims in [(16, 16, (384, 128, 128)), (32, 16, (384, 128, 128)), (16, 32, (192, 64, 128)), (32, 32, (192, 64, 128)), (64, 32, (192, 64, 128))]:
        pls = P.march_plans("conv_fwd", (cout, cin, 3, 3, 1), cls, dims, 2, cin, cout, cin, n=4)
        assert pls, (cin, cout)
        for pl in pls:
            lx, tyb, tz = pl.tile
            assert pl.depth in (-5, -6) and tyb == 64 * pl.mtw // tz and dims[1] % tyb == 0 and dims[2] % tz == 0 and 1 <= lx <= dims[0]
            g = cin // 8
            wbytes = 0 if pl.depth == -6 else ((9 * g + 3) // 4) * pl.nt * 1024  # depth -6: the packed weights live in registers
            assert pl.lds == wbytes + 4 * (tyb + 2) * tz * g * 16 + 5 * pl.nt * 16 * 4 + 16 <= 160 * 1024
            assert pl.depth == -5 or (cin, pl.nt, tz, pl.mtw) in P.MARCH_WREG_SHAPES
            assert pl.ksteps == (9 * g + 3) // 4 and pl.nchunks == 1 and pl.ck == cin
    # outside the domain: strided, 3x3x3, fp32
    assert P.march_plans("conv_fwd", (16, 16, 3, 3, 1), P.lattice_classes("conv_fwd", (3, 3, 1), (2, 2, 1))[0], (192, 64, 128), 2, 16, 16, 16) == []
    assert P.march_plans("conv_fwd", (16, 16, 3, 3, 1), cls, (384, 128, 128), 4, 16, 16, 16) == []


def test_chained_launch_plans_cover_the_benchmark_window_and_nothing_else():
    """planner.chain_plan / chain_pack_plan (csrc/chain.hip, inference): the three pairs of the 384x128x128 window get a plan whose workgroup owns ALL rows
    (y == waves * mtw * 16 / tz), about one workgroup per CU, and packed weights in the marching layout [K-steps][tiles][64][8]; every other shape gets None."""
    for cin, compact, cmid, dims in [(8, True, 16, (384, 128, 128)), (32, False, 16, (384, 128, 128)), (16, False, 32, (192, 64, 128))]:
        for n in (1, 2, 4):
            pc = P.chain_plan(cin, compact, dims, n, cmid)
            assert pc is not None, (cin, cmid, n)
            assert dims[1] == pc["waves"] * pc["mtw"] * 16 // pc["tz"] and dims[2] % pc["tz"] == 0 and 1 <= pc["lx"] <= dims[0]
            assert pc["waves"] in (8, 16) and pc["lead"] in (1, 3) and (pc["lead"] == 3 or not compact)
            wgs = n * -(-dims[0] // pc["lx"]) * (dims[2] // pc["tz"])
            assert 128 <= wgs <= 512, (cin, n, wgs)
    assert P.chain_plan(32, False, (384, 384, 64), 1) is None      # 384 rows do not fit one workgroup's rings
    assert P.chain_plan(8, True, (192, 64, 128), 1) is None        # the compact pair is instantiated for 128 rows
    assert P.chain_plan(16, False, (384, 128, 128), 1, 32) is None  # the level-1 unit for 64 rows
    assert P.chain_plan(32, False, (16, 128, 128), 1) is None      # too short to march
    assert P.chain_plan(64, False, (192, 64, 128), 1, 32) is None  # the level-1 attention block is not instantiated
    for wshape, kc in [((16, 1, 3, 3, 1), 8), ((16, 16, 3, 3, 1), 16), ((1, 16, 3, 3, 1), 16), ((32, 16, 3, 3, 1), 16), ((32, 32, 3, 3, 1), 32)]:
        dims = (384, 128, 128) if wshape[0] <= 16 else (192, 64, 128)
        pl = P.chain_pack_plan(wshape, dims, 2, kc, 1)
        assert pl is not None and pl.depth == -5 and pl.nt == (wshape[0] + 15) // 16
        assert pl.pack_map.size == ((9 * (kc // 8) + 3) // 4) * pl.nt * 64 * 8
        real = pl.pack_map[pl.pack_map >= 0]
        assert real.size == int(np.prod(wshape)) and len(set(real.tolist())) == real.size  # every weight element exactly once
    rm = P.residual_tile_pack_map(16, 2, (32, 16, 1, 1, 1))
    assert rm.size == 1 * 2 * 64 * 8 and (rm >= 0).sum() == 32 * 16


@pytest.mark.parametrize("kind,k,st,cin,cout,fine", [("convT_fwd", (3, 3, 3), (2, 2, 2), 96, 80, (8, 8, 8)), ("conv_dgrad", (3, 3, 3), (2, 2, 2), 80, 80, (16, 8, 16)),
                                                     ("convT_fwd", (3, 3, 1), (2, 2, 1), 48, 32, (8, 8, 4)), ("conv_dgrad", (3, 3, 3), (2, 2, 2), 64, 64, (8, 4, 8))])
def test_deep_class_plans_mirror_the_kernels_lds_request(kind, k, st, cin, cout, fine):
    """planner.deep_lds_bytes mirrors dc_check() of csrc/dconv.hip also for the all-classes launches, whose K-group table has one section per parity class
    (the planner sized it for one class; the untuned VSSEG_DEEP=force lowering could then be refused by the kernel)."""
    import ctypes

    from tests import gpu_harness as H
    from vs_seg_amd import _lib as L

    lib = L.lib()
    coarse = tuple(f // s for f, s in zip(fine, st))
    wshape = (cin, cout, *k) if kind == "convT_fwd" else (cout, cin, *k)
    kc, nreal = (cin, cout) if kind == "convT_fwd" else (cout, cin)
    plans = P.deep_class_plans(kind, wshape, k, st, coarse, 2, kc, nreal, nreal, 1)
    assert plans, "no all-classes plan for a level 3-5 transition"
    for pl in plans:
        inp = L.Tensor(4096, L.BF16, kc, kc, 1, *coarse)
        out = L.Tensor(8192, L.BF16, pl.nt * 16, pl.nt * 16, 1, *fine)
        d = H.igemm_desc(pl, torch.zeros(1), inp, out)
        assert lib.vsseg_igemm_lds_bytes(ctypes.byref(d)) == pl.lds, (pl.tile, pl.mtw, pl.nt, len(pl.classes), lib.vsseg_last_error())
