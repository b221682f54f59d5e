"""Helpers for the -m gpu parity tests: drive single kernels through the C ABI (ctypes) on torch-allocated device memory."""
import ctypes as C

import numpy as np
import torch

from vs_seg_amd import _lib as L
from vs_seg_amd import planner as P

DT = {"fp32": torch.float32, "bf16": torch.bfloat16}


FX_STAT = float(2 ** 20)  # csrc/common.h VSSEG_FX_STAT: the sharded statistics hold 64-bit fixed-point integers (order-independent atomics)


def stat_decode(t: torch.Tensor) -> torch.Tensor:
    """A statistics buffer of the C ABI (`double*`, filled by vsseg_fx_add) as real numbers."""
    return t.view(torch.int64).double() / FX_STAT


def stat_encode(t: torch.Tensor) -> torch.Tensor:
    """Real numbers -> the fixed-point bit pattern the finalisation kernels read (for tests that fill a statistics buffer by hand)."""
    return torch.round(t.double() * FX_STAT).to(torch.int64).view(torch.float64)


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdesc(t: torch.Tensor, c=None, c0=0) -> L.Tensor:
    """t is channels-last [N,X,Y,Z,C]."""
    n, x, y, z, pitch = t.shape
    return L.Tensor(t.data_ptr() + c0 * t.element_size(), L.F32 if t.dtype == torch.float32 else L.BF16, c or (pitch - c0), pitch, n, x, y, z)


def to_cl(x: torch.Tensor, dtype, cpad=None) -> torch.Tensor:
    """NCDHW fp32/64 cpu -> channels-last device tensor of dtype, channels zero-padded to cpad."""
    v = x.detach().permute(0, 2, 3, 4, 1).to(torch.float32)
    if cpad and cpad > v.shape[-1]:
        v = torch.cat([v, torch.zeros(*v.shape[:-1], cpad - v.shape[-1])], -1)
    return v.contiguous().to("cuda").to(dtype)


def from_cl(t: torch.Tensor, c=None) -> torch.Tensor:
    v = t.detach().to(torch.float32).cpu()
    if c:
        v = v[..., :c]
    return v.permute(0, 4, 1, 2, 3).contiguous()


def pack(plan: P.IgemmPlan, w: torch.Tensor, dtype) -> torch.Tensor:
    lib = L.lib()
    wf = w.detach().to(torch.float32).reshape(-1).contiguous().cuda()
    m = torch.from_numpy(plan.pack_map).cuda()
    out = torch.zeros(m.numel(), dtype=dtype, device="cuda")
    L.check(lib.vsseg_gather_cast(wf.data_ptr(), m.data_ptr(), None, out.data_ptr(), m.numel(), L.F32 if dtype == torch.float32 else L.BF16, stream()), "gather_cast")
    return out


def igemm_desc(plan: P.IgemmPlan, wpack: torch.Tensor, inp: L.Tensor, out: L.Tensor, **kw) -> L.IgemmDesc:
    d = L.IgemmDesc()
    d.inp, d.out = inp, out
    d.q, d.is_, d.os, d.oo = L.i3(plan.q), L.i3(plan.cls.is_), L.i3(plan.cls.os), L.i3(plan.cls.oo)
    d.ntaps = plan.ntaps
    for t, (off, _) in enumerate(plan.cls.taps):
        d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
    d.tile = L.i3(plan.tile)
    d.mtw, d.nt, d.nsplit, d.ck, d.nchunks, d.ksteps, d.depth = plan.mtw, plan.nt, plan.nsplit, plan.ck, plan.nchunks, plan.ksteps, plan.depth
    d.wpack = wpack.data_ptr()
    d.class_split = len(plan.classes) if plan.classes is not None else 0
    for s_, c_ in enumerate(plan.classes or ()):
        d.class_oo[s_][0], d.class_oo[s_][1], d.class_oo[s_][2] = c_.oo
        d.class_ntaps[s_] = len(c_.taps)
        for i, t in enumerate(plan.class_taps(s_)):
            d.class_tap[s_][i] = t
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _split_cl(t_cl: torch.Tensor, c0: int):
    return t_cl[..., :c0].contiguous(), t_cl[..., c0:].contiguous()


def two_part(a: torch.Tensor, b: torch.Tensor) -> L.Tensor:
    """Descriptor of the channel concatenation [a | b] of two dense channels-last tensors (nothing is copied)."""
    return L.Tensor.two_part(tdesc(a), tdesc(b))


def run_lattice_op(kind, w, x_cl, out_cl, stride, **epi):
    """Run every lattice class of a conv-like op; x_cl/out_cl are channels-last device tensors, or (a, b) pairs of them
    for a two-part tensor."""
    lib = L.lib()
    kernel = tuple(w.shape[2:])
    xin = two_part(*x_cl) if isinstance(x_cl, tuple) else tdesc(x_cl)
    xout = two_part(*out_cl) if isinstance(out_cl, tuple) else tdesc(out_cl)
    x0 = x_cl[0] if isinstance(x_cl, tuple) else x_cl
    es = x0.element_size()
    odims = (xout.x, xout.y, xout.z)
    keep = []
    mtw = epi.pop("mtw", None)
    for cls in P.lattice_classes(kind, kernel, stride):
        q = odims if kind in ("conv_fwd", "convT_dgrad") else tuple((o + s - 1) // s for o, s in zip(odims, stride))
        plan = P.plan_igemm(kind, tuple(w.shape), cls, q, es, kc_pad=xin.c, in_split=xin.csplit if xin.ptr2 else 0, mtw=mtw, aux_es=((4 if (epi.get("res") is not None and epi["res"].dtype == L.F32) else es) if (epi.get("accumulate") or epi.get("res_mode")) else 0) if mtw else 4, **({"lds_budget": epi.pop("lds_budget")} if "lds_budget" in epi else {}))
        wp = pack(plan, w, x0.dtype)
        d = igemm_desc(plan, wp, xin, xout, **epi)
        keep.append((wp, d))
        L.check(lib.vsseg_igemm(C.byref(d), stream()), "igemm")
    torch.cuda.synchronize()
    return keep


def run_wgrad(transposed, wshape, kernel, stride, p_cl, h_cl, cp_valid, ch_valid, hgroup=0, single_buffer=0, march_tile=None, dbias=None, h_gate=None, compute=0, blocks=None):
    lib = L.lib()
    es = p_cl.element_size()
    wp = P.plan_wgrad(transposed, wshape, kernel, stride, tuple(p_cl.shape[1:4]), es)
    dw = torch.zeros(int(np.prod(wshape)), dtype=torch.float32, device="cuda")
    d = L.WgradDesc()
    d.p, d.h, d.cp_valid, d.ch_valid = tdesc(p_cl), (two_part(*h_cl) if isinstance(h_cl, tuple) else tdesc(h_cl)), cp_valid, ch_valid
    d.q, d.hs, d.ntaps = L.i3(wp.q), L.i3(wp.hs), len(wp.taps)
    for t, (off, widx) in enumerate(wp.taps):
        d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
        d.tap_widx[t] = widx
    d.tile, d.ntp = L.i3(wp.tile), wp.ntp
    d.dw = dw.data_ptr()
    d.stride_p, d.stride_h, d.stride_tap = wp.stride_p, wp.stride_h, wp.stride_tap
    d.persistent_blocks = wp.blocks
    d.hgroup, d.single_buffer = hgroup, single_buffer
    if march_tile is not None:  # the marching kernel (csrc/mwgrad.hip): tile = (x steps per workgroup, rows, z slices)
        d.march, d.tile = 1, L.i3(march_tile)
    if compute:  # the compute kernel (csrc/cwgrad.hip): hgroup = 16-channel H chunks per workgroup (1 or 2)
        d.march, d.hgroup = 2, compute
    if blocks is not None:
        d.persistent_blocks = blocks
    if dbias is not None:
        d.dbias_p = dbias.data_ptr()
    if h_gate is not None:
        d.h_gate = h_gate.data_ptr()
    scr = torch.zeros((48 if compute else 8) * 1024 * 1024, dtype=torch.float32, device="cuda")
    d.scratch, d.scratch_elems = scr.data_ptr(), scr.numel()
    L.check(lib.vsseg_wgrad(C.byref(d), stream()), "wgrad")
    torch.cuda.synchronize()
    return dw.cpu().reshape(wshape)
