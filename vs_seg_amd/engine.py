"""Lowers the static network program (vs_seg_amd.graph) to HIP launches through the C ABI and runs them.

One `Plan` per (batch, spatial size, train/eval) signature holds every activation/gradient buffer, every prepared
ctypes descriptor and two launch lists (forward, backward).  Running a step is a flat loop over prepared launches on
torch's current HIP stream — no allocation, no host synchronisation, capturable in a hipGraph.

torch tensors are used only as device-memory containers.  There is no CPU/eager fallback: a missing extension raises.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import planner as P
from .graph import HP, AttGate, ConvBnAct, ConvPlain, Layer, Program, TensorSpec, build_program, level_dims

BN_EPS = 1e-5  # torch.nn.BatchNorm3d defaults (ref:params/networks/blocks/convolutions.py:152 passes no arguments)
BN_MOMENTUM = 0.1
ACT_CODE = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID}


def _tdtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


TUNED_DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")


def _tune_cache() -> dict:
    """launch signature -> [tile, nt, nsplit, ck] of the plan measured fastest.  Seeded from the in-tree file of plans tuned on an
    MI355X for the benchmark shapes and from $VSSEG_TUNE_CACHE; new measurements are written back to $VSSEG_TUNE_CACHE if set."""
    if getattr(_tune_cache, "data", None) is None:
        import json

        _tune_cache.data, _tune_cache.dirty = {}, False
        for f in (TUNED_DEFAULTS, os.environ.get("VSSEG_TUNE_CACHE")):
            if f and os.path.exists(f):
                try:
                    _tune_cache.data.update(json.load(open(f)))
                except (OSError, ValueError):
                    pass
    return _tune_cache.data


def _tune_cache_save():
    f = os.environ.get("VSSEG_TUNE_CACHE")
    if f and getattr(_tune_cache, "dirty", False):
        import json

        tmp = f"{f}.{os.getpid()}.tmp"
        json.dump(_tune_cache.data, open(tmp, "w"), indent=0, sort_keys=True)
        os.replace(tmp, f)
        _tune_cache.dirty = False


class ParamLayout:
    """Offsets of every state_dict entry inside the flat buffers owned by the model (params fp32, buffers fp32, counters int64)."""

    def __init__(self, manifest):
        self.param_off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.buf_off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.cnt_off: Dict[str, int] = {}
        self.order = [k for k, _ in manifest]
        po = bo = co = 0
        for key, shape in manifest:
            n = int(np.prod(shape)) if len(shape) else 1
            if key.endswith("num_batches_tracked"):
                self.cnt_off[key] = co
                co += 1
            elif "running_" in key:
                self.buf_off[key] = (bo, tuple(shape))
                bo += n
            else:
                self.param_off[key] = (po, tuple(shape))
                po += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the flat buffer
        self.n_param, self.n_buf, self.n_cnt = po, bo, co


@dataclass
class _Choice:
    """One implicit-GEMM launch of a layer (one lattice class): the candidate plans and, once lowered, the chosen one."""

    cands: List[P.IgemmPlan]  # default (heuristic) plan first
    woff: int  # offset of the layer's weight inside the flat parameter buffer
    wshape2: Optional[tuple] = None  # merged 1x1x1 residual convolution: its weight shape ...
    woff2: int = 0  # ... and flat offset
    chosen: Optional[P.IgemmPlan] = None
    map_off: int = -1  # element offset of the chosen plan's packed weights inside Plan.wpack
    tuned_ms: Optional[list] = None  # autotuner measurements, one per candidate
    wshape: tuple = ()
    cached: bool = False  # the choice came from the tuned-plan cache, nothing was measured
    fold: int = 0  # z-folded formulation (planner.FOLD z-neighbours as channels): the launch runs on reinterpreted tensors
    cmod: int = 0  # ... and output channel c is real channel c % cmod for the per-channel vectors
    alt: Optional["_Choice"] = None  # on the first lattice class of a multi-class op: all classes as ONE launch (planner.class_split_plans)
    probe_plan: Optional[P.IgemmPlan] = None  # plan of the last probe lowering (Plan._igemm(probe=True))
    woff_res: int = 0  # plans with residual tiles (planner.march_res_plans): flat offset of the 1x1x1 residual convolution's weight ...
    map_off_res: int = -1  # ... and the element offset of its packed weights inside Plan.wpack


@dataclass
class _ConvPlans:
    fwd: List[_Choice]
    dgrad: List[_Choice]
    wgrad: Optional[P.WgradPlan]
    fold_fwd: bool = False


class _Slot:
    """A launch argument that is filled in at run time (dropout seed, external gradient pointers)."""

    def __init__(self, kind, name=None):
        self.kind, self.name = kind, name




class Plan:
    def __init__(self, eng: "Engine", n: int, dims: Tuple[int, int, int], train: bool):
        self.eng, self.n, self.dims, self.train = eng, n, tuple(dims), train
        self.lv = level_dims(dims, eng.hp)
        self.compact_input: Optional[torch.Tensor] = None  # [N,X,Y,Z,1] copy of the network input for the z-folded first-layer launches
        self.needs_padded_input = False  # set by the first launch descriptor that reads the 8-channel zero-extended copy of the network input (_desc): staged per step only then
        self.fwd: List[list] = []
        self.fwd_pre: List[list] = []  # eval only: launches that depend on parameters / buffers alone (BatchNorm folding); re-run with the weight packing when those change
        self.params_key = None  # parameter version the packed weights + folded BatchNorm constants of an eval plan belong to
        self.bwd: List[list] = []
        self.keep: list = []  # ctypes objects that must outlive the launches
        self.bufs: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.flat_ptr = eng.flat.data_ptr()
        self.timer = None  # set to {'only': set|None, 'events': []} to time launches with HIP events
        self.generation = 0  # bumped by every training forward: a backward belongs to exactly one forward of this plan
        # The launch lists have FIXED kernel arguments (the dropout seed is read from `seed_dev`, external gradients are staged into
        # plan-owned buffers), so after two eager warm-up runs each list is captured in a hipGraph and replayed: one host call per list
        # instead of ~100-230 ctypes launches (VSSEG_GRAPHS=0 keeps the eager loop; the event-timed profiling mode always runs eagerly).
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=eng.device)
        self.gatt_buf: Dict[str, torch.Tensor] = {}
        self._gatt_set: Dict[str, bool] = {}
        self._fork_events: list = []  # events of the side-stream forks (created once, re-recorded every step)
        self.bwd_pre: List[list] = []  # backward launches that read caller-owned memory (the loss' gradient of the logits): never captured
        self._graphs: Dict[str, object] = {}
        self._graph_runs: Dict[str, int] = {}
        self._plan_layers()
        self._lower()
        self._index_slots()

    # ------------------------------------------------------------------ helpers
    def _alloc(self, spec: TensorSpec, store: Dict[str, torch.Tensor]) -> torch.Tensor:
        root = spec.root
        if root.name not in store:
            dt = torch.float32 if root.kind == "f32" else self.eng.tdtype
            store[root.name] = torch.zeros((self.n, *self.lv[root.level], root.c), dtype=dt, device=self.eng.device)
        return store[root.name]

    def _desc(self, spec: TensorSpec, store=None) -> L.Tensor:
        if spec.parts is not None:  # skip-connection concat: the pair of its dense operands
            return L.Tensor.two_part(self._desc(spec.parts[0], store), self._desc(spec.parts[1], store))
        buf = self._alloc(spec, self.bufs if store is None else store)
        if store is None and spec.root.name == self.eng.prog.input.name:
            self.needs_padded_input = True  # (every reader of the padded copy gets its descriptor here; the compact copy goes through _xdesc)
        x, y, z = self.lv[spec.level]
        return L.Tensor(buf.data_ptr() + spec.c0 * buf.element_size(), _tdtype(buf), spec.c, spec.root.c, self.n, x, y, z)

    def _xdesc(self, spec: TensorSpec, folded: bool) -> L.Tensor:
        """Input descriptor of a convolution; the z-folded launches of the 1-channel network input read its compact copy."""
        if folded and spec.root.name == self.eng.prog.input.name:
            self.compact_input = self._raw("input:c1", 0, 1)
            return self._tdesc(self.compact_input, 0)
        return self._desc(spec)

    def _raw(self, name: str, level: int, c: int, dtype=None) -> torch.Tensor:
        if name not in self.bufs:
            self.bufs[name] = torch.zeros((self.n, *self.lv[level], c), dtype=dtype or self.eng.tdtype, device=self.eng.device)
        return self.bufs[name]

    def _tdesc(self, buf: torch.Tensor, level: int, c: Optional[int] = None) -> L.Tensor:
        x, y, z = self.lv[level]
        return L.Tensor(buf.data_ptr(), _tdtype(buf), c or buf.shape[-1], buf.shape[-1], self.n, x, y, z)

    def _dpre_desc(self, buf: torch.Tensor, level: int) -> L.Tensor:
        """d(pre-sigmoid) as vsseg_att_apply_bwd writes it: the 8-channel group in front of a row of 8 or 16 channels."""
        t = self._tdesc(buf, level)
        return L.Tensor(t.ptr, t.dtype, 8, t.pitch, t.n, t.x, t.y, t.z)

    def _pp(self, key: str) -> int:  # device address of a parameter inside the flat fp32 buffer
        return self.flat_ptr + 4 * self.eng.layout.param_off[key][0]

    def _gp(self, key: str) -> int:  # ... of its gradient
        return self.eng.gflat.data_ptr() + 4 * self.eng.layout.param_off[key][0]

    def _bp(self, key: str) -> int:
        return self.eng.bflat.data_ptr() + 4 * self.eng.layout.buf_off[key][0]

    def _cp(self, key: str) -> int:
        return self.eng.cflat.data_ptr() + 8 * self.eng.layout.cnt_off[key]

    def _vox(self, level: int) -> int:
        x, y, z = self.lv[level]
        return self.n * x * y * z

    # ------------------------------------------------------------------ conv planning + weight pack buffer
    def _plan_layers(self):
        eng = self.eng
        self._maps: List[np.ndarray] = []
        self._maps2: List[np.ndarray] = []
        self._map_len = 0
        self._wpack_fixups: list = []  # (descriptor, element offset): wpack is allocated after every launch chose its plan
        self._res_fixups: list = []  # ... the same for the `wpack_res` field of the fused backward launches
        self._chain_fixups: list = []  # ... (descriptor, field, element offset) of the chained launches' two packed-weight pointers
        # Autotune (default on a GPU): every launch measures its candidate plans on the real buffers at lowering time and keeps
        # the fastest.  VSSEG_AUTOTUNE=0 keeps the heuristic plan (deterministic; what the CPU dry-run lowering always uses).
        self.tune = (not eng.dry_run) and os.environ.get("VSSEG_AUTOTUNE", "1") != "0"

        # a 1x1x1 residual conv added to a plain (no BatchNorm) conv of the same input merges into that conv's centre tap
        plain = {op.out.name: op for op in eng.prog.ops if isinstance(op, ConvPlain)}
        self.merged: Dict[str, ConvPlain] = {}  # residual-conv prefix -> the convolution that absorbed it
        self.absorbs: Dict[str, ConvPlain] = {}  # absorbing conv prefix -> the residual conv op
        for op in eng.prog.ops:
            if isinstance(op, ConvPlain) and op.res is not None and op.act == "none" and op.res.name in plain:
                a = plain[op.res.name]
                if a.x is op.x and a.act == "none" and a.layer.kernel == (1, 1, 1) and all(k % 2 == 1 for k in op.layer.kernel):
                    self.merged[a.layer.prefix] = op
                    self.absorbs[op.layer.prefix] = a

        def choices(kind, Lr, q, kc_pad, aux_es, in_split, absorbed, fold=False):
            out = []
            woff = eng.layout.param_off[Lr.wkey][0]
            if eng.fuse_classes and not fold and absorbed is None and not in_split:
                # the four output-parity classes of a stride-(2,2,1) transposed convolution / data gradient as ONE streaming-kernel launch
                # (planner.shuffle_plan): the per-class launches each read the whole input
                kreal, nreal = P.gemm_dims(kind, Lr.wshape)
                sps = P.shuffle_plans(kind, Lr.wshape, Lr.kernel, Lr.stride, q, eng.es, kc_pad, nreal, kreal)
                if sps is not None:  # (+ the marching variants where they exist: measured against the streaming launch by the tuner; the untuned lowering keeps the streaming one)
                    got = [_Choice([sp] + (P.march_shuffle_plans(sp, self.n) if (self.tune and eng.march_shuffle) else []), woff, wshape=tuple(Lr.wshape)) for sp in sps]
                    allc = P.march_shuffle_all_plans(kind, Lr.wshape, Lr.kernel, Lr.stride, q, eng.es, kc_pad, nreal, kreal, self.n) if (self.tune and eng.march_shuffle and len(sps) == 2) else []
                    if allc:  # 32 output channels: ONE marching launch for all four classes against the two streaming launches (decided by measurement, _use_class_split)
                        got[0].alt = _Choice(allc, woff, wshape=tuple(Lr.wshape))
                    return got
            alt = None
            if eng.class_split and not fold and absorbed is None:
                # ... and those of the 3x3x3 stride-(2,2,2) transitions of the deep levels as ONE launch of the general kernel (workgroup row = class)
                kreal, nreal = P.gemm_dims(kind, Lr.wshape)
                csp = P.class_split_plans(kind, Lr.wshape, Lr.kernel, Lr.stride, q, eng.es, kc_pad, nreal, kreal, aux_es=aux_es, in_split=in_split) if kind in ("convT_fwd", "conv_dgrad") else None
                # ... or of the deep-level kernel (csrc/dconv.hip: a workgroup loads the halo of its coarse tile once and runs the classes one after the other)
                dcp = P.deep_class_plans(kind, Lr.wshape, Lr.kernel, Lr.stride, q, eng.es, kc_pad, nreal, kreal, self.n, in_split) if (eng.deep != "0" and kind in ("convT_fwd", "conv_dgrad")) else []
                # ... or of the transition kernel (csrc/tconv.hip: levels 2 <-> 3, the waves split the voxels, a stage per class)
                tcp = P.transition_plans(kind, Lr.wshape, Lr.kernel, Lr.stride, q, eng.es, kc_pad, nreal, kreal, in_split) if (eng.transition and self.tune and kc_pad is not None and kind in ("convT_fwd", "conv_dgrad")) else []
                if eng.deep == "force" and dcp:
                    alt = _Choice(dcp if self.tune else dcp[:1], woff, wshape=tuple(Lr.wshape))
                elif csp or ((dcp or tcp) and self.tune):  # an alternative to the per-class launches below, decided per op at lowering time (Plan._use_class_split)
                    alt = _Choice(((csp or []) + dcp + tcp) if self.tune else csp[:1], woff, wshape=tuple(Lr.wshape))
            for cls in P.lattice_classes(kind, Lr.kernel, Lr.stride):
                if fold:  # one real input or output channel, no taps along z: 8 z-neighbours become the channel group (planner.FOLD)
                    cands = P.folded_candidate_plans(kind, Lr.wshape, cls, q, eng.es, aux_es=aux_es, heuristic_only=not self.tune)
                    out.append(_Choice(cands, woff, wshape=tuple(Lr.wshape), fold=P.FOLD, cmod=P.gemm_dims(kind, Lr.wshape)[1]))
                    continue
                if self.tune:
                    cands = P.candidate_plans(kind, Lr.wshape, cls, q, eng.es, kc_pad=kc_pad, aux_es=aux_es, in_split=in_split, n=self.n)
                    if eng.deep == "0":
                        cands = [pl for pl in cands if pl.depth != -7]
                    elif eng.deep == "force" and any(pl.depth == -7 for pl in cands):
                        cands = [pl for pl in cands if pl.depth == -7]
                else:
                    cands = [P.plan_igemm(kind, Lr.wshape, cls, q, eng.es, kc_pad=kc_pad, aux_es=aux_es, in_split=in_split)]
                    if eng.deep == "force":  # untuned lowering with the deep-level kernel wherever it is offered (tests: the whole network through csrc/dconv.hip's domain)
                        kr_, nr_ = P.gemm_dims(kind, Lr.wshape)
                        dp = P.deep_plans(kind, Lr.wshape, cls, q, eng.es, kc_pad if kc_pad is not None else P.round_up(kr_, 8), nr_, kr_, self.n, in_split)
                        if dp:
                            cands = dp[:1]
                out.append(_Choice(cands, woff, absorbed.layer.wshape if absorbed is not None else None, eng.layout.param_off[absorbed.layer.wkey][0] if absorbed is not None else 0, wshape=tuple(Lr.wshape)))
            out[0].alt = alt
            return out

        self.cplans: Dict[str, _ConvPlans] = {}
        self.wide_dpre: set = set()  # sigmoid convolutions whose d(pre-sigmoid) buffer is 16 channels wide (channel 0 real)
        for op in eng.prog.ops:
            if not isinstance(op, (ConvBnAct, ConvPlain)):
                continue
            Lr = op.layer
            if Lr.prefix in self.merged:
                self.cplans[Lr.prefix] = _ConvPlans([], [], None)
                continue
            absorbed = self.absorbs.get(Lr.prefix)
            dims_in = self.lv[Lr.level]
            kind = "convT_fwd" if Lr.transposed else "conv_fwd"
            dims_out = P.out_dims(kind, dims_in, Lr.kernel, Lr.stride)
            assert dims_out == self.lv[Lr.out_level]
            aux_es = 0 if (op.res is None or absorbed is not None) else (4 if op.res.kind == 'f32' else eng.es)
            can_fold = (eng.fold and absorbed is None and op.res is None and op.x.parts is None and op.x.base is None and op.out.base is None
                        and P.foldable(Lr.kernel, Lr.stride, Lr.transposed, Lr.cin, Lr.cout, dims_in))
            # Measured (tools: bench.py --profile, VSSEG_ZFOLD=all): folding pays on the narrow-OUTPUT side only — the attention sigmoid
            # convolution 16->1 drops from 0.51 to 0.33 ms (its input is read as 128 real channels, the 1-channel map is written as
            # 32-byte rows).  On the narrow-INPUT side (network input 1->16, dY of the sigmoid conv) the folded output rows are 256 B
            # wide and every 16-channel N-tile stores 32-byte fragments of them: 0.46 -> 0.50 ms and 0.51 -> 0.70 ms, so those stay unfolded.
            fold_fwd = can_fold and Lr.cout == 1
            fwd = choices(kind, Lr, dims_in if Lr.transposed else dims_out, op.x.c, aux_es, op.x.parts[0].c if op.x.parts else 0, absorbed, fold=fold_fwd)
            dgrad, wg = [], None
            if self.train:
                if op.x.root.name != eng.prog.input.name:  # the network input needs no gradient
                    dk = "convT_dgrad" if Lr.transposed else "conv_dgrad"
                    q = dims_in if Lr.transposed else tuple((d + s - 1) // s for d, s in zip(dims_in, Lr.stride))
                    # the data gradient of a 3x3x3 C -> 1 convolution (the sigmoid convolutions of levels 2-3) reads d(pre-sigmoid) as ONE 16-channel chunk so that the compute
                    # kernel (csrc/cconv.hip) is among its candidates: 0.16 -> 0.06 ms at level 2 (the general kernel on an 8-channel zero-extension was the only plan)
                    wide = (eng.wide_dpre and self.tune and eng.es == 2 and Lr.cout == 1 and not Lr.transposed and Lr.kernel == (3, 3, 3) and tuple(Lr.stride) == (1, 1, 1)
                            and not any(v % t for v, t in zip(q, P.COMPUTE_TILE)) and P.compute_split(Lr.cin) is not None)
                    if wide:
                        self.wide_dpre.add(Lr.prefix)
                    dgrad = choices(dk, Lr, q, 16 if wide else P.round_up(Lr.cout, 8), eng.es, 0, absorbed)
                wg = P.plan_wgrad(Lr.transposed, Lr.wshape, Lr.kernel, Lr.stride, dims_in if Lr.transposed else dims_out, eng.es)
            self.cplans[Lr.prefix] = _ConvPlans(fwd, dgrad, wg, fold_fwd)
        # ResidualUnit: the 1x1x1 residual convolution of the SAME input rides along in the unit's first 3x3x1 convolution (vsseg_igemm_desc.res_tiles, csrc/mconv.hip
        # NR: the input is read once for both, ref:params/networks/blocks/convolutions.py:241-255) where a marching plan with residual tiles is instantiated
        self.resn: Dict[str, ConvPlain] = {}     # unit convolution prefix -> the residual convolution op riding along
        self.resn_of: Dict[str, str] = {}        # residual convolution prefix -> unit convolution prefix
        if eng.resn and eng.es == 2:
            for op in eng.prog.ops:
                if not isinstance(op, ConvBnAct) or op.layer.transposed or tuple(op.layer.stride) != (1, 1, 1) or op.layer.kernel != (3, 3, 1) or op.x.parts is not None or op.x.base is not None:
                    continue
                Lr = op.layer
                rc = next((o for o in eng.prog.ops if isinstance(o, ConvPlain) and o.x is op.x and o.layer.kernel == (1, 1, 1) and o.layer.prefix.endswith(".residual") and o.layer.cout == Lr.cout
                           and o.layer.prefix not in self.merged and o.act == "none" and o.res is None and o.out.kind == "act"), None)
                if rc is None or len(self.cplans[Lr.prefix].fwd) != 1 or self.cplans[Lr.prefix].fold_fwd:
                    continue
                cls = P.lattice_classes("conv_fwd", Lr.kernel, Lr.stride)[0]
                plans = P.march_res_plans(Lr.wshape, rc.layer.wshape, cls, self.lv[Lr.level], eng.es, op.x.c, self.n)
                if not plans:
                    continue
                self.cplans[Lr.prefix].fwd = [_Choice(plans if self.tune else plans[:1], eng.layout.param_off[Lr.wkey][0], wshape=tuple(Lr.wshape), woff_res=eng.layout.param_off[rc.layer.wkey][0])]
                self.resn[Lr.prefix], self.resn_of[rc.layer.prefix] = rc, Lr.prefix

    def _register(self, ch: _Choice, pl: P.IgemmPlan):
        """Append the chosen plan's weight gather map(s) to the step's pack list."""
        ch.chosen, ch.map_off = pl, self._map_len
        self._maps.append(np.where(pl.pack_map >= 0, pl.pack_map + ch.woff, -1).astype(np.int32))
        if ch.wshape2 is not None:
            m2 = P.pack_map_centre(pl, ch.wshape2)
            self._maps2.append(np.where(m2 >= 0, m2 + ch.woff2, -1).astype(np.int32))
        else:
            self._maps2.append(np.full(pl.pack_map.size, -1, np.int32))
        self._map_len += pl.pack_map.size
        if pl.res_tiles:  # the residual convolution's tiles: their own gather map behind the main one
            ch.map_off_res = self._map_len
            self._maps.append(np.where(pl.pack_map_res >= 0, pl.pack_map_res + ch.woff_res, -1).astype(np.int32))
            self._maps2.append(np.full(pl.pack_map_res.size, -1, np.int32))
            self._map_len += pl.pack_map_res.size

    def _finish_pack(self):
        eng = self.eng
        self.pack_map = torch.from_numpy(np.concatenate(self._maps)).to(eng.device)
        self.pack_map2 = torch.from_numpy(np.concatenate(self._maps2)).to(eng.device) if self.absorbs else None
        assert self.pack_map2 is None or self.pack_map2.numel() == self.pack_map.numel()
        self.wpack = torch.zeros(self._map_len, dtype=eng.tdtype, device=eng.device)
        for d, off in self._wpack_fixups:
            d.wpack = self.wpack.data_ptr() + eng.es * off
        for d, off in self._res_fixups:
            d.wpack_res = self.wpack.data_ptr() + eng.es * off
        for d, attr, off in self._chain_fixups:
            setattr(d, attr, self.wpack.data_ptr() + eng.es * off)
        del self._maps, self._maps2, self._wpack_fixups, self._res_fixups, self._chain_fixups
        self._tune_gflat = None
        _tune_cache_save()

    @staticmethod
    def _fill_desc(d: L.IgemmDesc, pl: P.IgemmPlan):
        d.q, d.is_, d.os, d.oo = L.i3(pl.q), L.i3(pl.cls.is_), L.i3(pl.cls.os), L.i3(pl.cls.oo)
        d.ntaps = pl.ntaps
        for t, (off, _) in enumerate(pl.cls.taps):
            d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
        d.tile = L.i3(pl.tile)
        d.mtw, d.nt, d.nsplit, d.ck, d.nchunks, d.ksteps, d.depth = pl.mtw, pl.nt, pl.nsplit, pl.ck, pl.nchunks, pl.ksteps, pl.depth
        d.res_tiles = pl.res_tiles
        if P.is_shuffle(pl):  # fused output-parity classes: output channel tile t is class t, its channels are the real channels 0..nc-1
            d.cout_mod = pl.nc
        d.class_split = len(pl.classes) if pl.classes is not None else 0
        for s_, c_ in enumerate(pl.classes or ()):  # workgroup row s_ = lattice class s_: its output offset and its taps (indices into the union tap table above)
            d.class_oo[s_][0], d.class_oo[s_][1], d.class_oo[s_][2] = c_.oo
            d.class_ntaps[s_] = len(c_.taps)
            for i, t in enumerate(pl.class_taps(s_)):
                d.class_tap[s_][i] = t

    def _choose(self, ch: _Choice, d: L.IgemmDesc) -> P.IgemmPlan:
        """Tuned-plan cache lookup (same launch signature measured before, in this process or in a cache file), else measure."""
        p0 = ch.cands[0]
        key = (f"{p0.kind}|f{ch.fold}|w{ch.wshape}|is{p0.cls.is_}os{p0.cls.os}oo{p0.cls.oo}|q{p0.q}|n{self.n}|es{self.eng.es}|kc{p0.kc}|acc{int(d.accumulate)}|res{int(d.res_mode)}"
               f"|st{int(bool(d.stats))}|two{int(bool(d.inp.ptr2))}{int(bool(d.out.ptr2))}" + ("|gin" if bool(d.in_gate) else "") + ("|cs" if p0.classes is not None else "")
               + (f"|rn{p0.res_tiles}{int(bool(d.res_out.ptr))}" if p0.res_tiles else "") + ("|c1" if (d.inp.c == 1 and p0.kc == 8) else ("|c2" if (d.inp.c == 2 and d.inp.pitch == 2 and p0.kc == 8) else ""))
)
        cache = _tune_cache()
        hit = cache.get(key)
        # VSSEG_RETUNE_DEPTHS="-6": launches with a candidate plan of one of these depths are measured again although a choice is cached (how the plans of a
        # NEW kernel variant get into the cache without re-measuring every launch: tools/tune_shapes.py)
        retune = {int(v) for v in os.environ.get("VSSEG_RETUNE_DEPTHS", "").split(",") if v.strip()}
        if hit is not None and os.environ.get("VSSEG_AUTOTUNE", "1") != "force" and not any(pl.depth in retune for pl in ch.cands):
            for pl in ch.cands:
                if [list(pl.tile), pl.nt, pl.nsplit, pl.ck, pl.depth] == (hit if len(hit) > 4 else hit + [1]):
                    ch.cached = True
                    return pl
        pl = self._autotune(ch, d)
        # the choice always enters the IN-PROCESS cache: later plans of this process replay it, and under data parallel rank 0 broadcasts this cache so that every rank runs
        # the same kernels.  VSSEG_DEEP=0 / force are experiment modes: their restricted candidate lists are never written to the cache FILE
        cache[key] = [list(pl.tile), pl.nt, pl.nsplit, pl.ck, pl.depth]
        if self.eng.deep == "1":
            _tune_cache.dirty = True
        return pl

    def _autotune(self, ch: _Choice, d: L.IgemmDesc) -> P.IgemmPlan:
        """Measure every candidate of one launch with its real operands and epilogue (HIP events, best of 3 after a warm-up)."""
        eng, lib = self.eng, self.eng.lib
        stream = torch.cuda.current_stream().cuda_stream
        times = []
        for pl in ch.cands:
            m = torch.from_numpy(np.where(pl.pack_map >= 0, pl.pack_map + ch.woff, -1).astype(np.int32)).to(eng.device)
            wp = torch.empty(m.numel(), dtype=eng.tdtype, device=eng.device)
            L.check(lib.vsseg_gather_cast(eng.flat.data_ptr(), m.data_ptr(), None, wp.data_ptr(), m.numel(), L.BF16 if eng.es == 2 else L.F32, stream), "gather_cast")
            self._fill_desc(d, pl)
            d.wpack = wp.data_ptr()
            if pl.res_tiles:
                mr = torch.from_numpy(np.where(pl.pack_map_res >= 0, pl.pack_map_res + ch.woff_res, -1).astype(np.int32)).to(eng.device)
                wpr = torch.empty(mr.numel(), dtype=eng.tdtype, device=eng.device)
                L.check(lib.vsseg_gather_cast(eng.flat.data_ptr(), mr.data_ptr(), None, wpr.data_ptr(), mr.numel(), L.BF16 if eng.es == 2 else L.F32, stream), "gather_cast")
                d.wpack_res = wpr.data_ptr()
            if False:  # (debugging aid, was VSSEG_TUNE_TRACE: name every candidate before it runs, finish it before the next one)
                import sys
                print(f"[tune] {pl.kind} depth={pl.depth} q={pl.q} tile={pl.tile} mtw={pl.mtw} nt={pl.nt} ns={pl.nsplit} ck={pl.ck} kc={pl.kc} nc={pl.nc} in=({d.inp.c},{d.inp.pitch},{d.inp.n},{d.inp.x},{d.inp.y},{d.inp.z},two={bool(d.inp.ptr2)}) "
                      f"out=({d.out.c},{d.out.pitch},dt={d.out.dtype},two={bool(d.out.ptr2)}) acc={d.accumulate} res={d.res_mode} stats={bool(d.stats)} cls={d.class_split}", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
            if lib.vsseg_igemm(C.byref(d), stream):  # a candidate the kernel rejects is simply not chosen
                times.append(float("inf"))
                continue
            if False:
                torch.cuda.synchronize()
            best = float("inf")
            for _ in range(self.eng.tune_reps):  # best of N single launches (N = 5: with 3 the choice between near-equal plans flipped from run to run by up to 0.5 ms per step)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.vsseg_igemm(C.byref(d), stream)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            times.append(best)
        ch.tuned_ms = times
        i = int(np.argmin(times))
        if times[i] > 0.97 * times[0]:  # keep the default unless a candidate is clearly faster (timer noise on the tiny layers)
            i = 0
        return ch.cands[i]

    @staticmethod
    def _fold_desc(t: L.Tensor, fold: int) -> L.Tensor:
        """[N,X,Y,Z,C] dense -> the bit-identical [N,X,Y,Z/fold,fold*C] view."""
        assert t.c == t.pitch and not t.ptr2 and t.z % fold == 0, "z-folding needs a dense, single-part tensor"
        return L.Tensor(t.ptr, t.dtype, t.c * fold, t.pitch * fold, t.n, t.x, t.y, t.z // fold)

    @staticmethod
    def _sample(t: Optional[L.Tensor], b: int) -> Optional[L.Tensor]:
        """Sample b of a batched channels-last tensor (both parts of a two-part tensor), as a batch-1 descriptor."""
        if t is None:
            return None
        step = t.x * t.y * t.z * t.pitch * (2 if t.dtype == L.BF16 else 4)
        r = L.Tensor(t.ptr + b * step, t.dtype, t.c, t.pitch, 1, t.x, t.y, t.z)
        if t.ptr2:
            r.ptr2, r.csplit = t.ptr2 + b * step, t.csplit
        return r

    def _igemm_classes(self, lst, chs: List[_Choice], inp: L.Tensor, out: L.Tensor, *, res: Optional[L.Tensor] = None, **kw):
        """All lattice classes of one convolution (the output-parity classes of a transposed convolution / of a strided data gradient): each is
        its own launch and reads the WHOLE input (the stride-(2,2,1) transitions of levels 0-2 run all classes in one streaming-kernel launch
        instead, planner.shuffle_plans)."""
        if chs[0].alt is not None and self._use_class_split(chs, inp, out, res, kw):
            chs = [chs[0].alt]  # ... or every class in one launch of the general kernel, where that is faster (the deep levels)
        for ch in chs:
            self._igemm(lst, ch, inp, out, res=res, **kw)

    CLASS_SPLIT_MAX_VOXELS = 100_000  # heuristic of the untuned lowering: class-split launches below this many coarse voxels (measured: tools/bench_class_split.py)

    def _use_class_split(self, chs: List[_Choice], inp: L.Tensor, out: L.Tensor, res, kw) -> bool:
        """One class-split launch or one launch per class?  The split launch wins where the per-class launches are a handful of workgroups each
        (levels 3-5: 0.23 -> 0.09 ms, 0.18 -> 0.03 ms); at level 2 with batch 4 every workgroup of the split launch fetches the union halo and
        keeps the LDS of the 8-tap class, and the tuned per-class launches are faster (0.27 against 0.38 ms).  Measured once per launch signature."""
        alt = chs[0].alt
        p0 = alt.cands[0]
        nb = kw.get("nb") or self.n
        if not self.tune:
            return p0.depth == -7 or nb * p0.q[0] * p0.q[1] * p0.q[2] <= self.CLASS_SPLIT_MAX_VOXELS
        if alt.chosen is not None or chs[0].chosen is not None:  # a further launch of the same op (another sample): same decision
            return alt.chosen is not None
        key = (f"use_cs|{p0.kind}|w{alt.wshape}|q{p0.q}|n{nb}|es{self.eng.es}|kc{p0.kc}|acc{int(kw.get('accumulate', 0))}|res{int(kw.get('res_mode', 0))}"
               f"|st{int(bool(kw.get('stats')))}|two{int(bool(inp.ptr2))}{int(bool(out.ptr2))}" + ("|gate" if kw.get("gate") else ""))
        cache = _tune_cache()
        retune = {int(v) for v in os.environ.get("VSSEG_RETUNE_DEPTHS", "").split(",") if v.strip()}
        if key in cache and os.environ.get("VSSEG_AUTOTUNE", "1") != "force" and not any(pl.depth in retune for ch in (list(chs) + [alt]) for pl in ch.cands):
            return bool(cache[key])
        eng, lib = self.eng, self.eng.lib
        stream = torch.cuda.current_stream().cuda_stream
        times = []
        for variant in (chs, [alt]):  # lower both into scratch lists (plans chosen as usual, nothing registered), pack their weights, time the launches
            tmp, packs = [], []
            for ch in variant:
                self._igemm(tmp, ch, inp, out, res=res, probe=True, **kw)
                m = torch.from_numpy(np.where(ch.probe_plan.pack_map >= 0, ch.probe_plan.pack_map + ch.woff, -1).astype(np.int32)).to(eng.device)
                wp = torch.empty(m.numel(), dtype=eng.tdtype, device=eng.device)
                L.check(lib.vsseg_gather_cast(eng.flat.data_ptr(), m.data_ptr(), None, wp.data_ptr(), m.numel(), L.BF16 if eng.es == 2 else L.F32, stream), "gather_cast")
                tmp[-1][1][0]._obj.wpack = wp.data_ptr()
                packs.append(wp)
            best = float("inf")
            for _ in range(1 + eng.tune_reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for fn, args, _meta in tmp:
                    L.check(fn(*args, stream), "class-split decision")
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            times.append(best)
        use = times[1] < times[0]
        cache[key] = int(use)  # in-process always (every later plan and, under data parallel, every rank repeats the choice); the cache FILE only outside the experiment modes
        if self.eng.deep == "1":
            _tune_cache.dirty = True
        self.class_split_ms = getattr(self, "class_split_ms", []) + [(key, times)]
        return use

    def _igemm(self, lst, ch: _Choice, inp: L.Tensor, out: L.Tensor, *, bias=0, bias2=0, scale=0, shift=0, alpha=0, act=L.ACT_NONE, res: Optional[L.Tensor] = None,
               res_mode=L.RES_NONE, accumulate=0, stats=0, stats_stride=0, ncls=1, gate=0, nb: Optional[int] = None, in_gate=0, probe=False, res_out: Optional[L.Tensor] = None, bias_res=0,
               in1=None):
        """probe: build the launch (plan chosen / measured as usual) WITHOUT making it part of the step — no packed-weight registration, the descriptor
        is only referenced by `lst` (Plan._use_class_split times two alternative lowerings of an op this way; ch.probe_plan = the plan it used)."""
        nb = self.n if nb is None else nb
        d = L.IgemmDesc()
        d.gate = gate or None
        d.in_gate = in_gate or None
        if ch.fold:
            inp, out, res = self._fold_desc(inp, ch.fold), self._fold_desc(out, ch.fold), (self._fold_desc(res, ch.fold) if res is not None else None)
            d.cout_mod = ch.cmod
        d.inp, d.out = inp, out
        d.bias, d.bias2, d.scale, d.shift, d.alpha = bias or None, bias2 or None, scale or None, shift or None, alpha or None
        d.act, d.res_mode, d.accumulate = act, res_mode, accumulate
        if res is not None:
            d.res = res
        d.stats, d.stats_stride = stats or None, stats_stride
        d.bias_res = bias_res or None
        if in1 is not None:  # VSSEG_RES_IN1: (one-channel tensor, weights, bias) of a 1 -> C 1x1x1 convolution added behind the activation
            d.in1, d.in1_w, d.in1_b = in1
        if res_out is not None:
            d.res_out = res_out
        if ch.chosen is not None:  # a further launch of the same lattice class (another sample): same plan, same packed weights
            pl = ch.chosen
        else:
            pl = self._choose(ch, d) if (self.tune and len(ch.cands) > 1) else ch.cands[0]
            if probe:
                ch.probe_plan = pl
            else:
                self._register(ch, pl)
        self._fill_desc(d, pl)
        if not probe:
            self._wpack_fixups.append((d, ch.map_off))
            if pl.res_tiles:
                self._res_fixups.append((d, ch.map_off_res))
            self.keep.append(d)
        nvalid = nb  # output voxels this lattice class writes
        for a, oa in enumerate((out.x, out.y, out.z)):
            nvalid *= min(pl.q[a], -(-(oa - pl.cls.oo[a]) // pl.cls.os[a]))
        if P.is_shuffle(pl):
            nvalid = nb * out.x * out.y * out.z * (pl.nt * 16 // pl.nc) // 4  # the parity classes this launch writes
        taps_eff = 2.25 if P.is_shuffle(pl) else pl.ntaps
        if pl.classes is not None:  # every output voxel, by the class it belongs to (27 / 8 taps per voxel on average for 3x3x3 stride 2)
            nvalid, taps_eff = nb * out.x * out.y * out.z, sum(len(c_.taps) for c_ in pl.classes) / len(pl.classes)
        es_in, es_out = (2 if inp.dtype == L.BF16 else 4), (2 if out.dtype == L.BF16 else 4)
        tuned = " tuned[cache]" if ch.cached else ("" if ch.tuned_ms is None else f" tuned[{ch.cands.index(pl)}/{len(ch.cands)} {ch.tuned_ms[0]:.3f}->{min(ch.tuned_ms):.3f}ms]")
        fold_tag = f" zfold{ch.fold}" if ch.fold else ""
        rn = pl.res_tiles * 16 if pl.res_tiles else 0  # output channels of the 1x1x1 residual convolution riding along
        meta = dict(tag=f"{pl.kind}{fold_tag} q={pl.q} K={pl.kreal}x{pl.ntaps} N={pl.nc}{'+res' + str(rn) if rn else ''} tile={pl.tile} ck={pl.ck} ns={pl.nsplit} D={pl.depth} lds={pl.lds}{tuned}", name=self._igemm_name(pl, inp), kind="mfma",
                    flops=2.0 * nvalid * (taps_eff * pl.kreal * pl.nc + pl.kreal * rn) / max(ch.fold, 1),
                    # algorithmic bytes: input once + output once (+ the residual / mask / gated operand or the previous gradient an
                    # accumulating launch has to read: one more output-sized tensor)
                    bytes=float(nvalid) * pl.nc * es_out * (2 if (accumulate or res is not None) else 1) + float(nb) * inp.x * inp.y * inp.z * pl.kreal * es_in / (1 if pl.classes is not None else ncls)
                    + (float(nvalid) * rn * es_out if res_out is not None else 0.0))
        lst.append([self.eng.lib.vsseg_igemm, [C.byref(d)], meta])

    def _march_cands(self, ch: _Choice, Lr: Layer) -> List[P.IgemmPlan]:
        """The marching-kernel plans (depth -5) of a stride-1 3x3x1 forward launch: those among its candidates when the autotuner listed them, else
        planner.march_plans' deterministic list (the untuned lowering then takes the first one)."""
        got = [pl for pl in ch.cands if P.is_march(pl)]
        if got or len(ch.cands) != 1:
            return got
        p0 = ch.cands[0]
        got = P.march_plans(p0.kind, Lr.wshape, p0.cls, p0.q, self.eng.es, p0.kc, p0.nc, p0.kreal, self.n)
        for pl in got:
            if pl.pack_map is None:
                pl.pack_map = P.pack_map(pl, Lr.wshape)
        return got

    def _compact_choice(self, ch: _Choice, Lr: Layer) -> Optional[_Choice]:
        """A convolution launch whose input has ONE real channel (the network input, the pre-sigmoid gradient of an attention map), on the marching kernel reading the
        COMPACT one-channel tensor (csrc/mconv.hip C1: 2 bytes per voxel from HBM instead of the 16 of the zero-extended channel group): the marching plans (weights in
        LDS) of the launch as their own choice, or None where there is none / the switch is off."""
        if not self.eng.compact_c1 or self.eng.es != 2 or ch.fold or not ch.cands or ch.cands[0].kc != 8 or ch.cands[0].kreal not in (1, 2):
            return None  # (two real channels: the gradient of the logits, csrc/mconv.hip CC = 2)
        got = getattr(ch, "_compact", None)
        if got is None:
            cands = [pl for pl in self._march_cands(ch, Lr) if pl.depth == -5]
            got = ch._compact = _Choice(cands, ch.woff, wshape=ch.wshape, wshape2=ch.wshape2, woff2=ch.woff2) if cands else False  # (a merged residual convolution stays merged)
            if got:
                got._is_compact, got._compact = True, got  # (asked again with the new choice: itself)
        return got or None

    def _logits_compact(self, Lr: Layer, x: TensorSpec) -> bool:
        """Whether BOTH backward launches of the logits convolution can read the gradient of the two logits compact: a marching data-gradient plan (csrc/mconv.hip CC = 2)
        and a marching weight-gradient tile with the compact P operand (csrc/mwgrad.hip PC2, instantiated for 32 input channels)."""
        cp = self.cplans[Lr.prefix]
        if (not self.eng.compact_c1 or self.eng.es != 2 or Lr.transposed or tuple(Lr.stride) != (1, 1, 1) or Lr.kernel != (3, 3, 1) or Lr.cout != 2 or Lr.cin != 32
                or len(cp.dgrad) != 1 or x.parts is not None or x.base is not None or self._compact_choice(cp.dgrad[0], Lr) is None):
            return False
        return bool(P.march_wgrad_tiles(Lr.cin, 8, cp.wgrad.q, self.n, self.eng.wgrad_scratch().numel()))

    @staticmethod
    def _igemm_name(pl: P.IgemmPlan, inp: L.Tensor) -> str:
        """Kernel group of a convolution launch in the profiles: which of the kernels its plan runs on."""
        if pl.depth in (-2, -4):
            return f"sconv<bf16,{pl.nt}>"
        if pl.depth == -3:
            return f"cconv<bf16,{pl.nt}>"
        if P.is_march(pl):
            return f"mconv<bf16,{pl.nt}>"
        if pl.depth == -7:
            return f"dconv<bf16,{pl.mtw},{pl.nt}>"
        if pl.depth == -8:
            return f"tconv<bf16,{pl.ck // 8}>"
        if pl.depth == -9:
            return f"gconv<bf16,{pl.nt}>"
        return f"igemm<{'bf16' if inp.dtype == L.BF16 else 'f32'},{pl.nt},{pl.mtw}>"

    def _ew_meta(self, name: str, level: int, passes_c: int, dtype_es: Optional[int] = None) -> dict:
        """Launch metadata of a streaming kernel: algorithmic bytes = (channels read + written per voxel, summed over its tensor
        passes) x voxels x element size.  `passes_c` = that per-voxel channel count."""
        es = self.eng.es if dtype_es is None else dtype_es
        return dict(name=name, kind="hbm", flops=0.0, bytes=float(self._vox(level)) * passes_c * es, tag=f"L{level} {passes_c} ch/voxel")

    # ------------------------------------------------------------------ lowering
    def _lower(self):
        eng, prog, lib, dev = self.eng, self.eng.prog, self.eng.lib, self.eng.device
        ops = prog.ops
        bn_layers = [op.layer for op in ops if isinstance(op, ConvBnAct)]
        cpad = {Lr.prefix: P.round_up(Lr.cout, 16) for Lr in bn_layers}
        tot_c = max(sum(cpad.values()), 1)
        nsh = L.STAT_SHARDS
        # fp64 sharded statistics: row 0 forward (sum, sumsq), row 1 backward (sum dz, sum dz*xhat, sum dout) + PReLU-slope accumulators
        self.stats = torch.zeros(2, nsh * 3 * tot_c + nsh * max(len(bn_layers), 1), dtype=torch.float64, device=dev)
        self.vec = torch.zeros(6, tot_c, dtype=torch.float32, device=dev)  # mean, invstd, scale, shift, mean_dz, mean_dzx
        st_off, v_off, a_off = {}, {}, {}
        o = v = 0
        for i, Lr in enumerate(bn_layers):
            st_off[Lr.prefix], v_off[Lr.prefix], a_off[Lr.prefix] = o, v, nsh * 3 * tot_c + nsh * i
            o += nsh * 3 * cpad[Lr.prefix]
            v += cpad[Lr.prefix]
        row = self.stats.shape[1]
        sptr = lambda which, pre: self.stats.data_ptr() + 8 * (which * row + st_off[pre])
        aptr = lambda pre: self.stats.data_ptr() + 8 * (row + a_off[pre])
        vptr = lambda r, pre: self.vec.data_ptr() + 4 * (r * tot_c + v_off[pre])
        salt = {Lr.prefix: (i + 1) | L.SEED_INDIRECT for i, Lr in enumerate(bn_layers)}  # the kernels read the seed through seed_dev
        SEED = self.seed_dev.data_ptr()
        p_drop = float(eng.dropout_p) if self.train else 0.0
        self.bn_info = {Lr.prefix: (Lr, salt[Lr.prefix] & ~L.SEED_INDIRECT) for Lr in bn_layers}

        def keep_ptr(Lr: Layer):
            """Dropout keep-mask bytes of a BatchNorm layer (one byte per voxel and 8-channel group = 1/16 of the bf16 activation): written by
            the forward, read by both backward passes, which are otherwise VALU-bound on running Philox4x32-10 again (measured: the three
            BN kernels ran at the same ~90-130 G iterations/s whatever they moved; VSSEG_KEEPMASK=0 regenerates instead)."""
            if not (self.train and p_drop > 0.0 and eng.keepmask):
                return None
            return self._raw("keep:" + Lr.prefix, Lr.out_level, Lr.cout // 8, dtype=torch.uint8).data_ptr()

        # First encoder ResidualUnit (in_channels = 1): its 1x1x1 residual convolution is x1[v]*w[c] + b[c]; in training it is
        # computed inside the BN/dropout/PReLU kernel that adds it (vsseg_bn_act_fwd_res1) instead of by an igemm launch that
        # writes a 16-channel tensor for that kernel to read back.  (Eval adds the residual in the conv epilogue and keeps the launch.)
        plain_by_out = {op.out.name: op for op in ops if isinstance(op, ConvPlain)}
        res1_fused: Dict[str, ConvPlain] = {}
        if self.train and eng.res1_fuse:
            for op in ops:
                if isinstance(op, ConvBnAct) and op.res is not None and op.res.name in plain_by_out:
                    pr = plain_by_out[op.res.name]
                    if pr.layer.cin == 1 and pr.layer.kernel == (1, 1, 1) and pr.x.root.name == prog.input.name and pr.act == "none" and pr.res is None:
                        res1_fused[pr.layer.prefix] = pr

        # ... and in eval inside the epilogue of the convolution it is added to (VSSEG_RES_IN1), where that convolution has a marching plan
        self.eval_in1: Dict[str, ConvPlain] = {}  # prefix of the convolution whose epilogue adds the residual -> the residual convolution op
        if not self.train and eng.res1_fuse and eng.es == 2 and not eng.dry_run:
            for op in ops:
                if isinstance(op, ConvBnAct) and op.res is not None and op.res.name in plain_by_out and op.x.parts is None and op.x.base is None:
                    pr = plain_by_out[op.res.name]
                    cpo = self.cplans[op.layer.prefix]
                    if (pr.layer.cin == 1 and pr.layer.kernel == (1, 1, 1) and pr.x.root.name == prog.input.name and pr.act == "none" and pr.res is None and op.layer.kernel == (3, 3, 1)
                            and tuple(op.layer.stride) == (1, 1, 1) and not op.layer.transposed and len(cpo.fwd) == 1 and not cpo.fold_fwd and op.layer.prefix not in self.resn and self._march_cands(cpo.fwd[0], op.layer)):
                        self.eval_in1[op.layer.prefix] = pr
                        res1_fused[pr.layer.prefix] = pr  # (its own launch is skipped)

        # Attention gates applied ON LOAD (ref:params/networks/blocks/attentionblock.py:43-47: out = att.repeat(C) * x + x): where the gated tensor's only
        # reader is ONE stride-1 3x3x1 convolution (with its merged 1x1x1 residual) that runs on the marching kernel, that convolution and its weight
        # gradient read x and the attention map and multiply in LDS (csrc/mconv.hip MODE 3, csrc/mwgrad.hip GIN): vsseg_att_apply_fwd is not launched and
        # the gated tensor (2c channels at the level's resolution) is neither written nor read back.  Level 0 of this network (the logits convolution).
        self.gate_onload: Dict[str, AttGate] = {}
        if eng.gate_onload and eng.es == 2:  # (independent of the autotuner: VSSEG_AUTOTUNE=0 lowers the same graph, with the first marching plan)
            for g in ops:
                if not isinstance(g, AttGate):
                    continue
                readers = [o for o in ops if isinstance(o, (ConvBnAct, ConvPlain)) and (o.x is g.out or (o.x.parts is not None and g.out in o.x.parts) or o.res is g.out)]
                readers += [o for o in ops if isinstance(o, AttGate) and (o.x is g.out or (o.x.parts is not None and g.out in o.x.parts))]
                # (b) a ResidualUnit whose first 3x3x1 convolution carries the unit's 1x1x1 residual convolution as residual tiles (self.resn) and whose backward is the
                #     fused launch (csrc/mbwd.hip, x gated on load): the level-1 decoder unit.  Both readers of the gated tensor are then one forward launch.
                unit = [o for o in readers if isinstance(o, ConvBnAct) and o.layer.prefix in self.resn]
                if (eng.gate_onload_units and len(unit) == 1 and len(readers) == 2 and self.resn[unit[0].layer.prefix] in readers and unit[0].x is g.out and (unit[0].layer.cin, unit[0].layer.cout) == (64, 32)
                        and g.x.c == 64 and (not self.train or (eng.fused_bwd == "1" and eng.fused_bwd_res and (p_drop == 0.0 or eng.keepmask)
                                                                and P.fused_bwd_tiles(32, 64, self.lv[unit[0].layer.level], self.n, 48 * 1024 * 1024, res=True)))):
                    self.gate_onload[g.out.name] = g
                    continue
                main = [o for o in readers if isinstance(o, ConvPlain) and o.layer.prefix not in self.merged]
                if len(main) != 1 or any(o is not main[0] and not (isinstance(o, ConvPlain) and self.merged.get(o.layer.prefix) is main[0]) for o in readers):
                    continue
                c = main[0]
                Lc, cpc = c.layer, self.cplans[c.layer.prefix]
                if Lc.transposed or tuple(Lc.stride) != (1, 1, 1) or Lc.kernel != (3, 3, 1) or Lc.cin != 32 or Lc.cout > 8 or c.x is not g.out or len(cpc.fwd) != 1 or cpc.fold_fwd or g.x.c != Lc.cin:
                    continue
                if not self._march_cands(cpc.fwd[0], Lc):
                    continue
                if self.train and not P.march_wgrad_tiles(Lc.cin, 8, self.lv[Lc.level], self.n, eng.wgrad_scratch().numel()):
                    continue
                self.gate_onload[g.out.name] = g

        # Chained marching convolutions (csrc/chain.hip, vsseg_conv_chain; inference only): two consecutive stride-1 3x3x1 convolutions whose intermediate 16-channel tensor has
        # no other reader run as ONE launch with that tensor in LDS — the first ResidualUnit of the encoder (1 -> 16 -> 16 with the residual convolution of the network input in
        # the second epilogue) and the attention block of the finest decoder level (32 -> 16 + ReLU -> 1 + sigmoid).  201 MB per patch are neither written nor read back, per pair.
        self.chain_first: Dict[str, tuple] = {}  # prefix of the first convolution -> (first op, second op, plan)
        self.chain_second: set = set()           # prefixes of the second convolutions: emitted with the first
        if eng.chain != "0" and eng.es == 2 and not eng.dry_run:
            def march3(Lr):
                return not Lr.transposed and tuple(Lr.stride) == (1, 1, 1) and Lr.kernel == (3, 3, 1)
            for a in ops:
                if self.train:  # inference only (in training a BatchNorm needs the whole tensor between the two convolutions; the attention block, which has none, measured no gain: DESIGN 3.11)
                    continue
                if not isinstance(a, (ConvBnAct, ConvPlain)) or not march3(a.layer) or a.layer.cout not in (16, 32) or a.res is not None or a.x.base is not None or a.out.base is not None:
                    continue
                pa_ = a.layer.prefix
                unit = eng.chain != "l0" and pa_ in self.resn and (a.layer.cin, a.layer.cout) == (16, 32) and a.x.parts is None  # a two-sub-unit ResidualUnit whose residual convolution rides along
                if (pa_ in self.resn) != unit or (a.layer.cout == 32) != unit or pa_ in self.merged or pa_ in self.absorbs or pa_ in self.resn_of or a.x.name in self.gate_onload:
                    continue
                readers = [o for o in ops if o is not a and ((getattr(o, "x", None) is a.out) or (getattr(getattr(o, "x", None), "parts", None) is not None and a.out in o.x.parts)
                                                             or getattr(o, "res", None) is a.out or getattr(o, "att", None) is a.out)]
                if len(readers) != 1 or not isinstance(readers[0], type(a)) or readers[0].x is not a.out:
                    continue
                b = readers[0]
                pb_ = b.layer.prefix
                if not march3(b.layer) or b.layer.cin != a.layer.cout or pb_ in self.resn or pb_ in self.merged or pb_ in self.absorbs or b.out.base is not None or a.out in (prog.logits, *prog.att_maps):
                    continue
                if unit:  # 16 -> 32 -> 32 + the residual convolution's output (which then never exists): the residual tensor's only reader must be the second sub-unit
                    rcv = self.resn[pa_]
                    ok = (isinstance(b, ConvBnAct) and b.layer.cout == 32 and b.res is rcv.out
                          and sum(1 for o in ops if getattr(o, "res", None) is rcv.out or getattr(o, "x", None) is rcv.out) == 1)
                    compact, cin = False, 16
                elif isinstance(a, ConvBnAct):  # the first ResidualUnit: compact network input, the unit's residual convolution already folded into the second epilogue (eval_in1)
                    ok = a.x.root.name == prog.input.name and a.x.real == 1 and eng.compact_c1 and b.layer.cout == 16 and pb_ in self.eval_in1 and b.res is not None
                    compact, cin = True, 8
                else:  # the attention block: conv + ReLU -> conv + sigmoid -> the fp32 attention map
                    ok = a.act == "relu" and b.act == "sigmoid" and b.layer.cout == 1 and b.res is None and b.out.kind == "f32" and a.layer.cin == 32 and (a.x.parts is None or a.x.parts[0].c % 16 == 0)
                    compact, cin = False, a.layer.cin
                plan_c = P.chain_plan(cin, compact, self.lv[a.layer.level], self.n, a.layer.cout) if ok else None
                if plan_c is None:
                    continue
                self.chain_first[pa_] = (a, b, plan_c)
                self.chain_second.add(pb_)

        def chain_launch(a, b, plan_c):
            La, Lb = a.layer, b.layer
            q = self.lv[La.level]
            d = L.ChainDesc()
            bn = isinstance(a, ConvBnAct)
            rcv = self.resn.get(La.prefix)
            compact = bn and rcv is None
            d.inp, d.out, d.cmid = (self._xdesc(a.x, True) if compact else self._desc(a.x)), self._desc(b.out), La.cout
            d.bias_a, d.bias_b = self._pp(La.bkey), self._pp(Lb.bkey)
            if bn:
                for Lr in (La, Lb):
                    pre = Lr.prefix
                    self.fwd_pre.append([lib.vsseg_bn_fold_eval, [self._pp(pre + ".norm.weight"), self._pp(pre + ".norm.bias"), self._bp(pre + ".norm.running_mean"), self._bp(pre + ".norm.running_var"), BN_EPS,
                                                                  vptr(2, pre), vptr(3, pre), Lr.cout]])
                d.scale_a, d.shift_a, d.alpha_a, d.act_a = vptr(2, La.prefix), vptr(3, La.prefix), self._pp(La.prefix + ".act.weight"), L.ACT_PRELU
                d.scale_b, d.shift_b, d.alpha_b, d.act_b = vptr(2, Lb.prefix), vptr(3, Lb.prefix), self._pp(Lb.prefix + ".act.weight"), L.ACT_PRELU
                if compact:
                    pr1 = self.eval_in1[Lb.prefix]
                    d.in1_w, d.in1_b = self._pp(pr1.layer.wkey), self._pp(pr1.layer.bkey)
                else:
                    d.res_tiles, d.bias_res = 2, self._pp(rcv.layer.bkey)
            else:
                d.act_a, d.act_b = L.ACT_RELU, L.ACT_SIGMOID
            d.tz, d.mtw, d.lx, d.waves, d.lead = plan_c["tz"], plan_c["mtw"], plan_c["lx"], plan_c["waves"], plan_c["lead"]
            for attr, Lr, kc in (("wpack_a", La, 8 if compact else La.cin), ("wpack_b", Lb, La.cout)):
                pl = P.chain_pack_plan(tuple(Lr.wshape), q, eng.es, kc, self.n)
                assert pl is not None, "no marching pack layout for a chained convolution"
                ch = _Choice([pl], eng.layout.param_off[Lr.wkey][0], wshape=tuple(Lr.wshape))
                if attr == "wpack_a" and rcv is not None:  # + the residual convolution's centre-tap tiles behind the first convolution's weights
                    pl = dataclasses.replace(pl, res_tiles=2)
                    pl.pack_map_res = P.residual_tile_pack_map(kc, 2, tuple(rcv.layer.wshape))
                    ch.woff_res = eng.layout.param_off[rcv.layer.wkey][0]
                self._register(ch, pl)
                self._chain_fixups.append((d, attr, ch.map_off))
                if pl.res_tiles:
                    self._chain_fixups.append((d, "wpack_res", ch.map_off_res))
            dummy = torch.zeros(16, dtype=eng.tdtype, device=dev)
            d.wpack_a = d.wpack_b = dummy.data_ptr()
            d.wpack_res = dummy.data_ptr() if rcv is not None else None  # (placeholders for the domain check; the real pointers are set once the packed-weight buffer exists)
            if lib.vsseg_conv_chain_lds_bytes(C.byref(d)) < 0:
                raise RuntimeError("vsseg_conv_chain rejected a launch the planner offered: " + lib.vsseg_last_error().decode())
            self.keep.append(d)
            nvox = float(self.n) * q[0] * q[1] * q[2]
            cin_r = a.x.real
            F.append([lib.vsseg_conv_chain, [C.byref(d)], dict(tag=f"chain q={q} K={cin_r}x9 -> {La.cout}x9 -> N={Lb.cout}{'+res' if rcv is not None else ''} tz={plan_c['tz']} waves={plan_c['waves']} mtw={plan_c['mtw']} lead={plan_c['lead']} lx={plan_c['lx']}",
                                                               name=f"chain<bf16,{8 if compact else La.cin}>", kind="mfma", flops=2.0 * nvox * (9 * (cin_r * La.cout + La.cout * Lb.cout) + (cin_r * Lb.cout if rcv is not None else 0)),
                                                               bytes=nvox * (cin_r * eng.es + Lb.cout * (4 if b.out.kind == "f32" else eng.es) + (La.cout * eng.es if (self.train and not bn) else 0)))])

        # ---- forward
        F = self.fwd
        grad_alias: Dict[str, TensorSpec] = {}  # residual-conv output -> the tensor it is added into (shares its gradient)
        for op in ops:
            if isinstance(op, (ConvBnAct, ConvPlain)) and op.layer.prefix in self.chain_second:
                continue
            if isinstance(op, (ConvBnAct, ConvPlain)) and op.layer.prefix in self.chain_first:
                chain_launch(*self.chain_first[op.layer.prefix])
                continue
            if isinstance(op, ConvBnAct):
                Lr, cp, pre = op.layer, self.cplans[op.layer.prefix], op.layer.prefix
                glu = self.gate_onload.get(op.x.name)  # the attention gate in front of the unit is applied on load: read x (the concat) and the attention map
                cc1 = self._compact_choice(cp.fwd[0], Lr) if (len(cp.fwd) == 1 and op.x.root.name == prog.input.name and pre not in self.resn) else None
                if cc1 is not None:  # the network input as a compact one-channel tensor (marching plans only)
                    cp.fwd[0] = cc1
                xin = self._desc(glu.x) if glu is not None else self._xdesc(op.x, cp.fold_fwd or cc1 is not None)
                out = self._desc(op.out)
                fused_res = plain_by_out[op.res.name] if (op.res is not None and op.res.name in plain_by_out and plain_by_out[op.res.name].layer.prefix in res1_fused) else None
                res = self._desc(op.res) if (op.res is not None and fused_res is None) else None
                gam, bet, alp = self._pp(pre + ".norm.weight"), self._pp(pre + ".norm.bias"), self._pp(pre + ".act.weight")
                rm, rv = self._bp(pre + ".norm.running_mean"), self._bp(pre + ".norm.running_var")
                rcv = self.resn.get(pre)  # the unit's 1x1x1 residual convolution rides along (same input): stored to its own tensor, or (eval, single-subunit units) added in the epilogue
                rkw = dict(bias_res=self._pp(rcv.layer.bkey)) if rcv is not None else {}
                if glu is not None:
                    assert rcv is not None
                    rkw["in_gate"] = self._alloc(glu.att, self.bufs).data_ptr()
                if rcv is not None and (self.train or op.res is not rcv.out):
                    rkw["res_out"] = self._desc(rcv.out)
                if self.train:
                    yd = self._tdesc(self._raw("y:" + pre, Lr.out_level, Lr.cout), Lr.out_level)
                    self._igemm_classes(F, cp.fwd, xin, yd, bias=self._pp(Lr.bkey), stats=sptr(0, pre), stats_stride=cpad[pre], ncls=len(cp.fwd), **rkw)
                    F.append([lib.vsseg_bn_finalize, [sptr(0, pre), cpad[pre], Lr.cout, float(self._vox(Lr.out_level)), gam, bet, BN_EPS, BN_MOMENTUM, rm, rv,
                                                      self._cp(pre + ".norm.num_batches_tracked"), vptr(0, pre), vptr(1, pre), vptr(2, pre), vptr(3, pre)]])
                    if fused_res is not None:
                        x1 = self._xdesc(fused_res.x, True)  # compact 1-channel copy of the network input
                        F.append([lib.vsseg_bn_act_fwd_res1, [yd, vptr(2, pre), vptr(3, pre), alp, p_drop, SEED, salt[pre], x1.ptr, self._pp(fused_res.layer.wkey), self._pp(fused_res.layer.bkey), out, keep_ptr(Lr)],
                                  self._ew_meta("bn_act_fwd", Lr.out_level, 2 * Lr.cout + 1)])
                    else:
                        F.append([lib.vsseg_bn_act_fwd, [yd, vptr(2, pre), vptr(3, pre), alp, p_drop, SEED, salt[pre], res if res is not None else L.Tensor(), 1 if res is not None else 0, out, keep_ptr(Lr)],
                                  self._ew_meta("bn_act_fwd", Lr.out_level, (3 if res is not None else 2) * Lr.cout)])
                else:
                    self.fwd_pre.append([lib.vsseg_bn_fold_eval, [gam, bet, rm, rv, BN_EPS, vptr(2, pre), vptr(3, pre), Lr.cout]])  # depends on parameters only
                    if rcv is not None and op.res is rcv.out:  # out = act(bn(conv(x))) + residual(x) entirely inside the launch: the residual tensor does not exist
                        res = None
                    pr1 = self.eval_in1.get(pre)
                    if pr1 is not None:  # the first ResidualUnit: its 1 -> C residual convolution of the network input is x1 * w + b in this launch's epilogue (marching plans only)
                        ch0 = cp.fwd[0]
                        cp.fwd[0] = _Choice(self._march_cands(ch0, Lr), ch0.woff, wshape=ch0.wshape)
                        x1 = self._xdesc(pr1.x, True)
                        self._igemm(F, cp.fwd[0], xin, out, bias=self._pp(Lr.bkey), scale=vptr(2, pre), shift=vptr(3, pre), alpha=alp, act=L.ACT_PRELU, res_mode=L.RES_IN1,
                                    in1=(x1.ptr, self._pp(pr1.layer.wkey), self._pp(pr1.layer.bkey)))
                        continue
                    self._igemm_classes(F, cp.fwd, xin, out, bias=self._pp(Lr.bkey), scale=vptr(2, pre), shift=vptr(3, pre), alpha=alp, act=L.ACT_PRELU, res=res,
                                        res_mode=L.RES_ADD if res is not None else L.RES_NONE, ncls=len(cp.fwd), **rkw)
            elif isinstance(op, ConvPlain):
                Lr, cp = op.layer, self.cplans[op.layer.prefix]
                if Lr.prefix in self.merged or Lr.prefix in res1_fused or Lr.prefix in self.resn_of:  # computed inside the convolution / elementwise kernel it is added to
                    continue
                absorbed = self.absorbs.get(Lr.prefix)
                gl = self.gate_onload.get(op.x.name)
                if gl is not None:  # the attention gate in front of this convolution is applied on load: read x and the attention map, marching plans only
                    ch0 = cp.fwd[0]
                    gch = _Choice(self._march_cands(ch0, Lr), ch0.woff, ch0.wshape2, ch0.woff2, wshape=ch0.wshape)
                    cp.fwd[0] = gch
                    self._igemm(F, gch, self._desc(gl.x), self._desc(op.out), bias=self._pp(Lr.bkey), bias2=self._pp(absorbed.layer.bkey) if absorbed is not None else 0, act=ACT_CODE[op.act],
                                in_gate=self._alloc(gl.att, self.bufs).data_ptr())
                    continue
                # the C -> 1 stride-1 3x3x1 convolution that closes an attention block of the two finest levels: a bandwidth kernel on the vector ALUs (csrc/nconv.hip: every
                # input voxel read once, partial sums exchanged between neighbouring threads) instead of an MFMA launch with one real output channel
                yext, zext = self.lv[Lr.level][1], self.lv[Lr.level][2]
                if (eng.narrow_fwd and eng.es == 2 and not eng.dry_run and Lr.cout == 1 and Lr.cin in (16, 32) and Lr.kernel == (3, 3, 1) and tuple(Lr.stride) == (1, 1, 1) and not Lr.transposed
                        and absorbed is None and op.res is None and op.x.parts is None and op.x.base is None and op.out.base is None and op.act in ("none", "sigmoid")
                        and yext in (16, 32, 64, 128, 256) and zext % (512 // yext) == 0):
                    xd, od = self._desc(op.x), self._desc(op.out)
                    if od.c == 1 and od.pitch == 1 and not xd.ptr2:
                        nzb = zext // (512 // yext)
                        nxs = max(1, round(256 / (self.n * nzb)))  # about one 512-thread workgroup per CU (measured best at batch 1 and 4 on both levels: tools/bench_nconv.py)
                        lx = -(-self.lv[Lr.level][0] // nxs)
                        nq = float(self._vox(Lr.level))
                        F.append([lib.vsseg_conv_to1, [xd, self._pp(Lr.wkey), self._pp(Lr.bkey), ACT_CODE[op.act], od, lx],
                                  dict(name="nconv<bf16>", kind="hbm", flops=2.0 * nq * 9 * Lr.cin, bytes=nq * (2.0 * Lr.cin + (4.0 if od.dtype == L.F32 else 2.0)), tag=f"{Lr.prefix[-40:]} {Lr.cin}->1 k={Lr.kernel} lx={lx}")])
                        continue
                xin, out = self._xdesc(op.x, cp.fold_fwd), self._desc(op.out)
                res = self._desc(op.res) if (op.res is not None and absorbed is None) else None
                for ch in cp.fwd:
                    self._igemm(F, ch, xin, out, bias=self._pp(Lr.bkey), bias2=self._pp(absorbed.layer.bkey) if absorbed is not None else 0, act=ACT_CODE[op.act], res=res,
                                res_mode=L.RES_ADD if res is not None else L.RES_NONE)
            elif isinstance(op, AttGate):
                if op.out.name in self.gate_onload:
                    continue
                F.append([lib.vsseg_att_apply_fwd, [self._desc(op.x), self._alloc(op.att, self.bufs).data_ptr(), self._desc(op.out)], self._ew_meta("att_apply_fwd", op.x.level, 2 * op.x.c + 2)])
            if isinstance(op, (ConvBnAct, ConvPlain)) and op.res is not None and op.res.name.endswith(":res"):
                grad_alias[op.res.name] = op.out
        self.out_logits = self._alloc(prog.logits, self.bufs)
        self.out_atts = [self._alloc(a, self.bufs) for a in prog.att_maps]
        if not self.train:
            self._finish_pack()
            return

        # ---- backward
        B = self.bwd
        written: Dict[str, bool] = {}
        x0, y0, z0 = self.lv[0]
        staged: Dict[str, L.Tensor] = {}

        def logits_grad(compact: bool) -> L.Tensor:
            """The loss' fp32 gradient of the logits staged in the compute dtype by the eager prelude: zero-extended to one 8-channel group, or (the marching kernels'
            compact operands, csrc/mconv.hip CC = 2 / csrc/mwgrad.hip PC2) as it is, two channels per voxel."""
            assert not staged or ("c" in staged) == compact, "the gradient of the logits is staged in one layout"
            if not staged:
                if compact:
                    buf = self._raw("g:logits2", 0, prog.logits.c)
                    staged["c"] = self._tdesc(buf, 0)
                    self.glogits_dst = staged["c"]
                else:
                    buf = self._raw("g:logits8", 0, 8)  # channels 2..7 stay zero
                    staged["p"] = self._tdesc(buf, 0)
                    self.glogits_dst = L.Tensor(buf.data_ptr(), _tdtype(buf), prog.logits.c, 8, self.n, x0, y0, z0, None, 0, L.ZERO_PADDED)
                self.bwd_pre.append([lib.vsseg_copy_cast, [_Slot("glogits"), self.glogits_dst]])
            return staged["c" if compact else "p"]

        def gdesc(spec: TensorSpec) -> L.Tensor:
            return self._desc(spec, self.grads)

        def contribution(spec: TensorSpec) -> int:
            """accumulate flag for adding a gradient contribution into g[spec] (first full-width contribution overwrites)."""
            if spec.parts is not None:  # both operands of a concat receive their first contribution together
                flags = [contribution(p) for p in spec.parts]
                assert flags[0] == flags[1], f"operands of {spec.name} have inconsistent gradient state"
                return flags[0]
            root = spec.root.name
            if written.get(root):
                return 1
            written[root] = True
            if spec.c != spec.root.c:  # first contribution covers only a channel slice: start from zero
                gbuf = self._alloc(spec, self.grads)
                B.append([lib.vsseg_memset_zero, [gbuf.data_ptr(), gbuf.numel() * gbuf.element_size()]])
                return 1
            return 0

        def grad_of_out(t: TensorSpec) -> L.Tensor:
            t = grad_alias.get(t.name, t)
            if t.kind == "f32":  # logits: the loss' fp32 gradient staged in an 8-channel compute-dtype buffer
                assert t.name == prog.logits.name
                return logits_grad(False)
            assert written.get(t.root.name), f"gradient of {t.name} is consumed before it is produced"
            return gdesc(t)

        def narrow_wgrad(Lr: Layer, x: TensorSpec, dy: L.Tensor, dy_compact: Optional[L.Tensor], bias_grad: bool = False) -> bool:
            """One input or one output channel, 3x3x1 / 1x1x1, stride 1: the weight gradient is a bandwidth reduction over the C-channel
            operand (vsseg_wgrad_narrow) instead of an MFMA launch on a zero-extended one (narrow-channel tails off the matrix cores)."""
            if not eng.narrow_wgrad or Lr.transposed or tuple(Lr.stride) != (1, 1, 1) or Lr.kernel not in ((3, 3, 1), (1, 1, 1)) or self.lv[Lr.level][1] % 4:
                return False
            k3 = Lr.kernel[0]
            scr = self.eng.wgrad_scratch()  # partial-sum slabs (shared with vsseg_wgrad: the weight-gradient launches serialise on one stream)
            if Lr.cin == 1 and Lr.cout in (8, 16, 32, 64) and x.root.name == prog.input.name and dy.c == Lr.cout and not dy.ptr2:
                x1 = self._xdesc(x, True)  # the compact one-channel copy of the network input
                assert not bias_grad, "the 1 -> C narrow weight gradient does not reduce a bias gradient (those convolutions sit in front of a BatchNorm)"
                B.append([lib.vsseg_wgrad_narrow, [dy, x1.ptr, k3, 1, self._gp(Lr.wkey), k3 * k3, None, scr.data_ptr(), scr.numel()],
                          dict(name="wgrad_narrow", kind="hbm", side=True, late_ok=(k3 == 1), flops=0.0, bytes=float(self.eng.es) * self._vox(Lr.level) * (Lr.cout + 1), tag=f"{Lr.prefix[-40:]} 1->{Lr.cout} k={Lr.kernel}")])
                return True
            if Lr.cout == 1 and Lr.cin in (8, 16, 32, 64) and dy_compact is not None and x.parts is None and x.base is None:
                # (the bias gradient of the C -> 1 convolution = sum of its one-channel dY rides in the same slabs: fixed summation order)
                B.append([lib.vsseg_wgrad_narrow, [self._desc(x), dy_compact.ptr, k3, -1, self._gp(Lr.wkey), k3 * k3, self._gp(Lr.bkey) if bias_grad else None, scr.data_ptr(), scr.numel()],
                          dict(name="wgrad_narrow", kind="hbm", side=True, flops=0.0, bytes=float(self.eng.es) * self._vox(Lr.level) * (Lr.cin + 1), tag=f"{Lr.prefix[-40:]} {Lr.cin}->1 k={Lr.kernel}")])
                return True
            return False

        def conv_backward(Lr: Layer, x: TensorSpec, dy: L.Tensor, bias_grad: bool, relumask: Optional[TensorSpec] = None, dy_compact: Optional[L.Tensor] = None, gate=None, own_dy: bool = False):
            cp = self.cplans[Lr.prefix]
            wg = cp.wgrad
            if narrow_wgrad(Lr, x, dy, dy_compact, bias_grad):
                conv_backward_data(Lr, x, dy, relumask, dy_compact, gate)
                return
            gl = self.gate_onload.get(x.name)  # the convolution's input is an attention-gated tensor that was never materialised: H = x, gated on load
            xin = self._desc(gl.x if gl is not None else x)
            d = L.WgradDesc()
            if gl is not None:
                d.h_gate = self._alloc(gl.att, self.bufs).data_ptr()
            pc2 = dy_compact is not None and dy_compact.c == 2  # P = the compact two-channel gradient of the logits (marching kernel only, csrc/mwgrad.hip PC2)
            if Lr.transposed:
                d.p, d.h, d.cp_valid, d.ch_valid = xin, dy, Lr.cin, Lr.cout
            else:
                d.p, d.h, d.cp_valid, d.ch_valid = (dy_compact if pc2 else dy), xin, Lr.cout, Lr.cin
            d.q, d.hs, d.ntaps = L.i3(wg.q), L.i3(wg.hs), len(wg.taps)
            for t, (off, widx) in enumerate(wg.taps):
                d.tap_off[t][0], d.tap_off[t][1], d.tap_off[t][2] = off
                d.tap_widx[t] = widx
            d.tile, d.ntp = L.i3(wg.tile), wg.ntp
            d.dw = self._gp(Lr.wkey)
            if bias_grad and not Lr.transposed:  # bias gradient = sum of dY, reduced inside the weight-gradient kernel (P = dY)
                d.dbias_p = self._gp(Lr.bkey)
            d.stride_p, d.stride_h, d.stride_tap = wg.stride_p, wg.stride_h, wg.stride_tap
            hch = (d.ch_valid + 15) // 16
            tiles = self.n
            for a in range(3):
                tiles *= -(-wg.q[a] // wg.tile[a])
            # few persistent workgroups with many tiles each: the per-workgroup flush is as large as the weight gradient itself
            scr = self.eng.wgrad_scratch()
            d.scratch, d.scratch_elems = scr.data_ptr(), scr.numel()
            self.keep.append(d)
            tuned = ""
            # H-chunk group: one workgroup multiplies the P tile it fetched with `hgroup` 16-channel chunks of H (P is then read
            # ceil(chunks / hgroup) times instead of once per chunk); the library clamps the request to a divisor of the chunk count
            # that fits registers and LDS.  Heuristic: as large as allowed.
            hgs = [g for g in (4, 3, 2, 1) if hch % g == 0]
            d.hgroup = hgs[0]

            def set_blocks(wpc):  # persistent workgroups = wpc per CU over all H-chunk groups (the library clamps to what is resident)
                d.persistent_blocks = max(1, min(tiles, (256 * wpc) // max(1, hch // max(1, d.hgroup))))

            wpc = 4
            # the marching kernel (csrc/mwgrad.hip: both operands fetched once) where it is instantiated: stride-1 3x3x1 bf16, 16/32/64 input channels
            mtiles = []
            if eng.es == 2 and not Lr.transposed and tuple(Lr.stride) == (1, 1, 1) and Lr.kernel == (3, 3, 1) and d.h.c == Lr.cin and (d.p.c in (8, 16, 32) or pc2) and not d.p.ptr2:
                mtiles = P.march_wgrad_tiles(Lr.cin, 8 if pc2 else d.p.c, wg.q, self.n, scr.numel())
            assert mtiles or not pc2
            # the compute kernel (csrc/cwgrad.hip, march = 2) on the MFMA-bound stride-1 3x3x3 layers of levels 2-3: H chunks per workgroup (hgroup) 1 or 2
            cgs = []
            if (eng.compute_wgrad and eng.es == 2 and not Lr.transposed and tuple(Lr.stride) == (1, 1, 1) and Lr.kernel == (3, 3, 3) and gl is None and not pc2 and d.p.c in (48, 64) and d.p.c == Lr.cout
                    and d.h.c == Lr.cin and Lr.cin % 16 == 0 and wg.q[1] % 8 == 0 and wg.q[2] % 32 == 0 and not d.p.ptr2):
                cgs = [g for g in ((2, 1) if d.p.c == 48 else (1,)) if hch % g == 0 and hch // g <= 8]
            live_tile = L.i3(wg.tile)
            if self.tune:  # measured per launch: {double-buffered DMA pipeline | one buffer} x H-chunk group x workgroups per CU, and the marching kernel's tiles
                key = f"wgrad{4 if cgs else 3}|w{tuple(Lr.wshape)}|T{int(Lr.transposed)}|q{wg.q}|n{self.n}|es{self.eng.es}|two{int(bool(d.h.ptr2))}|pc{d.p.c}" + ("|gin" if gl is not None else "") + ("|c2" if pc2 else "")
                cache = _tune_cache()
                if key in cache and os.environ.get("VSSEG_AUTOTUNE", "1") != "force":
                    hit = cache[key]
                    if hit[0] == "m":
                        d.march, live_tile = 1, L.i3(hit[1:4])
                    elif hit[0] == "c":
                        d.march, d.hgroup = 2, int(hit[1])
                    else:
                        d.single_buffer, d.hgroup, wpc = (int(v) for v in hit)
                    tuned = " tuned[cache]"
                else:
                    stream = torch.cuda.current_stream().cuda_stream
                    # measured launches accumulate into a scratch copy of the gradient buffer, never into the live one
                    if getattr(self, "_tune_gflat", None) is None:
                        self._tune_gflat = torch.zeros_like(self.eng.gflat)
                    delta = self._tune_gflat.data_ptr() - self.eng.gflat.data_ptr()
                    live_dw, live_db = d.dw, d.dbias_p
                    d.dw = live_dw + delta
                    d.dbias_p = (live_db + delta) if live_db else None

                    def measure():
                        if lib.vsseg_wgrad(C.byref(d), stream):
                            return float("inf")
                        best = float("inf")
                        for _ in range(self.eng.tune_reps):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            lib.vsseg_wgrad(C.byref(d), stream)
                            e1.record()
                            e1.synchronize()
                            best = min(best, e0.elapsed_time(e1))
                        return best

                    ms = {}
                    for hg in (hgs if (gl is None and not pc2) else []):  # (a gated H operand / a compact P operand: marching kernel only)
                        for sb in (0, 1):
                            for w in (2, 3, 4):
                                d.march, d.single_buffer, d.hgroup = 0, sb, hg
                                set_blocks(w)
                                ms[(sb, hg, w)] = measure()
                    for mt in mtiles:
                        d.march, d.tile = 1, L.i3(mt)
                        ms[("m", *mt)] = measure()
                    for g in cgs:
                        d.march, d.hgroup, d.tile = 2, g, L.i3(wg.tile)
                        set_blocks(4)
                        ms[("c", g)] = measure()
                    d.dw, d.dbias_p, d.tile = live_dw, live_db, L.i3(wg.tile)
                    bestk = min(ms, key=ms.get)
                    if bestk[0] == "m":
                        d.march, live_tile = 1, L.i3(bestk[1:4])
                    elif bestk[0] == "c":
                        d.march, d.hgroup = 2, bestk[1]
                    else:
                        d.march = 0
                        d.single_buffer, d.hgroup, wpc = bestk
                    cache[key] = list(bestk)
                    _tune_cache.dirty = True
                    tuned = f" tuned[best of {len(ms)}: {min(ms.values()):.3f} ms, default {ms.get((0, hgs[0], 4), float('nan')):.3f}]"
            elif cgs:  # untuned lowering: the compute kernel wherever it applies, two chunks per workgroup where they divide
                d.march, d.hgroup = 2, cgs[0]
            elif gl is not None or pc2:  # untuned lowering of a gated H operand / a compact P: the marching kernel is the only one that reads them — its first tile
                assert mtiles, "gate-on-load / a compact P was enabled for a layer without a marching weight-gradient tile"
                d.march, live_tile = 1, L.i3(mtiles[0])
            d.tile = live_tile
            set_blocks(wpc)
            nq = self.n * wg.q[0] * wg.q[1] * wg.q[2]
            B.append([lib.vsseg_wgrad, [C.byref(d)], dict(own_dy=own_dy, tag=f"{Lr.prefix[-40:]} q={wg.q} taps={len(wg.taps)} cin={Lr.cin} cout={Lr.cout} " + (f"compute chunks/wg={d.hgroup}" if d.march == 2 else f"march tile={tuple(d.tile)}" if d.march else f"tile={wg.tile} blocks={d.persistent_blocks}x{hch} sb={d.single_buffer} hg={d.hgroup} lds={wg.lds}") + tuned,
                                                          name=(f"cwgrad<bf16,{wg.ntp}>" if d.march == 2 else f"mwgrad<bf16,{wg.ntp}>" if d.march else f"wgrad<{'bf16' if self.eng.es == 2 else 'f32'},{wg.ntp}>"), kind="mfma", side=True, flops=2.0 * nq * len(wg.taps) * Lr.cin * Lr.cout,
                                                          bytes=float(self.eng.es) * (nq * (Lr.cout if not Lr.transposed else Lr.cin) + self._vox(Lr.level if not Lr.transposed else Lr.out_level) * (Lr.cin if not Lr.transposed else Lr.cout)))])
            if bias_grad and Lr.transposed:  # (does not occur in this network: transposed convolutions are followed by BatchNorm)
                B.append([lib.vsseg_channel_sum, [L.Tensor(dy.ptr, dy.dtype, Lr.cout, dy.pitch, dy.n, dy.x, dy.y, dy.z), self._gp(Lr.bkey)]])
            conv_backward_data(Lr, x, dy, relumask, dy_compact, gate)

        def conv_backward_data(Lr: Layer, x: TensorSpec, dy: L.Tensor, relumask, dy_compact, gate):
            cp = self.cplans[Lr.prefix]
            if cp.dgrad:
                acc = contribution(x)
                gx = gdesc(x)
                if gate is not None:  # d(x) = conv^T(dy) + d(gated) * (1 + att): the attention gate's backward rides in this launch's epilogue
                    assert acc == 0 and relumask is None and len(cp.dgrad) == 1
                    self._igemm(B, cp.dgrad[0], dy, gx, res=gate[0], res_mode=L.RES_GATE, gate=gate[1])
                    return
                if len(cp.dgrad) > 1 and not any(ch.fold for ch in cp.dgrad):
                    self._igemm_classes(B, cp.dgrad, dy, gx, accumulate=acc, res=self._desc(relumask) if relumask is not None else None, res_mode=L.RES_RELUMASK if relumask is not None else L.RES_NONE, ncls=len(cp.dgrad))
                    return
                for ci, ch in enumerate(cp.dgrad):
                    cc1 = self._compact_choice(ch, Lr) if (dy_compact is not None and len(cp.dgrad) == 1) else None
                    if cc1 is not None:  # the one-channel gradient read compact by the marching kernel (its 8-channel zero-extension is then never written)
                        cp.dgrad[ci] = ch = cc1
                    self._igemm(B, ch, dy_compact if (ch.fold or cc1 is not None) else dy, gx, accumulate=acc, res=self._desc(relumask) if relumask is not None else None, res_mode=L.RES_RELUMASK if relumask is not None else L.RES_NONE, ncls=len(cp.dgrad))

        def fused_backward(op: ConvBnAct, yd: L.Tensor, dA: L.Tensor) -> bool:
            """The layer's vsseg_bn_act_bwd_apply + data gradient + weight gradient as ONE marching launch (csrc/mbwd.hip), where it is instantiated and this
            data gradient is the first contribution to d(x) (the launch overwrites): dy is then never written.  Returns False when the layer is not eligible."""
            Lr, pre = op.layer, op.layer.prefix
            x = op.x
            want = eng.fused_bwd
            if (want == "0" or eng.es != 2 or Lr.transposed or tuple(Lr.stride) != (1, 1, 1) or Lr.kernel != (3, 3, 1) or x.parts is not None or x.base is not None or x.kind != "act"
                    or x.root.name == prog.input.name or x.c != Lr.cin or dA.ptr2 or (p_drop > 0.0 and keep_ptr(Lr) is None)):
                assert x.name not in self.gate_onload, "a unit behind an attention gate applied on load must run the fused backward"
                return False
            glx = self.gate_onload.get(x.name)  # x is an attention-gated tensor that was never materialised: the launch reads the concat and the attention map
            if want != "1" and f"{Lr.cin}x{Lr.cout}" not in want.split(","):
                return False
            scr = self.eng.fused_scratch()  # (its own slabs: the launch runs on the main stream, concurrently with the side stream's weight gradients)
            # The ResidualUnit's 1x1x1 residual convolution of the same input (ref:params/networks/blocks/convolutions.py:241-255) rides along: its output gradient is
            # the gradient of the tensor it is added into — this block's own dA (single-subunit decoder units) or the unit's output gradient (encoder units, where
            # the add sits behind the second convolution; complete long before this launch).  Its own data- and weight-gradient launches are then skipped.
            rc = next((o for o in ops if isinstance(o, ConvPlain) and o.x is x and o.layer.kernel == (1, 1, 1) and o.layer.prefix.endswith(".residual") and o.layer.cout == Lr.cout
                       and o.layer.prefix not in self.merged and o.act == "none" and o.res is None), None)
            sink = next((o for o in ops if rc is not None and isinstance(o, ConvBnAct) and o.res is rc.out), None)
            dres = None
            if eng.fused_bwd_res and rc is not None and sink is not None and (Lr.cout == 32 or Lr.cin > Lr.cout) and rc.layer.prefix in folded_bias:
                dres = dA if sink is op else grad_of_out(sink.out)
                if dres.ptr2 or dres.dtype != L.BF16 or dres.c != Lr.cout or not P.fused_bwd_tiles(Lr.cout, Lr.cin, self.lv[Lr.level], self.n, scr.numel(), res=True):
                    dres = None
            tiles = P.fused_bwd_tiles(Lr.cout, Lr.cin, self.lv[Lr.level], self.n, scr.numel(), res=dres is not None)
            cls = P.lattice_classes("conv_dgrad", Lr.kernel, Lr.stride)[0]
            mps = P.march_plans("conv_dgrad", Lr.wshape, cls, self.lv[Lr.level], eng.es, Lr.cout, Lr.cin, Lr.cout, self.n)  # (the packed-weight layout of the data gradient)
            if not tiles or not mps or written.get(x.root.name) or (glx is not None and dres is None):
                assert glx is None, "a unit behind an attention gate applied on load must run the fused backward with its residual convolution"
                return False
            assert contribution(x) == 0
            mp = mps[0]
            mp.pack_map = P.pack_map(mp, Lr.wshape)
            ch = _Choice([mp], eng.layout.param_off[Lr.wkey][0], wshape=tuple(Lr.wshape))
            self._register(ch, mp)
            d = L.ConvBwdDesc()
            d.y, d.dout, d.x, d.dx = yd, dA, self._desc(glx.x if glx is not None else x), gdesc(x)
            if glx is not None:
                d.x_gate = self._alloc(glx.att, self.bufs).data_ptr()
            d.mean, d.invstd, d.gamma, d.scale, d.shift, d.alpha = vptr(0, pre), vptr(1, pre), self._pp(pre + ".norm.weight"), vptr(2, pre), vptr(3, pre), self._pp(pre + ".act.weight")
            d.mean_dz, d.mean_dzx, d.p_drop, d.keep = vptr(4, pre), vptr(5, pre), p_drop, keep_ptr(Lr)
            d.dw = self._gp(Lr.wkey)
            d.scratch, d.scratch_elems = scr.data_ptr(), scr.numel()
            self._wpack_fixups.append((d, ch.map_off))
            self.keep.append(d)
            res_tag = ""
            if dres is not None:
                rp = P.residual_dgrad_pack_plan(rc.layer.wshape, self.lv[Lr.level])
                rch = _Choice([rp], eng.layout.param_off[rc.layer.wkey][0], wshape=tuple(rc.layer.wshape))
                self._register(rch, rp)
                d.dres, d.dw_res = dres, self._gp(rc.layer.wkey)
                self._res_fixups.append((d, rch.map_off))
                absorbed_res.add(rc.layer.prefix)
                res_tag = f" +res[{'dA' if sink is op else 'unit dA'}]"
            tile, tuned = tiles[0], ""
            if self.tune and len(tiles) > 1:
                key = f"fbwd|w{tuple(Lr.wshape)}|q{self.lv[Lr.level]}|n{self.n}" + ("|res" if res_tag else "") + ("|xg" if glx is not None else "")
                cache = _tune_cache()
                if key in cache and os.environ.get("VSSEG_AUTOTUNE", "1") != "force":
                    tile, tuned = tuple(cache[key]), " tuned[cache]"
                else:  # measured on the real buffers; the weight gradient accumulates into a scratch copy of the gradient buffer
                    stream = torch.cuda.current_stream().cuda_stream
                    if getattr(self, "_tune_gflat", None) is None:
                        self._tune_gflat = torch.zeros_like(self.eng.gflat)
                    wp = torch.zeros(mp.pack_map.size + 8192, dtype=eng.tdtype, device=eng.device)
                    delta = self._tune_gflat.data_ptr() - self.eng.gflat.data_ptr()
                    d.dw, d.wpack = d.dw + delta, wp.data_ptr()
                    if res_tag:
                        d.dw_res, d.wpack_res = d.dw_res + delta, wp.data_ptr() + eng.es * mp.pack_map.size
                    ms = {}
                    for t in tiles:
                        d.tile = L.i3(t)
                        if lib.vsseg_conv_bwd_fused(C.byref(d), stream):
                            ms[t] = float("inf")
                            continue
                        best = float("inf")
                        for _ in range(self.eng.tune_reps):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            lib.vsseg_conv_bwd_fused(C.byref(d), stream)
                            e1.record()
                            e1.synchronize()
                            best = min(best, e0.elapsed_time(e1))
                        ms[t] = best
                    d.dw = self._gp(Lr.wkey)
                    if res_tag:
                        d.dw_res = self._gp(rc.layer.wkey)
                    tile = min(ms, key=ms.get)
                    cache[key] = list(tile)
                    _tune_cache.dirty = True
                    tuned = f" tuned[best of {len(ms)}: {ms[tile]:.3f} ms]"
            d.tile = L.i3(tile)
            nq = float(self._vox(Lr.level))
            B.append([lib.vsseg_conv_bwd_fused, [C.byref(d)], dict(tag=f"{pre[-40:]} q={self.lv[Lr.level]} cin={Lr.cin} cout={Lr.cout} tile={tuple(tile)}{res_tag}{tuned}", name=f"mbwd<bf16,{Lr.cout // 16},{Lr.cin // 16}>", kind="mfma",
                                                                  flops=2.0 * 2.0 * nq * (10 if res_tag else 9) * Lr.cin * Lr.cout,
                                                                  bytes=eng.es * nq * ((3 if res_tag.endswith("[unit dA]") else 2) * Lr.cout + 2 * Lr.cin) + nq * Lr.cout / 8)])
            return True

        def fused_narrow(op: ConvBnAct, yd: L.Tensor, dA: L.Tensor) -> bool:
            """The 1 -> C 3x3x1 block on the network input (no data gradient): vsseg_bn_act_bwd_apply is applied on load by the narrow weight-gradient reduction."""
            Lr, pre = op.layer, op.layer.prefix
            if (eng.fused_bwd == "0" or eng.es != 2 or not eng.narrow_wgrad or Lr.transposed or tuple(Lr.stride) != (1, 1, 1) or Lr.kernel != (3, 3, 1) or Lr.cin != 1 or Lr.cout not in (8, 16, 32, 64)
                    or op.x.root.name != prog.input.name or self.lv[Lr.level][1] % 4 or dA.ptr2 or dA.c != Lr.cout or (p_drop > 0.0 and keep_ptr(Lr) is None)):
                return False
            x1 = self._xdesc(op.x, True)  # the compact one-channel copy of the network input
            scr = self.eng.wgrad_scratch()
            B.append([lib.vsseg_wgrad_narrow_bn, [yd, dA, keep_ptr(Lr), vptr(0, pre), vptr(1, pre), self._pp(pre + ".norm.weight"), vptr(2, pre), vptr(3, pre), self._pp(pre + ".act.weight"), vptr(4, pre), vptr(5, pre),
                                                   p_drop, x1.ptr, self._gp(Lr.wkey), 9, scr.data_ptr(), scr.numel()],
                      dict(name="wgrad_narrow", kind="hbm", side=True, flops=0.0, bytes=float(self.eng.es) * self._vox(Lr.level) * (2 * Lr.cout + 1) + self._vox(Lr.level) * Lr.cout / 8, tag=f"{pre[-40:]} 1->{Lr.cout} k={Lr.kernel} +bn on load")])
            return True

        absorbed_res = set()  # residual convolutions whose data / weight gradient a fused launch produces
        early_res = set()  # residual convolutions whose backward was issued with the block their output is added to
        gate_fuse: Dict[str, tuple] = {}  # relu-conv prefix -> (d(gated) descriptor, attention map pointer) of the gate fused into its data gradient
        relu_out = {op.out.name: op.out for op in ops if isinstance(op, ConvPlain) and op.act == "relu"}
        producer = {op.out.name: op for op in ops if isinstance(op, ConvPlain)}  # residual convs / attention convs by output tensor
        folded_bias = set()  # plain convolutions whose bias gradient is produced by another kernel's reduction
        for op in reversed(ops):
            if isinstance(op, ConvBnAct):
                Lr, pre = op.layer, op.layer.prefix
                yd = self._tdesc(self.bufs["y:" + pre], Lr.out_level)
                dA = grad_of_out(op.out)
                gam, bet, alp = self._pp(pre + ".norm.weight"), self._pp(pre + ".norm.bias"), self._pp(pre + ".act.weight")
                B.append([lib.vsseg_bn_act_bwd_reduce, [yd, dA, vptr(0, pre), vptr(1, pre), gam, bet, vptr(2, pre), vptr(3, pre), alp, p_drop, SEED, salt[pre], sptr(1, pre), cpad[pre], aptr(pre), keep_ptr(Lr)],
                          self._ew_meta("bn_act_bwd_reduce", Lr.out_level, 2 * Lr.cout)])
                dres_bias = None
                if op.res is not None and op.res.name in producer:  # residual conv: d(out)/d(res) = 1, its bias gradient is sum(dA) (reduced above)
                    dres_bias = self._gp(producer[op.res.name].layer.bkey)
                    folded_bias.add(producer[op.res.name].layer.prefix)
                B.append([lib.vsseg_bn_act_bwd_finalize, [sptr(1, pre), cpad[pre], aptr(pre), Lr.cout, float(self._vox(Lr.out_level)), self._gp(pre + ".norm.weight"), self._gp(pre + ".norm.bias"),
                                                          self._gp(pre + ".act.weight"), vptr(4, pre), vptr(5, pre), dres_bias]])
                if op.res is not None and op.res.name in producer and eng.early_res_wgrad == "1":
                    # The 1x1x1 residual convolution of the network input (first ResidualUnit): its weight gradient needs d(out) of the unit only — which exists from here on — but in list
                    # order it came LAST, behind the unit's first block, where nothing else is left to run beside it (0.13 ms at the end of the step with the main stream idle,
                    # profiles/r06_loss_phase.txt's run).  Issued here it shares the GPU with the unit's own backward instead.
                    pr = producer[op.res.name]
                    if pr.layer.cin == 1 and pr.x.root.name == prog.input.name and pr.layer.prefix not in self.merged and pr.act != "sigmoid":
                        conv_backward(pr.layer, pr.x, dA, bias_grad=pr.layer.prefix not in folded_bias)
                        early_res.add(pr.layer.prefix)
                fused = (op.res is None or op.res.name.endswith(":res")) and (fused_backward(op, yd, dA) or fused_narrow(op, yd, dA))  # (an identity residual re-uses dA's buffer below: keep those unfused)
                if not fused:
                    dyd = self._tdesc(self._raw("dy:" + pre, Lr.out_level, Lr.cout), Lr.out_level)
                    B.append([lib.vsseg_bn_act_bwd_apply, [yd, dA, vptr(0, pre), vptr(1, pre), gam, bet, vptr(2, pre), vptr(3, pre), alp, p_drop, SEED, salt[pre], vptr(4, pre), vptr(5, pre), dyd, keep_ptr(Lr)],
                              self._ew_meta("bn_act_bwd_apply", Lr.out_level, 3 * Lr.cout)])
                if op.res is not None and not op.res.name.endswith(":res"):  # identity residual: d(res) += d(out)
                    r, o = op.res, grad_alias.get(op.out.name, op.out)
                    if (r.parts is None and r.base is None and o.parts is None and o.base is None and r.kind == o.kind == "act" and (r.level, r.c) == (o.level, o.c)
                            and not written.get(r.name) and r.name not in self.grads and o.name in self.grads):
                        # first contribution to d(res), and d(out) is dead from here on (reduce / apply above were its last readers): d(res) IS the
                        # buffer of d(out) — later contributions accumulate into it in place — instead of a copy of it (one tensor round trip less)
                        self.grads[r.name] = self.grads[o.name]
                        written[r.name] = True
                    elif contribution(op.res):
                        B.append([lib.vsseg_add_inplace, [gdesc(op.res), dA], self._ew_meta("grad_add/copy", Lr.out_level, 3 * Lr.cout)])
                    else:
                        B.append([lib.vsseg_copy_cast, [dA, gdesc(op.res)], self._ew_meta("grad_add/copy", Lr.out_level, 2 * Lr.cout)])
                if not fused:
                    conv_backward(Lr, op.x, dyd, bias_grad=False, own_dy=True)  # a bias in front of a training-mode BatchNorm has zero gradient
            elif isinstance(op, ConvPlain):
                Lr = op.layer
                if Lr.prefix in self.merged:  # gradients of a merged residual conv = centre-tap slice / bias gradient of the absorbing conv
                    big = self.merged[Lr.prefix].layer
                    kx, ky, kz = big.kernel
                    centre = ((kx // 2) * ky + ky // 2) * kz + kz // 2
                    B.append([lib.vsseg_merge_residual_grads, [self._gp(big.wkey), self._gp(big.bkey), self._gp(Lr.wkey), self._gp(Lr.bkey), Lr.cout, Lr.cin, kx * ky * kz, centre], dict(name="vsseg_merge_residual_grads", kind="hbm", flops=0.0, bytes=0.0, side=True)])
                    continue
                if Lr.prefix in early_res:  # issued with the unit's last block (above)
                    continue
                if Lr.prefix in absorbed_res:  # data + weight gradient came out of the fused launch of the unit's 3x3x1 block (csrc/mbwd.hip, RES); bias gradient: folded_bias
                    continue
                assert op.res is None or op.res.name.endswith(":res"), "identity residual on a plain convolution is not part of this network"
                dyc = self._tdesc(self.bufs["dpre1:" + op.out.name], Lr.level) if (op.act == "sigmoid" and ("dpre1:" + op.out.name) in self.bufs) else None
                if op.act != "sigmoid" and op.out.name == prog.logits.name and self._logits_compact(Lr, op.x):
                    dyc = logits_grad(True)  # both launches of the logits convolution read the two-channel gradient compact (4 bytes per voxel instead of 16)
                dy = (self._tdesc(self.bufs["dpre:" + op.out.name], Lr.level) if ("dpre:" + op.out.name) in self.bufs else dyc) if op.act == "sigmoid" else (dyc if dyc is not None else grad_of_out(op.out))
                conv_backward(Lr, op.x, dy, bias_grad=Lr.prefix not in folded_bias, relumask=relu_out.get(op.x.name), dy_compact=dyc, gate=gate_fuse.get(Lr.prefix))
            elif isinstance(op, AttGate):
                gout = grad_of_out(op.out)
                # The gate's d(x) = d(gated) * (1 + att) and the data gradient of the attention branch's first convolution (relu
                # conv on the same x) both flow into d(x): fused, the gate kernel skips its full-width d(x) store and the igemm
                # launch reads d(gated) where it would have read-modify-written d(x) (one tensor pass less per level).
                c1 = next((o for o in ops if isinstance(o, ConvPlain) and o.act == "relu" and o.x is op.x), None)
                xs = op.x.parts if op.x.parts else (op.x,)
                fuse = eng.gate_fuse and c1 is not None and len(self.cplans[c1.layer.prefix].dgrad) == 1 and not any(written.get(t.root.name) for t in xs)
                if fuse:
                    gate_fuse[c1.layer.prefix] = (gout, self._alloc(op.att, self.bufs).data_ptr())
                    acc = 2  # att_apply_bwd: do not write d(x)
                else:
                    acc = contribution(op.x)
                sigop = producer[op.att.name]
                sig = sigop.layer  # the sigmoid convolution (its bias gradient sum(dpre) is reduced by its weight-gradient launch, in a fixed order)
                want_c1 = (eng.narrow_wgrad and sig.kernel in ((3, 3, 1), (1, 1, 1)) and sig.cin in (8, 16, 32, 64) and self.lv[sig.level][1] % 4 == 0)
                dpre1 = self._raw("dpre1:" + op.att.name, op.att.level, 1).data_ptr() if want_c1 else None  # compact copy of d(pre-sigmoid): z-folded data gradient / narrow weight gradient
                # the 8-channel zero-extension of d(pre-sigmoid) (16 bytes per voxel for 2 real ones) is written only if a launch reads it: not when the sigmoid convolution's
                # weight gradient is the narrow reduction and its data gradient reads the compact copy (marching kernel, csrc/mconv.hip C1)
                cps = self.cplans[sig.prefix]
                all_compact = (want_c1 and not sig.transposed and tuple(sig.stride) == (1, 1, 1) and sigop.x.parts is None and sigop.x.base is None and len(cps.dgrad) == 1
                               and self._compact_choice(cps.dgrad[0], sig) is not None)
                dpre = None if all_compact else self._raw("dpre:" + op.att.name, op.att.level, 16 if sig.prefix in self.wide_dpre else 8)  # (channels 8..15 of the wide rows are never written: zero)
                gbuf = self.gatt_buf[op.att.name] = torch.zeros((self.n, *self.lv[op.att.level]), dtype=torch.float32, device=dev)  # the loss' gradient of this attention map is staged here
                B.append([lib.vsseg_att_apply_bwd, [self._desc(op.x), self._alloc(op.att, self.bufs).data_ptr(), gout, gbuf.data_ptr(), gdesc(op.x), acc, self._dpre_desc(dpre, op.att.level) if dpre is not None else L.Tensor(), None, dpre1],
                          self._ew_meta("att_apply_bwd", op.x.level, (2 if acc == 2 else (4 if acc else 3)) * op.x.c + (8 if dpre is not None else 1) + 4)])
        if eng.late_wgrad > 0:
            # The LAST weight-gradient launches of the list leave the side stream for the END of the main stream's list: the step ends with the first block's chain
            # (statistics pass -> finalize -> narrow weight gradient, ~0.7 ms) in which the main stream has nothing left to issue and the side stream runs one VALU-bound
            # launch; weight gradients that would have shared the GPU with the main stream's last convolutions run beside that launch instead.  Only launches whose operands
            # nothing writes again (P = the block's own dy buffer, H = a forward activation); their slabs go to the main stream's scratch (the fused launches are done by then).
            cands = [i for i, rec in enumerate(B) if len(rec) > 2 and rec[2].get("side") and rec[2].get("own_dy") and rec[0] is lib.vsseg_wgrad]
            late = set(cands[-eng.late_wgrad:])
            fs = eng.fused_scratch()
            moved = []
            for i in sorted(late):
                rec = B[i]
                d = rec[1][0]._obj
                d.scratch, d.scratch_elems = fs.data_ptr(), fs.numel()
                rec[2] = dict(rec[2], side=False, late=True)
                moved.append(rec)
            if eng.early_res_wgrad == "late":  # + the first unit's 1x1x1 residual-convolution weight gradient (the last launch of the side stream's list)
                for i, rec in enumerate(B):
                    if len(rec) > 2 and rec[2].get("late_ok") and rec[0] is lib.vsseg_wgrad_narrow and i not in late:
                        rec[1][7], rec[1][8] = fs.data_ptr(), fs.numel()
                        rec[2] = dict(rec[2], side=False, late=True)
                        moved.append(rec)
                        late.add(i)
            B[:] = [r for j, r in enumerate(B) if j not in late] + moved
        self._finish_pack()

    def _index_slots(self):
        self.ext_slots = []
        for lst in (self.fwd, self.bwd, self.bwd_pre):
            for rec in lst:
                for i, a in enumerate(rec[1]):
                    if isinstance(a, _Slot):
                        assert lst is self.bwd_pre, "a captured launch list may not depend on caller-owned memory"
                        self.ext_slots.append((rec[1], i, a))

    # ------------------------------------------------------------------ run
    def set_seed(self, seed: int, stream=None):
        stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        L.check(self.eng.lib.vsseg_store_u64(self.seed_dev.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF, stream), "store_u64")

    def set_external_grads(self, glogits: L.Tensor, gatt: Dict[str, Optional[int]], stream=None):
        """glogits: descriptor of the loss' fp32 gradient of the logits (read by the eager prelude); gatt: device address of the fp32
        gradient of each attention map (or None), copied into the plan-owned buffers the captured backward reads."""
        stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        lib = self.eng.lib
        for args, i, slot in self.ext_slots:
            args[i] = glogits
        for name, buf in self.gatt_buf.items():
            src = gatt.get(name)
            nbytes = buf.numel() * 4
            if src is not None:
                L.check(lib.vsseg_copy_bytes(src, buf.data_ptr(), nbytes, stream), "copy_bytes")
                self._gatt_set[name] = True
            elif self._gatt_set.get(name):  # no external gradient this time: back to zeros
                L.check(lib.vsseg_memset_zero(buf.data_ptr(), nbytes, stream), "memset_zero")
                self._gatt_set[name] = False

    def grad_landing(self):
        """Where a loss that runs between this plan's forward and backward may write its gradients itself (the fused train step, vs_seg_amd.parallel):
        the descriptor the eager prelude stages the gradient of the logits in (compute dtype; compact two channels, or the first two of VSSEG_ZERO_PADDED rows
        of 8) and, per attention map in the network's order, the fp32 buffer the captured backward reads (None: that map has no gradient path)."""
        assert self.train and getattr(self, "glogits_dst", None) is not None, "no backward was lowered for this plan"
        return self.glogits_dst, [self.gatt_buf.get(spec.name) for spec in self.eng.prog.att_maps]

    def grads_landed(self, att_written, stream=None):
        """The loss wrote the staged gradient of the logits and the attention-map buffers named in `att_written` in place: what set_external_grads + the
        eager prelude would have copied is already there.  Buffers of maps without a gradient this time go back to zeros."""
        stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        for name, buf in self.gatt_buf.items():
            if name in att_written:
                self._gatt_set[name] = True
            elif self._gatt_set.get(name):
                L.check(self.eng.lib.vsseg_memset_zero(buf.data_ptr(), buf.numel() * 4, stream), "memset_zero")
                self._gatt_set[name] = False

    def zero_stats(self, stream, row=None):
        """Zero the sharded fp64 statistics (both rows before a training forward, the backward row before a backward)."""
        t = self.stats if row is None else self.stats[row]
        L.check(self.eng.lib.vsseg_memset_zero(t.data_ptr(), t.numel() * t.element_size(), stream), "memset_zero")

    def pack_weights(self, stream):
        m2 = self.pack_map2.data_ptr() if self.pack_map2 is not None else None
        L.check(self.eng.lib.vsseg_gather_cast(self.eng.flat.data_ptr(), self.pack_map.data_ptr(), m2, self.wpack.data_ptr(), self.pack_map.numel(), L.BF16 if self.eng.es == 2 else L.F32, stream), "gather_cast")

    def run(self, lst, stream, graph_key: Optional[str] = None):
        tm = self.timer
        if tm is not None and tm["only"] is None:  # the per-kernel profile: every launch between two events, each kernel alone on the GPU
            return self._run_timed(lst, stream)
        # tm with a set of names (bench.py's timed region: the dominant kernel's launches measured live): the product's schedule — hipGraph replay of the lists that hold none
        # of those launches, bound forks in the eager ones — with two events around the named launches only
        timed_here = tm is not None and tm.get("active", True) and any(_rec_name(r) in tm["only"] for r in lst)
        if graph_key is not None and self.eng.use_graphs and not timed_here and not (self.eng.overlap and any(len(r) > 2 and r[2].get("side") for r in lst)):
            g = self._graphs.get(graph_key)
            if g is not None:
                g.replay()
                return
            runs = self._graph_runs[graph_key] = self._graph_runs.get(graph_key, 0) + 1
            if runs > 2:  # two eager runs first: plan measurement, tile-descriptor tables, function attributes are all settled
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._run_eager(lst, torch.cuda.current_stream().cuda_stream)
                    self._graphs[graph_key] = g
                    g.replay()  # the capture recorded the launches without executing them
                    return
                except RuntimeError as e:  # capture is an optimisation of HOW the same kernels are launched: fall back to the eager loop, loudly
                    import warnings

                    self._graphs.pop(graph_key, None)
                    if isinstance(e, L.VssegError) and not e.during_capture:  # the library rejected a launch (bad arguments, a real launch failure): an error of the step,
                        raise                                                  # not of the capture — only calls HIP refuses BECAUSE the stream is capturing fall back
                    torch.cuda.synchronize()
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError(f"vs_seg_amd: hipGraph capture of the {graph_key} launch list failed ({e}) and left the stream capturing") from e
                    warnings.warn(f"vs_seg_amd: hipGraph capture of the {graph_key} launch list failed ({e}); launching eagerly")
                    self.eng.use_graphs = False
        self._run_eager(lst, stream, tm if timed_here else None)

    def _run_eager(self, lst, stream, tm=None):
        """Launches in list order on `stream` (torch's current stream).  With Engine.overlap, the launches marked side=True — the weight
        gradients, which nothing in the backward pass reads (they feed the optimizer) — go to a second HIP stream: each forks from the main
        stream where the list places it (its operands are final there; every gradient tensor has its own buffer, so nothing later on the
        main stream overwrites them), they serialise among themselves (shared partial-sum scratch), and the main stream joins the side
        stream at the end of the list.  Under hipGraph capture the fork / join events become graph edges."""
        side = None
        overlap = self.eng.overlap
        lib = self.eng.lib
        # A fork = the side stream waits for the main-stream launch in front of it.  With Engine.bound_forks that launch's kernels carry the fork's event as the stop event of
        # their own dispatch (vsseg_fork_arm: no marker packet on the main stream, whose next kernel a hipEventRecord delays by 5-8 us, 42 times per step: DESIGN 3.18)
        bound = overlap and self.eng.bound_forks and not torch.cuda.is_current_stream_capturing()
        is_side = [overlap and len(rec) > 2 and bool(rec[2].get("side")) for rec in lst]
        events = self._fork_events if bound else None
        nfork, ready = 0, None  # ready: the event the launch in front of the next side launch carried
        for i, rec in enumerate(lst):
            if is_side[i]:
                if side is None:
                    side = self.eng.side_stream()
                if ready is not None:
                    L.check(lib.vsseg_stream_wait_event(side.cuda_stream, ready), "stream_wait_event")
                elif i == 0 or not is_side[i - 1]:  # (consecutive side launches share the fork of the first)
                    side.wait_stream(torch.cuda.current_stream())
                ready = None
                timed = tm is not None and _rec_name(rec) in tm["only"]
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                rc = rec[0](*rec[1], side.cuda_stream)
                if timed:
                    e1.record(side)
                    tm["events"].append((_rec_name(rec), rec[2], e0, e1))
            else:
                timed = tm is not None and _rec_name(rec) in tm["only"]
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                arm = bound and i + 1 < len(lst) and is_side[i + 1]
                if arm:
                    if nfork == len(events):
                        events.append(lib.vsseg_fork_event_create())
                    ev = events[nfork]
                    nfork += 1
                    L.check(lib.vsseg_fork_arm(ev), "fork_arm")
                try:
                    rc = rec[0](*rec[1], stream)
                finally:
                    if arm:
                        ready = ev if lib.vsseg_fork_disarm() > 0 else None  # 0: the record launched no kernel of the library (a memset): plain fork
                if timed:
                    e1.record()
                    tm["events"].append((_rec_name(rec), rec[2] if len(rec) > 2 else None, e0, e1))
            if rc:
                L.check(rc, getattr(rec[0], "__name__", "launch"))
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def _run_timed(self, lst, stream):
        """Per-launch HIP-event timing (events are recorded on the stream the kernels are launched on).  With Engine.overlap the side
        launches go to the second stream exactly as in _run_eager — unless EVERY launch is timed (tm['only'] is None: the per-kernel profile
        wants each kernel alone on the GPU, not sharing it with a concurrent weight gradient)."""
        tm = self.timer
        overlap = self.eng.overlap and tm["only"] is not None
        side = None
        for rec in lst:
            name = rec[2]["name"] if len(rec) > 2 else getattr(rec[0], "__name__", "memset")
            on_side = overlap and len(rec) > 2 and rec[2].get("side")
            if on_side:
                if side is None:
                    side = self.eng.side_stream()
                side.wait_stream(torch.cuda.current_stream())
            s_obj = side if on_side else None
            s_raw = side.cuda_stream if on_side else stream
            if tm["only"] is not None and name not in tm["only"]:
                rc = rec[0](*rec[1], s_raw)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s_obj) if s_obj is not None else e0.record()
                rc = rec[0](*rec[1], s_raw)
                e1.record(s_obj) if s_obj is not None else e1.record()
                tm["events"].append((name, rec[2] if len(rec) > 2 else None, e0, e1))
            if rc:
                L.check(rc, name)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def memory_bytes(self) -> int:
        tot = sum(t.numel() * t.element_size() for t in self.bufs.values()) + sum(t.numel() * t.element_size() for t in self.grads.values())
        return tot + self.wpack.numel() * self.wpack.element_size() + self.pack_map.numel() * 4


def _rec_name(rec) -> str:
    return rec[2]["name"] if len(rec) > 2 and "name" in rec[2] else getattr(rec[0], "__name__", "memset")


class Engine:
    """Owns the per-signature plans over the model's flat parameter / gradient / buffer storage."""

    def __init__(self, attention: bool, dtype: str, flat: torch.Tensor, gflat: torch.Tensor, bflat: torch.Tensor, cflat: torch.Tensor, layout: ParamLayout, hp=HP, dropout_p: Optional[float] = None, dry_run: bool = False):
        self.lib = L.lib()
        if not flat.is_cuda and not dry_run:  # dry_run: build plans on CPU to check the lowering (tests); nothing can be launched
            raise RuntimeError("vs_seg_amd: parameters are not on a GPU — this engine has no CPU path (move the model with .to('cuda'))")
        self.device = flat.device
        self.dry_run = dry_run
        # Lowering features that were environment switches while they were being measured (rounds 2-5) and are simply how the engine lowers now — round 6 removed the
        # switches whose other arm no test, tool or measurement used any more (VSSEG_ZFOLD, _RES1_FUSE, _CLASS_SPLIT, _FUSE_CLASSES, _KEEPMASK, _NARROW_WGRAD, _GATE_FUSE,
        # _FUSED_BWD, _FUSED_BWD_RES, _COMPACT_C1, _RESN, _MARCH_SHUFFLE, _GATE_ONLOAD, _GATE_ONLOAD_UNITS, _TUNE_TRACE; VSSEG_CHAIN_TRAIN went with its code).  The
        # attributes stay: the lowering consults them together with "is the kernel instantiated for this shape", and the CPU dry run turns some off.
        self.fold = True  # z-folded launch of the attention sigmoid convolutions (planner.FOLD)
        self.res1_fuse = True  # 1-channel residual conv computed inside bn_act_fwd (training)
        self.tune_reps = int(os.environ.get("VSSEG_TUNE_REPS", "5"))  # timed launches per candidate plan (best of)
        self.class_split = True  # all parity classes of the stride-(2,2,2) transitions as one launch of the general kernel (planner.class_split_plans), where that measures faster
        self.fuse_classes = not dry_run  # output-parity classes of the stride-(2,2,1) level transitions as one launch (depth -4)
        self.keepmask = True  # dropout keep-masks stored by the forward (1 bit per element) instead of regenerated twice in backward
        self.compute_wgrad = os.environ.get("VSSEG_COMPUTE_WGRAD", "1") != "0"  # A/B switch of round 6: the compute weight-gradient kernel (csrc/cwgrad.hip) as a candidate for the 3x3x3 layers of levels 2-3
        self.wide_dpre = os.environ.get("VSSEG_WIDE_DPRE", "1") != "0"  # A/B switch of round 6: 16-channel rows for d(pre-sigmoid) of the 3x3x3 sigmoid convolutions (compute-kernel data gradient)
        self.early_res_wgrad = os.environ.get("VSSEG_EARLY_RES_WGRAD", "late")  # round 6, where the first unit's 1x1x1 residual-convolution weight gradient runs: "late" = with the late launches (next line), "1" = with the unit's last block, "0" = last on the side stream
        self.late_wgrad = int(os.environ.get("VSSEG_LATE_WGRAD", "1"))  # the last N tile / marching / compute weight-gradient launches run at the end of the main stream's list (see the end of the backward lowering)
        self.bound_forks = os.environ.get("VSSEG_BOUND_FORKS", "1") != "0"  # round 6: the side stream forks on events bound to the main-stream kernels' own completion (no marker packets); 0: hipEventRecord per fork
        self.transition = os.environ.get("VSSEG_TRANSITION", "1") != "0"  # A/B switch of round 6: the level 2 <-> 3 transition kernel (csrc/tconv.hip, depth -8)
        self.narrow_fwd = os.environ.get("VSSEG_NARROW_FWD", "1") != "0"  # A/B switch of round 6: the C -> 1 attention convolutions of levels 0-1 on the vector ALUs (csrc/nconv.hip)
        self.narrow_wgrad = True  # weight gradients of the 1-channel-input / 1-channel-output convolutions as bandwidth reductions
        self.gate_fuse = True  # attention-gate backward fused into the attention conv's data gradient
        self.fused_bwd = "1"  # BatchNorm-backward apply + data gradient + weight gradient of the stride-1 3x3x1 blocks of levels 0-1 in ONE launch (csrc/mbwd.hip), every instantiated shape
        self.fused_bwd_res = True  # ... with the unit's 1x1x1 residual convolution riding along
        self.chain = os.environ.get("VSSEG_CHAIN", "1")  # inference: pairs of 3x3x1 convolutions as one launch, the tensor between them in LDS (csrc/chain.hip); "0": off, "l0": level 0 only
        self.compact_c1 = not dry_run  # one-real-channel convolution inputs read compact by the marching kernel (csrc/mconv.hip C1)
        self.resn = not dry_run  # forward: the unit's 1x1x1 residual convolution as extra output tiles of its first 3x3x1 convolution
        # the deep-level kernel (csrc/dconv.hip, plans with depth -7) on the small launches of levels 3-5: "1" = a candidate the tuner measures, "0" = off, "force" = every launch
        # it is offered for runs on it (the untuned lowering then too: how the tests send a whole network through it)
        self.deep = os.environ.get("VSSEG_DEEP", "1")
        self.march_shuffle = True  # marching variants of the fused-parity-classes launch of the level-1 -> level-0 transposed convolution as tuner candidates
        self.gate_onload_units = True  # attention gate on load also in front of the level-1 decoder ResidualUnit (residual tiles + fused backward)
        self.gate_onload = True  # attention-gate forward applied on load by the (marching) convolution behind it and its weight gradient
        # weight gradients on a second HIP stream, concurrent with the data-gradient chain: on the deep levels neither chain fills the 256 CUs
        # (145 launches of 20-50 us), together they do: 37.3 -> 36.1 ms per step (tools/time_step.py).  The backward list is then launched
        # eagerly: replayed as ONE hipGraph the two branches ran no faster than serially (measured 37.6 ms)
        self.overlap = os.environ.get("VSSEG_OVERLAP", "1") == "1" and not dry_run
        # (ALL weight gradients go to the side stream: they share one partial-sum scratch and must serialise on one stream; keeping the finest
        # levels on the main stream measured slower anyway, 37.0 / 37.1 / 37.6 / 37.9 ms for side-stream levels >= 0 / 1 / 2 / 3)
        self._side = None
        self.use_graphs = os.environ.get("VSSEG_GRAPHS", "1") != "0" and not dry_run  # replay the launch lists as hipGraphs after two eager runs
        self.attention, self.hp = attention, hp
        self.tdtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[dtype]
        self.es = 2 if dtype == "bf16" else 4
        self.flat, self.gflat, self.bflat, self.cflat, self.layout = flat, gflat, bflat, cflat, layout
        self.prog: Program = build_program(attention, hp)
        self.dropout_p = hp["dropout"] if dropout_p is None else dropout_p
        self.plans: Dict[tuple, Plan] = {}

    def side_stream(self) -> "torch.cuda.Stream":
        if self._side is None:
            from ._streams import side_stream  # the process-wide pool: the sliding-window lanes reuse this stream instead of adding a fifth one (see _streams.py)

            self._side = side_stream(self.device, 0)
        return self._side

    def wgrad_scratch(self) -> torch.Tensor:
        if getattr(self, "_wg_scratch", None) is None:
            self._wg_scratch = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device=self.device)  # 192 MB of partial-sum slabs, shared by all layers
        return self._wg_scratch

    def fused_scratch(self) -> torch.Tensor:
        if getattr(self, "_fb_scratch", None) is None:
            self._fb_scratch = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device=self.device)  # partial-sum slabs of the fused backward launches (main stream)
        return self._fb_scratch

    def plan(self, n, dims, train, slot: int = 0) -> Plan:
        """The lowered launch lists + activation buffers for one (batch, size, mode).  `slot` > 0: a further, independent set of buffers for
        the same signature (eval forwards issued on different HIP streams run concurrently: sliding-window windows, inferers.py)."""
        key = (int(n), tuple(int(d) for d in dims), bool(train), int(slot))
        pl = self.plans.get(key)
        if pl is None:
            for d, m in zip(key[1], self.min_multiple()):
                if d % m:
                    raise ValueError(f"spatial size {key[1]} must be a multiple of {self.min_multiple()} (product of the network strides)")
            pl = self._lower_rank0_first(key) if train else Plan(self, *key[:3])
            self.plans[key] = pl
        return pl

    def _lower_rank0_first(self, key) -> Plan:
        """Data parallel: rank 0 lowers the training plan (measuring whatever the shipped / cached plans do not cover), broadcasts its measured
        choices, and the other ranks lower from those — every rank then runs the same kernels with the same memory footprint, and the step stays
        bit-reproducible across ranks and processes.  Training plans only: every rank creates them at the same step (DataParallelTrainer), which an
        eval plan (a rank without test cases never lowers one) does not guarantee.  Single process: plain lowering.

        COLLECTIVE: under data parallel the first train-mode forward of a (batch, size) signature issues a broadcast, so every rank must reach it (they do: the
        trainer steps in lock step).  If rank 0 fails while lowering (out of memory, a rejected launch) it broadcasts the error instead of its plan choices and every
        rank raises — nobody is left waiting in the broadcast until the RCCL timeout; ranks whose VSSEG_AUTOTUNE mode differs from rank 0's are an error, too."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) or self.dry_run:
            return Plan(self, *key[:3])
        mode = os.environ.get("VSSEG_AUTOTUNE", "1")
        if dist.get_rank() == 0:
            msg = dict(mode=mode, error=None, cache=None)
            pl = None
            try:
                pl = Plan(self, *key[:3])
                if mode != "0":
                    msg["cache"] = dict(_tune_cache())
            except BaseException as e:  # (re-raised below, after the other ranks have been told)
                msg["error"] = f"{type(e).__name__}: {e}"
                dist.broadcast_object_list([msg], src=0)
                raise
            dist.broadcast_object_list([msg], src=0)
            return pl
        box = [None]
        dist.broadcast_object_list(box, src=0)
        msg = box[0]
        if msg["error"] is not None:
            raise RuntimeError(f"vs_seg_amd: rank 0 failed while lowering the training plan {key[:3]}: {msg['error']}")
        if msg["mode"] != mode:
            raise RuntimeError(f"vs_seg_amd: VSSEG_AUTOTUNE differs between ranks (rank 0: {msg['mode']!r}, rank {dist.get_rank()}: {mode!r}): the ranks would run different kernels")
        if msg["cache"] is not None:
            _tune_cache().update(msg["cache"])
        return Plan(self, *key[:3])

    def min_multiple(self):
        m = [1, 1, 1]
        for s in self.hp["strides"]:
            m = [a * b for a, b in zip(m, s)]
        return tuple(m)
