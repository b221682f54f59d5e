"""`UNet2d5_spvPA` — drop-in for ref:params/networks/nets/unet2d5_spvPA.py:24-206, executed by hand-written gfx950 kernels.

Same constructor signature, `forward(x) -> (logits, att_maps)`, `.train()/.eval()`, `.parameters()` and a state_dict with
the reference's 256 keys/shapes (ref call sites: params/VSparams.py:343-374, 451, 458, 474, 508, 526, 549, 556-560), so
`best_metric_model.pth` files interchange.  Parameters are `nn.Parameter` views into ONE flat fp32 buffer (gradients
likewise), which is what the fused Adam kernel and the single-bucket RCCL all-reduce operate on.

No torch compute op runs in forward/backward: tensors are containers, the work is `libvsseg_hip.so`.  There is no CPU
fallback — calling the model with CPU tensors raises.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import Engine, ParamLayout, Plan
from ...graph import HP, state_manifest


class _Node(nn.Module):
    """Name-only container so that state_dict keys match the reference's nn.Sequential nesting."""


def _supported(dimensions, in_channels, out_channels, channels, strides, kernel_sizes, sample_kernel_sizes, num_res_units, norm, act):
    ok = dimensions == 3 and num_res_units == 2 and str(norm).lower() == "batch" and str(act).lower() == "prelu"
    ok = ok and len(channels) == len(kernel_sizes) == len(strides) + 1 == len(sample_kernel_sizes) + 1
    return ok


class UNet2d5_spvPA(nn.Module):
    def __init__(self, dimensions, in_channels, out_channels, channels, strides, kernel_sizes, sample_kernel_sizes, num_res_units=0, act="prelu", norm="instance", dropout=0,
                 attention_module=True, compute_dtype: Optional[str] = None):
        super().__init__()
        assert len(channels) == len(kernel_sizes) == (len(strides)) + 1 == len(sample_kernel_sizes) + 1  # ref :41
        if not _supported(dimensions, in_channels, out_channels, channels, strides, kernel_sizes, sample_kernel_sizes, num_res_units, norm, act):
            raise NotImplementedError("vs_seg_amd accelerates the configuration VSparams instantiates (3-D, batch norm, PReLU, num_res_units=2)")
        self.dimensions, self.in_channels, self.out_channels = dimensions, in_channels, out_channels
        self.channels, self.strides, self.kernel_sizes, self.sample_kernel_sizes = channels, strides, kernel_sizes, sample_kernel_sizes
        self.num_res_units, self.act, self.norm, self.dropout, self.attention_module = num_res_units, act, norm, dropout, attention_module
        self.att_maps: List[torch.Tensor] = []
        self.hp = dict(in_channels=in_channels, out_channels=out_channels, channels=tuple(channels), strides=tuple(tuple(s) for s in strides),
                       kernel_sizes=tuple(tuple(k) for k in kernel_sizes), sample_kernel_sizes=tuple(tuple(k) for k in sample_kernel_sizes), num_res_units=num_res_units,
                       dropout=float(dropout or 0.0))
        # compute dtype of the conv stack: 'bf16' (MFMA bf16, fp32 accumulate; default) or 'fp32' (exact-fp32 MFMA; parity mode)
        self.compute_dtype = compute_dtype or os.environ.get("VSSEG_DTYPE", "bf16")
        assert self.compute_dtype in ("bf16", "fp32")
        self.reuse_output_buffers = False  # True: forward returns views of plan-owned buffers (valid until the next forward)
        self._manifest = state_manifest(attention_module, self.hp)
        self._layout = ParamLayout(self._manifest)
        self._flat = torch.zeros(self._layout.n_param)
        self._bflat = torch.zeros(max(self._layout.n_buf, 1))
        self._cflat = torch.zeros(max(self._layout.n_cnt, 1), dtype=torch.int64)
        self._gflat: Optional[torch.Tensor] = None
        self._engine: Optional[Engine] = None
        self._anchor: Optional[torch.Tensor] = None
        self._step = 0
        self._eval_slots = {}  # HIP stream -> eval plan slot
        self._seed_base = int(torch.initial_seed()) & 0xFFFFFFFF
        self._params: Dict[str, nn.Parameter] = {}
        self._register_state()
        self.reset_parameters()

    # ------------------------------------------------------------------ state
    def _register_state(self):
        for key, shape in self._manifest:
            parts = key.split(".")
            node = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            if key in self._layout.param_off:
                off, _ = self._layout.param_off[key]
                n = max(1, math.prod(shape))
                p = nn.Parameter(self._flat[off : off + n].view(shape))
                p._vsseg_owner = self
                node.register_parameter(parts[-1], p)
                self._params[key] = p
            elif key in self._layout.buf_off:
                off, _ = self._layout.buf_off[key]
                node.register_buffer(parts[-1], self._bflat[off : off + math.prod(shape)].view(shape))
            else:
                node.register_buffer(parts[-1], self._cflat[self._layout.cnt_off[key]])

    @torch.no_grad()
    def reset_parameters(self):
        """torch defaults of the modules the reference builds: Conv/ConvT kaiming-uniform(a=sqrt(5)) + U(±1/sqrt(fan_in)) bias, BN (1, 0, 0, 1), PReLU 0.25."""
        sd = dict(self.named_parameters())
        for key, shape in self._manifest:
            leaf2 = key.rsplit(".", 2)[-2:]
            if key in self._layout.cnt_off or key in self._layout.buf_off:
                continue
            p = sd[key]
            if len(shape) == 5:
                fan_in = shape[1] * math.prod(shape[2:])  # torch uses size(1)*receptive field for both Conv3d and ConvTranspose3d weights
                bound = 1.0 / math.sqrt(fan_in)
                p.uniform_(-bound, bound)
                bkey = key[: -len("weight")] + "bias"
                if bkey in sd:
                    sd[bkey].uniform_(-bound, bound)
            elif leaf2 == ["norm", "weight"]:
                p.fill_(1.0)
            elif leaf2 == ["norm", "bias"]:
                p.zero_()
            elif leaf2 == ["act", "weight"]:
                p.fill_(0.25)
        for key, _ in self._manifest:
            if key.endswith("running_mean"):
                self._buffer(key).zero_()
            elif key.endswith("running_var"):
                self._buffer(key).fill_(1.0)
            elif key.endswith("num_batches_tracked"):
                self._buffer(key).zero_()

    def _buffer(self, key):
        node = self
        parts = key.split(".")
        for name in parts[:-1]:
            node = node._modules[name]
        return node._buffers[parts[-1]]

    def _set_buffer(self, key, t):
        node = self
        parts = key.split(".")
        for name in parts[:-1]:
            node = node._modules[name]
        node._buffers[parts[-1]] = t

    def _ensure_flat(self):
        """(Re)establish the flat storage after `.to(device)` / `.cuda()` replaced the individual tensors."""
        first = next(self.parameters())
        if first is not self._params[self._manifest[0][0]]:  # Module._apply may replace Parameter objects (e.g. cross-device .to())
            self._params = dict(self.named_parameters())
            for p in self._params.values():
                p._vsseg_owner = self
        dev = first.device
        ok = self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            for key, (off, _) in self._layout.param_off.items():
                if self._params[key].data_ptr() != base + 4 * off:
                    ok = False
                    break
        if ok and self._engine is not None:
            return
        if not ok:
            with torch.no_grad():
                flat = torch.zeros(self._layout.n_param, device=dev)
                bflat = torch.zeros(max(self._layout.n_buf, 1), device=dev)
                cflat = torch.zeros(max(self._layout.n_cnt, 1), dtype=torch.int64, device=dev)
                for key, (off, shape) in self._layout.param_off.items():
                    p = self._params[key]
                    n = p.numel()
                    flat[off : off + n].copy_(p.detach().reshape(-1).to(torch.float32))
                    p.data = flat[off : off + n].view(shape)
                    p.grad = None
                for key, (off, shape) in self._layout.buf_off.items():
                    n = math.prod(shape)
                    bflat[off : off + n].copy_(self._buffer(key).reshape(-1))
                    self._set_buffer(key, bflat[off : off + n].view(shape))
                for key, off in self._layout.cnt_off.items():
                    cflat[off] = self._buffer(key).to(dev)
                    self._set_buffer(key, cflat[off])
                self._flat, self._bflat, self._cflat = flat, bflat, cflat
        self._gflat = torch.zeros_like(self._flat)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._engine = Engine(self.attention_module, self.compute_dtype, self._flat, self._gflat, self._bflat, self._cflat, self._layout, self.hp, self.hp["dropout"])

    def flat_parameters(self):
        """(flat fp32 parameter buffer, flat fp32 gradient buffer) — what fused Adam and the DP all-reduce consume."""
        self._ensure_flat()
        return self._flat, self._gflat

    def train(self, mode: bool = True):
        self._mode_epoch = getattr(self, "_mode_epoch", 0) + 1  # entering eval re-validates the cached eval constants (covers writes through `.data`)
        return super().train(mode)

    def decorrelate_dropout(self, rank: int):
        """Data parallel: mix the rank into the Philox seed base so that ranks seeded identically draw different keep-masks."""
        if rank and not getattr(self, "_seed_rank_mixed", False):
            self._seed_base = (self._seed_base + 0x9E3779B1 * int(rank)) & 0xFFFFFFFF
            self._seed_rank_mixed = True

    def invalidate_cache(self):
        """Force eval plans to re-pack weights / re-fold BatchNorm on their next forward (after out-of-band parameter writes)."""
        self._mode_epoch = getattr(self, "_mode_epoch", 0) + 1

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("vs_seg_amd.UNet2d5_spvPA runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input [B,{self.in_channels},X,Y,Z], got {tuple(x.shape)}")
        self._ensure_flat()
        if self.training and torch.is_grad_enabled():
            outs = _UNetFn.apply(self, x, self._anchor)
        else:
            outs = self._run_forward(x, self.training)[1]
        logits, atts = outs[0], list(outs[1:])
        if self.attention_module:  # same list semantics as the reference's forward hooks (ref :101-104, :204-206)
            self.att_maps = atts
        return logits, self.att_maps

    def segmentation_predictor(self):
        """The predictor for `sliding_window_inference` (the reference hands it `lambda x: model(x)[0]`, ref:params/VSparams.py:560), marked `stream_safe`:
        eval forwards of this model issued on different HIP streams use separate plans (activation buffers, packed weights, hipGraph), so the
        inferer may keep two window groups in flight (vs_seg_amd.inferers.sliding_window_inference, `concurrent_groups`).

        ALIASING CONTRACT — this is NOT a drop-in for `lambda x: model(x)[0]` outside an inferer: in eval mode the returned logits are a VIEW of the calling stream's
        plan buffer and the NEXT call on the same HIP stream overwrites them (whatever `model.reuse_output_buffers` says).  A caller that keeps two outputs alive
        (test-time augmentation, ensembling, a custom inferer that batches its blends) must `.clone()` each result — or use `lambda x: model(x)[0]`, which copies.
        The inferers of this package consume a group's logits before they let that stream run its next group."""

        def predictor(x):
            # logits only, as a VIEW of the stream's own plan buffer (no copy of the logits, no copies of the six attention maps the inferer never looks at):
            # valid until the next call on the same HIP stream — the inferer blends a group's segmentation before it lets that stream run its next group
            if self.training or torch.is_grad_enabled():
                return self(x)[0]
            if not x.is_cuda:
                raise RuntimeError("vs_seg_amd.UNet2d5_spvPA runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
            if x.dim() != 5 or x.shape[1] != self.in_channels:  # (the same validation as forward())
                raise ValueError(f"expected input [B,{self.in_channels},X,Y,Z], got {tuple(x.shape)}")
            self._ensure_flat()
            keep, self.reuse_output_buffers = self.reuse_output_buffers, True
            try:
                return self._run_forward(x, False)[1][0]
            finally:
                self.reuse_output_buffers = keep

        predictor.stream_safe = True
        return predictor

    MAX_EVAL_SLOTS = 4  # independent sets of eval activation buffers (one per HIP stream that runs eval forwards)

    def _eval_slot(self, stream: int) -> int:
        slots = self._eval_slots
        if stream not in slots:
            if len(slots) >= self.MAX_EVAL_SLOTS:  # a fifth stream takes over the buffers of the stream seen longest ago, once that one has drained
                torch.cuda.synchronize()
                slots[stream] = slots.pop(next(iter(slots)))
            else:
                slots[stream] = len(slots)
        return slots[stream]

    def _run_forward(self, x: torch.Tensor, train: bool):
        eng = self._engine
        n, _, X, Y, Z = x.shape
        stream = torch.cuda.current_stream().cuda_stream
        # eval forwards on different HIP streams may overlap (the sliding-window inferer issues consecutive windows on two streams):
        # each stream gets its own plan (activation buffers, packed weights, hipGraph); training plans are stream-independent
        slot = 0 if train else self._eval_slot(stream)
        plan: Plan = eng.plan(n, (X, Y, Z), train, slot)
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.to(torch.float32).contiguous()
        if train:
            plan.pack_weights(stream)
            self._step += 1
            plan.step_seed = (self._seed_base << 32) | (self._step & 0xFFFFFFFF)
            plan.set_seed(plan.step_seed, stream)
            plan.generation += 1
            plan.zero_stats(stream)
            plan.bwd_prepared = True  # the dropout seed is stored and the backward's statistics row is zero: the backward of THIS forward need not do either again
            torch.autograd.graph.increment_version(self._bflat)  # bn_finalize updates the running statistics through raw pointers
        else:
            # eval: packed weights and folded BatchNorm constants depend on parameters / buffers only — the 14 windows of a
            # sliding-window volume (ref:params/VSparams.py:568-574) reuse them; refreshed when a tensor version changes
            # (Parameters / buffers re-pointed into the flat storage via `.data` keep their own version counters, so theirs are
            # summed in; the flat tensors' counters are the ones the raw-pointer kernels — Adam, bn_finalize — bump)
            key = (self._flat.data_ptr(), self._flat._version, self._bflat.data_ptr(), self._bflat._version, getattr(self, "_mode_epoch", 0),
                   sum(p._version for p in self._params.values()), sum(b._version for b in self.buffers()))
            if plan.params_key != key:
                plan.pack_weights(stream)
                plan.run(plan.fwd_pre, stream)
                plan.params_key = key
        if plan.needs_padded_input:  # 8-channel zero-extended copy (one MFMA K-group) for the launches that are not z-folded
            inp = plan._desc(eng.prog.input)
            L.check(eng.lib.vsseg_stage_input(xin.data_ptr(), n, L.i3((X, Y, Z)), L.i3((0, 0, 0)), inp, stream), "stage_input")
        if plan.compact_input is not None:  # 1-channel copy in the compute dtype for the z-folded first-layer launches
            L.check(eng.lib.vsseg_stage_input(xin.data_ptr(), n, L.i3((X, Y, Z)), L.i3((0, 0, 0)), plan._tdesc(plan.compact_input, 0), stream), "stage_input")
        plan.run(plan.fwd, stream, "fwd")
        logits = plan.out_logits.permute(0, 4, 1, 2, 3)  # [B,2,X,Y,Z] view of channels-last storage (torch.channels_last_3d strides)
        atts = [a.permute(0, 4, 1, 2, 3) for a in plan.out_atts]
        if not self.reuse_output_buffers:
            logits = logits.clone(memory_format=torch.preserve_format)
            atts = [a.clone(memory_format=torch.preserve_format) for a in atts]
        return plan, (logits, *atts)

    # ---- the fused train step (vs_seg_amd.parallel.DataParallelTrainer): forward, a loss that writes its gradients where the backward reads them, backward ----
    def train_forward_landing(self, x: torch.Tensor):
        """Training forward outside autograd.  Returns ((logits, att_maps), landing): the outputs are VIEWS of the plan's buffers (valid until the next training
        forward of this shape), `landing` = (descriptor of the staged gradient of the logits, [fp32 gradient buffer per attention map]) — see Plan.grad_landing.
        A loss with `forward_backward_into` writes there; `backward_landed()` then runs the backward without the copy / cast passes of the autograd route."""
        if not self.training:
            raise RuntimeError("train_forward_landing: the module is in eval mode")
        if not x.is_cuda:
            raise RuntimeError("vs_seg_amd.UNet2d5_spvPA runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input [B,{self.in_channels},X,Y,Z], got {tuple(x.shape)}")
        self._ensure_flat()
        keep, self.reuse_output_buffers = self.reuse_output_buffers, True
        try:
            with torch.no_grad():
                plan, outs = self._run_forward(x, True)
        finally:
            self.reuse_output_buffers = keep
        self._landed = (plan, plan.generation)
        logits, atts = outs[0], list(outs[1:])
        if self.attention_module:
            self.att_maps = atts
        return (logits, self.att_maps), plan.grad_landing()

    def backward_landed(self, att_written):
        """Backward of the last `train_forward_landing`; `att_written`: indices of the attention maps whose gradient buffer the loss wrote."""
        plan, generation = getattr(self, "_landed", (None, None))
        if plan is None or plan.generation != generation:
            raise RuntimeError("vs_seg_amd.UNet2d5_spvPA.backward_landed: no training forward of train_forward_landing is pending (or a later one overwrote its activations)")
        self._landed = (None, None)
        names = {spec.name for i, spec in enumerate(self._engine.prog.att_maps) if i in set(att_written)}
        self._run_backward(plan, None, (), landed=names)

    def _run_backward(self, plan: Plan, g_logits: Optional[torch.Tensor], g_atts, landed=None):
        eng = self._engine
        stream = torch.cuda.current_stream().cuda_stream
        n, (X, Y, Z) = plan.n, plan.dims
        keep = []
        if landed is not None:  # the loss wrote its gradients where the backward reads them
            plan.grads_landed(landed, stream)
        else:
            if g_logits is None:
                g_logits = torch.zeros((n, self.out_channels, X, Y, Z), device=self._flat.device)
            gl = g_logits.permute(0, 2, 3, 4, 1)
            if gl.dtype != torch.float32 or not gl.is_contiguous():
                gl = gl.to(torch.float32).contiguous()
            keep.append(gl)
            gatt = {}
            for spec, g in zip(eng.prog.att_maps, g_atts):
                if g is None:
                    continue
                g = g.to(torch.float32).contiguous() if (g.dtype != torch.float32 or not g.is_contiguous()) else g
                keep.append(g)
                gatt[spec.name] = g.data_ptr()
            plan.set_external_grads(L.Tensor(gl.data_ptr(), L.F32, self.out_channels, self.out_channels, n, X, Y, Z), gatt, stream)
        if not getattr(plan, "bwd_prepared", False):  # a second backward of the same forward (retain_graph): the first one used the row up
            plan.set_seed(plan.step_seed, stream)
            plan.zero_stats(stream, 1)
        plan.bwd_prepared = False
        lib, nbytes = eng.lib, self._gflat.numel() * 4
        accumulate = any(p.grad is not None for p in self._params.values())
        if accumulate:  # gradient accumulation (no zero_grad between two backwards): this backward's gradients are added to the previous sum
            if getattr(self, "_gprev", None) is None or self._gprev.device != self._gflat.device:
                self._gprev = torch.empty_like(self._gflat)
            L.check(lib.vsseg_copy_bytes(self._gflat.data_ptr(), self._gprev.data_ptr(), nbytes, stream), "copy_bytes")
        L.check(lib.vsseg_memset_zero(self._gflat.data_ptr(), nbytes, stream), "memset_zero")
        if landed is None:
            plan.run(plan.bwd_pre, stream)  # reads the caller's gradient tensor: always eager
        plan.run(plan.bwd, stream, "bwd")
        if accumulate:
            n4 = self._gflat.numel() // 4  # every tensor is padded to 4 elements inside the flat buffer
            flat_desc = lambda t: L.Tensor(t.data_ptr(), L.F32, 4, 4, 1, 1, 1, n4)  # noqa: E731
            L.check(lib.vsseg_add_inplace(flat_desc(self._gflat), flat_desc(self._gprev), stream), "add_inplace")
        for key, (off, shape) in self._layout.param_off.items():
            p = self._params[key]
            if p.grad is None or p.grad.data_ptr() != self._gflat.data_ptr() + 4 * off:
                p.grad = self._gflat[off : off + p.numel()].view(shape)
        del keep

    # debug / parity: the keep-masks the last training forward used, keyed like the oracle expects (layer prefix -> [B,C,X,Y,Z])
    def dropout_masks(self) -> Dict[str, torch.Tensor]:
        eng = self._engine
        plan = next(p for k, p in eng.plans.items() if k[2] and hasattr(p, "step_seed"))
        out = {}
        stream = torch.cuda.current_stream().cuda_stream
        for pre, (Lr, salt) in plan.bn_info.items():
            x, y, z = plan.lv[Lr.out_level]
            m = torch.empty((plan.n, x, y, z, Lr.cout), device=self._flat.device)
            L.check(eng.lib.vsseg_dropout_mask(m.data_ptr(), plan.n * x * y * z, Lr.cout, float(eng.dropout_p), plan.step_seed, salt, stream), "dropout_mask")
            out[pre] = m.permute(0, 4, 1, 2, 3)
        return out


class _UNetFn(torch.autograd.Function):
    """Glue so that `loss.backward()` (ref:params/VSparams.py:461) drives the hand-written backward."""

    @staticmethod
    def forward(ctx, module: UNet2d5_spvPA, x, anchor):
        plan, outs = module._run_forward(x, True)
        ctx.module, ctx.plan, ctx.generation = module, plan, plan.generation
        return outs

    @staticmethod
    def backward(ctx, g_logits, *g_atts):
        if ctx.plan.generation != ctx.generation:
            # activations, BatchNorm statistics and the dropout seed live in the shape-keyed plan: a later training forward of the
            # same shape has overwritten what this graph's backward needs (the reference nn.Module has no such restriction)
            raise RuntimeError("vs_seg_amd.UNet2d5_spvPA: backward of a forward pass whose activations were overwritten by a later training forward of the same "
                               "shape; call backward() before the next forward (two live graphs of one shape are not supported)")
        ctx.module._run_backward(ctx.plan, g_logits, g_atts)
        return None, None, torch.zeros(1, device=ctx.module._flat.device)


Unet2d5_spvPA = unet2d5_spvPA = UNet2d5_spvPA
