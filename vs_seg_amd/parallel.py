"""Single-node data parallelism over RCCL/xGMI: one process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).

The reference is single-device (`"cuda:0"`, ref:params/VSparams.py:83); the semantics below are the build's:

* training — pure data parallel.  Every rank steps on its own shard of the shuffled index list; ONE collective per step:
  all-reduce(sum) of the flat 3.45 M-element fp32 gradient buffer (13.8 MB, a single bucket — xGMI links are
  point-to-point, so one large transfer per step beats 178 per-tensor ones), folded into a mean by the fused Adam kernel
  (`grad_scale = 1/world`).  BatchNorm uses per-rank batch statistics (PyTorch-DDP default); running statistics and
  checkpoints come from rank 0.
* inference — cases are sharded round-robin over ranks (`shard_indices`), Dice scalars all-gathered at the end; for a
  single volume the windows are sharded (`sharded_sliding_window_inference`): window i -> rank i mod W, window logits
  all-gathered, then every rank blends ALL windows in the reference's window order, which keeps the sequential fp32
  accumulation bit-identical to the single-GPU result (an all-reduce of partial accumulators would not).

All functions work with the gloo backend on CPU tensors too; that is how the world_size-2 tests run without GPUs.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist

from .inferers import window_geometry


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Smoke-testing the multi-rank code path on a single-GPU box: VSSEG_DIST_BACKEND=gloo VSSEG_SHARE_DEVICE=1 lets several ranks
    # share cuda:0 (RCCL refuses two ranks on one device; gloo moves CUDA tensors through the host).  Never used for measurements.
    backend = backend or os.environ.get("VSSEG_DIST_BACKEND") or None
    if os.environ.get("VSSEG_SHARE_DEVICE") == "1" and torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    if (world > 1 or _force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def _force_collectives() -> bool:
    """VSSEG_FORCE_COLLECTIVES=1: a single process still creates its process group and issues every collective (a world-size-1 RCCL
    all-reduce / all-gather / broadcast is a real RCCL call on the device tensor, ordered on the HIP stream like the 8-GPU one).  It is how
    the `nccl` branch, and its stream hand-off with the side-stream weight gradients, is executed on a 1-GPU box (tests/test_gpu_parallel.py)."""
    return os.environ.get("VSSEG_FORCE_COLLECTIVES") == "1"


def _collectives_on() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force_collectives())


def _via_host(x: torch.Tensor) -> bool:
    """gloo (CPU tests, or several ranks sharing one GPU with VSSEG_SHARE_DEVICE=1) is given host tensors: its CUDA/HIP support
    depends on the torch build, a staged copy does not.  RCCL ("nccl") always works on the device tensor directly."""
    return x.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum(x: torch.Tensor, async_op: bool = False):
    if _via_host(x):
        h = x.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        x.copy_(h)
        return None
    return dist.all_reduce(x, op=dist.ReduceOp.SUM, async_op=async_op)


def _broadcast(x: torch.Tensor, src: int):
    if _via_host(x):
        h = x.detach().cpu()
        dist.broadcast(h, src)
        x.copy_(h)
    else:
        dist.broadcast(x, src)


def _all_gather(x: torch.Tensor) -> List[torch.Tensor]:
    if _via_host(x):
        h = x.detach().cpu()
        out = [torch.empty_like(h) for _ in range(dist.get_world_size())]
        dist.all_gather(out, h)
        return [o.to(x.device) for o in out]
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x)
    return out


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def shard_indices(n: int, rank: int | None = None, world: int | None = None) -> List[int]:
    """Round-robin shard of range(n): rank r owns r, r+W, r+2W, ..."""
    rank = get_rank() if rank is None else rank
    world = world_size() if world is None else world
    return list(range(rank, n, world))


def broadcast_parameters(flat: torch.Tensor, src: int = 0):
    if _collectives_on():
        _broadcast(flat, src)


def broadcast_buffers(model, src: int = 0):
    """BatchNorm running statistics / batch counters of rank `src` to every rank (per-rank batch statistics make them drift apart;
    validation and checkpoints must describe ONE model).  No-op in a single process."""
    if _collectives_on():
        model.flat_parameters()
        _broadcast(model._bflat, src)
        _broadcast(model._cflat, src)


def allreduce_gradients(gflat: torch.Tensor, async_op: bool = False):
    """Sum the flat gradient buffer over ranks (single bucket).  The mean is applied by Adam's `grad_scale`."""
    if _collectives_on():
        return _all_reduce_sum(gflat, async_op)
    return None


def allreduce_scalar_mean(x: torch.Tensor) -> torch.Tensor:
    if _collectives_on():
        x = x.clone()
        _all_reduce_sum(x)
        x /= world_size()
    return x


def allreduce_max_float(value: float, device="cpu") -> float:
    """Maximum of one host scalar over the ranks (the max-over-ranks wall time of a timed region); identity in a single process."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def allreduce_sum(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (identity in a single process); returns x."""
    if _collectives_on():
        _all_reduce_sum(x)
    return x


def all_gather_scalars(values: Sequence[float], total: int, device="cpu") -> List[float]:
    """Each rank passes the scores of its `shard_indices(total)`; returns the `total` scores in case order on every rank."""
    W, r = world_size(), get_rank()
    if not _collectives_on():
        return list(values)
    per = (total + W - 1) // W
    buf = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
    buf[: len(values)] = torch.tensor(list(values), dtype=torch.float64, device=device)
    out = _all_gather(buf)
    res = [float("nan")] * total
    for rr in range(W):
        for j, idx in enumerate(shard_indices(total, rr, W)):
            res[idx] = float(out[rr][j])
    return res


def sharded_window_logits(inputs: torch.Tensor, roi_size, predictor: Callable, overlap: float, crop_fn: Callable) -> tuple:
    """Runs this rank's windows (i mod W == rank) through `predictor` and all-gathers every window's logits.

    Returns (windows [(b, start)], logits [n_windows, C, *roi] in reference window order).  `crop_fn(vol, [(b, start)], roi,
    pad_before)` produces the window batch (HIP crop on GPU, a torch slice in the gloo tests).
    """
    W, r = world_size(), get_rank()
    B = inputs.shape[0]
    img = tuple(int(v) for v in inputs.shape[2:])
    roi, padded, pad_before, _, starts = window_geometry(img, roi_size, overlap)
    windows = [(b, s) for b in range(B) for s in starts]
    mine = shard_indices(len(windows), r, W)
    per = (len(windows) + W - 1) // W
    local = None
    for j, wi in enumerate(mine):
        seg = predictor(crop_fn(inputs, [windows[wi]], roi, pad_before))
        if local is None:
            local = torch.zeros((per, *seg.shape[1:]), dtype=torch.float32, device=seg.device)
        local[j] = seg[0]
    if local is None:  # more ranks than windows: still take part in the collective
        probe = predictor(crop_fn(inputs, [windows[0]], roi, pad_before))
        local = torch.zeros((per, *probe.shape[1:]), dtype=torch.float32, device=probe.device)
    local = local.contiguous()
    if not _collectives_on():
        return windows, local[: len(windows)], (roi, padded, pad_before)
    gathered = _all_gather(local)
    out = torch.empty((len(windows), *local.shape[1:]), dtype=torch.float32, device=local.device)
    for rr in range(W):
        idx = shard_indices(len(windows), rr, W)
        if idx:
            out[idx] = gathered[rr][: len(idx)]
    return windows, out, (roi, padded, pad_before)


def sharded_sliding_window_inference(inputs: torch.Tensor, roi_size, predictor: Callable, overlap: float = 0.25, mode: str = "constant") -> torch.Tensor:
    """`sliding_window_inference` with the windows of each volume spread over the ranks (latency mode)."""
    from . import _lib as L
    from .inferers import _as_cl, crop_windows, importance_map

    windows, logits, (roi, padded, pad_before) = sharded_window_logits(inputs, roi_size, predictor, overlap, crop_windows)
    lib = L.lib()
    stream = torch.cuda.current_stream().cuda_stream
    B = inputs.shape[0]
    img = tuple(int(v) for v in inputs.shape[2:])
    seg = _as_cl(logits)
    C = seg.shape[-1]
    imap = importance_map(roi, mode, inputs.device)
    out = torch.zeros((B, *padded, C), dtype=torch.float32, device=inputs.device)
    cnt = torch.zeros((B, *padded), dtype=torch.float32, device=inputs.device)
    per, pvox, ivox = roi[0] * roi[1] * roi[2], padded[0] * padded[1] * padded[2], img[0] * img[1] * img[2]
    for i, (b, s) in enumerate(windows):  # reference window order on every rank
        L.check(lib.vsseg_swi_accumulate(seg.data_ptr() + 4 * i * per * C, imap.data_ptr(), L.i3(roi), L.i3(s), C, out.data_ptr() + 4 * b * pvox * C, cnt.data_ptr() + 4 * b * pvox, L.i3(padded), stream), "swi_accumulate")
    final = torch.empty((B, *img, C), dtype=torch.float32, device=inputs.device)
    for b in range(B):
        L.check(lib.vsseg_swi_finalize(out.data_ptr() + 4 * b * pvox * C, cnt.data_ptr() + 4 * b * pvox, L.i3(padded), L.i3(pad_before), L.i3(img), C, final.data_ptr() + 4 * b * ivox * C, stream), "swi_finalize")
    return final.permute(0, 4, 1, 2, 3)


class DataParallelTrainer:
    """fwd + loss + bwd + gradient all-reduce + fused Adam for one rank (the train step of ref:params/VSparams.py:454-463 under DP)."""

    def __init__(self, model, loss_fn, optimizer):
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.world = world_size()
        flat, _ = model.flat_parameters()
        broadcast_parameters(flat, 0)
        self.fused_mean = hasattr(optimizer, "grad_scale")
        # VSSEG_FUSED_LOSS=0: the autograd route (loss.backward() through fp32 gradient tensors), for A/B measurements
        self.fused_loss = hasattr(model, "train_forward_landing") and hasattr(loss_fn, "forward_backward_into") and os.environ.get("VSSEG_FUSED_LOSS", "1") != "0"
        if self.fused_mean:
            optimizer.grad_scale = 1.0 / self.world  # the fused Adam kernel multiplies the summed gradient by 1/world
        if self.world > 1 and hasattr(model, "decorrelate_dropout"):
            model.decorrelate_dropout(get_rank())  # every rank seeds torch identically (VS_train.py): give each its own keep-masks

    def step(self, inputs, labels, sync: bool = True):
        """Returns the loss tensor (device scalar).  `sync` is accepted for symmetry with the reference loop, which reads
        `loss.item()` every step (ref:params/VSparams.py:463); nothing here forces a host read either way."""
        self.opt.zero_grad()
        # logits and attention maps never leave this function: the loss reads them and its backward has run before the next forward overwrites the plan's buffers, so the
        # forward may hand out views of them instead of clones (7 device copies, 0.15 ms of a 28 ms step at the benchmark shape)
        if getattr(self, "fused_loss", False) and self.model.training:
            # the loss writes its gradients where the network's backward reads them, in the backward's layout and dtype: no fp32 gradient tensors, no copies of the six
            # attention-map gradients, no cast pass over the gradient of the logits (7 launches and 0.7 GB of traffic per step at the benchmark shape)
            outputs, landing = self.model.train_forward_landing(inputs)
            loss, written = self.loss_fn.forward_backward_into(outputs, labels, landing)
            self.model.backward_landed(written)
        else:
            keep = getattr(self.model, "reuse_output_buffers", None)
            if keep is not None:
                self.model.reuse_output_buffers = True
            try:
                outputs = self.model(inputs)
                loss = self.loss_fn(outputs, labels)
                loss.backward()
            finally:
                if keep is not None:
                    self.model.reuse_output_buffers = keep
        _, gflat = self.model.flat_parameters()
        allreduce_gradients(gflat)
        if self.world > 1 and not self.fused_mean:
            gflat.div_(self.world)  # any other optimizer (torch.optim.Adam / SGD ...) must see the mean, not the sum over ranks
        self.opt.step()
        return loss.detach()
