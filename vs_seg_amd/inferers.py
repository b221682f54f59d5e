"""`sliding_window_inference` — drop-in for the MONAI 0.4.0 function the reference calls at ref:params/VSparams.py:568-574.

MONAI is pinned (ref:requirements.txt:7) but not vendored, so the algorithm follows its published behaviour (SURVEY.md
App. B): symmetric zero padding up to the ROI, scan interval int(roi*(1-overlap)), clamp-to-edge window starts in
row-major order, Gaussian (sigma = 0.125*roi) or constant importance map, `out += map*seg; count += map`, divide, crop.
The index arithmetic is host-side integer code (bit-exact with the oracle and the C restatement); crop, blend and
normalise are HIP kernels; windows are blended one by one in reference order, so the fp32 result does not depend on
how windows are spread over GPUs (`vs_seg_amd.parallel.sharded_sliding_window_inference`).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


def fall_back_tuple(roi_size, image_size):
    return tuple(int(r) if r and r > 0 else int(s) for r, s in zip(roi_size, image_size))


def window_geometry(image_size: Sequence[int], roi_size: Sequence[int], overlap: float):
    """(roi, padded size, pad_before, scan interval, window starts) — integer arithmetic only."""
    roi = fall_back_tuple(roi_size, image_size)
    padded = tuple(max(int(s), r) for s, r in zip(image_size, roi))
    pad_before = tuple(max(r - int(s), 0) // 2 for s, r in zip(image_size, roi))
    interval = tuple(r if r == p else max(int(r * (1 - overlap)), 1) for r, p in zip(roi, padded))
    per_dim: List[List[int]] = []
    for size, r, iv in zip(padded, roi, interval):
        num = int(math.ceil(float(size) / iv))
        scan = next(d for d in range(num) if d * iv + r >= size) + 1
        per_dim.append([d * iv - max(d * iv + r - size, 0) for d in range(scan)])
    starts = [(a, b, c) for a in per_dim[0] for b in per_dim[1] for c in per_dim[2]]  # first spatial dim slowest
    return roi, padded, pad_before, interval, starts


def gaussian_taps(sigma: float) -> torch.Tensor:
    tail = int(max(float(sigma) * 4.0, 0.5) + 0.5)
    xs = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / abs(sigma)
    return (0.5 * ((t * (xs + 0.5)).erf() - (t * (xs - 0.5)).erf())).clamp(min=0)


_IMAP_CACHE: Dict[tuple, torch.Tensor] = {}


def importance_map(roi: Sequence[int], mode: str, device) -> torch.Tensor:
    key = (tuple(roi), mode, str(device))
    if key not in _IMAP_CACHE:
        if mode == "constant":
            m = torch.ones(tuple(roi), dtype=torch.float32)
        elif mode == "gaussian":
            # separable zero-padded convolutions of a unit impulse at roi//2 == outer product of the (shifted) 1-D taps,
            # multiplied in the same x, y, z order (every output is a single fp32 product chain, so this is exact)
            axes = []
            for r in roi:
                taps = gaussian_taps(r * 0.125)
                tail = (taps.numel() - 1) // 2
                line = torch.zeros(r, dtype=torch.float32)
                c = r // 2
                lo, hi = max(0, c - tail), min(r, c + tail + 1)
                line[lo:hi] = taps[lo - c + tail : hi - c + tail]
                axes.append(line)
            m = (axes[0][:, None, None] * axes[1][None, :, None]) * axes[2][None, None, :]
            m = m / m.max()
            m[m == 0] = m[m != 0].min()
        else:
            raise ValueError(f"unsupported blend mode {mode!r} (constant | gaussian)")
        _IMAP_CACHE[key] = m.contiguous().to(device)
    return _IMAP_CACHE[key]


def _as_cl(seg: torch.Tensor) -> torch.Tensor:
    v = seg.detach().permute(0, 2, 3, 4, 1)
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.to(torch.float32).contiguous()
    return v


def crop_windows(vol: torch.Tensor, b_and_starts, roi, pad_before) -> torch.Tensor:
    """[n,1,rx,ry,rz] fp32 windows of the (virtually zero-padded) volume `vol` [B,1,X,Y,Z] — one HIP crop per window."""
    lib = L.lib()
    stream = torch.cuda.current_stream().cuda_stream
    B, _, X, Y, Z = vol.shape
    out = torch.empty((len(b_and_starts), 1, *roi), dtype=torch.float32, device=vol.device)
    per = roi[0] * roi[1] * roi[2]
    for i, (b, s) in enumerate(b_and_starts):
        dst = L.Tensor(out.data_ptr() + 4 * i * per, L.F32, 1, 1, 1, roi[0], roi[1], roi[2])
        origin = tuple(si - pb for si, pb in zip(s, pad_before))
        L.check(lib.vsseg_stage_input(vol.data_ptr() + 4 * b * X * Y * Z, 1, L.i3((X, Y, Z)), L.i3(origin), dst, stream), "stage_input")
    return out


def crop_all_windows(vol: torch.Tensor, b_and_starts, roi, pad_before) -> torch.Tensor:
    """Every window of the (virtually zero-padded) volumes in ONE launch (`vsseg_crop_flip` with one job per window, no flip):
    [n_windows,1,rx,ry,rz] fp32.  The per-group predictor inputs are then views of this buffer."""
    lib = L.lib()
    stream = torch.cuda.current_stream().cuda_stream
    B, _, X, Y, Z = vol.shape
    n = len(b_and_starts)
    jobs = (L.CropJob * n)()
    for i, (b, s) in enumerate(b_and_starts):
        j = jobs[i]
        j.src, j.sdims, j.origin, j.flip_x = vol.data_ptr() + 4 * b * X * Y * Z, L.i3((X, Y, Z)), L.i3(tuple(si - pb for si, pb in zip(s, pad_before))), 0
    jbuf = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(vol.device)
    out = torch.empty((n, 1, *roi), dtype=torch.float32, device=vol.device)
    L.check(lib.vsseg_crop_flip(jbuf.data_ptr(), n, out.data_ptr(), L.i3(roi), stream), "crop_flip")
    jbuf.record_stream(torch.cuda.current_stream())
    return out


_CNT_CACHE: Dict[tuple, torch.Tensor] = {}  # (device, padded extents, roi, window starts, importance map) -> sum of the window weights per voxel


def _weight_sum(lib, device, padded, roi, starts, imap: torch.Tensor, stream, mode_key=None) -> torch.Tensor:
    """The denominator of the blend, sum of the importance maps of all windows that cover a voxel (MONAI's count_map): a function of the window geometry only, so it is
    accumulated once per geometry — in window order, exactly as the per-volume buffer was — and kept (the four most recent geometries)."""
    key = (str(device), tuple(padded), tuple(roi), tuple(tuple(s) for s in starts), mode_key)
    hit = _CNT_CACHE.get(key)
    cnt = hit[0] if hit is not None else None
    if cnt is None:
        cnt = torch.zeros(tuple(padded), dtype=torch.float32, device=device)
        for s0 in starts:
            L.check(lib.vsseg_swi_accumulate(None, imap.data_ptr(), L.i3(roi), L.i3(s0), 1, None, cnt.data_ptr(), L.i3(padded), stream), "swi_accumulate")
        torch.cuda.current_stream(device).synchronize()  # (once per geometry: later calls may read it from any stream; `stream` IS this device's current stream, see the caller)
        while len(_CNT_CACHE) >= 4:
            _CNT_CACHE.pop(next(iter(_CNT_CACHE)))
        _CNT_CACHE[key] = (cnt, imap)  # keyed by what defines the map (roi, mode, device — as _IMAP_CACHE), and holding the map itself: an address can be reused, a key cannot
    return cnt


def _side_stream(device, i: int) -> "torch.cuda.Stream":
    from ._streams import side_stream  # one pool for the whole process (the training step's weight-gradient stream is its lane 0): see _streams.py

    return side_stream(device, i)


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable, overlap: float = 0.25, mode: str = "constant", padding_mode: str = "constant",
                             cval: float = 0.0, device=None, concurrent_groups: Optional[int] = None) -> torch.Tensor:
    """`concurrent_groups` (not a MONAI argument; 1 = strictly serial, the reference's schedule): consecutive window groups run their predictor on that
    many HIP streams, so the latency-bound deep levels of one window's forward overlap the bandwidth-bound outer levels of the next; the blend
    (`out += map*seg`) stays on the caller's stream in window order, so the result is bit-identical to the serial schedule.

    The default is 1 for an arbitrary `predictor` — a callable that reuses buffers between calls, or returns views of them, would race on two
    streams — and 2 for a predictor that declares itself `stream_safe` (`UNet2d5_spvPA.segmentation_predictor()`: one set of eval activation
    buffers, packed weights and hipGraph per stream, i.e. twice the eval memory and lowering time of the serial schedule).  Three groups (round 5, with the shorter launch
    list): 42.7 against 41.8 volumes/s in an inference-only process, 38.6-41.0 against 40.8-42.9 in a process that has also trained (bench.py); round 6: 44.3-44.8 against 42.1-42.2
    in an inference-only process, 44.3-44.5 against 44.1-44.9 inside bench.py's (four groups: 42.1) — not the default."""
    if concurrent_groups is None:
        concurrent_groups = 2 if getattr(predictor, "stream_safe", False) else 1
    if not inputs.is_cuda:
        raise RuntimeError("vs_seg_amd.sliding_window_inference runs on an MI355X only (got a CPU tensor); there is no CPU fallback")
    if inputs.dim() != 5 or inputs.shape[1] != 1:
        raise ValueError("expected inputs [B,1,X,Y,Z]")
    if padding_mode != "constant" or cval != 0.0:
        raise NotImplementedError("only constant zero padding (the MONAI default the reference relies on)")
    mode = getattr(mode, "value", mode)
    lib = L.lib()
    B = inputs.shape[0]
    img = tuple(int(v) for v in inputs.shape[2:])
    roi, padded, pad_before, _, starts = window_geometry(img, roi_size, overlap)
    vol = inputs.detach()
    if vol.dtype != torch.float32 or not vol.is_contiguous():
        vol = vol.to(torch.float32).contiguous()
    imap = importance_map(roi, mode, inputs.device)
    slices = [(b, s) for b in range(B) for s in starts]
    out = None
    main = torch.cuda.current_stream(inputs.device)  # launch stream and synchronisation point are the SAME stream object: the current stream of the inputs' device
    stream = main.cuda_stream
    cnt = _weight_sum(lib, inputs.device, padded, roi, starts, imap, stream, mode_key=str(mode))  # [padded]: the same for every batch element
    per_win = roi[0] * roi[1] * roi[2] * 4
    windows = crop_all_windows(vol, slices, roi, pad_before) if len(slices) * per_win <= (8 << 30) else None  # one crop launch per call (<= 8 GB of windows), else per group
    ngroups = -(-len(slices) // sw_batch_size)
    lanes = [_side_stream(inputs.device, i) for i in range(min(int(concurrent_groups), ngroups))] if concurrent_groups > 1 and ngroups > 1 else []
    ready = main.record_event() if lanes else None  # the windows (and the volume) are complete on the caller's stream
    blended: Dict[int, "torch.cuda.Event"] = {}  # lane -> its previous group's segmentation has been blended (the predictor may reuse its output buffers)
    for gi, g in enumerate(range(0, len(slices), sw_batch_size)):
        grp = slices[g : g + sw_batch_size]
        if lanes:
            lane = lanes[gi % len(lanes)]
            lane.wait_event(blended.get(gi % len(lanes), ready))
            with torch.cuda.stream(lane):
                win = windows[g : g + len(grp)] if windows is not None else crop_windows(vol, grp, roi, pad_before)
                seg = _as_cl(predictor(win))  # [n,rx,ry,rz,C]
                done = lane.record_event()
            seg.record_stream(main)
            main.wait_event(done)
        else:
            win = windows[g : g + len(grp)] if windows is not None else crop_windows(vol, grp, roi, pad_before)
            seg = _as_cl(predictor(win))  # [n,rx,ry,rz,C]
        C = seg.shape[-1]
        if out is None:
            out = torch.zeros((B, *padded, C), dtype=torch.float32, device=inputs.device)
        per = roi[0] * roi[1] * roi[2]
        pvox = padded[0] * padded[1] * padded[2]
        for i, (b, s) in enumerate(grp):
            L.check(lib.vsseg_swi_accumulate(seg.data_ptr() + 4 * i * per * C, imap.data_ptr(), L.i3(roi), L.i3(s), C, out.data_ptr() + 4 * b * pvox * C, None, L.i3(padded), stream), "swi_accumulate")
        if lanes:
            blended[gi % len(lanes)] = main.record_event()
    C = out.shape[-1]
    final = torch.empty((B, *img, C), dtype=torch.float32, device=inputs.device)
    pvox, ivox = padded[0] * padded[1] * padded[2], img[0] * img[1] * img[2]
    for b in range(B):
        L.check(lib.vsseg_swi_finalize(out.data_ptr() + 4 * b * pvox * C, cnt.data_ptr(), L.i3(padded), L.i3(pad_before), L.i3(img), C, final.data_ptr() + 4 * b * ivox * C, stream), "swi_finalize")
    return final.permute(0, 4, 1, 2, 3)


def compute_dice_score(predicted_probabilities: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """Hard Dice of argmax vs label on the foreground channel, shape [1,1] (ref:params/VSparams.py:393-408): one fused reduction."""
    if not predicted_probabilities.is_cuda:
        raise RuntimeError("vs_seg_amd.compute_dice_score runs on an MI355X only; there is no CPU fallback")
    lib = L.lib()
    stream = torch.cuda.current_stream().cuda_stream
    lg = _as_cl(predicted_probabilities)
    assert lg.shape[-1] == 2
    lab = label.detach()
    if lab.dtype != torch.float32 or not lab.is_contiguous():
        lab = lab.to(torch.float32).contiguous()
    B = lg.shape[0]
    nv = lg.numel() // (2 * B)
    counts = torch.zeros((B, 3), dtype=torch.float64, device=lg.device)
    for b in range(B):  # DiceLoss(reduction="mean") averages the per-sample scores (the reference always calls it with batch 1)
        L.check(lib.vsseg_hard_dice_counts(lg.data_ptr() + 8 * b * nv, 2, lab.data_ptr() + 4 * b * nv, nv, counts.data_ptr() + 24 * b, stream), "hard_dice_counts")
    return ((2.0 * counts[:, 0] + 1e-5) / (counts[:, 1] + counts[:, 2] + 1e-5)).mean().to(torch.float32).reshape(1, 1)


def argmax_segmentation(outputs: torch.Tensor) -> torch.Tensor:
    """uint8 [B,X,Y,Z] = argmax over the 2 class channels of [B,2,X,Y,Z] logits / probabilities (the export path of
    ref:params/VSparams.py:582-594); ties go to class 0 like torch.argmax."""
    if not outputs.is_cuda:
        raise RuntimeError("vs_seg_amd.argmax_segmentation runs on an MI355X only; there is no CPU fallback")
    lg = _as_cl(outputs)
    assert lg.shape[-1] == 2
    seg = torch.empty(lg.shape[:-1], dtype=torch.uint8, device=lg.device)
    L.check(L.lib().vsseg_argmax2(lg.data_ptr(), 2, lg.numel() // 2, seg.data_ptr(), torch.cuda.current_stream().cuda_stream), "argmax2")
    return seg
