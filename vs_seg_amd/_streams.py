"""The process-wide pool of side HIP streams.

HIP multiplexes its streams onto a few hardware queues (4 by default); work on two streams that share a queue serialises.  The product uses streams in two places — the
weight-gradient stream of the training step (engine.Engine.side_stream) and the window-group lanes of the sliding-window inferer — and both draw from this one pool, lane 0
first, so that a process that trains AND runs inference (bench.py; the reference's training script validates between epochs, ref:params/VSparams.py:496-541) holds the caller's
stream plus at most three side streams.  Measured (round 5): with a private weight-gradient stream beside three inference lanes the sliding window ran at 40.9 volumes/s against
42.9 with two lanes; in an inference-only process (four streams in all) three lanes gave 42.7 against 41.8."""
from __future__ import annotations

from typing import Dict

import torch

_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def side_stream(device, i: int) -> "torch.cuda.Stream":
    key = (str(torch.device(device)), int(i))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]
