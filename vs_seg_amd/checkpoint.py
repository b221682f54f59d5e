"""Checkpoint compatibility: the reference's `torch.save(model.state_dict())` files
(ref:params/VSparams.py:508,526; the published `best_metric_model.pth`, ref:README.md:161-170) load into the HIP-backed
module unchanged — same 256 keys and shapes (tests/golden/manifest.json).  Helpers here only add tolerance for the
wrappers people save around a state_dict (`module.` prefixes from DataParallel, an outer {"state_dict": ...} dict) and a
strict report of what does not match.
"""
from __future__ import annotations

from typing import Dict

import torch


def normalise_state_dict(obj) -> Dict[str, torch.Tensor]:
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict) and all(torch.is_tensor(v) for v in obj["model"].values()):
        obj = obj["model"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError("not a state_dict (expected a mapping of tensors)")
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in obj.items()}


def load_checkpoint(model, path: str, strict: bool = True):
    """Load a reference-format checkpoint file into `model` (UNet2d5_spvPA); returns (missing, unexpected) key lists."""
    sd = normalise_state_dict(torch.load(path, map_location="cpu"))
    want = model.state_dict()
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    bad_shape = [k for k in sd if k in want and tuple(sd[k].shape) != tuple(want[k].shape)]
    if bad_shape:
        raise ValueError(f"{path}: shape mismatch for {bad_shape[:5]} (is the checkpoint from the attention / no-attention variant?)")
    if strict and (missing or unexpected):
        raise KeyError(f"{path}: missing keys {missing[:5]}, unexpected keys {unexpected[:5]}")
    model.load_state_dict({k: v for k, v in sd.items() if k in want}, strict=strict)
    return missing, unexpected


def save_checkpoint(model, path: str):
    """What the reference writes: the bare state_dict, fp32 CPU tensors."""
    torch.save({k: v.detach().to("cpu") for k, v in model.state_dict().items()}, path)
