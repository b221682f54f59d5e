"""CPU restatement of the data-side arithmetic of the reference's MONAI transform chain (ref:params/VSparams.py:205-245): the checker of the HIP data
kernels (`vsseg_normalize_intensity`, `vsseg_crop_flip`, vs_seg_amd/data/transforms.py).

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE ***  Only `tests/` imports this module.

**Parity unpinned**: the arithmetic lives in MONAI 0.4.0 (ref:requirements.txt:7: `NormalizeIntensityd`, `SpatialPadd`, `RandFlipd`, `RandSpatialCropd`), which is neither
vendored in the reference nor installed in this image; these functions restate its published behaviour and are pinned only by hand-computed cases
(tests/test_data_pipeline.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def host_normalize_intensity(img: np.ndarray) -> np.ndarray:
    """NormalizeIntensityd(keys=["image"]): (x - mean) / std over the whole image, population std, no division if std == 0."""
    m, s = float(img.mean(dtype=np.float64)), float(img.std(dtype=np.float64))
    out = img.astype(np.float32) - np.float32(m)
    return out / np.float32(s) if s != 0.0 else out


def pad_widths(shape: Sequence[int], spatial_size: Sequence[int]) -> List[Tuple[int, int]]:
    """SpatialPadd(method="symmetric"): width w = max(target - size, 0) split as (w // 2, w - w // 2)."""
    out = []
    for d, t in zip(shape, spatial_size):
        w = max(int(t) - int(d), 0)
        out.append((w // 2, w - w // 2))
    return out


def host_spatial_pad(vol: np.ndarray, spatial_size: Sequence[int]) -> np.ndarray:
    return np.pad(vol, pad_widths(vol.shape, spatial_size), mode="constant", constant_values=0)


def host_flip_crop(vol: np.ndarray, flip: bool, start: Sequence[int], roi: Sequence[int]) -> np.ndarray:
    v = vol[::-1] if flip else vol
    return np.ascontiguousarray(v[start[0] : start[0] + roi[0], start[1] : start[1] + roi[1], start[2] : start[2] + roi[2]])
