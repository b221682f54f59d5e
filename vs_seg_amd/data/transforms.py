"""The reference's MONAI transform chains (ref:params/VSparams.py:205-245), restated and moved to the GPU.

Deterministic head (cached once per case, like `CacheDataset(cache_rate=1.0)`, ref:params/VSparams.py:305-334):
    LoadNiftid → AddChanneld → Orientationd("RAS") → NormalizeIntensityd(image) → SpatialPadd(pad_crop_shape)
Random tail (per sample, per epoch):
    RandFlipd(prob=0.5, spatial_axis=0) → RandSpatialCropd(roi=pad_crop_shape, random_center=True, random_size=False)

The numpy restatement of MONAI 0.4.0's arithmetic that checks the HIP path (`vsseg_normalize_intensity`, `vsseg_crop_flip`) is test
infrastructure and lives in `oracle/data_oracle.py` (parity unpinned — MONAI is not installed).  `PatchSampler`
is the product path: cached volumes live in HBM, one launch crops/flips image and label of a whole batch.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L
from . import nifti

MAX_SEED = np.iinfo(np.uint32).max + 1


# ---------------------------------------------------------------------------------------------------------------
# geometry shared with the checker (the numpy restatement of the MONAI arithmetic lives in oracle/data_oracle.py: test infrastructure)
# ---------------------------------------------------------------------------------------------------------------
def pad_widths(shape: Sequence[int], spatial_size: Sequence[int]) -> List[Tuple[int, int]]:
    """SpatialPadd(method="symmetric"): width w = max(target - size, 0) split as (w // 2, w - w // 2)."""
    out = []
    for d, t in zip(shape, spatial_size):
        w = max(int(t) - int(d), 0)
        out.append((w // 2, w - w // 2))
    return out


class RandomTail:
    """The random decisions of RandFlipd + RandSpatialCropd with MONAI's per-transform RandomState layout:
    `Compose.set_random_state(seed)` seeds its own state and gives every Randomizable transform, in order, the seed
    `R.randint(MAX_SEED, dtype=uint32)`; RandFlipd draws `R.random_sample() < prob`, RandSpatialCropd draws
    `R.randint(0, size - roi + 1)` per axis where size > roi."""

    def __init__(self, roi: Sequence[int], flip_prob: Optional[float] = 0.5, seed: Optional[int] = None):
        self.roi = tuple(int(r) for r in roi)
        self.flip_prob = flip_prob
        self.set_random_state(seed)

    def set_random_state(self, seed: Optional[int] = None):
        R = np.random.RandomState(seed)
        self._flipR = np.random.RandomState(R.randint(MAX_SEED, dtype="uint32")) if self.flip_prob is not None else None
        self._cropR = np.random.RandomState(R.randint(MAX_SEED, dtype="uint32"))
        return self

    def draw(self, shape: Sequence[int]) -> Tuple[bool, Tuple[int, int, int]]:
        flip = bool(self._flipR.random_sample() < self.flip_prob) if self._flipR is not None else False
        start = tuple(int(self._cropR.randint(0, s - r + 1)) if s > r else 0 for s, r in zip(shape, self.roi))
        return flip, start


# ---------------------------------------------------------------------------------------------------------------
# product path: cached volumes in HBM, HIP kernels
# ---------------------------------------------------------------------------------------------------------------
def load_case(files: Dict[str, str], pad_to: Optional[Sequence[int]] = None, device="cuda") -> Dict:
    """Deterministic head of the chain for one {"image": path, "label": path} entry → cached device tensors [X,Y,Z] fp32
    (RAS, image normalised on the GPU, both zero-padded to at least `pad_to`) + the metadata NIfTI export needs."""
    lib = L.lib()
    out: Dict = {"files": dict(files)}
    stream = torch.cuda.current_stream().cuda_stream
    for key in ("image", "label"):
        arr, aff, hdr = nifti.read_nifti(files[key])
        ras, ras_aff, ornt = nifti.to_ras(arr, aff)
        t = torch.from_numpy(np.ascontiguousarray(ras, dtype=np.float32)).to(device)
        if key == "image":
            acc = torch.zeros(2, dtype=torch.float64, device=device)
            y = torch.empty_like(t)
            L.check(lib.vsseg_normalize_intensity(t.data_ptr(), y.data_ptr(), t.numel(), acc.data_ptr(), stream), "normalize_intensity")
            t = y
        if pad_to is not None:
            pw = pad_widths(t.shape, pad_to)
            if any(a or b for a, b in pw):
                t = torch.nn.functional.pad(t, (pw[2][0], pw[2][1], pw[1][0], pw[1][1], pw[0][0], pw[0][1]))
        out[key] = t.contiguous()
        out[key + "_meta"] = dict(affine=ras_aff, original_affine=aff, ornt=ornt, filename_or_obj=files[key], spatial_shape=tuple(arr.shape))
    assert out["image"].shape == out["label"].shape, f"image/label shapes differ for {files}"
    return out


class PatchSampler:
    """RandFlipd + RandSpatialCropd over cached cases, one `vsseg_crop_flip` launch per batch.

    `sample(indices)` → (inputs [B,1,*roi], labels [B,1,*roi]) fp32 on the device, the tensors the reference's DataLoader
    yields as batch["image"], batch["label"] (ref:params/VSparams.py:455).  `flip_prob=None` = the validation chain
    (no RandFlipd, ref:params/VSparams.py:224-236)."""

    def __init__(self, cases: List[Dict], roi: Sequence[int], flip_prob: Optional[float] = 0.5, seed: Optional[int] = 0):
        self.cases, self.roi = cases, tuple(int(r) for r in roi)
        self.tail = RandomTail(self.roi, flip_prob, seed)
        self.lib = L.lib()
        self.last_draws: List[Tuple[bool, Tuple[int, int, int]]] = []

    def __len__(self):
        return len(self.cases)

    def sample(self, indices: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = self.cases[indices[0]]["image"].device
        B = len(indices)
        jobs = (L.CropJob * (2 * B))()
        self.last_draws = []
        for b, i in enumerate(indices):
            case = self.cases[i]
            shape = tuple(case["image"].shape)
            flip, start = self.tail.draw(shape)
            self.last_draws.append((flip, start))
            for k, key in enumerate(("image", "label")):
                j = jobs[b + k * B]  # dst = [image_0..image_{B-1} | label_0..label_{B-1}]
                j.src, j.sdims, j.origin, j.flip_x = case[key].data_ptr(), L.i3(shape), L.i3(start), int(flip)
        jbuf = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        out = torch.empty((2, B, 1, *self.roi), dtype=torch.float32, device=dev)
        L.check(self.lib.vsseg_crop_flip(jbuf.data_ptr(), 2 * B, out.data_ptr(), L.i3(self.roi), torch.cuda.current_stream().cuda_stream), "crop_flip")
        jbuf.record_stream(torch.cuda.current_stream())
        return out[0], out[1]


def epoch_batches(n: int, batch_size: int, shuffle: bool, rng: np.random.RandomState, rank: int = 0, world: int = 1, pad: bool = True) -> List[List[int]]:
    """Index batches of one epoch: DataLoader(shuffle=True) order (ref:params/VSparams.py:311-318), sharded over ranks
    (rank r takes positions r, r+world, … of the shuffled list — SURVEY §8e) and cut into batches (last one may be short).

    `pad=True` (TRAINING loaders only): every rank gets the SAME number of indices, hence of batches — when n is not a multiple of
    `world` the list is padded by wrapping around to its own beginning (torch's DistributedSampler(drop_last=False) rule), because each
    training step issues a gradient all-reduce and unequal step counts would pair mismatched collectives.  Validation and test loaders
    pass `pad=False`: they issue ONE collective after their loop, so ranks may run different numbers of cases, and a wrapped case would
    be counted twice in the all-reduced Dice / loss sums (rank r then owns exactly `shard_indices(n, r, world)`)."""
    order = rng.permutation(n) if shuffle else np.arange(n)
    if pad and world > 1 and n % world and n > 0:
        pad = world - n % world
        order = np.concatenate([order, np.resize(order, pad)])
    mine = [int(i) for i in order[rank::world]]
    return [mine[i : i + batch_size] for i in range(0, len(mine), batch_size)]
