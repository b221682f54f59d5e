"""NIfTI-1 reader / writer and RAS re-orientation in plain numpy.

The reference reads and writes its volumes through MONAI's `LoadNiftid` / `NiftiSaver`, i.e. nibabel
(ref:params/VSparams.py:208-210, 582-594); neither is installed here, so the parts of the format and of nibabel's
orientation logic the pipeline needs are restated: single-file `.nii` / `.nii.gz`, the numeric datatypes the VS dataset
uses, `scl_slope/inter`, sform-then-qform affine selection, `io_orientation` + axis flips/permutations to RAS and back.
Parity unpinned (no nibabel to compare with): pinned by round trips and hand-built known-answer headers in
tests/test_data_pipeline.py.
"""
from __future__ import annotations

import gzip
import struct
from typing import Dict, Tuple

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


def _quaternion_affine(b, c, d, qfac, pixdim, offset):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-12 else 0.0
    if a == 0.0:  # renormalise a (b, c, d) of length >= 1 to a unit quaternion with a = 0
        n = np.sqrt(b * b + c * c + d * d)
        b, c, d = b / n, c / n, d / n
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    S = np.diag([pixdim[0], pixdim[1], pixdim[2] * (-1.0 if qfac < 0 else 1.0)])
    A = np.eye(4)
    A[:3, :3] = R @ S
    A[:3, 3] = offset
    return A


def read_nifti(path: str) -> Tuple[np.ndarray, np.ndarray, Dict]:
    """-> (float32 array indexed [x, y, z] like nibabel's get_fdata order, 4x4 affine, header fields)."""
    raw = gzip.open(path, "rb").read() if str(path).endswith(".gz") else open(path, "rb").read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError(f"{path}: sizeof_hdr != 348 (not NIfTI-1)")
    magic = raw[344:348]
    if magic[:3] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad magic {magic!r}")
    if magic[:3] == b"ni1":
        raise ValueError(f"{path}: header/image pairs (.hdr/.img) are not supported")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    nd = dim[0]
    if not 1 <= nd <= 7:
        raise ValueError(f"{path}: bad dim[0] = {nd}")
    shape = tuple(int(d) for d in dim[1 : 1 + nd])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    arr = data.astype(np.float32)
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0 and np.isfinite(slope) and np.isfinite(inter):
            arr = arr * np.float32(slope) + np.float32(inter)
    if sform_code > 0:  # nibabel get_best_affine: sform, then qform, then the pixel sizes
        affine = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        affine = _quaternion_affine(qb, qc, qd, pixdim[0], pixdim[1:4], (qx, qy, qz))
    else:
        affine = np.diag([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0, 1.0])
        affine[:3, 3] = -0.5 * (np.array(shape[:3]) - 1) * np.diag(affine)[:3]
    hdr = dict(shape=shape, datatype=int(datatype), bitpix=int(bitpix), pixdim=tuple(float(p) for p in pixdim), qform_code=int(qform_code), sform_code=int(sform_code),
               scl_slope=float(slope), scl_inter=float(inter), endianness=end)
    return np.ascontiguousarray(arr), affine.astype(np.float64), hdr


def write_nifti(path: str, array: np.ndarray, affine: np.ndarray, dtype=None) -> None:
    """Single-file NIfTI-1 with the affine in the sform (code 2, as nibabel's Nifti1Image(data, affine) writes it)."""
    arr = np.asarray(array)
    if dtype is not None:
        arr = arr.astype(dtype)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    key = arr.dtype.str[1:]
    if key not in _CODES:
        raise ValueError(f"cannot write dtype {arr.dtype} as NIfTI")
    if not 1 <= arr.ndim <= 7:
        raise ValueError("NIfTI arrays have 1..7 dimensions")
    A = np.asarray(affine, dtype=np.float64)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dims = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dims)
    struct.pack_into("<2h", hdr, 70, _CODES[key], arr.dtype.itemsize * 8)
    vox = np.sqrt((A[:3, :3] ** 2).sum(0))
    struct.pack_into("<8f", hdr, 76, 1.0, *[float(v) for v in vox], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 0, 2)
    struct.pack_into("<12f", hdr, 280, *[float(v) for v in A[:3].reshape(-1)])
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + np.asfortranarray(arr.astype(arr.dtype.newbyteorder("<"))).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=3) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


# ---- orientation (nibabel.orientations.io_orientation / ornt_transform / apply_orientation, restated) ---------------
def io_orientation(affine: np.ndarray, tol=None) -> np.ndarray:
    """[3][2]: for every input axis the output (R, A, S) axis it is closest to, and the direction (+1 / -1)."""
    A = np.asarray(affine, dtype=np.float64)
    q, p = A.shape[0] - 1, A.shape[1] - 1
    RZS = A[:q, :p]
    zooms = np.sqrt((RZS * RZS).sum(0))
    zooms[zooms == 0] = 1.0
    RS = RZS / zooms
    P, S, Qs = np.linalg.svd(RS, full_matrices=False)
    if tol is None:
        tol = S.max() * max(RS.shape) * np.finfo(S.dtype).eps
    keep = S > tol
    R = P[:, keep] @ Qs[keep]
    ornt = np.full((p, 2), np.nan)
    for in_ax in range(p):
        col = R[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1.0 if col[out_ax] < 0 else 1.0
            R[out_ax, :] = 0  # an output axis is used once
    return ornt


def to_ras(array: np.ndarray, affine: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Flip / permute the first three axes so that they run Left→Right, Posterior→Anterior, Inferior→Superior
    (MONAI Orientationd(axcodes="RAS")).  -> (array, new affine, ornt) with ornt the transform that was applied."""
    ornt = io_orientation(affine)
    if np.isnan(ornt).any():
        raise ValueError("affine has a dropped axis; cannot orient")
    arr = array
    shape = np.array(arr.shape[:3])
    for ax in range(3):  # flips first (in input axes), then the permutation — nibabel.apply_orientation
        if ornt[ax, 1] < 0:
            arr = np.flip(arr, axis=ax)
    order = np.argsort(ornt[:, 0]).astype(int)  # output axis o takes input axis order[o]
    arr = np.transpose(arr, tuple(order) + tuple(range(3, arr.ndim)))
    # inverse affine of the orientation (nibabel.orientations.inv_ornt_aff)
    T = np.zeros((4, 4))
    T[3, 3] = 1.0
    for in_ax in range(3):
        out_ax, flip = int(ornt[in_ax, 0]), ornt[in_ax, 1]
        T[in_ax, out_ax] = flip
        if flip < 0:
            T[in_ax, 3] = shape[in_ax] - 1
    return np.ascontiguousarray(arr), np.asarray(affine, dtype=np.float64) @ T, ornt


def from_ras(array: np.ndarray, ornt: np.ndarray) -> np.ndarray:
    """Undo `to_ras` (used when a segmentation is exported in the image's original orientation)."""
    order = np.argsort(ornt[:, 0]).astype(int)
    inv = np.argsort(order)
    arr = np.transpose(array, tuple(inv) + tuple(range(3, array.ndim)))
    for ax in range(3):
        if ornt[ax, 1] < 0:
            arr = np.flip(arr, axis=ax)
    return np.ascontiguousarray(arr)
