"""CPU tests of the data side: NIfTI IO, RAS orientation, the numpy restatement of the MONAI transform
chain, epoch sharding, checkpoint normalisation, the driver's host logic.

MONAI / nibabel are not installed (parity unpinned), so these are known-answer and round-trip tests: headers
built by hand with `struct`, affines whose orientation is known by construction, world-coordinate invariance.
"""
import gzip
import struct

import numpy as np
import pytest
import torch

from vs_seg_amd.data import nifti
from vs_seg_amd.data import transforms as T
from oracle import data_oracle as DO  # noqa: E402


def _raw_header(shape, datatype, bitpix, *, qform=None, sform=None, pixdim=(1, 1, 1), slope=0.0, inter=0.0, endian="<"):
    h = bytearray(352)
    struct.pack_into(endian + "i", h, 0, 348)
    struct.pack_into(endian + "8h", h, 40, len(shape), *shape, *([1] * (7 - len(shape))))
    struct.pack_into(endian + "2h", h, 70, datatype, bitpix)
    struct.pack_into(endian + "8f", h, 76, qform[0] if qform else 1.0, *pixdim, 1, 1, 1, 1)
    struct.pack_into(endian + "3f", h, 108, 352.0, slope, inter)
    struct.pack_into(endian + "2h", h, 252, 1 if qform else 0, 1 if sform is not None else 0)
    if qform:
        struct.pack_into(endian + "6f", h, 256, *qform[1:])
    if sform is not None:
        struct.pack_into(endian + "12f", h, 280, *np.asarray(sform, dtype=np.float64)[:3].reshape(-1))
    h[344:348] = b"n+1\x00"
    return bytes(h)


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64, np.uint16])
@pytest.mark.parametrize("gz", [True, False])
def test_nifti_round_trip(tmp_path, dtype, gz):
    rng = np.random.default_rng(0)
    arr = (rng.random((7, 5, 3)) * 100).astype(dtype)
    aff = np.array([[0.5, 0, 0, -10], [0, 0.6, 0, 20], [0, 0, 1.5, -30], [0, 0, 0, 1.0]])
    p = tmp_path / ("a.nii.gz" if gz else "a.nii")
    nifti.write_nifti(str(p), arr, aff)
    got, gaff, hdr = nifti.read_nifti(str(p))
    np.testing.assert_array_equal(got, arr.astype(np.float32))
    np.testing.assert_allclose(gaff, aff, atol=1e-6)
    assert hdr["shape"] == (7, 5, 3) and hdr["sform_code"] == 2
    if gz:
        assert gzip.open(p, "rb").read(4) == struct.pack("<i", 348)


def test_nifti_reads_fortran_order_scaling_big_endian_and_qform(tmp_path):
    shape = (4, 3, 2)
    vals = np.arange(24, dtype=np.int16)
    # x is the fastest axis on disk: element (x, y, z) = x + 4*y + 12*z
    raw = _raw_header(shape, 4, 16, sform=np.diag([2.0, 3.0, 4.0, 1.0]), slope=0.5, inter=1.0) + vals.astype("<i2").tobytes()
    p = tmp_path / "f.nii"
    p.write_bytes(raw)
    arr, aff, _ = nifti.read_nifti(str(p))
    assert arr.shape == shape and arr[1, 0, 0] == 0.5 * 1 + 1 and arr[0, 1, 0] == 0.5 * 4 + 1 and arr[0, 0, 1] == 0.5 * 12 + 1
    np.testing.assert_allclose(aff, np.diag([2.0, 3.0, 4.0, 1.0]))
    # big endian, qform only: quaternion (b,c,d) = (0,0,1) is a rotation by pi about z -> diag(-1,-1,1); qfac=-1 flips z
    raw = _raw_header(shape, 4, 16, qform=(-1.0, 0.0, 0.0, 1.0, 5.0, 6.0, 7.0), pixdim=(2, 3, 4), endian=">") + vals.astype(">i2").tobytes()
    p.write_bytes(raw)
    arr, aff, hdr = nifti.read_nifti(str(p))
    assert hdr["endianness"] == ">" and arr[3, 2, 1] == 23
    want = np.array([[-2.0, 0, 0, 5], [0, -3.0, 0, 6], [0, 0, -4.0, 7], [0, 0, 0, 1]])
    np.testing.assert_allclose(aff, want, atol=1e-6)


def test_nifti_rejects_garbage(tmp_path):
    p = tmp_path / "bad.nii"
    p.write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError):
        nifti.read_nifti(str(p))


@pytest.mark.parametrize("perm,flips", [((0, 1, 2), (1, 1, 1)), ((0, 1, 2), (-1, -1, 1)), ((2, 0, 1), (1, -1, 1)), ((1, 2, 0), (-1, 1, -1)), ((0, 2, 1), (1, 1, -1))])
def test_to_ras_known_orientations_and_world_coordinates(perm, flips):
    """Affine whose input axis a points along world axis perm[a] with sign flips[a] (plus a small shear that must not matter)."""
    rng = np.random.default_rng(1)
    shape = (5, 4, 3)
    arr = rng.random(shape).astype(np.float32)
    A = np.zeros((4, 4))
    for a in range(3):
        A[perm[a], a] = flips[a] * (1.0 + 0.5 * a)
    A[:3, :3] += 0.01 * rng.standard_normal((3, 3))
    A[:3, 3] = [3.0, -7.0, 11.0]
    A[3, 3] = 1.0
    ornt = nifti.io_orientation(A)
    np.testing.assert_array_equal(ornt[:, 0], perm)
    np.testing.assert_array_equal(ornt[:, 1], flips)
    ras, ras_aff, ornt2 = nifti.to_ras(arr, A)
    assert ras.shape == tuple(shape[list(perm).index(o)] for o in range(3))
    # the re-oriented affine is RAS+: dominant diagonal, positive
    o3 = nifti.io_orientation(ras_aff)
    np.testing.assert_array_equal(o3, [[0, 1], [1, 1], [2, 1]])
    # every voxel keeps its value and its world coordinate
    for idx in [(0, 0, 0), (4, 3, 2), (2, 1, 1), (1, 3, 0)]:
        world = A @ np.array([*idx, 1.0])
        new = [0, 0, 0]
        for a in range(3):
            new[perm[a]] = idx[a] if flips[a] > 0 else shape[a] - 1 - idx[a]
        assert ras[tuple(new)] == arr[idx]
        np.testing.assert_allclose(ras_aff @ np.array([*new, 1.0]), world, atol=1e-9)
    np.testing.assert_array_equal(nifti.from_ras(ras, ornt2), arr)


def test_normalize_pad_and_flip_crop_semantics():
    rng = np.random.default_rng(2)
    v = (rng.random((6, 5, 4)) * 50 + 10).astype(np.float32)
    n = DO.host_normalize_intensity(v)
    assert abs(float(n.mean())) < 1e-5 and abs(float(n.std()) - 1.0) < 1e-5
    c = np.full((3, 3, 3), 4.0, np.float32)
    np.testing.assert_array_equal(DO.host_normalize_intensity(c), np.zeros_like(c))  # std == 0: subtract only
    assert T.pad_widths((6, 5, 4), (8, 8, 4)) == [(1, 1), (1, 2), (0, 0)]  # odd difference: the extra voxel goes after
    assert T.pad_widths((10, 5, 4), (8, 8, 4))[0] == (0, 0)  # never crops
    p = DO.host_spatial_pad(v, (8, 8, 4))
    assert p.shape == (8, 8, 4) and p[0].sum() == 0 and p[1, 1, 0] == v[0, 0, 0]
    # RandFlipd acts on the padded volume, the crop start is in flipped coordinates
    out = DO.host_flip_crop(p, True, (1, 2, 0), (4, 3, 4))
    np.testing.assert_array_equal(out, p[::-1][1:5, 2:5, 0:4])
    assert out[0, 0, 0] == p[8 - 1 - 1, 2, 0]


def test_random_tail_is_reproducible_in_range_and_uses_separate_states():
    a, b = T.RandomTail((4, 4, 2), 0.5, seed=3), T.RandomTail((4, 4, 2), 0.5, seed=3)
    da = [a.draw((9, 4, 6)) for _ in range(200)]
    assert da == [b.draw((9, 4, 6)) for _ in range(200)]
    flips = [f for f, _ in da]
    assert 60 < sum(flips) < 140
    for _, s in da:
        assert 0 <= s[0] <= 5 and s[1] == 0 and 0 <= s[2] <= 4  # size == roi: start 0 without consuming a draw
    assert {s[0] for _, s in da} == set(range(6))  # upper bound size - roi inclusive
    # the validation chain has no RandFlipd: the crop state is then the FIRST seed drawn from the compose state
    v = T.RandomTail((4, 4, 2), None, seed=3)
    R = np.random.RandomState(3)
    crop = np.random.RandomState(R.randint(T.MAX_SEED, dtype="uint32"))
    assert v.draw((9, 4, 6)) == (False, (int(crop.randint(0, 6)), 0, int(crop.randint(0, 5))))


def test_epoch_batches_shard_and_cover():
    rng = np.random.RandomState(0)
    e0 = T.epoch_batches(11, 4, True, np.random.RandomState(0))
    assert sorted(i for b in e0 for i in b) == list(range(11)) and [len(b) for b in e0] == [4, 4, 3]
    r0 = T.epoch_batches(11, 2, True, np.random.RandomState(5), 0, 2)
    r1 = T.epoch_batches(11, 2, True, np.random.RandomState(5), 1, 2)
    # 11 cases over 2 ranks: padded to 12 by wrapping around (both ranks run the same number of steps), every case covered
    assert sorted(set(i for b in r0 + r1 for i in b)) == list(range(11)) and sum(len(b) for b in r0) == sum(len(b) for b in r1) == 6
    assert len(set(i for b in r0 for i in b) & set(i for b in r1 for i in b)) <= 1
    assert T.epoch_batches(5, 1, False, rng) == [[0], [1], [2], [3], [4]]


def test_center_of_mass_slice_known_answers():
    from vs_seg_amd.params import VSparams

    lab = np.zeros((4, 4, 10))
    assert VSparams.get_center_of_mass_slice(lab) == 5  # empty label: uniform weights, left-to-right float sum 4.500000000000001 -> 5
    lab[1, 1, 7] = 1
    assert VSparams.get_center_of_mass_slice(lab) == 7
    lab[2, 2, 2] = 3  # masses 3 @ z=2, 1 @ z=7 -> 3.25 -> 3
    assert VSparams.get_center_of_mass_slice(torch.from_numpy(lab)) == 3


def test_checkpoint_normalisation(tmp_path):
    from vs_seg_amd import checkpoint as CK

    sd = {"model.0.conv.unit0.conv.weight": torch.zeros(2, 2), "model.0.conv.unit0.norm.num_batches_tracked": torch.tensor(3)}
    assert CK.normalise_state_dict(sd).keys() == sd.keys()
    wrapped = {"state_dict": {"module." + k: v for k, v in sd.items()}}
    assert CK.normalise_state_dict(wrapped).keys() == sd.keys()
    with pytest.raises(ValueError):
        CK.normalise_state_dict({"epoch": 3})


def test_epoch_batches_give_every_rank_the_same_number_of_steps():
    """with n % world != 0 the ranks ran different numbers of training steps and paired mismatched collectives."""
    from vs_seg_amd.data.transforms import epoch_batches

    for n, world, bs in [(176, 8, 1), (10, 4, 1), (7, 2, 2), (5, 8, 1), (242, 8, 4), (3, 2, 1)]:
        per_rank = [epoch_batches(n, bs, True, np.random.RandomState(0), r, world) for r in range(world)]
        assert len({len(b) for b in per_rank}) == 1, (n, world, [len(b) for b in per_rank])
        assert len({sum(len(x) for x in b) for b in per_rank}) == 1
        seen = [i for b in per_rank for x in b for i in x]
        assert set(seen) == set(range(n)) and len(seen) == -(-n // world) * world  # every case at least once, padded by wrap-around
    # a single process is untouched: the reference's DataLoader order, nothing padded or dropped
    one = epoch_batches(7, 2, False, np.random.RandomState(0))
    assert one == [[0, 1], [2, 3], [4, 5], [6]]


def test_validation_and_test_loaders_are_not_padded():
    """wrap-around padding is for the training loader only.  The validation / test loaders issue one collective after
    their loop; a padded case would be counted twice in the all-reduced Dice sums (46 test cases on 8 ranks: cases 0 and 1 twice)."""
    from vs_seg_amd.data.transforms import epoch_batches
    from vs_seg_amd.parallel import shard_indices

    for n, world in [(46, 8), (20, 8), (5, 2), (3, 4)]:
        per_rank = [epoch_batches(n, 1, False, np.random.RandomState(0), r, world, pad=False) for r in range(world)]
        for r in range(world):  # exactly the shard run_inference writes its Dice scores into
            assert [i for b in per_rank[r] for i in b] == shard_indices(n, r, world)
        seen = sorted(i for b in per_rank for x in b for i in x)
        assert seen == list(range(n))  # every case exactly once over all ranks


def test_cached_loader_pads_only_when_it_shuffles(monkeypatch):
    from vs_seg_amd import params as PR

    monkeypatch.setattr(PR.DP, "get_rank", lambda: 1)
    monkeypatch.setattr(PR.DP, "world_size", lambda: 2)
    cases = [dict(image=None, label=None, image_meta={}, label_meta={}) for _ in range(5)]
    train = PR.CachedLoader(cases, None, 1, True, None)
    val = PR.CachedLoader(cases, None, 1, False, None)
    assert train.pad and not val.pad
    from vs_seg_amd.data.transforms import epoch_batches

    assert len(epoch_batches(5, 1, True, np.random.RandomState(0), 1, 2, pad=train.pad)) == 3  # padded to 6 -> 3 per rank
    assert epoch_batches(5, 1, False, np.random.RandomState(0), 1, 2, pad=val.pad) == [[1], [3]]


def test_seeded_weights_helper_equals_the_oracle_generator():
    import torch

    from oracle import vsseg_oracle as O
    from tests.helpers import seeded_weights_for

    ref = O.seeded_state_dict(True, 24)
    got = seeded_weights_for({k: torch.empty(v.shape) for k, v in ref.items()}, 24)
    assert list(got) == list(ref)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


def test_split_files_ship_with_the_repo():
    """ref:params/VSparams.py:72-75 defaults: ./params/split_TCIA.csv (242 cases) and ./params/split_debug.csv (6 cases)."""
    import csv
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = list(csv.reader(open(os.path.join(root, "params", "split_TCIA.csv"))))
    assert len(rows) == 242 and {r[1] for r in rows} == {"training", "validation", "test"}
    assert [sum(1 for r in rows if r[1] == s) for s in ("training", "validation", "test")] == [176, 20, 46]
    dbg = list(csv.reader(open(os.path.join(root, "params", "split_debug.csv"))))
    assert [r[1] for r in dbg] == ["training", "training", "validation", "validation", "test", "test"]


def test_make_debug_data_writes_readable_cases(tmp_path):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_debug_data as M
    from vs_seg_amd.data import nifti

    M.main(["--data_root", str(tmp_path), "--size", "24", "20", "16"])
    d = os.path.join(str(tmp_path), "input_data", "vs_gk_182")
    img, aff, _ = nifti.read_nifti(os.path.join(d, "vs_gk_t1_refT1.nii.gz"))
    lab, _, _ = nifti.read_nifti(os.path.join(d, "vs_gk_seg_refT1.nii.gz"))
    assert img.shape == lab.shape == (24, 20, 16) and 0 < lab.sum() < lab.size and aff[0, 0] < 0
    assert len(os.listdir(os.path.join(str(tmp_path), "input_data"))) == 6
