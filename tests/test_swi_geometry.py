"""Known-answer tests for the sliding-window index arithmetic (MONAI 0.4.0 behaviour).

`sliding_window_inference` lives in un-vendored MONAI (ref:requirements.txt:7; call site ref:params/VSparams.py:568-574),
so these tables — not reference goldens — pin the oracle ("parity unpinned" in DESIGN.md).  The product's host-side
geometry (vs_seg_amd.inferers) is checked against the same tables and against the C oracle bit-exactly.
"""
import numpy as np
import pytest
import torch

from oracle import c_binding as C
from oracle import vsseg_oracle as O

TABLE = [
    # volume, roi, overlap, padded, pad_before, interval, n_windows, per-dim starts
    ((512, 512, 120), (384, 384, 64), 0.25, (512, 512, 120), (0, 0, 0), (288, 288, 48), 12, ([0, 128], [0, 128], [0, 48, 56])),
    ((512, 512, 120), (384, 128, 128), 0.5, (512, 512, 128), (0, 0, 4), (192, 64, 128), 14, ([0, 128], [0, 64, 128, 192, 256, 320, 384], [0])),
    ((448, 448, 80), (384, 384, 64), 0.25, (448, 448, 80), (0, 0, 0), (288, 288, 48), 8, ([0, 64], [0, 64], [0, 16])),
    ((384, 384, 80), (384, 384, 64), 0.25, (384, 384, 80), (0, 0, 0), (384, 384, 48), 2, ([0], [0], [0, 16])),
    ((448, 448, 80), (384, 128, 128), 0.5, (448, 448, 128), (0, 0, 24), (192, 64, 128), 12, ([0, 64], [0, 64, 128, 192, 256, 320], [0])),
    ((384, 384, 80), (384, 128, 128), 0.5, (384, 384, 128), (0, 0, 24), (384, 64, 128), 5, ([0], [0, 64, 128, 192, 256], [0])),
    ((64, 64, 64), (128, 128, 32), 0.25, (128, 128, 64), (32, 32, 0), (128, 128, 24), 3, ([0], [0], [0, 24, 32])),
]


@pytest.mark.parametrize("vol,roi,ov,padded,padb,iv,n,starts", TABLE)
def test_window_tables(vol, roi, ov, padded, padb, iv, n, starts):
    r, p, pb, interval, wins = O.swi_geometry(vol, roi, ov)
    assert tuple(p) == padded and tuple(pb) == padb and tuple(interval) == iv and len(wins) == n
    expect = [(a, b, c) for a in starts[0] for b in starts[1] for c in starts[2]]  # first spatial dim slowest
    assert wins == expect
    for d in range(3):  # plain-C restatement agrees bit-exactly
        cp, cpb, civ, cst = C.swi_starts(vol[d], roi[d], ov)
        assert (cp, cpb, civ, cst) == (padded[d], padb[d], iv[d], starts[d])


def test_float_truncation_of_interval():
    # int(roi*(1-overlap)) truncates in double precision: 10*(1-0.7) = 3.0000000000000004 -> 3, 10*(1-0.9) = 0.99.. -> max(0,1)=1
    assert O.swi_geometry((40,) * 3, (10,) * 3, 0.7)[3] == [3, 3, 3]
    assert O.swi_geometry((40,) * 3, (10,) * 3, 0.9)[3] == [1, 1, 1]
    assert C.swi_starts(40, 10, 0.7)[2] == 3 and C.swi_starts(40, 10, 0.9)[2] == 1


def _taps(sigma):
    tail = int(max(sigma * 4.0, 0.5) + 0.5)
    xs = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / abs(sigma)
    return (0.5 * ((t * (xs + 0.5)).erf() - (t * (xs - 0.5)).erf())).clamp(min=0).numpy()


def test_gaussian_importance_map_properties():
    m = O.gaussian_importance_map((128, 128, 32))
    assert m.shape == (128, 128, 32) and float(m.max()) == 1.0 and float(m.min()) > 0
    assert float(m[64, 64, 16]) == 1.0
    assert 1e-11 < float(m[0, 0, 0]) < 1e-10  # SURVEY.md App. B.1 step 5 probe: corner weight 3.9e-11
    for r in (128, 128, 32):
        np.testing.assert_allclose(C.gaussian_1d(r * 0.125), _taps(r * 0.125), rtol=2e-6, atol=1e-7)  # erff vs torch.erf differ in the last ulp before the subtraction


def test_swi_identity_predictor_roundtrip():
    """Blend of identical window predictions reproduces the input (weights cancel): size-independent property."""
    x = torch.randn(1, 1, 40, 36, 20)
    out = O.sliding_window_inference(x, (16, 16, 32), 1, lambda w: torch.cat([w, -w], 1), overlap=0.5, mode="gaussian")
    assert out.shape == (1, 2, 40, 36, 20)
    np.testing.assert_allclose(out[:, :1].numpy(), x.numpy(), atol=1e-5)
    np.testing.assert_allclose(out[:, 1:].numpy(), -x.numpy(), atol=1e-5)


def test_c_conv_matches_torch_functional():
    import torch.nn.functional as F

    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 6, 5, 4)).astype(np.float32)
    for k, s in (((3, 3, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2)), ((3, 3, 1), (2, 2, 1)), ((1, 1, 1), (1, 1, 1))):
        pad = tuple((kk - 1) // 2 for kk in k)
        w = rng.standard_normal((4, 3, *k)).astype(np.float32)
        b = rng.standard_normal(4).astype(np.float32)
        np.testing.assert_allclose(C.conv3d(x, w, b, s, pad), F.conv3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=pad).numpy(), atol=2e-5)
    for k, s in (((3, 3, 1), (2, 2, 1)), ((3, 3, 3), (2, 2, 2))):
        pad = tuple((kk - 1) // 2 for kk in k)
        opad = tuple(ss + 2 * p - (kk - 1) - 1 for ss, p, kk in zip(s, pad, k))
        w = rng.standard_normal((3, 4, *k)).astype(np.float32)
        b = rng.standard_normal(4).astype(np.float32)
        want = F.conv_transpose3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=pad, output_padding=opad).numpy()
        np.testing.assert_allclose(C.conv_transpose3d(x, w, b, s, pad, opad), want, atol=2e-5)
