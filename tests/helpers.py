"""Shared synthetic-input helpers for the parity tests (same generators as tests/golden/make_goldens.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth_input(seed, shape):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def synth_label(seed, shape):
    rng = np.random.default_rng(seed + 7919)
    B, _, X, Y, Z = shape
    lab = np.zeros(shape, np.float32)
    for b in range(B):
        if B > 1 and b == 0:
            continue
        c = [rng.integers(s // 4, max(s // 4 + 1, 3 * s // 4)) for s in (X, Y, Z)]
        r = [max(1, s // 6) for s in (X, Y, Z)]
        gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
        lab[b, 0] = (((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0).astype(np.float32)
    return torch.from_numpy(lab)


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def check_summary(t, meta_json, sub, atol, rtol=0.0):
    """Compare a tensor with a golden summary (checksums + strided sub-sample)."""
    meta = json.loads(str(meta_json))
    a = t.detach().double().flatten()
    assert a.numel() == meta["n"]
    got = a[:: meta["stride"]].float().numpy()
    np.testing.assert_allclose(got, sub, atol=atol, rtol=rtol)
    assert abs(float(a.sum()) - meta["sum"]) <= atol * meta["n"] ** 0.5 * 4 + rtol * meta["abssum"]
    assert abs(float(a.abs().sum()) - meta["abssum"]) <= atol * meta["n"] + rtol * meta["abssum"]


def seeded_weights_for(state_dict, seed):
    """The goldens' weight generator (one numpy stream in state_dict order; the goldens store no weights) applied to the
    key/shape list of a model's own state_dict — same values as `oracle.vsseg_oracle.seeded_state_dict` (pinned by
    tests/test_host.py) without importing anything from oracle/, so bench.py's parity check can use it."""
    import math

    rng = np.random.default_rng(seed)
    sd = {}
    for key, t in state_dict.items():
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 2)[-2:]
        if key.endswith("num_batches_tracked"):
            v = np.zeros((), np.int64)
        elif leaf == ["norm", "weight"] or key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == ["norm", "bias"] or key.endswith("running_mean"):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == ["act", "weight"]:
            v = rng.uniform(0.1, 0.4, shape).astype(np.float32)
        elif key.endswith("bias"):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            is_t = key.endswith(".2.conv.weight") and ".submodule.2." in key
            fan_in = (shape[0] if is_t else shape[1]) * int(np.prod(shape[2:]))
            if is_t:
                fan_in = max(fan_in // 4, 1)
            v = (rng.standard_normal(shape) / math.sqrt(fan_in)).astype(np.float32)
        sd[key] = torch.from_numpy(np.asarray(v))
    return sd
