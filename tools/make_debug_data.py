#!/usr/bin/env python
"""Synthetic stand-ins for the six cases of params/split_debug.csv so that BASELINE config 1 runs without the TCIA data set:

    python tools/make_debug_data.py [--data_root ./data/VS_defaced/] [--size 64 64 64] [--dataset T1]
    python VS_train.py --debug --num_epochs 2          # then: python VS_inference.py --debug

Each case is a 64^3 (default) float volume with Gaussian noise and one bright ellipsoid "tumour" plus its binary label, written
as NIfTI-1 .nii.gz under <data_root>/input_data/<case>/ with the reference's file names (ref:params/VSparams.py:162-193),
stored LPS-oriented like the TCIA exports so that the RAS re-orientation has work to do.  numpy + the in-tree NIfTI writer only.
"""
import argparse
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vs_seg_amd.data import nifti  # noqa: E402

NAMES = {"T1": ("vs_gk_t1_refT1.nii.gz", "vs_gk_seg_refT1.nii.gz"), "T2": ("vs_gk_t2_refT2.nii.gz", "vs_gk_seg_refT2.nii.gz")}


def make_case(path, names, shape, rng):
    os.makedirs(path, exist_ok=True)
    gx, gy, gz = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    c = [s * rng.uniform(0.35, 0.65) for s in shape]
    r = [max(2.0, s * rng.uniform(0.08, 0.14)) for s in shape]
    lab = (((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0).astype(np.uint8)
    img = (100 + 20 * rng.standard_normal(shape) + 150 * lab).astype(np.float32)
    aff = np.diag([-0.5, -0.5, 1.5, 1.0])  # LPS, anisotropic like the T1 scans (in-plane 0.5 mm, 1.5 mm slices)
    aff[:3, 3] = [20.0, 30.0, -10.0]
    nifti.write_nifti(os.path.join(path, names[0]), img, aff)
    nifti.write_nifti(os.path.join(path, names[1]), lab, aff)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", default="./data/VS_defaced/")
    ap.add_argument("--split", default=os.path.join(ROOT, "params", "split_debug.csv"))
    ap.add_argument("--size", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--dataset", default="T1", choices=["T1", "T2"])
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    n = 0
    with open(a.split) as f:
        for row in csv.reader(f):
            if len(row) >= 2:
                make_case(os.path.join(a.data_root, "input_data", row[0]), NAMES[a.dataset], tuple(a.size), rng)
                n += 1
    print(f"wrote {n} synthetic {a.dataset} cases of {tuple(a.size)} under {os.path.join(a.data_root, 'input_data')}")


if __name__ == "__main__":
    main()
