"""-m gpu: the data-side HIP kernels against their numpy restatement, and the drop-in driver end to end
(VSparams: NIfTI cases → GPU cache → train epochs → on-device validation → sliding-window inference → NIfTI export)."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd.data import nifti  # noqa: E402
from vs_seg_amd.data import transforms as T  # noqa: E402
from oracle import data_oracle as DO  # noqa: E402


def test_normalize_intensity_matches_host():
    lib = L.lib()
    rng = np.random.default_rng(0)
    for shape, scale in (((67, 45, 23), 300.0), ((8, 8, 8), 0.0)):
        v = (rng.random(shape) * scale + 40.0).astype(np.float32)
        x = torch.from_numpy(v).cuda()
        y, acc = torch.empty_like(x), torch.zeros(2, dtype=torch.float64, device="cuda")
        L.check(lib.vsseg_normalize_intensity(x.data_ptr(), y.data_ptr(), x.numel(), acc.data_ptr(), torch.cuda.current_stream().cuda_stream))
        np.testing.assert_allclose(y.cpu().numpy(), DO.host_normalize_intensity(v), atol=2e-5)


@pytest.mark.parametrize("dims,origin,c,dtype", [
    ((16, 12, 24), (0, 0, 0), 1, torch.bfloat16),      # the network input of a training step: plain cast into the compact layout (four z per thread)
    ((16, 12, 24), (-3, 5, -2), 1, torch.float32),     # a sliding-window crop hanging over three faces of the volume (zero padding), unaligned rows
    ((16, 12, 24), (4, -7, 6), 8, torch.bfloat16),     # zero-extended to one 8-channel group
    ((16, 12, 22), (2, 1, 3), 1, torch.bfloat16),      # z % 4 != 0: the scalar kernel
    ((8, 8, 12), (-2, -2, 13), 1, torch.float32),      # z range partly / rows entirely outside
])
def test_stage_input_crops_pads_and_casts(dims, origin, c, dtype):
    """vsseg_stage_input (the window crop of the sliding-window predictor, ref:params/VSparams.py:553-567, and the cast of the network input): out[b][q] = vol[b][q + origin], zero
    outside the volume, channel 0 of a c-channel row — both kernels (four z-consecutive voxels per thread / one voxel per thread), bit-exact against a numpy restatement."""
    lib = L.lib()
    rng = np.random.default_rng(5)
    n, src = 2, (19, 17, 29)
    vol = rng.standard_normal((n, *src)).astype(np.float32)
    ref = np.zeros((n, *dims), np.float32)
    for ax_x in range(dims[0]):
        gx = ax_x + origin[0]
        if not 0 <= gx < src[0]:
            continue
        for ax_y in range(dims[1]):
            gy = ax_y + origin[1]
            if not 0 <= gy < src[1]:
                continue
            z0, z1 = max(0, -origin[2]), min(dims[2], src[2] - origin[2])
            if z1 > z0:
                ref[:, ax_x, ax_y, z0:z1] = vol[:, gx, gy, z0 + origin[2]:z1 + origin[2]]
    out = torch.full((n, *dims, c), 7.0, dtype=dtype, device="cuda")
    dst = L.Tensor(out.data_ptr(), L.F32 if dtype == torch.float32 else L.BF16, c, c, n, *dims)
    L.check(lib.vsseg_stage_input(torch.from_numpy(vol).cuda().data_ptr(), n, L.i3(src), L.i3(origin), dst, torch.cuda.current_stream().cuda_stream))
    got = out.float().cpu().numpy()
    want = torch.from_numpy(ref).to(dtype).float().numpy()
    np.testing.assert_array_equal(got[..., 0], want)
    assert not got[..., 1:].any()


def _case(vol, lab):
    return {"image": torch.from_numpy(vol).cuda(), "label": torch.from_numpy(lab).cuda()}


def test_patch_sampler_matches_host_flip_crop_including_padding():
    rng = np.random.default_rng(1)
    cases, host = [], []
    for shape in ((40, 36, 20), (33, 50, 16), (64, 64, 24)):
        v, l = rng.standard_normal(shape).astype(np.float32), (rng.random(shape) > 0.9).astype(np.float32)
        cases.append(_case(v, l))
        host.append((v, l))
    roi = (32, 32, 16)
    s = T.PatchSampler(cases, roi, flip_prob=0.5, seed=7)
    seen_flip = set()
    for idx in ([0, 1, 2], [2, 2, 0, 1], [1]):
        img, lab = s.sample(idx)
        assert img.shape == (len(idx), 1, *roi) and lab.shape == img.shape
        for b, i in enumerate(idx):
            flip, start = s.last_draws[b]
            seen_flip.add(flip)
            np.testing.assert_array_equal(img[b, 0].cpu().numpy(), DO.host_flip_crop(host[i][0], flip, start, roi))
            np.testing.assert_array_equal(lab[b, 0].cpu().numpy(), DO.host_flip_crop(host[i][1], flip, start, roi))
    assert seen_flip == {True, False}
    # negative origin / roi larger than the volume = SpatialPadd's zero padding, through the C ABI directly
    lib = L.lib()
    v = host[0][0]
    jobs = (L.CropJob * 1)()
    jobs[0].src, jobs[0].sdims, jobs[0].origin, jobs[0].flip_x = cases[0]["image"].data_ptr(), L.i3(v.shape), L.i3((-3, -2, -1)), 1
    jb = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).cuda()
    out = torch.empty((48, 40, 24), device="cuda")
    L.check(lib.vsseg_crop_flip(jb.data_ptr(), 1, out.data_ptr(), L.i3((48, 40, 24)), torch.cuda.current_stream().cuda_stream))
    want = np.pad(v[::-1], ((3, 5), (2, 2), (1, 3)))
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def _write_cases(root, n, rng):
    """Small synthetic T1-like cases with a bright ellipsoid 'tumour', stored LPS-oriented so that Orientationd has work to do."""
    for i in range(n):
        d = os.path.join(root, "input_data", f"vs_gk_{i + 1}")
        os.makedirs(d, exist_ok=True)
        shape = (72 + 8 * (i % 2), 64, 20)
        gx, gy, gz = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        c = [shape[0] * (0.4 + 0.05 * i), shape[1] * 0.5, shape[2] * 0.5]
        lab = (((gx - c[0]) / 9) ** 2 + ((gy - c[1]) / 8) ** 2 + ((gz - c[2]) / 4) ** 2 <= 1.0).astype(np.uint8)
        img = (100 + 20 * rng.standard_normal(shape) + 150 * lab).astype(np.float32)
        aff = np.diag([-0.5, -0.5, 1.5, 1.0])  # LPS
        aff[:3, 3] = [20.0, 30.0, -10.0]
        nifti.write_nifti(os.path.join(d, "vs_gk_t1_refT1.nii.gz"), img, aff)
        nifti.write_nifti(os.path.join(d, "vs_gk_seg_refT1.nii.gz"), lab, aff)
    split = os.path.join(root, "split.csv")
    with open(split, "w") as f:
        for i in range(n):
            f.write(f"vs_gk_{i + 1},{'training' if i < n - 2 else ('validation' if i == n - 2 else 'test')}\n")
    return split


def test_vsparams_train_validate_infer_export_end_to_end(tmp_path):
    from vs_seg_amd.params import VSparams

    rng = np.random.default_rng(3)
    root = str(tmp_path)
    split = _write_cases(root, 5, rng)
    argv = ["--split", split, "--data_root", root, "--results_folder_name", "t", "--train_batch_size", "2", "--compute_dtype", "fp32", "--num_epochs", "2"]
    p = VSparams(argparse.ArgumentParser(), argv)
    p.pad_crop_shape = p.pad_crop_shape_test = [64, 64, 16]  # the reference's debug sizes, smaller still
    p.sliding_window_inferer_roi_size = [64, 64, 16]
    p.create_results_folders()
    p.set_up_logger("training_log.txt")
    p.log_parameters()
    train_files, val_files, test_files = p.load_T1_or_T2_data()
    assert (len(train_files), len(val_files), len(test_files)) == (3, 1, 1)
    ttf, vtf, stf = p.get_transforms()
    train_loader, val_loader, test_loader = p.cache_transformed_train_data(train_files, ttf), p.cache_transformed_val_data(val_files, vtf), p.cache_transformed_test_data(test_files, stf)
    # cached head of the chain: RAS (the LPS file got both in-plane axes flipped), normalised, padded
    c0 = train_loader.cases[0]
    raw, aff, _ = nifti.read_nifti(train_files[0]["image"])
    np.testing.assert_array_equal(c0["image_meta"]["ornt"], [[0, -1], [1, -1], [2, 1]])
    want = DO.host_spatial_pad(DO.host_normalize_intensity(raw[::-1, ::-1]), p.pad_crop_shape)
    np.testing.assert_allclose(c0["image"].cpu().numpy(), want, atol=2e-5)
    batches = list(train_loader)
    assert [b["image"].shape[0] for b in batches] == [2, 1] and batches[0]["image"].shape[1:] == (1, 64, 64, 16)

    model, loss_fn = p.set_and_get_model(), p.set_and_get_loss_function()
    opt = p.set_and_get_optimizer(model)
    p.val_interval = 1
    epoch_losses, metrics = p.run_training_algorithm(model, loss_fn, opt, train_loader, val_loader)
    assert len(epoch_losses) == 2 and len(metrics) == 2 and all(np.isfinite(epoch_losses)) and all(0.0 <= m <= 1.0 for m in metrics)
    assert os.path.isfile(os.path.join(p.model_path, "best_metric_model.pth")) and os.path.isfile(os.path.join(p.model_path, "last_epoch_model.pth"))
    # validation accounting: mean Dice of the single case, loss doubled (reference quirk)
    model.eval()
    with torch.no_grad():
        vb = next(iter(p.cache_transformed_val_data(val_files, vtf)))
        out = model(vb["image"])
        d1, l1 = float(p.compute_dice_score(out[0], vb["label"])), float(loss_fn(out, vb["label"]))
    m, lv = p.validate(model, loss_fn, p.cache_transformed_val_data(val_files, vtf))
    assert abs(m - d1) < 1e-6 and abs(lv - 2 * l1) < 1e-5

    # checkpoint round trip + inference + export in the ORIGINAL orientation
    model2 = p.load_trained_state_of_model(p.set_and_get_model())
    scores = p.run_inference(model2, test_loader)
    assert scores.shape == (1,) and 0.0 <= scores[0] <= 1.0
    out_dir = os.path.join(p.results_folder_path, "inferred_segmentations_nifti", "vs_gk_5", "vs_gk_seg_refT1")
    seg, saff, _ = nifti.read_nifti(os.path.join(out_dir, "vs_gk_seg_refT1.nii.gz"))
    lab, laff, _ = nifti.read_nifti(test_files[0]["label"])
    assert seg.shape == lab.shape and set(np.unique(seg)) <= {0.0, 1.0}
    np.testing.assert_allclose(saff, laff, atol=1e-5)
    # the exported mask is the argmax of the same sliding-window output, un-oriented
    with torch.no_grad():
        data = next(iter(test_loader))
        from vs_seg_amd import sliding_window_inference

        o = sliding_window_inference(data["image"], p.sliding_window_inferer_roi_size, 1, lambda x: model2(x)[0], mode="gaussian")
    want_seg = nifti.from_ras(torch.argmax(o, 1)[0].cpu().numpy().astype(np.uint8), data["label_meta_dict"]["ornt"])
    np.testing.assert_array_equal(seg, want_seg.astype(np.float32))


def test_vs_train_debug_script_runs_end_to_end(tmp_path):
    """BASELINE config 1 plumbing: `VS_train.py --debug` then `VS_inference.py --debug` as the user runs them, on the six synthetic
    64^3 cases of params/split_debug.csv written by tools/make_debug_data.py (cwd = a scratch copy of the layout the scripts expect)."""
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "params"))
    for f in ("split_debug.csv", "split_TCIA.csv"):
        shutil.copy(os.path.join(root, "params", f), os.path.join(work, "params", f))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), VSSEG_AUTOTUNE="0")
    run = lambda *a: subprocess.run([sys.executable, *a], cwd=work, env=env, capture_output=True, text=True, timeout=900)  # noqa: E731
    r = run(os.path.join(root, "tools", "make_debug_data.py"), "--data_root", "./data/VS_defaced/")
    assert r.returncode == 0, r.stderr[-2000:]
    r = run(os.path.join(root, "VS_train.py"), "--debug", "--num_epochs", "2", "--compute_dtype", "fp32")
    assert r.returncode == 0, r.stderr[-3000:]
    log = open(os.path.join(work, "data/VS_defaced/results/debug/logs/training_log.txt")).read()
    assert "Check the transforms on the first validation set image and label" in log and "Validation image shape = torch.Size([128, 128, 32])" in log
    assert "epoch 2 average loss" in log and "current epoch 2 current mean dice" in log and "Train completed" in log
    assert os.path.isfile(os.path.join(work, "data/VS_defaced/results/debug/model/best_metric_model.pth"))
    r = run(os.path.join(root, "VS_inference.py"), "--debug", "--compute_dtype", "fp32")
    assert r.returncode == 0, r.stderr[-3000:]
    tlog = open(os.path.join(work, "data/VS_defaced/results/debug/logs/test_log.txt")).read()
    assert "mean_dice_score" in tlog
    segs = [f for _, _, fs in os.walk(os.path.join(work, "data/VS_defaced/results/debug/inferred_segmentations_nifti")) for f in fs]
    assert len(segs) == 2


def test_argmax_export_matches_torch_argmax_bit_exactly():
    from vs_seg_amd.inferers import argmax_segmentation

    torch.manual_seed(5)
    out = torch.randn(2, 2, 40, 24, 12, device="cuda")
    out[0, :, 3, 4, 5] = 0.25  # a tie: class 0, like torch.argmax
    out[1, 1, 7] = out[1, 0, 7]
    got = argmax_segmentation(out)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (2, 40, 24, 12)
    assert torch.equal(got, torch.argmax(out, dim=1).to(torch.uint8))
    cl = out.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)  # the network's channels-last logits view
    assert torch.equal(argmax_segmentation(cl), got)
