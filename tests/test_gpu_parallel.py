"""-m gpu: the data-parallel path with a REAL model, two ranks sharing the one GPU of the test box (gloo, VSSEG_SHARE_DEVICE=1).

RCCL refuses two ranks on one device, so the collectives go through gloo (staged through the host) — what is verified is the
data-parallel semantics the 8-GPU runs rely on: the all-reduced gradient equals the sum of the two
single-rank gradients (per-rank BatchNorm statistics), fused Adam applies its mean so parameters stay identical across ranks,
and the window-sharded sliding-window inference is bit-identical to the single-process result.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import vs_seg_amd as V  # noqa: E402
from oracle import vsseg_oracle as O  # noqa: E402
from tests.helpers import synth_input, synth_label  # noqa: E402

HP = O.HP
SEED, SHAPE, VOL, ROI = 61, (2, 1, 128, 64, 32), (1, 1, 96, 80, 20), (64, 32, 16)


def _model(dropout=0.0):
    torch.manual_seed(77)
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, channels=HP["channels"], strides=HP["strides"], kernel_sizes=HP["kernel_sizes"], sample_kernel_sizes=HP["sample_kernel_sizes"],
                        num_res_units=2, norm="batch", dropout=dropout, attention_module=True, compute_dtype="fp32")
    m.load_state_dict(O.seeded_state_dict(True, SEED))
    return m.to("cuda")


def _model_dt(dtype, dropout):
    torch.manual_seed(77)
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, channels=HP["channels"], strides=HP["strides"], kernel_sizes=HP["kernel_sizes"], sample_kernel_sizes=HP["sample_kernel_sizes"],
                        num_res_units=2, norm="batch", dropout=dropout, attention_module=True, compute_dtype=dtype)
    m.load_state_dict(O.seeded_state_dict(True, SEED))
    return m.to("cuda")


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("supervised", [True, False])
def test_fused_loss_train_step_is_bit_identical_to_the_autograd_route(dtype, supervised):
    """DataParallelTrainer.step with the loss writing its gradients where the backward reads them (train_forward_landing / forward_backward_into / backward_landed)
    against the same step through loss.backward(): losses, gradients, parameters and BatchNorm buffers after two steps, bit for bit.  `supervised=False` leaves the
    attention-map buffers without a gradient in both routes (after a supervised step has filled them: they must go back to zeros)."""
    from vs_seg_amd import parallel as DP

    x, y = _batch(0)
    res = []
    for fused in (True, False):
        m = _model_dt(dtype, 0.1)
        trainer = DP.DataParallelTrainer(m.train(), V.Dice_spvPA(to_onehot_y=True, softmax=True), V.Adam(m.parameters(), lr=1e-3, weight_decay=1e-7))
        assert trainer.fused_loss
        trainer.fused_loss = fused
        losses = [float(trainer.step(x, y))]
        trainer.loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=supervised)
        losses.append(float(trainer.step(x, y)))
        flat, gflat = m.flat_parameters()
        torch.cuda.synchronize()
        res.append((losses, gflat.clone(), flat.clone(), m._bflat.clone()))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)
    assert float(res[0][1].abs().max()) > 0


def test_late_weight_gradient_launches_change_the_order_and_nothing_else(monkeypatch):
    """The schedule of DESIGN 3.18 — the last weight-gradient launch (and the first unit's residual-convolution one) moved from the side stream to the end of the main
    stream's list, with the main stream's slab scratch — against the list order (VSSEG_LATE_WGRAD=0, VSSEG_EARLY_RES_WGRAD=0) and the early variant: two bf16 steps with
    dropout, losses / gradients / parameters / BatchNorm buffers bit for bit; and the moved launches are where they should be."""
    from vs_seg_amd import parallel as DP

    x, y = _batch(0)
    res = []
    for late, early, bound in (("0", "0", "0"), ("1", "late", "1"), ("2", "1", "1")):  # (bound: the side stream forks on events bound to the main-stream kernels, or on event records)
        monkeypatch.setenv("VSSEG_LATE_WGRAD", late)
        monkeypatch.setenv("VSSEG_EARLY_RES_WGRAD", early)
        monkeypatch.setenv("VSSEG_BOUND_FORKS", bound)
        m = _model_dt("bf16", 0.1)
        trainer = DP.DataParallelTrainer(m.train(), V.Dice_spvPA(to_onehot_y=True, softmax=True), V.Adam(m.parameters(), lr=1e-3, weight_decay=1e-7))
        losses = [float(trainer.step(x, y)) for _ in range(2)]
        flat, gflat = m.flat_parameters()
        torch.cuda.synchronize()
        plan = next(p for k, p in m._engine.plans.items() if k[2])
        assert (len(plan._fork_events) > 20) == (bound == "1")
        moved = [rec for rec in plan.bwd if len(rec) > 2 and rec[2].get("late")]
        assert len(moved) == {"0": 0, "1": 2, "2": 2}[late]
        if moved:
            assert plan.bwd[-len(moved):] == moved and not any(rec[2].get("side") for rec in moved)
        res.append((losses, gflat.clone(), flat.clone(), m._bflat.clone()))
    for other in res[1:]:
        assert other[0] == res[0][0]
        for a, b in zip(res[0][1:], other[1:]):
            assert torch.equal(a, b)


def _batch(rank):
    return synth_input(SEED + 10 * rank, SHAPE).cuda(), synth_label(SEED + 10 * rank, SHAPE).cuda()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VSSEG_SHARE_DEVICE="1", VSSEG_DIST_BACKEND="gloo",
                      VSSEG_AUTOTUNE="0", VSSEG_NO_POISON="1")
    from vs_seg_amd import parallel as DP

    r, w, local = DP.init_distributed()
    assert (r, w, local) == (rank, world, 0)
    torch.cuda.set_device(local)
    m = _model()
    if rank == 1:  # the trainer must broadcast rank 0's parameters
        with torch.no_grad():
            next(iter(m.parameters())).add_(1.0)
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True)
    trainer = DP.DataParallelTrainer(m.train(), loss_fn, V.Adam(m.parameters(), lr=1e-3, weight_decay=1e-7))
    x, y = _batch(rank)
    loss = trainer.step(x, y)
    flat, gflat = m.flat_parameters()
    res = dict(loss=float(loss), gsum=gflat.cpu().clone(), params=flat.cpu().clone(), bn=m._bflat.cpu().clone())
    DP.broadcast_buffers(m)
    res["bn_after_bcast"] = m._bflat.cpu().clone()
    m.eval()
    vol = synth_input(SEED + 5, VOL).cuda()
    with torch.no_grad():
        res["swi"] = DP.sharded_sliding_window_inference(vol, ROI, lambda t: m(t)[0], overlap=0.5, mode="gaussian").cpu().clone()
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_step_and_sharded_inference(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    [p.start() for p in procs]
    [p.join(600) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]

    # single-process references: each rank's batch through a fresh model (same weights), gradients summed by hand
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True)
    single, bn_single = [], []
    for r in range(world):
        m = _model().train()
        x, y = _batch(r)
        loss = loss_fn(m(x), y)
        loss.backward()
        single.append((float(loss), m.flat_parameters()[1].cpu().clone()))
        bn_single.append(m._bflat.cpu().clone())
    gsum = single[0][1] + single[1][1]
    for r in range(world):
        assert abs(res[r]["loss"] - single[r][0]) < 1e-6
        rel = float((res[r]["gsum"] - gsum).norm() / gsum.norm())
        assert rel < 1e-3, rel  # fp32 atomics: summation order only (1.2e-4 measured; training-mode BatchNorm amplifies it a little)
        torch.testing.assert_close(res[r]["bn"], bn_single[r], atol=1e-6, rtol=1e-6)  # BatchNorm statistics are per rank
    assert not torch.equal(res[0]["bn"], res[1]["bn"])
    # one all-reduce, one fused Adam with grad_scale 1/world: bit-identical parameters on both ranks, equal to Adam on the mean gradient
    assert torch.equal(res[0]["params"], res[1]["params"]) and torch.equal(res[0]["gsum"], res[1]["gsum"])
    ref = _model()
    p0 = ref.flat_parameters()[0].cpu().double()
    g = (res[0]["gsum"].double() / world) + 1e-7 * p0
    mhat, vhat = g, g * g  # first step: m/(1-b1) = g, v/(1-b2) = g^2
    want = p0 - 1e-3 * mhat / (vhat.sqrt() + 1e-8)
    torch.testing.assert_close(res[0]["params"].double(), want, atol=2e-6, rtol=1e-5)
    # buffers after the broadcast are rank 0's on both ranks
    assert torch.equal(res[0]["bn_after_bcast"], res[0]["bn"]) and torch.equal(res[1]["bn_after_bcast"], res[0]["bn"])
    # window-sharded inference: both ranks hold the same blended volume, bit-identical to one process running every window
    m = _model()
    with torch.no_grad():
        m.flat_parameters()[0].copy_(res[0]["params"].cuda())
        m._bflat.copy_(res[0]["bn"].cuda())
    m.invalidate_cache()
    m.eval()
    vol = synth_input(SEED + 5, VOL).cuda()
    with torch.no_grad():
        want_swi = V.sliding_window_inference(vol, ROI, 1, lambda t: m(t)[0], overlap=0.5, mode="gaussian").cpu()
    assert torch.equal(res[0]["swi"], res[1]["swi"])
    assert torch.equal(res[0]["swi"], want_swi)


# ---- the RCCL ("nccl") branch itself, on one GPU: a world-size-1 process group with VSSEG_FORCE_COLLECTIVES=1 ------------------------
def _rccl_worker(out_path, dtype):
    """One process, one GPU.  First the non-distributed results, then the same step / inference with a world-size-1 `nccl` process
    group whose collectives are really issued (parallel._collectives_on): init_process_group("nccl", device_id=...), the RCCL
    all-reduce of the flat gradient on the device buffer ordered after the side-stream weight gradients, the RCCL broadcast of
    parameters / buffers, and the RCCL all-gather of window logits and Dice scalars."""
    os.environ.update(VSSEG_AUTOTUNE="0", VSSEG_NO_POISON="1", VSSEG_OVERLAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VSSEG_DIST_BACKEND", "VSSEG_SHARE_DEVICE", "VSSEG_FORCE_COLLECTIVES"):
        os.environ.pop(k, None)
    import torch.distributed as dist

    from vs_seg_amd import parallel as DP

    def run():
        torch.manual_seed(77)
        m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, channels=HP["channels"], strides=HP["strides"], kernel_sizes=HP["kernel_sizes"], sample_kernel_sizes=HP["sample_kernel_sizes"],
                            num_res_units=2, norm="batch", dropout=0.0, attention_module=True, compute_dtype=dtype)
        m.load_state_dict(O.seeded_state_dict(True, SEED))
        m = m.to("cuda")
        loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True)
        trainer = DP.DataParallelTrainer(m.train(), loss_fn, V.Adam(m.parameters(), lr=1e-3, weight_decay=1e-7))
        x, y = _batch(0)
        losses = [float(trainer.step(x, y))]
        flat, gflat = m.flat_parameters()
        g1 = gflat.cpu().clone()  # the (all-reduced) gradient of the first step: same parameters in every run
        losses += [float(trainer.step(x, y)) for _ in range(2)]  # 3 steps: the third one replays the captured forward graph
        out = dict(losses=losses, g=g1, p=flat.cpu().clone())
        DP.broadcast_buffers(m)
        m.eval()
        vol = synth_input(SEED + 5, VOL).cuda()
        with torch.no_grad():
            swi = DP.sharded_sliding_window_inference(vol, ROI, lambda t: m(t)[0], overlap=0.5, mode="gaussian")
            out["swi"] = swi.cpu().clone()
            lab = torch.zeros(VOL, device="cuda")
            lab[..., 30:60, 20:50, 5:15] = 1.0
            d = float(V.compute_dice_score(swi, lab).reshape(()))
        out["dice"] = DP.all_gather_scalars([d], 1, device="cuda" if DP._collectives_on() else "cpu")
        out["mean"] = float(DP.allreduce_scalar_mean(torch.tensor([d], device="cuda")))
        return out

    plain = run()
    plain2 = run()  # a second non-distributed run: the step is run-to-run bit-identical, with or without the collectives
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), VSSEG_FORCE_COLLECTIVES="1")
    r, w, local = DP.init_distributed()
    assert (r, w, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl" and DP._collectives_on()
    forced = run()
    dist.destroy_process_group()
    torch.save(dict(plain=plain, plain2=plain2, forced=forced), out_path)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_rccl_branch_world_size_one_matches_the_non_distributed_run(tmp_path, dtype):
    ctx = mp.get_context("spawn")
    out = os.path.join(str(tmp_path), "rccl.pt")
    p = ctx.Process(target=_rccl_worker, args=(out, dtype))
    p.start()
    p.join(900)
    assert p.exitcode == 0, p.exitcode
    res = torch.load(out)
    a, a2, b = res["plain"], res["plain2"], res["forced"]
    # The collectives are identities at world size 1 and every reduction of a step is order-independent (fixed-point statistics, fixed-order slab
    # sums: tests/test_gpu_network.py::test_training_step_is_run_to_run_bit_identical), so the run through RCCL must reproduce the non-distributed
    # run BIT FOR BIT: a race between the side-stream weight gradients and the all-reduce that corrupts any part of the gradient fails here.
    for k in ("g", "p", "swi"):
        assert torch.equal(a[k], a2[k]), f"two non-distributed runs differ in {k}"
        assert torch.equal(a[k], b[k]), f"the world-size-1 RCCL run differs from the non-distributed run in {k}: max |d| {float((a[k] - b[k]).abs().max())}"
    assert a["losses"] == a2["losses"] == b["losses"], (a["losses"], a2["losses"], b["losses"])
    assert len(b["dice"]) == 1 and abs(b["dice"][0] - b["mean"]) < 1e-6 and a["dice"][0] == b["dice"][0]


# ---- BASELINE config 5's harness (bench.py --swi-cases) at 6 cases, Dice scalars gathered through RCCL --------------------------------
def _c5_worker(out_path):
    os.environ.update(VSSEG_NO_POISON="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), VSSEG_FORCE_COLLECTIVES="1", VSSEG_AUTOTUNE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VSSEG_DIST_BACKEND", "VSSEG_SHARE_DEVICE"):
        os.environ.pop(k, None)
    import bench
    from vs_seg_amd import parallel as DP

    rank, world, local = DP.init_distributed()
    dev = torch.device("cuda", local)
    model = bench.build_model("bf16", dev).eval()
    block = bench.sharded_cases(model, 6, rank, world, dev, torch.cuda.synchronize)
    # the same case computed directly, no collective anywhere
    vol = torch.from_numpy(np.random.default_rng(100).standard_normal((1, 1, 448, 448, 80), dtype=np.float32)).to(dev)
    lab = torch.zeros((1, 1, 448, 448, 80), device=dev)
    lab[..., 200:260, 210:250, 30:50] = 1.0
    with torch.no_grad():
        out = V.sliding_window_inference(vol, bench.PATCH, 1, lambda w: model(w)[0], overlap=0.5, mode="gaussian")
        direct = float(V.compute_dice_score(out, lab).reshape(()))
    torch.distributed.destroy_process_group()
    torch.save(dict(block=block, direct=direct, shape=tuple(out.shape)), out_path)


def test_config5_harness_six_cases_through_rccl(tmp_path):
    """bench.py's `sharded_cases` block (BASELINE config 5: T2-shaped cases sharded by shard_indices, hard Dice per case, scores all-gathered inside
    the timed region) at 6 cases of 448x448x80 = 12 windows each, with the gather going through a world-size-1 RCCL group."""
    ctx = mp.get_context("spawn")
    out = os.path.join(str(tmp_path), "c5.pt")
    p = ctx.Process(target=_c5_worker, args=(out,))
    p.start()
    p.join(900)
    assert p.exitcode == 0, p.exitcode
    res = torch.load(out)
    b = res["block"]
    assert b["cases"] == 6 and b["windows"] == 12 and len(b["scores"]) == 6 and b["volumes_per_sec"] > 0
    assert res["shape"] == (1, 2, 448, 448, 80)
    s = b["scores"]
    assert all(np.isfinite(v) and 0.0 <= v <= 1.0 for v in s)
    assert s[0] == s[4] and s[1] == s[5]  # cases 0 / 4 and 1 / 5 are the same volume: eval forward + blend are deterministic
    assert abs(s[0] - res["direct"]) < 1e-6
    assert abs(b["mean_dice"] - float(np.mean(s))) < 1e-6
