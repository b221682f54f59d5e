"""-m gpu: the data-parallel path with a REAL model, two ranks sharing the one GPU of the test box (gloo, VSSEG_SHARE_DEVICE=1).

RCCL refuses two ranks on one device, so the collectives go through gloo (staged through the host) — what is verified is the
data-parallel semantics the 8-GPU runs rely on: the all-reduced gradient equals the sum of the two
single-rank gradients (per-rank BatchNorm statistics), fused Adam applies its mean so parameters stay identical across ranks,
and the window-sharded sliding-window inference is bit-identical to the single-process result.
"""
import os
import socket

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
