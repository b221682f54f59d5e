"""world_size-2 tests of the data-parallel host logic on CPU (gloo): index sharding, flat-gradient all-reduce + mean,
window-sharded sliding-window inference (gather order == reference window order), score all-gather.

The GPU kernels are not involved (there is no GPU here); the predictor / crop are torch-on-CPU stand-ins, and the blend that
consumes the gathered windows is the oracle's.  What is verified is exactly what the 8-GPU run relies on: every rank ends
up with every window's logits in the reference order, so the sequential fp32 blend is identical to the single-process one.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vsseg_oracle as O
from vs_seg_amd import parallel as DP
from vs_seg_amd.inferers import window_geometry


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _predictor(w):
    return torch.cat([w * 2.0 + 1.0, torch.tanh(w) - 0.5], 1)


def _crop(vol, wins, roi, pad_before):
    pad = []
    for k in range(2, -1, -1):
        pad += [pad_before[k], max(roi[k] - vol.shape[2 + k], 0) - pad_before[k]]
    x = torch.nn.functional.pad(vol, pad)
    return torch.cat([x[b : b + 1, :, s[0] : s[0] + roi[0], s[1] : s[1] + roi[1], s[2] : s[2] + roi[2]] for b, s in wins])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = DP.init_distributed("gloo")
    assert (r, w) == (rank, world) and DP.world_size() == world and DP.get_rank() == rank
    res = {}
    # 1. shards partition the index set
    res["shard"] = DP.shard_indices(11)
    # 2. flat-gradient all-reduce (sum) + 1/world scale == mean of the per-rank gradients
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    DP.allreduce_gradients(g)
    res["gsum"] = g.clone()
    flat = torch.full((10,), float(rank))
    DP.broadcast_parameters(flat, 0)
    res["bcast"] = flat.clone()
    res["lossmean"] = float(DP.allreduce_scalar_mean(torch.tensor(float(rank + 1))))
    # 3. window-sharded sliding window: logits of every window, in reference order, on every rank
    torch.manual_seed(0)
    vol = torch.randn(1, 1, 40, 36, 20)
    wins, logits, (roi, padded, pad_before) = DP.sharded_window_logits(vol, (16, 16, 32), _predictor, 0.5, _crop)
    res["wins"], res["logits"] = wins, logits.clone()
    # 4. per-case scores gathered in case order
    mine = [10.0 * i for i in DP.shard_indices(7)]
    res["scores"] = DP.all_gather_scalars(mine, 7)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_results():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    return dict(out)


def test_shards_partition_indices(two_rank_results):
    a, b = two_rank_results[0]["shard"], two_rank_results[1]["shard"]
    assert a == [0, 2, 4, 6, 8, 10] and b == [1, 3, 5, 7, 9]


def test_gradient_allreduce_is_a_sum_and_mean_via_scale(two_rank_results):
    want = torch.arange(1000, dtype=torch.float32) * 3  # (1 + 2) * base
    for r in (0, 1):
        torch.testing.assert_close(two_rank_results[r]["gsum"], want)
        assert two_rank_results[r]["lossmean"] == 1.5
        assert float(two_rank_results[r]["bcast"].abs().max()) == 0.0  # parameters come from rank 0


def test_window_sharded_inference_matches_single_process(two_rank_results):
    torch.manual_seed(0)
    vol = torch.randn(1, 1, 40, 36, 20)
    roi, ov = (16, 16, 32), 0.5
    want, starts = O.sliding_window_inference(vol, roi, 1, _predictor, overlap=ov, mode="gaussian", return_windows=True)
    for r in (0, 1):
        wins, logits = two_rank_results[r]["wins"], two_rank_results[r]["logits"]
        assert [s for _, s in wins] == starts == window_geometry((40, 36, 20), roi, ov)[4]
        # blend the gathered windows sequentially in reference order (oracle blend) -> identical to the single-process result
        it = iter(logits)
        got = O.sliding_window_inference(vol, roi, 1, lambda w: next(it)[None], overlap=ov, mode="gaussian")
        np.testing.assert_array_equal(got.numpy(), want.numpy())
    np.testing.assert_array_equal(two_rank_results[0]["logits"].numpy(), two_rank_results[1]["logits"].numpy())


def test_scores_gathered_in_case_order(two_rank_results):
    for r in (0, 1):
        assert two_rank_results[r]["scores"] == [0.0, 10.0, 20.0, 30.0, 40.0, 50.0, 60.0]


def _lower_worker(rank, world, port, out, fail, modes):
    """Engine._lower_rank0_first with a stand-in Plan (no GPU here): what the other ranks do when rank 0 fails while lowering, or runs another VSSEG_AUTOTUNE mode."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VSSEG_AUTOTUNE=modes[rank])
    DP.init_distributed("gloo")
    from vs_seg_amd import engine as E

    class FakePlan:
        def __init__(self, eng, n, dims, train):
            if fail and dist.get_rank() == 0:
                raise MemoryError("out of memory while measuring plans")
            self.key = (n, tuple(dims), train)

    real = E.Plan
    E.Plan = FakePlan
    eng = E.Engine.__new__(E.Engine)
    eng.dry_run = False
    try:
        pl = eng._lower_rank0_first((2, (64, 64, 32), True, 0))
        out[rank] = ("ok", pl.key)
    except BaseException as e:  # noqa: BLE001
        out[rank] = (type(e).__name__, str(e))
    finally:
        E.Plan = real
        dist.destroy_process_group()


@pytest.mark.parametrize("fail,modes", [(True, ("1", "1")), (False, ("1", "0")), (False, ("1", "1"))])
def test_rank0_lowering_failure_reaches_every_rank(fail, modes):
    """rank 0 lowers the training plan first and broadcasts its plan choices — if it fails (or the ranks disagree on VSSEG_AUTOTUNE) every rank must
    raise at once instead of blocking in the broadcast until the collective times out."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_lower_worker, args=(world, port, out, fail, modes), nprocs=world, join=True)
    if fail:
        assert out[0][0] == "MemoryError" and out[1][0] == "RuntimeError" and "rank 0 failed while lowering" in out[1][1] and "out of memory" in out[1][1]
    elif modes[0] != modes[1]:
        assert out[0][0] == "ok" and out[1][0] == "RuntimeError" and "VSSEG_AUTOTUNE differs" in out[1][1]
    else:
        assert out[0] == out[1] == ("ok", (2, (64, 64, 32), True))
