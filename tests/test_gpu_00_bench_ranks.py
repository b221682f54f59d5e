"""-m gpu: `python bench.py --gpus 2` exactly as the driver invokes it for N > 1 without a launcher environment (BASELINE configs 4-5).

Named to run FIRST among the GPU tests: its two ranks are separate processes sharing the test box's one GPU with this pytest process, and late in a session
the parent's caching allocator holds most of the 288 GB (a batch-4 training plan is ~30 GB per model under test)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


# ---- `python bench.py --gpus 2` as the driver invokes it: no launcher environment, bench.py starts its own ranks -----------------------------
def test_bench_gpus_2_starts_two_ranks_and_prints_one_line():
    """SURVEY §8(e) / BASELINE configs 4-5: `python bench.py --gpus N` must run N ranks.  Two ranks share the one GPU of the test box
    (VSSEG_SHARE_DEVICE=1, gloo): what is checked is the launch path and the `world > 1` branches of bench.py (barriers, max-over-ranks timing,
    sharded cases, rank-0-only output), not a measurement."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()  # the ranks are other processes on the same GPU: hand the memory cached by this session's earlier tests back to the driver
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSSEG_SHARE_DEVICE="1", VSSEG_DIST_BACKEND="gloo", VSSEG_NO_POISON="1", VSSEG_AUTOTUNE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VSSEG_FORCE_COLLECTIVES", "VSSEG_TUNE_CACHE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--swi-volumes", "1", "--swi-cases", "2"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp2"
    assert r["sharded_cases"]["cases"] == 2 and len(r["sharded_cases"]["scores"]) == 2
    assert r["value"] > 0 and r["sliding_window"]["volumes_per_sec"] > 0 and "cpu_baseline" not in r
    # the latency mode of BASELINE config 5 ("patches scattered + logits all-gather"): one volume's 14 windows over the two ranks, window logits gathered, timed
    ws = r["sliding_window"]["window_sharded"]
    assert ws["ms_per_volume"] > 0 and ws["windows_per_rank"] == 7 and abs(ws["allgather_mb"] - 2 * 7 * 2 * 384 * 128 * 128 * 4 / 1e6) < 1e-6
