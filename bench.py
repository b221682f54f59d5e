#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: training patches/s (fwd + Dice_spvPA + bwd + Adam) on synthetic
384x128x128 bf16 patches, batch 4 per GPU, plus sliding-window volumes/s (512x512x120, roi 384x128x128, overlap 0.5).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task description): whole-job patches/s, the roofline of the dominant
kernel (HIP-event timed inside the timed region) and, at N=1, the CPU baseline (oracle on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATCH = (384, 128, 128)
HP = dict(channels=(16, 32, 48, 64, 80, 96), strides=((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)), kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
          sample_kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3)))
PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
FWD_BWD_GFLOP_PER_PATCH = 2053.9  # convolutions only, 2 FLOP/MAC


_T0 = time.perf_counter()
_PHASES = []


def phase(name):
    """Wall-clock bookkeeping of the bench command itself (stderr + the `bench_phases_s` field): the timed region is a fraction of a second,
    everything else here is set-up (import, plan lowering, parity check against the golden, CPU baseline)."""
    now = time.perf_counter()
    _PHASES.append((name, round(now - _T0, 2)))
    print(f"[bench {now - _T0:7.2f} s] {name}", file=sys.stderr, flush=True)


def synth_batch(batch, patch, seed, device):
    rng = np.random.default_rng(seed)
    img = torch.from_numpy(rng.standard_normal((batch, 1, *patch), dtype=np.float32))
    lab = np.zeros((batch, 1, *patch), np.float32)
    for b in range(1, batch):  # sample 0 stays all-background; the others carry a small blob (tumour << 1 % of the voxels)
        c = [int(rng.integers(s // 4, 3 * s // 4)) for s in patch]
        r = [max(2, s // 12) for s in patch]
        sl = tuple(slice(max(0, ci - ri), ci + ri) for ci, ri in zip(c, r))
        gx, gy, gz = np.meshgrid(*[np.arange(s.start, s.stop) for s in sl], indexing="ij")
        lab[b, 0][sl] = (((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0).astype(np.float32)
    return img.to(device), torch.from_numpy(lab).to(device)


def build_model(dtype, device, attention=True, dropout=0.1):
    import vs_seg_amd as V

    torch.manual_seed(0)
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, num_res_units=2, norm="batch", dropout=dropout, attention_module=attention, compute_dtype=dtype, **HP)
    return m.to(device)


def summarize_events(events):
    torch.cuda.synchronize()
    agg = {}
    for name, meta, e0, e1 in events:
        a = agg.setdefault(name, dict(ms=0.0, n=0, flops=0.0, bytes=0.0, kind=(meta or {}).get("kind", "hbm")))
        a["ms"] += e0.elapsed_time(e1)
        a["n"] += 1
        if meta:
            a["flops"] += meta.get("flops", 0.0)
            a["bytes"] += meta.get("bytes", 0.0)
    return agg


def parity_block(args, dev):
    """Parity of exactly what is timed below (compute dtype, batch, tuned launch plans) against the reference's own golden of one
    training step at 384x128x128 (tests/golden/net_train_b1_384x128x128.npz, produced by importing /root/reference in the build
    container): the golden's input is replicated over the batch (tests/parity_check.py explains why that is exact).  Dropout is
    off for this check (torch's dropout stream cannot be reproduced); it is on in the timed region."""
    import vs_seg_amd as V
    from tests import parity_check as PC
    from tests.helpers import seeded_weights_for

    g, seed, shape = PC.golden_train_case()
    torch.manual_seed(1000 + seed)
    m = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, num_res_units=2, norm="batch", dropout=0.0, attention_module=True, compute_dtype=args.dtype, **HP)
    m.load_state_dict(seeded_weights_for(m.state_dict(), seed))
    m = m.to(dev)
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    met = PC.train_step_metrics(m, loss_fn, batch=args.batch)
    bars = PC.BARS if args.dtype == "bf16" else PC.BARS_FP32
    out = {k: (round(v, 8) if isinstance(v, float) else v) for k, v in met.items()}
    out.update({"pass": PC.passes(met, bars), "failed_bars": PC.failures(met, bars), "bars": bars, "golden": "tests/golden/net_train_b1_384x128x128.npz (reference fwd+Dice_spvPA+bwd, fp32 CPU)",
                "config": f"{args.dtype}, batch {args.batch} (golden input replicated), tuned launch plans, dropout 0"})
    if not getattr(args, "no_c3_parity", False):  # BASELINE config 3 at its own size against its fixture (hard Dice + blended logits), same weights' seed as the fixture
        from tests.helpers import load

        torch.manual_seed(1000 + int(load("c3_swi_512x512x120.npz")["seed"]))
        m3 = V.UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, num_res_units=2, norm="batch", dropout=0.0, attention_module=True, compute_dtype=args.dtype, **HP)
        m3.load_state_dict(seeded_weights_for(m3.state_dict(), int(load("c3_swi_512x512x120.npz")["seed"])))
        c3 = PC.c3_metrics(m3.to(dev))
        c3.pop("out")
        cb = PC.C3_BARS["bf16" if args.dtype == "bf16" else "fp32"]
        out["c3_sliding_window"] = dict({k: round(v, 8) for k, v in c3.items()}, bars=cb, golden="tests/golden/c3_swi_512x512x120.npz (reference network per window x oracle blend)")
        out["c3_sliding_window"]["pass"] = bool(c3["dice_abs"] < cb["dice_abs"] and c3["logits_rel_l2"] < cb["logits_rel_l2"])
        del m3
    del m
    torch.cuda.empty_cache()
    return out


def roofline_table(full, peak, traffic):
    """Every kernel group of one event-timed training step: time, algorithmic flops / bytes, which roof bounds it (the MFMA peak when
    its arithmetic intensity is right of the ridge, else 8 TB/s HBM), the achieved fraction of that roof, and the PMC traffic."""
    ridge = peak * 1e12 / 8e12
    tot = sum(a["ms"] for a in full.values())
    rows = []
    for k, a in sorted(full.items(), key=lambda kv: -kv[1]["ms"]):
        row = dict(kernel=k, launches=a["n"], ms=round(a["ms"], 4), pct=round(100 * a["ms"] / tot, 2))
        if a["bytes"] > 0 or a["flops"] > 0:
            ai = a["flops"] / a["bytes"] if a["bytes"] else float("inf")
            tf, gbs = a["flops"] / a["ms"] / 1e9, a["bytes"] / a["ms"] / 1e6
            if a["kind"] == "mfma" and ai >= ridge:
                row.update(bound="mfma", achieved=round(tf, 1), unit="TFLOP/s", peak=peak, frac=round(tf / peak, 4))
            else:
                row.update(bound="hbm", achieved=round(gbs, 1), unit="GB/s", peak=8000.0, frac=round(gbs / 8000.0, 4))
            row.update(alg_gb=round(a["bytes"] / 1e9, 3), alg_tflop=round(a["flops"] / 1e12, 4))
            t = (traffic.get(k) or {}).get("hbm_bytes_per_launch")
            row["pmc_gb"] = round(t * a["n"] / 1e9, 3) if t else None
        rows.append(row)
    return rows


LIVE_EVERY = 4  # steps of the timed region between two steps whose dominant-kernel launches carry HIP events (see the timed loop)
CONV_GROUPS = ("igemm", "sconv", "cconv", "mconv", "dconv", "tconv", "gconv", "nconv", "wgrad", "mwgrad", "cwgrad", "mbwd")


def roofline_fractions(events, peak):
    """The step against its own unfused roofline, launch by launch: ideal time of a launch = max(algorithmic bytes / 8 TB/s, algorithmic flops /
    MFMA peak); step_roofline_frac = sum(ideal) / sum(measured) over every launch of one fully event-timed step, conv_stack_roofline_frac the same
    over the convolution launches (forward, data and weight gradients).  Launches without metadata (finalize kernels, memsets) count as pure overhead."""
    tot = ideal = ctot = cideal = 0.0
    n = 0
    for name, meta, e0, e1 in events:
        ms = e0.elapsed_time(e1)
        idl = max((meta or {}).get("bytes", 0.0) / 8e12, (meta or {}).get("flops", 0.0) / (peak * 1e12)) * 1e3
        tot += ms
        ideal += idl
        n += 1
        if name.split("<")[0] in CONV_GROUPS or name == "wgrad_narrow":
            ctot += ms
            cideal += idl
    return dict(step_roofline_frac=ideal / tot, conv_stack_roofline_frac=cideal / max(ctot, 1e-9), launches_per_step=n, step_ideal_ms=ideal, step_event_ms=tot, conv_stack_ideal_ms=cideal, conv_stack_event_ms=ctot)


def _cpu_step_fn():
    """step(shape) -> seconds of one fwd + Dice_spvPA + bwd + Adam step of the oracle (CPU restatement, pinned to the reference's goldens) at batch 1."""
    from oracle import vsseg_oracle as O

    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in O.seeded_state_dict(True, 0).items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-7)

    def step(shape):
        rng = np.random.default_rng(0)
        x = torch.from_numpy(rng.standard_normal((1, 1, *shape), dtype=np.float32))
        y = torch.zeros(1, 1, *shape)
        y[..., shape[0] // 3 : shape[0] // 2, shape[1] // 3 : shape[1] // 2, shape[2] // 3 : shape[2] // 2] = 1.0
        t0 = time.perf_counter()
        opt.zero_grad()
        logits, atts, _ = O.unet_forward(sd, x, train=True, attention_module=True, rng=torch.Generator().manual_seed(0))
        loss = O.dice_spvpa(logits, atts, y)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    return step


def cpu_baseline(budget_s=25.0):
    """The oracle timed on the host cores: one fwd+loss+bwd+Adam step, batch 1 (bounded sample)."""
    ncpu = os.cpu_count() or 1
    step = _cpu_step_fn()
    # A bounded sample (about 15 s of CPU work) on the cores the container may use, at most 16 threads per process: one warm-up step on a small patch, then one full
    # training step on half a benchmark patch, scaled by voxels.  (Rounds 2-4 found "all 256 host threads" 5-800x SLOWER than 16 and blamed oversubscription of the
    # oracle's small convolutions; round 5 read the cgroup: the box grants 16 cores of CPU time, 256 threads were simply throttled.)
    # the cores this process may actually USE: the container's cgroup CPU quota (cpu.max) — the GPU boxes of this pool show 256 hardware threads and grant 16 cores of
    # CPU time (measured: cpu.max = 1600000 100000), which is why 256 threads were 800x slower than 16 — and the affinity mask
    quota = float(ncpu)
    try:
        q_, per_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q_ != "max":
            quota = float(q_) / float(per_)
    except (OSError, ValueError):
        pass
    usable = max(1, min(ncpu, int(quota + 0.5)))
    cores = min(16, usable)
    torch.set_num_threads(cores)
    step((64, 64, 32))
    shape = (PATCH[0] // 2, PATCH[1], PATCH[2])
    t = step(shape)
    frac = (shape[0] * shape[1] * shape[2]) / (PATCH[0] * PATCH[1] * PATCH[2])
    # "all host cores, same box" (north_star) as a NUMBER: one process with every host thread oversubscribes the oracle's small convolutions (measured once: 163 s for a
    # 0.2 s step at 256 threads), so the box is filled the way a user would fill it — P = host_cores // 16 processes x 16 threads, each pinned to its own 16 cores, each
    # running one training step on its own quarter patch AT THE SAME TIME (they share the memory system; ~5 GB of RAM each, P capped by MemAvailable).  Throughput of
    # the box = P x 0.25 patch / slowest process.  Bounded: the children get 90 s.
    import subprocess

    q_shape = (PATCH[0] // 4, PATCH[1], PATCH[2])
    qfrac = (q_shape[0] * q_shape[1] * q_shape[2]) / (PATCH[0] * PATCH[1] * PATCH[2])
    try:
        avail_gb = next(int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")) / 1e6
    except (OSError, StopIteration):
        avail_gb = 32.0
    nproc = max(1, min(usable // cores, int(avail_gb // 8), 16))
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(ncpu))
    code = ("import os, sys, time, torch; sys.path.insert(0, %r); idx = int(sys.argv[1]); cpus = %r; mine = cpus[idx * %d:(idx + 1) * %d];\n"
            "try:\n    os.sched_setaffinity(0, mine)\nexcept (AttributeError, OSError):\n    pass\n"
            "import bench; torch.set_num_threads(%d); f = bench._cpu_step_fn(); f((32, 32, 16)); print('READY', flush=True); sys.stdin.readline(); t = f(%r); print('T', t, flush=True)"
            % (ROOT, cpus, cores, cores, cores, q_shape))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(nproc)]
    times, note = [], ""
    t_dead = time.perf_counter() + 90.0
    try:
        for pr in procs:  # every child has imported torch and run its warm-up step before any of them starts the timed one
            ln = pr.stdout.readline()
            if not ln.startswith("READY"):
                raise RuntimeError("a child process failed: " + pr.stderr.read()[-200:])
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        for pr in procs:
            out, _ = pr.communicate(timeout=max(1.0, t_dead - time.perf_counter()))
            times.append(next(float(ln.split()[1]) for ln in out.splitlines() if ln.startswith("T ")))
    except (subprocess.TimeoutExpired, RuntimeError, StopIteration) as e:
        note = f"{type(e).__name__} (90 s bound)" if isinstance(e, subprocess.TimeoutExpired) else f"{type(e).__name__}: {e}"[:160]
        times = []
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    tq = step(q_shape)  # the same quarter patch alone on the headline's 16 threads, for scale
    all_cores = dict(cores=nproc * cores, processes=nproc, threads_per_process=cores, host_cores=ncpu, usable_cores=usable, cgroup_cpu_quota=quota, value=(nproc * qfrac / max(times)) if times else None, unit="patches/s",
                     single_process_same_sample=qfrac / tq,
                     sample=(f"this container may use {usable} of the host's {ncpu} hardware threads (cgroup cpu.max / affinity): {nproc} processes x {cores} threads (each pinned to its own cores), one training step each on a {q_shape[0]}x{q_shape[1]}x{q_shape[2]} patch ({qfrac:.2f} of a benchmark patch) at the same time: "
                             + (f"slowest {max(times):.2f} s, fastest {min(times):.2f} s" if times else f"no number ({note})") + f"; one such process alone: {tq:.2f} s; scaled by voxels"))
    return dict(value=frac / t, unit="patches/s", cores=cores, host_cores=ncpu, kind="port", all_cores=all_cores,
                sample=f"1 training step (fwd+Dice_spvPA+bwd+Adam, fp32, batch 1) of the oracle on a {shape[0]}x{shape[1]}x{shape[2]} patch = {frac:.3f} of a 384x128x128 patch in {t:.2f} s, scaled by voxels")


def sharded_cases(model, n_cases, rank, world, dev, barrier, t2_shape=(448, 448, 80)):
    """BASELINE config 5's shape on the ranks that are there: `n_cases` synthetic T2-sized cases (the TCIA T2 matrix size is not recorded in the
    reference, SURVEY §8d: fixed at 448x448x80 here; z pads to 128 -> 12 windows at roi 384x128x128 / overlap 0.5) sharded round-robin over the
    ranks (`shard_indices`, as VSparams.run_inference does), hard Dice per case, the scores all-gathered INSIDE the timed region.  242 = the
    size of params/split_TCIA.csv.  Returns the `sharded_cases` block of the bench line (every rank takes part; all ranks return it)."""
    import vs_seg_amd as V
    from vs_seg_amd import parallel as DP

    was_training = model.training
    model.eval()
    mine = DP.shard_indices(n_cases, rank, world)
    pred = model.segmentation_predictor()  # `lambda x: model(x)[0]`, stream-safe: two window groups in flight (the product's default for this model)
    vols = [torch.from_numpy(np.random.default_rng(100 + i % 4).standard_normal((1, 1, *t2_shape), dtype=np.float32)).to(dev) for i in range(4)]  # 4 distinct volumes reused round-robin (HBM-resident inputs)
    lab = torch.zeros((1, 1, *t2_shape), device=dev)
    lab[..., 200:260, 210:250, 30:50] = 1.0
    on_device = world > 1 or DP._collectives_on()
    with torch.no_grad():
        V.compute_dice_score(V.sliding_window_inference(vols[0], PATCH, 1, pred, overlap=0.5, mode="gaussian"), lab)
        barrier()
        s0 = time.perf_counter()
        scores = torch.zeros(len(mine), dtype=torch.float32, device=dev)
        for j, ci in enumerate(mine):
            out = V.sliding_window_inference(vols[ci % 4], PATCH, 1, pred, overlap=0.5, mode="gaussian")
            scores[j] = V.compute_dice_score(out, lab).reshape(())
        all_scores = DP.all_gather_scalars(scores.double().cpu().tolist(), n_cases, device=dev if on_device else "cpu")  # one host read per rank, then the gather
        barrier()
        cdt = time.perf_counter() - s0
    cdt = DP.allreduce_max_float(cdt, dev)
    if was_training:
        model.train()
    return dict(volumes_per_sec=n_cases / cdt, cases=n_cases, volume="x".join(str(v) for v in t2_shape) + " (synthetic T2 shape)", roi="384x128x128", overlap=0.5, windows=12, mean_dice=float(np.mean(all_scores)),
                scores=[round(float(v), 6) for v in all_scores][:16], sharding="cases round-robin over ranks (shard_indices), Dice scalars all-gathered inside the timed region")


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment starts its own N ranks: the process replaces itself by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    (one rank per GPU over RCCL; rank 0 prints the one JSON line with n_gpus = N).  Fewer than N visible GPUs is an error, not a warning —
    except with VSSEG_SHARE_DEVICE=1 (+ VSSEG_DIST_BACKEND=gloo), the smoke-test mode in which the ranks share the GPUs that are there."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("VSSEG_SHARE_DEVICE") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible")
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {args.gpus} without a launcher environment: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU (BASELINE config 2)")
    ap.add_argument("--dtype", default=os.environ.get("VSSEG_DTYPE", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--swi-volumes", type=int, default=2, help="sliding-window volumes per GPU timed after the training steps (0 = skip)")
    ap.add_argument("--swi-cases", type=int, default=8, help="BASELINE config 5: N synthetic T2-shaped cases (448x448x80 -> 12 windows at roi 384x128x128 / overlap 0.5) sharded over the ranks, "
                    "hard Dice per case, scores all-gathered; 242 = the size of params/split_TCIA.csv; default 8 so that every driver line carries the block (0 = skip)")
    ap.add_argument("--dropout", type=float, default=0.1, help="dropout probability of the timed network (reference: 0.1; other values are experiments and are named in config.workload)")
    ap.add_argument("--fp32-steps", type=int, default=3, help="also time N steps of the fp32 parity mode (exact-fp32 MFMA, the mode that meets the 1e-3 logits bar) at the same batch; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity check of the benchmarked configuration against the reference golden")
    ap.add_argument("--no-c3-parity", action="store_true", help="skip the BASELINE config 3 part of the parity block (sliding window at 512x512x120 + hard Dice against its fixture)")
    ap.add_argument("--profile", action="store_true", help="print the per-kernel HIP-event breakdown of one step to stderr")
    args = ap.parse_args()
    self_launch(args)

    import torch.distributed as dist

    import vs_seg_amd as V
    from vs_seg_amd import parallel as DP

    rank, world, local = DP.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    phase("imports + process group done")
    parity = None
    if rank == 0 and not args.no_parity and args.batch >= 1:
        parity = parity_block(args, dev)  # before the timed region, same launch-plan signatures (dtype, batch) as the timed steps
        phase("parity check against the reference golden done")
    model = build_model(args.dtype, dev, dropout=args.dropout)
    model.reuse_output_buffers = True
    loss_fn = V.Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=True, hardness_weighting=True)
    opt = V.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    trainer = DP.DataParallelTrainer(model.train(), loss_fn, opt)
    img, lab = synth_batch(args.batch, PATCH, 1000 + rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(max(1, args.warmup)):
        trainer.step(img, lab)
    plan = next(p for k, p in model._engine.plans.items() if k[2])
    # one fully event-timed step: finds the dominant kernel (not part of the timed region)
    plan.timer = dict(only=None, events=[])
    trainer.step(img, lab)
    torch.cuda.synchronize()
    fractions = roofline_fractions(plan.timer["events"], PEAK[args.dtype])
    full = summarize_events(plan.timer["events"])
    dominant = max(full, key=lambda k: full[k]["ms"])
    phase("warm-up + event-timed step done")
    if args.profile and rank == 0:
        tot = sum(a["ms"] for a in full.values())
        print(f"--- per-kernel HIP-event time of one training step (sum {tot:.2f} ms) ---", file=sys.stderr)
        for k, a in sorted(full.items(), key=lambda kv: -kv[1]["ms"]):
            extra = f"{a['flops'] / a['ms'] / 1e9:9.1f} TFLOP/s" if a["flops"] else ""
            extra += f" {a['bytes'] / a['ms'] / 1e6:9.1f} GB/s(alg)" if a["bytes"] else ""
            print(f"{k:34s} n={a['n']:4d} {a['ms']:9.3f} ms {100 * a['ms'] / tot:5.1f}%  {extra}", file=sys.stderr)
        torch.cuda.synchronize()
        rows = sorted(((e0.elapsed_time(e1), name, (meta or {}).get("tag", ""), meta) for name, meta, e0, e1 in plan.timer["events"]), key=lambda r: -r[0])
        nrows = int(os.environ.get("VSSEG_PROFILE_ROWS", "45"))
        print(f"--- {nrows} slowest launches ---", file=sys.stderr)
        for ms, name, tag, meta in rows[:nrows]:
            tf = f"{meta['flops'] / ms / 1e9:7.1f} TF" if meta and meta.get("flops") else ""
            gb = f"{meta['bytes'] / ms / 1e6:7.0f} GB/s" if meta and meta.get("bytes") else ""
            print(f"{ms:8.3f} ms {name:18s} {tf} {gb} {tag}", file=sys.stderr)
    plan.timer = dict(only={dominant}, events=[])

    # The dominant kernel's launches are timed LIVE, by two HIP events each, in every LIVE_EVERY-th step of the timed region: an event record is a marker packet in the stream's
    # queue and delays the next kernel by ~5 us — 14 of them per step lengthen it by 0.19 ms (tools/time_step.py with VSSEG_TIME_LIVE) —, so the other steps run
    # exactly as a user's do (hipGraph replay of the forward, no events)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        plan.timer["active"] = i % LIVE_EVERY == 0
        loss = trainer.step(img, lab)
    barrier()
    dt = time.perf_counter() - t0
    dt = DP.allreduce_max_float(dt, dev)
    dom = summarize_events(plan.timer["events"])[dominant]
    plan.timer = None
    phase(f"timed region done ({dt:.2f} s)")
    loss_val = float(loss)
    patches_per_s = args.steps * args.batch * world / dt

    # ---- the same step in the fp32 parity mode (exact-fp32 MFMA: logits within 1e-3 of the reference, tests/test_gpu_network.py), not the headline
    fp32 = None
    if args.fp32_steps > 0 and args.dtype != "fp32" and world == 1:
        m32 = build_model("fp32", dev, dropout=args.dropout)
        m32.reuse_output_buffers = True
        t32 = DP.DataParallelTrainer.__new__(DP.DataParallelTrainer)  # single-rank timing: no parameter broadcast / all-reduce on the side
        t32.model, t32.loss_fn, t32.opt, t32.world, t32.fused_mean = m32.train(), loss_fn, V.Adam(m32.parameters(), lr=1e-4, weight_decay=1e-7), 1, True
        step32 = lambda: (t32.opt.zero_grad(), loss_fn(m32(img), lab).backward(), t32.opt.step())  # noqa: E731
        for _ in range(3):
            step32()
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(args.fp32_steps):
            step32()
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - s0) / args.fp32_steps
        fp32 = dict(ms_per_step=1e3 * d32, patches_per_sec=args.batch / d32, steps=args.fp32_steps, conv_stack_mfma_frac=FWD_BWD_GFLOP_PER_PATCH * args.batch / d32 / 1e3 / PEAK["fp32"],
                    note="compute_dtype fp32: fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s peak); one rank, same batch and patch")
        del m32, t32, step32
        torch.cuda.empty_cache()
        phase("fp32 parity-mode steps done")

    # ---- sliding-window inference (BASELINE configs 3/5): every rank blends its own volumes, no data-path collective
    swi = None
    if args.swi_volumes > 0:
        model.eval()
        vol = torch.from_numpy(np.random.default_rng(7 + rank).standard_normal((1, 1, 512, 512, 120), dtype=np.float32)).to(dev)
        pred = model.segmentation_predictor()  # the predictor the product's inference script builds (vs_seg_amd/params.py): `lambda w: model(w)[0]` without the per-window copies of the logits and attention maps
        def time_swi(swb, lanes=2):
            with torch.no_grad():
                for _ in range(3):  # every (lane, group size) plan has had its eager runs and its hipGraph capture (third run) before the timed region
                    V.sliding_window_inference(vol, PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
                barrier()
                s0 = time.perf_counter()
                for _ in range(args.swi_volumes):
                    V.sliding_window_inference(vol, PATCH, swb, pred, overlap=0.5, mode="gaussian", concurrent_groups=lanes)
                barrier()
                sdt = time.perf_counter() - s0
            return DP.allreduce_max_float(sdt, dev)

        sdt = time_swi(1)  # the reference's setting (ref:params/VSparams.py:571)
        swi = dict(volumes_per_sec=args.swi_volumes * world / sdt, ms_per_volume=1e3 * sdt / args.swi_volumes, volume="512x512x120", roi="384x128x128", overlap=0.5, windows=14, sw_batch_size=1,
                   mode="gaussian", sharding="volumes round-robin over ranks", concurrent_groups=2)
        t = time_swi(1, lanes=1)
        swi["serial_schedule"] = dict(volumes_per_sec=args.swi_volumes * world / t, ms_per_volume=1e3 * t / args.swi_volumes, note="concurrent_groups=1: one window's forward at a time")
        # per window: one eval forward of the 384x128x128 patch = 685.31 GFLOP and 5.63 GB of convolution-boundary bytes + the blend (2 RMW passes of 50.3 MB)
        wms = 1e3 * sdt / args.swi_volumes / 14
        swi["roofline"] = dict(ms_per_window=wms, alg_gflop_per_window=685.31, alg_gb_per_window=5.63, mfma_frac=685.31 / wms / PEAK[args.dtype], hbm_frac=5.63 / wms / 8.0,
                               note="unfused roofline of one window's eval forward: max(5.63 GB / 8 TB/s, 685.31 GFLOP / MFMA peak) = 0.70 ms")
        for swb in (2, 4):  # the same blend with 2 / 4 windows per predictor call (identical result: eval-mode BatchNorm has no cross-sample term)
            t = time_swi(swb)
            swi[f"sw_batch_size_{swb}"] = dict(volumes_per_sec=args.swi_volumes * world / t, ms_per_volume=1e3 * t / args.swi_volumes)
        # BASELINE config 5's other half — "patches scattered + logits RCCL all-gather": ONE volume's windows spread over the ranks (latency mode), the window logits
        # all-gathered, every rank blending in reference order (parallel.sharded_sliding_window_inference; bit-identical to one GPU).  Timed where there is a group to
        # gather through: world > 1, or one rank with VSSEG_FORCE_COLLECTIVES=1 (the RCCL branch on a single GPU).
        if world > 1 or DP._collectives_on():
            with torch.no_grad():
                for _ in range(2):
                    DP.sharded_sliding_window_inference(vol, PATCH, pred, overlap=0.5, mode="gaussian")
                barrier()
                s0 = time.perf_counter()
                for _ in range(args.swi_volumes):
                    DP.sharded_sliding_window_inference(vol, PATCH, pred, overlap=0.5, mode="gaussian")
                barrier()
                wdt = DP.allreduce_max_float(time.perf_counter() - s0, dev)
            per_rank = -(-14 // world)
            swi["window_sharded"] = dict(ms_per_volume=1e3 * wdt / args.swi_volumes, volumes_per_sec=args.swi_volumes / wdt, windows_per_rank=per_rank,
                                         allgather_mb=world * per_rank * 2 * PATCH[0] * PATCH[1] * PATCH[2] * 4 / 1e6,
                                         note="one 512x512x120 volume at a time: its 14 windows round-robin over the ranks, fp32 window logits all-gathered (RCCL), blended on every rank in reference order")
        model.train()

    # ---- BASELINE config 5: a TCIA-shaped synthetic T2 set, cases sharded over ranks, Dice scores all-gathered (timed with the gather)
    c5 = sharded_cases(model, args.swi_cases, rank, world, dev, barrier) if args.swi_cases > 0 else None

    # ---- data side: RandFlipd + RandSpatialCropd of image+label batches from volumes cached in HBM
    from vs_seg_amd.data.transforms import PatchSampler

    rng = np.random.default_rng(11 + rank)
    cached = [dict(image=torch.from_numpy(rng.standard_normal((448, 448, 128), dtype=np.float32)).to(dev), label=torch.zeros((448, 448, 128), device=dev)) for _ in range(2)]
    sampler = PatchSampler(cached, PATCH, flip_prob=0.5, seed=rank)
    sampler.sample([0, 1, 0, 1])
    barrier()
    s0 = time.perf_counter()
    for _ in range(20):
        sampler.sample([0, 1, 0, 1])
    barrier()
    sdt2 = time.perf_counter() - s0
    data_side = dict(patches_per_sec=20 * args.batch * world / sdt2, note="image+label crops of 384x128x128 from GPU-cached 448x448x128 volumes, one vsseg_crop_flip launch per batch of 4",
                     gb_per_sec=20 * args.batch * 2 * 2 * 4 * PATCH[0] * PATCH[1] * PATCH[2] / sdt2 / 1e9)
    del cached, sampler

    if rank != 0:
        return
    peak = PEAK[args.dtype]
    ridge = peak * 1e12 / 8e12  # FLOP per byte at which the MFMA roof meets the 8 TB/s HBM roof
    ai = dom["flops"] / dom["bytes"] if dom["bytes"] else float("inf")
    if dom["kind"] == "mfma" and dom["flops"] > 0 and ai >= ridge:
        ach = dom["flops"] / dom["ms"] / 1e9
        roof = dict(bound="mfma", kernel=dominant, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None, launches=dom["n"], avg_launch_ms=dom["ms"] / dom["n"],
                    alg_gflop_per_launch=dom["flops"] / dom["n"] / 1e9, alg_flop_per_byte=ai)
    else:  # the dominant kernel's launches sit left of the ridge (16/32-channel full-resolution layers): HBM is the roof
        ach = dom["bytes"] / dom["ms"] / 1e6
        roof = dict(bound="hbm", kernel=dominant, achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0, traffic=None, launches=dom["n"], avg_launch_ms=dom["ms"] / dom["n"],
                    alg_bytes_per_launch=dom["bytes"] / dom["n"], alg_gflop_per_launch=dom["flops"] / dom["n"] / 1e9, alg_flop_per_byte=ai if ai != float("inf") else None)
    # The timed region runs the product's default schedule: the weight gradients on a second HIP stream, concurrent with the data-gradient
    # chain (vs_seg_amd/engine.py, VSSEG_OVERLAP).  A kernel's live duration there includes whatever shared the GPU with it; the same kernel
    # group's time in the fully event-timed step — every launch alone on one stream — is reported beside it.
    roof["live_timing"] = (f"two HIP events around every launch of the kernel group in every {LIVE_EVERY}th step of the timed region ({dom['n']} launches of {args.steps} steps); the other steps carry no "
                           "events (an event record delays the stream's next kernel by ~5 us: 0.19 ms per step if every step carried them)")
    iso = full[dominant]
    roof["isolated"] = dict(ms=iso["ms"], achieved=(iso["flops"] / iso["ms"] / 1e9) if roof["bound"] == "mfma" else (iso["bytes"] / iso["ms"] / 1e6),
                            frac=((iso["flops"] / iso["ms"] / 1e9) / peak) if roof["bound"] == "mfma" else (iso["bytes"] / iso["ms"] / 1e6 / 8000.0),
                            note="same kernel group, one fully event-timed step, single stream (no concurrent weight gradient)")
    tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    traffic = {}
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile))
            roof["traffic"] = (traffic.get(dominant) or {}).get("hbm_bytes_per_launch")  # PMC FETCH_SIZE(x2)+WRITE_SIZE, see profiles/
        except Exception:
            pass
    res = {
        "metric": "train_patches_per_sec_fwd_bwd_384x128x128",
        "value": patches_per_s,
        "unit": "patches/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: 2.5D attention-UNet fwd + Dice_spvPA(attention+hardness) + bwd + Adam on random 384x128x128 patches, batch {args.batch} per GPU, dropout {args.dropout:g}, random-init weights",
                   "global_batch": args.batch * world, "patch": "384x128x128", "parallelism": f"dp{world}"},
        "conv_stack_mfma_frac": FWD_BWD_GFLOP_PER_PATCH * patches_per_s / world / 1e3 / peak,
        **{k: round(v, 4) if isinstance(v, float) else v for k, v in fractions.items()},
        "loss": loss_val,
        "roofline": roof,
        "roofline_table": roofline_table(full, peak, traffic),
        "parity": parity,
        "fp32_mode": fp32,
        "sliding_window": swi,
        "sharded_cases": c5,
        "data_side": data_side,
    }
    phase("sliding window / sharded cases / data side done")
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
        phase("CPU baseline done")
    res["bench_phases_s"] = _PHASES
    print(json.dumps(res))


if __name__ == "__main__":
    main()
