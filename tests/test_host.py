"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, plans lower without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from vs_seg_amd import _lib as L
from vs_seg_amd.engine import Engine, ParamLayout
from vs_seg_amd.graph import build_program, state_manifest


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    header = open("include/vsseg_hip.h").read()
    header = re.sub(r"static inline[^{]*\{.*?\n\}", "", header, flags=re.S)  # (the inline fixed-point encode / decode helpers are not exports)
    declared = set(re.findall(r"\b(vsseg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vsseg_version() >= 1


def test_invalid_descriptor_is_rejected_with_message():
    lib = L.lib()
    d = L.IgemmDesc()
    assert lib.vsseg_igemm_lds_bytes(ctypes.byref(d)) < 0
    assert b"vsseg_igemm" in lib.vsseg_last_error()


@pytest.mark.parametrize("att", [True, False])
@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_plans_lower_on_cpu(att, dt):
    lay = ParamLayout(state_manifest(att))
    flat = torch.zeros(lay.n_param)
    eng = Engine(att, dt, flat, torch.zeros_like(flat), torch.zeros(lay.n_buf), torch.zeros(lay.n_cnt, dtype=torch.int64), lay, dry_run=True)
    tr = eng.plan(2, (64, 32, 16), True)
    ev = eng.plan(1, (64, 32, 16), False)
    prog = build_program(att)
    n_conv = len(prog.layers)
    igemms = [r for r in tr.fwd if r[0] is eng.lib.vsseg_igemm]
    # one igemm launch per convolution: the output-parity classes of the transposed convolutions share one launch (planner.class_split_plans);
    # the merged residual conv has none and the 1-channel attention map convolutions run on the narrow kernel
    assert n_conv - 2 <= len(igemms) <= n_conv
    assert sum(1 for r in igemms if r[1][0]._obj.class_split == 8) == 3 and sum(1 for r in tr.bwd if r[0] is eng.lib.vsseg_igemm and r[1][0]._obj.class_split == 8) == 3
    # (bf16: the stride-1 3x3x1 blocks of the finest levels run BatchNorm-backward apply + data gradient + weight gradient as ONE launch, csrc/mbwd.hip)
    n_fused = sum(1 for r in tr.bwd if r[0] in (eng.lib.vsseg_conv_bwd_fused, eng.lib.vsseg_wgrad_narrow_bn))
    assert (n_fused >= 2) == (dt == "bf16")
    assert sum(1 for r in tr.bwd if r[0] in (eng.lib.vsseg_wgrad, eng.lib.vsseg_wgrad_narrow)) + n_fused == n_conv - len(tr.merged)  # the final 1x1x1 residual conv is merged into the final 3x3x1 conv
    assert len(tr.merged) == 1 and sum(1 for r in tr.bwd if r[0] is eng.lib.vsseg_merge_residual_grads) == 1
    assert len(ev.bwd) == 0 and len(ev.fwd) < len(tr.fwd)
    # every dropout launch reads the seed through the plan's device scalar (fixed arguments: the lists can be captured as hipGraphs)
    seeded = [r for lst in (tr.fwd, tr.bwd) for r in lst if tr.seed_dev.data_ptr() in [a for a in r[1] if isinstance(a, int)]]
    assert len(seeded) == 3 * sum(1 for L_ in prog.layers if L_.has_bn) - n_fused and len(tr.bwd_pre) == 1 and len(tr.ext_slots) == 1  # (a fused launch reads the stored keep-mask: no seed)
    # the LDS request the C side computes equals the planner's (mirrored formula)
    for rec in igemms[:10]:
        d = rec[1][0]._obj
        assert eng.lib.vsseg_igemm_lds_bytes(ctypes.byref(d)) > 0


def test_bad_spatial_size_raises():
    lay = ParamLayout(state_manifest(True))
    flat = torch.zeros(lay.n_param)
    eng = Engine(True, "bf16", flat, torch.zeros_like(flat), torch.zeros(lay.n_buf), torch.zeros(lay.n_cnt, dtype=torch.int64), lay, dry_run=True)
    with pytest.raises(ValueError):
        eng.plan(1, (48, 32, 8), False)


def test_shipped_tuned_plans_still_name_existing_candidates():
    """vs_seg_amd/tuned_gfx950.json (plans measured on an MI355X for the benchmark shapes) is only useful while its entries still
    match what `planner.candidate_plans` generates: a planner / kernel change that invalidates them must be noticed (the engine
    would silently fall back to measuring every launch at start-up)."""
    import ast
    import json
    import os

    from vs_seg_amd import engine as E
    from vs_seg_amd import planner as P

    data = json.load(open(E.TUNED_DEFAULTS))
    ig = {k: v for k, v in data.items() if not k.startswith(("wgrad", "use_cs", "fbwd"))}  # (use_cs|...: per-class or class-split launch, 0 / 1; fbwd|...: tile of a fused backward launch)
    assert len(ig) > 150 and sum(1 for k in data if k.startswith("wgrad")) >= 40
    checked = hits = 0
    for key, choice in list(ig.items())[::5]:
        parts = key.split("|")
        kind, fold = parts[0], int(parts[1][1:])
        wshape = ast.literal_eval(parts[2][1:])
        is_, os_, oo = (ast.literal_eval(t) for t in re.findall(r"\([^)]*\)", parts[3]))
        q = ast.literal_eval(parts[4][1:])
        es, kc = int(parts[6][2:]), int(parts[7][2:])
        acc, res, two = int(parts[8][3:]), int(parts[9][3:]), parts[11][3:]
        kernel = tuple(wshape[2:])
        cls = next((c for s in ((1, 1, 1), (2, 2, 1), (2, 2, 2)) for c in P.lattice_classes(kind, kernel, s) if (c.is_, c.os, c.oo) == (is_, os_, oo)), None)
        assert cls is not None, key
        aux_es = es if (acc or res) else 0
        if key.endswith("|cs"):  # every parity class in one launch
            kreal, nreal = P.gemm_dims(kind, wshape)
            cands = P.class_split_plans(kind, wshape, kernel, os_, q, es, kc, nreal, kreal, aux_es=aux_es)
        elif fold:
            cands = P.folded_candidate_plans(kind, wshape, cls, (q[0], q[1], q[2] * fold), es, aux_es=aux_es)
        else:
            cands = P.candidate_plans(kind, wshape, cls, q, es, kc_pad=kc, aux_es=aux_es, in_split=kc // 2 if two[0] == "1" else 0, n=int(parts[5][1:]))  # (the marching shapes depend on the batch)
        checked += 1
        want = choice if len(choice) > 4 else choice + [1]
        hits += any([list(c.tile), c.nt, c.nsplit, c.ck, c.depth] == want for c in cands)
    assert checked >= 30 and hits >= 0.9 * checked, (hits, checked)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` outside a launcher environment starts its own ranks (bench.self_launch) — and fails loudly, before anything is
    launched, when the box has fewer GPUs than ranks (there is none in the build container)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("needs a box with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VSSEG_SHARE_DEVICE")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but only" in p.stderr and not p.stdout.strip()
