"""Turn rocprofv3 rocpd (.db) outputs into the small text summaries committed under profiles/.

    python tools/rocprof_summary.py kernel <results.db>            -> per-kernel calls / total / average / share
    python tools/rocprof_summary.py pmc <fetch.db> <write.db>      -> per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE

FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950 wide coalesced reads
(the counter tallies 128-byte requests as 64 bytes); WRITE_SIZE is reported as is (uncalibrated there).  Both are in KiB.
"""
import glob
import json
import sqlite3
import sys


def short(name):
    name = name.replace("unsigned short", "bf16").replace("void ", "")
    return name.split("(")[0]


def bench_name(short_name):
    """rocprof kernel name -> the name bench.py uses for its roofline block (igemm<bf16,NT,MTW>, wgrad<bf16,NTP>)."""
    import re
    m = re.match(r"igemm_kernel<(bf16|float), (\d+), (\d+), \d+(, \d+)?>", short_name)  # the MODE / unrolled-K variants of one (type, NT, MTW) share a bench name
    if m:
        return f"igemm<{'bf16' if m.group(1) == 'bf16' else 'f32'},{m.group(2)},{m.group(3)}>"
    m = re.match(r"sconv_kernel<(\d+), (\d+), (\d+), \d+>", short_name)  # streaming kernel: the MODE / tap / input-channel variants of one NT share a bench name
    if m:
        return f"sconv<bf16,{m.group(2)}>"
    m = re.match(r"mconv_kernel<(\d+), (\d+), (\d+), (\d+), \d+(, \w+)*>", short_name)  # marching kernel: (CIN, NT, TZ, MT, MODE, WREG, NR) variants of one NT share a bench name
    if m:
        return f"mconv<bf16,{m.group(2)}>"
    m = re.match(r"mbwd_kernel<(\d+), (\d+), ", short_name)  # fused backward (BatchNorm backward on load + data gradient + weight gradient): (tiles of the outputs, tiles of the inputs)
    if m:
        return f"mbwd<bf16,{int(m.group(1)) // 16},{int(m.group(2)) // 16}>"
    m = re.match(r"mwgrad_kernel<(\d+), (\d+), ", short_name)  # marching weight gradient: bench name by the P tiles (ntp = max(1, CP / 16))
    if m:
        return f"mwgrad<bf16,{max(1, int(m.group(2)) // 16)}>"
    m = re.match(r"cconv_kernel<(\d+), \d+>", short_name)  # compute-bound kernel: the MODE variants of one NT share a bench name
    if m:
        return f"cconv<bf16,{m.group(1)}>"
    m = re.match(r"wgrad_kernel<(bf16|float), (\d+), (\d+)(, \d+)?>", short_name)  # MAXT / H-group variants of one (type, NTP) share a bench name
    if m:
        return f"wgrad<{'bf16' if m.group(1) == 'bf16' else 'f32'},{m.group(3)}>"
    m = re.match(r"cwgrad_kernel<(\d+), (\d+), ", short_name)  # compute weight gradient (3x3x3 stride-1 layers): bench name by the P tiles (NPW x PS)
    if m:
        return f"cwgrad<bf16,{int(m.group(1)) * int(m.group(2))}>"
    if short_name.startswith("nconv_kernel"):  # narrow-output convolution (C -> 1, vector ALUs)
        return "nconv<bf16>"
    if short_name.startswith("wgrad_narrow_kernel"):
        return "wgrad_narrow"
    m = re.match(r"chain_kernel<(\d+),", short_name)  # chained marching convolution (inference): input channels of its first stage
    if m:
        return f"chain<bf16,{m.group(1)}>"
    m = re.match(r"gconv_kernel<(\d+), (\d+), ", short_name)  # gathering marching kernel (stride-(2,2,1) launches fine -> coarse): output channel tiles
    if m:
        return f"gconv<bf16,{m.group(2)}>"
    m = re.match(r"tconv_kernel<(\d+), ", short_name)  # transition kernel (levels 2 <-> 3): 8-channel groups of its input
    if m:
        return f"tconv<bf16,{m.group(1)}>"
    m = re.match(r"dconv_kernel<(\d+), (\d+)>", short_name)  # deep-level kernel: (M-tiles per workgroup, channel tiles per workgroup)
    if m:
        return f"dconv<bf16,{m.group(1)},{m.group(2)}>"
    m = re.match(r"(bn_act_fwd|bn_act_bwd_reduce|bn_act_bwd_apply|att_apply_fwd|att_apply_bwd)_kernel<", short_name)  # streaming kernels: bench.py's group names
    if m:
        return m.group(1)
    return short_name


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
    print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for n, c, t, a, p in rows:
        print(f"{short(n)[:60]:60s} {c:7d} {t:12.1f} {a:10.2f} {p:6.2f}")


def pmc(fetch_db, write_db):
    out = {}
    for label, db in (("fetch", fetch_db), ("write", write_db)):
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name"
        for n, c, v in cur.execute(q, ("FETCH_SIZE" if label == "fetch" else "WRITE_SIZE",)):
            d = out.setdefault(short(n), {})
            d[label + "_kib_per_launch"] = v / c
            d["launches"] = c
    print(f"{'kernel':60s} {'launches':>8s} {'fetch MB/launch (x2 corrected)':>32s} {'write MB/launch':>16s}")
    res = {}
    for n, d in sorted(out.items(), key=lambda kv: -(kv[1].get("fetch_kib_per_launch", 0) * kv[1].get("launches", 0))):
        f = 2.0 * d.get("fetch_kib_per_launch", 0.0) * 1024 / 1e6
        w = d.get("write_kib_per_launch", 0.0) * 1024 / 1e6
        print(f"{n[:60]:60s} {d.get('launches', 0):8d} {f:32.2f} {w:16.2f}")
        key = bench_name(n)
        if key in res:  # several rocprof instantiations map to one bench name (wgrad MAXT variants): launch-weighted mean
            o = res[key]
            tot = o["launches"] + d.get("launches", 0)
            o["fetch_mb"] = (o["fetch_mb"] * o["launches"] + f * d.get("launches", 0)) / max(tot, 1)
            o["write_mb"] = (o["write_mb"] * o["launches"] + w * d.get("launches", 0)) / max(tot, 1)
            o["launches"] = tot
        else:
            res[key] = dict(fetch_mb=f, write_mb=w, launches=d.get("launches", 0))
    for v in res.values():
        v["hbm_bytes_per_launch"] = (v["fetch_mb"] + v["write_mb"]) * 1e6
    return res


def sq(dbs):
    """Per-kernel-group averages of the SQ counters (MFMA busy fraction, instruction mix per MFMA, wave-cycle breakdown, LDS bank conflicts)."""
    agg = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for n, cn, c, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            a = agg.setdefault(bench_name(short(n)), {})
            a[cn] = a.get(cn, 0.0) + v
            a["launches:" + cn] = a.get("launches:" + cn, 0) + c
    rows = []
    for k, a in agg.items():
        busy = a.get("SQ_BUSY_CU_CYCLES", 0.0)
        mf = a.get("SQ_INSTS_MFMA", 0.0)
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((busy, k, a, mf, wc))
    print(f"{'kernel group':34s} {'launches':>8s} {'MFMA busy / (4 SIMD x CU busy)':>32s} {'VALU/MFMA':>10s} {'SALU/MFMA':>10s} {'LDS/MFMA':>9s} {'wait%':>7s} {'stall%':>7s} {'issue%':>7s} {'LDS conflict%':>14s}")
    for busy, k, a, mf, wc in sorted(rows, key=lambda r: -r[0])[:40]:
        frac = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy) if busy else 0.0
        per = lambda c: (a.get(c, 0.0) / mf) if mf else float("nan")  # noqa: E731
        pc = lambda c: (100.0 * a.get(c, 0.0) / wc) if wc else float("nan")  # noqa: E731
        conf = 100.0 * a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        print(f"{k[:34]:34s} {a.get('launches:SQ_BUSY_CU_CYCLES', 0):8d} {frac:32.3f} {per('SQ_INSTS_VALU') - (1 if mf else 0):10.2f} {per('SQ_INSTS_SALU'):10.2f} {per('SQ_INSTS_LDS'):9.2f} {pc('SQ_WAIT_ANY'):7.1f} {pc('SQ_WAIT_INST_ANY'):7.1f} {pc('SQ_ACTIVE_INST_ANY'):7.1f} {conf:14.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_stats(sys.argv[2])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2:])
    else:
        r = pmc(sys.argv[2], sys.argv[3])
        if len(sys.argv) > 4:
            json.dump(r, open(sys.argv[4], "w"), indent=1)
