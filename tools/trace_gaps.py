"""GPU idle time inside the training steps of a rocprofv3 kernel trace (rocpd .db): union of the kernel intervals against the wall time between the
first and the last kernel of the steady-state steps, and the largest gaps with the kernels either side.
    rocprofv3 --kernel-trace -d /tmp/kt -- python tools/time_step.py 10;  python tools/trace_gaps.py /tmp/kt/*/*.db"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tabs if t.startswith("kernels") or t == "kernels")
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
name_c = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute(f"select {name_c}, start, end from {kd} order by start"))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = adam[len(adam) // 2], adam[-1]  # steady state: from the middle optimizer step to the last
seg = rows[lo + 1 : hi + 1]
steps = len([i for i in adam if lo < i <= hi])
wall = seg[-1][2] - seg[0][1]
busy, cur_end, gaps = 0, seg[0][1], []
prev = seg[0]
for r in seg:
    if r[1] > cur_end:
        gaps.append((r[1] - cur_end, prev[0][:50], r[0][:50]))
        busy += 0
        cur_s = r[1]
    s = max(r[1], cur_end)
    if r[2] > s:
        busy += r[2] - s
        cur_end = r[2]
    prev = r
print(f"{steps} steps: wall {wall / steps / 1e6:.3f} ms per step, GPU busy {busy / steps / 1e6:.3f} ms, idle {(wall - busy) / steps / 1e6:.3f} ms in {len(gaps) / steps:.0f} gaps per step")
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg[k] = agg.get(k, 0) + g
for (a, b), g in sorted(agg.items(), key=lambda kv: -kv[1])[:15]:
    print(f"{g / steps / 1e3:8.1f} us per step   {a}  ->  {b}")
