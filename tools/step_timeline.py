"""Every kernel of the last steady-state training step of a rocprofv3 kernel trace (rocpd .db) in start order: queue / stream, start offset, TRUE duration, gap to the
previous kernel on the same queue; and the per-queue busy / idle totals.   python tools/step_timeline.py <results.db> [n_steps_back]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tabs if t.startswith("kernels") or t == "kernels")
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
name_c = "name" if "name" in cols else "kernel_name"
qc = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
print("columns:", cols, file=sys.stderr)
rows = list(cur.execute(f"select {name_c}, start, end, {qc or '0'} from {kd} order by start"))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
seg = rows[adam[-2] + 1 : adam[-1] + 1]
t0 = seg[0][1]
last_end = {}
busy = {}
for name, s, e, q in seg:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    short = name.replace("unsigned short", "bf16").replace("void ", "").split("(")[0][:60]
    print(f"q{q} {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {short}")
wall = (seg[-1][2] - t0) / 1e3
print(f"=== step wall {wall:.1f} us; " + "; ".join(f"q{q} busy {b / 1e3:.1f} us" for q, b in busy.items()))
