"""Per-kernel sums of every counter in rocprofv3 --pmc result databases, per launch:  python tools/pmc_dump.py <filter> a.db b.db ..."""
import sqlite3
import sys

flt = sys.argv[1]
acc = {}
for db in sys.argv[2:]:
    cur = sqlite3.connect(db).cursor()
    for n, cn, c, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        if flt in n:
            acc.setdefault(n.split("(")[0][:60], {})[cn] = v / max(c, 1)
for n, d in acc.items():
    print(n)
    for k in sorted(d):
        print(f"    {k:32s} {d[k]:16.0f}")
