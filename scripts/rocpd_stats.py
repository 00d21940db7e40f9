"""Aggregate a rocprofv3 rocpd SQLite database (default output of `rocprofv3 --kernel-trace`) per kernel.
usage: rocpd_stats.py results.db [steps]  -> name, calls, ms/step, avg us, share"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':72s} {'calls':>6s} {'ms/step':>9s} {'avg us':>9s} {'share':>6s}")
for n, c, t in rows[:40]:
    print(f"{n[:72]:72s} {c:6d} {t/1e6/steps:9.3f} {t/c/1e3:9.1f} {100*t/tot:5.1f}%")
print(f"total kernel time per step: {tot/1e6/steps:.3f} ms")
