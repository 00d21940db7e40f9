"""Per-kernel totals of the depthwise kernels (and the whole step) from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
for n, c, t in rows:
    if "dw_" in n:
        print(f"{n[8:60]:54s} {c:5d} {t/c/1e3:8.1f} us avg {t/1e6:8.2f} ms total")
