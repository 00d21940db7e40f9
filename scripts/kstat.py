"""Average duration per kernel out of a rocprofv3 --kernel-trace database (rocpd .db): python scripts/kstat.py <dir-or-db> [substring ...]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
pats = sys.argv[2:]
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kt} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc"
    rows = list(c.execute(q))
    tot = sum(r[3] for r in rows)
    print(f"{os.path.basename(db)}: {tot:.1f} ms of kernels")
    for name, n, avg, ms in rows:
        if pats and not any(p in name for p in pats): continue
        print(f"  {name[:100]:100s} {n:6d} {avg:9.1f} us {ms:9.2f} ms")
