"""Two-stream step timeline out of a rocprofv3 --kernel-trace database: per step (delimited by adam_kernel) the busy time of each HSA
queue, when each queue's last kernel ends relative to the step's end, and the gaps of the main queue.
    python scripts/timeline.py <dir-or-db>"""
import glob, os, sqlite3, sys
path = sys.argv[1]
db = path if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.queue_id, d.start, d.end from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
for k in range(max(1, len(adam) - 3), len(adam)):
    lo, hi = adam[k - 1] + 1, adam[k]
    step = rows[lo:hi + 1]
    t0, t1 = step[0][2], step[-1][3]
    qs = sorted(set(r[1] for r in step))
    print(f"step {k}: {1e-6 * (t1 - t0):.3f} ms, {len(step)} kernels, queues {qs}")
    for q in qs:
        rq = [r for r in step if r[1] == q]
        busy = sum(r[3] - r[2] for r in rq)
        print(f"  queue {q}: {len(rq):4d} kernels, busy {1e-6 * busy:.3f} ms, first start +{1e-6 * (rq[0][2] - t0):.3f} ms, last end {1e-6 * (t1 - rq[-1][3]):.3f} ms before the step's end ({rq[-1][0][:50]})")
    # when does the forward end: first kernel on the second queue
    if len(qs) > 1:
        mainq = max(qs, key=lambda q: sum(1 for r in step if r[1] == q))
        rm = [r for r in step if r[1] == mainq]
        gaps = sorted(((rm[i + 1][2] - rm[i][3], rm[i][0][:40], rm[i + 1][0][:40]) for i in range(len(rm) - 1)), reverse=True)[:6]
        print("  largest gaps on the main queue:", [(round(1e-3 * g, 1), a, b) for g, a, b in gaps])
        side = [r for r in step if r[1] != mainq]
        # overlap: time when both busy
        ev = []
        for r in rm: ev += [(r[2], 0, 1), (r[3], 0, -1)]
        for r in side: ev += [(r[2], 1, 1), (r[3], 1, -1)]
        ev.sort()
        cnt = [0, 0]; last = ev[0][0]; both = only_m = only_s = idle = 0
        for t, w, d in ev:
            dt = t - last
            if cnt[0] > 0 and cnt[1] > 0: both += dt
            elif cnt[0] > 0: only_m += dt
            elif cnt[1] > 0: only_s += dt
            else: idle += dt
            cnt[w] += d; last = t
        print(f"  both busy {1e-6 * both:.3f} ms, main only {1e-6 * only_m:.3f}, side only {1e-6 * only_s:.3f}, idle {1e-6 * idle:.3f}")
