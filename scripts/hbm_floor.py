"""HBM floor of one training step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py: total bytes per step, per
kernel class, against the 6.3 TB/s copy ceiling and the 8 TB/s peak (MI355X_MICROARCH.md; FETCH_SIZE doubled: gfx950 correction).
usage: hbm_floor.py <fetch_dir> <write_dir> <steps in the traced run (warm-up + timed)>"""
import collections, csv, glob, re, sys
fd, wd, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
def load(d, name):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"\s+", "", r["Kernel_Name"].split("(")[0].replace("void uwm::", "").replace("uwm::", ""))
        agg[k] += float(r["Counter_Value"]) * 1024
    return agg
fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
tot = {k: 2 * fe.get(k, 0) + wr.get(k, 0) for k in set(fe) | set(wr)}
T = sum(tot.values()) / steps
print(f"HBM bytes per step (FETCH x2 + WRITE): {T / 1e9:.2f} GB -> {T / 6.3e12 * 1e3:.2f} ms at 6.3 TB/s (copy ceiling), {T / 8e12 * 1e3:.2f} ms at 8 TB/s")
for k in sorted(tot, key=lambda k: -tot[k])[:25]:
    print(f"  {tot[k] / steps / 1e9:7.3f} GB/step  {k}")
