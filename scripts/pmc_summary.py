"""Digest rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE, WRITE_SIZE, MFMA-busy) into profiles/:
   usage: scripts/pmc_summary.py <fetch_dir> <write_dir> <mfma_dir> <tag>
   writes profiles/<tag>_pmc_summary.json and profiles/pmc_traffic.json (read by bench.py for roofline.traffic; carries the
   sha256 of the kernel sources it was measured on, and GIT_SHA from the environment)."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_sha256
fd, wd, md, tag = sys.argv[1:5]
def norm(k):
    k = k.split("(")[0].replace("void uwm::", "").replace("uwm::", "")
    return re.sub(r"\s+", "", k)
def load(d):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = norm(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}
fa, fn = load(fd); wa, wn = load(wd); ma, mn = load(md)
tr = glob.glob(f"{md}/*/*_kernel_trace.csv")[0]
dur = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(tr)):
    k = norm(r["Kernel_Name"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[k] += 1
rows, traffic = [], {}
alias = {"wgrad_patch_kernel<64,32>": "wgrad_patch_kernel<64>", "conv_wino_x3_kernel": "conv_wino_x3_kernel", "wgrad_patch_kernel<32,32>": "wgrad_patch_kernel<32>",
         "wgrad_patch_kernel<16,32>": "wgrad_patch_kernel<16>",
         # template parameter is channel tiles of 16; the profiler class names say channels
         "conv_wino_kernel<4>": "conv_wino_kernel<64>", "conv_wino_kernel<2>": "conv_wino_kernel<32>",
         "conv_wino_kernel<1>": "conv_wino_kernel<16>"}
for k in sorted(dur, key=lambda k: -dur[k])[:16]:
    fe = fa[k].get("FETCH_SIZE", 0) / max(1, fn.get(k, 1)) * 1024; wr = wa[k].get("WRITE_SIZE", 0) / max(1, wn.get(k, 1)) * 1024
    gui = ma[k].get("GRBM_GUI_ACTIVE", 0); busy = ma[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    # GRBM_GUI_ACTIVE / 8 / duration reads high on short dispatches (the counter window is wider than the kernel: bn_finalize came out
    # at 6.8 GHz) — MI355X_MICROARCH.md: "reads high on dispatches shorter than about 0.3 ms".  The clock and the busy fraction derived
    # from it are printed only for kernels whose average launch is >= 20 us; below that both are null.
    avg_us = dur[k] / cnt[k]
    trust = avg_us >= 20.0
    rows.append(dict(kernel=k, launches=cnt[k], avg_us=round(avg_us, 1), fetch_bytes_raw=int(fe), fetch_bytes_x2=int(2 * fe),
                     write_bytes=int(wr), mfma_busy_frac=round(busy / (gui / 8 * 1024), 3) if (gui and trust) else None,
                     clock_GHz=round(gui / 8 / (dur[k] * 1e3), 2) if (dur[k] and trust) else None))
    traffic[alias.get(k, k)] = {"hbm_bytes_per_launch": int(2 * fe + wr), "fetch_bytes_per_launch": int(2 * fe), "write_bytes_per_launch": int(wr), "launches": cnt[k]}
json.dump(rows, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
json.dump({"source": f"profiles/{tag}_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; FETCH_SIZE x2 (gfx950), per launch",
           "source_sha256": source_sha256(),          # the kernel sources these counters were measured on (bench.py refuses another)
           "git_sha": os.environ.get("GIT_SHA"),      # commit of that tree (handed in by the caller: the GPU box has no .git)
           "kernels": traffic}, open("profiles/pmc_traffic.json", "w"), indent=1)
for r in rows[:8]: print(r)
