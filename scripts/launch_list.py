"""One training step of a rocprofv3 kernel trace, launch by launch in start order (the per-kernel averages of the statistics CSV
hide a single mis-routed or under-filled launch; this listing is what exposed them).  usage: launch_list.py <rocprof output dir> [min_us]
Works on a UWM_SIDE_STREAM=0 trace (every kernel alone); a step is delimited by two consecutive adam_kernel launches."""
import csv, glob, sys
d = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"{len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from the first start to the last end")
for r in step:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us < min_us: continue
    wgs = int(r.get("Grid_Size_X", "0") or 0) // max(1, int(r.get("Workgroup_Size_X", "1") or 1))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {us:8.1f} us  {wgs:6d} x {r.get('Workgroup_Size_X', '?'):>4}  {r['Kernel_Name'].replace('void uwm::', '').replace('uwm::', '')[:90]}")
