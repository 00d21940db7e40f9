#!/bin/bash
# quick per-kernel time table of the bench step (two streams, then serial): scripts/quick_stats.sh <tag> [bench args]
T=${1:-q}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --alt-steps 0 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_qs -o s -- python3 bench.py --steps 10 --warmup 3 $B > gpurun_out/${T}_qs.log 2>&1 &&
UWM_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_qs_serial -o s -- python3 bench.py --steps 10 --warmup 3 $B > gpurun_out/${T}_qs_serial.log 2>&1 &&
python3 - <<PY
import csv,glob
for d in ("${T}_qs","${T}_qs_serial"):
    f=glob.glob("gpurun_out/%s/**/*kernel_stats.csv"%d,recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(d,"total ms/step",tot/1e6/13)
    for r in rows[:28]: print("  %-62s %5d %8.3f ms/step %8.1f us"%(r["Name"][:62], int(r["Calls"]), float(r["TotalDurationNs"])/1e6/13, float(r["AverageNs"])/1e3))
PY
