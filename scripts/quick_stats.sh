#!/bin/bash
# quick per-kernel time table of the bench step: scripts/quick_stats.sh <tag> <mode: both|two|serial> [bench args]
T=${1:-q}; MODE=${2:-both}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --alt-steps 0 --serial-steps 0 --steps 10 --warmup 3 $*"
DIRS=""
if [ $MODE != serial ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_qs -o s -- python3 bench.py $B > gpurun_out/${T}_qs.log 2>&1 || exit 1
  DIRS="${T}_qs"
fi
if [ $MODE != two ]; then
  UWM_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_qs_serial -o s -- python3 bench.py $B > gpurun_out/${T}_qs_serial.log 2>&1 || exit 1
  DIRS="$DIRS ${T}_qs_serial"
fi
python3 - $DIRS <<PY
import csv,glob,sys,json
for d in sys.argv[1:]:
    f=glob.glob("gpurun_out/%s/**/*kernel_stats.csv"%d,recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    line=[l for l in open("gpurun_out/%s.log"%d) if l.startswith("{")][-1]; j=json.loads(line)
    print(d,"img/s",j["value"],"ms/step",j["ms_per_step"],"| kernel ms/step",round(tot/1e6/13,3), "launches/step", sum(int(r["Calls"]) for r in rows)/13)
    for r in rows[:34]: print("  %-66s %6.1f/step %8.3f ms/step %8.1f us"%(r["Name"][:66], int(r["Calls"])/13, float(r["TotalDurationNs"])/1e6/13, float(r["AverageNs"])/1e3))
PY
