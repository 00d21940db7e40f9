#!/bin/bash
# usage: scripts/bench_ab.sh VAR v1 v2 ...   -> runs bench.py with VAR=v for each v, prints value / ms / TF
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_$v.json
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v.json")); print("$var=$v", d["value"], "img/s", d["ms_per_step"], "ms", d["model_tflops"], "TF")
PY
done
