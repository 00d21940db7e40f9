# PMC passes over a kernel-at-a-time timing script: where do a kernel's wave cycles go?
# usage (GPU box): bash scripts/pmc_wgrad.sh <tag> <layers> [script] [first arg of the script] [kernel-name filter]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-x}; LAYERS=${2:-layer3,dec1c1}; SCRIPT=${3:-scripts/time_wgrad.py}; ARG1=${4:-0}; FILT=${5:-wgrad}
rocprofv3 -L > gpurun_out/${T}_counters.txt 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${T}_pmcw$i/x -o p -- python3 $SCRIPT $ARG1 $LAYERS > gpurun_out/${T}_pmcw$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/${T}_pmcw*/x/*_counter_collection.csv") + glob.glob("gpurun_out/${T}_pmcw*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void uwm::", "")
        if "${FILT}" not in k: continue
        key = (k, r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add((f, r["Dispatch_Id"]))
for key in sorted(agg):
    cnt = {}
    for f in set(x[0] for x in n[key]): cnt[f] = len([1 for x in n[key] if x[0] == f])
    d = max(cnt.values())
    print(key, "dispatches/pass", d)
    for c, v in sorted(agg[key].items()): print(f"   {c:32s} {v / d:16.0f}")
PY
