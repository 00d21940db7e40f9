# Regenerates one set of measurement evidence on the GPU box (gpurun): rocprofv3 kernel statistics + PMC passes of bench.py,
# the bench lines of every configuration, logit-error table.  usage: GIT_SHA=<commit> [PART=a|b] bash scripts/evidence.sh <tag> [tests]
#   <tag>_bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the two-stream step (co-resident kernels)
#   <tag>_bench_serial_kernel_stats.csv   the same with UWM_SIDE_STREAM=0: every kernel alone (compare roofline.alone)
#   <tag>_pmc_summary.json, pmc_traffic.json   FETCH_SIZE / WRITE_SIZE / MFMA-busy passes digested by scripts/pmc_summary.py
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04_z}
export GIT_SHA=${GIT_SHA:-unknown}      # the GPU box has no .git: pass the commit in (GIT_SHA=$(git rev-parse --short HEAD) gpurun ...)
if [ "$2" = "tests" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_gpu_tests.log; tail -3 gpurun_out/${T}_gpu_tests.log
fi
B="--no-cpu-baseline --serial-steps 0 --alt-steps 0 --prof-steps 0"
if [ "${PART:-all}" != "b" ]; then      # PART=a: profiles + headline lines; PART=b: secondary configurations and tables (two gpurun calls of < 20 min each)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -o s -- python3 bench.py --steps 10 --warmup 3 $B > gpurun_out/${T}_stats.log 2>&1
cp gpurun_out/${T}_stats/*kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv 2>/dev/null || cp gpurun_out/${T}_stats/*/*kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv
UWM_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_serial -o s -- python3 bench.py --steps 10 --warmup 3 $B > gpurun_out/${T}_stats_serial.log 2>&1
cp gpurun_out/${T}_stats_serial/*kernel_stats.csv gpurun_out/${T}_bench_serial_kernel_stats.csv 2>/dev/null || cp gpurun_out/${T}_stats_serial/*/*kernel_stats.csv gpurun_out/${T}_bench_serial_kernel_stats.csv
python3 scripts/launch_list.py gpurun_out/${T}_stats_serial > gpurun_out/${T}_launch_list.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_fetch/x -o p -- python3 bench.py --steps 3 --warmup 1 $B > gpurun_out/${T}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_write/x -o p -- python3 bench.py --steps 3 --warmup 1 $B > gpurun_out/${T}_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${T}_mfma/x -o p -- python3 bench.py --steps 3 --warmup 1 $B > gpurun_out/${T}_mfma.log 2>&1
python scripts/pmc_summary.py gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_mfma ${T} | head -8
cp profiles/${T}_pmc_summary.json profiles/pmc_traffic.json gpurun_out/
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench_1gpu.json 2> gpurun_out/${T}_bench.err; tail -c 600 gpurun_out/${T}_bench_1gpu.json
timeout -k 10 300 python bench_predict.py > gpurun_out/${T}_bench_predict.json 2>/dev/null; tail -c 400 gpurun_out/${T}_bench_predict.json
fi
S="--steps 20 --warmup 5 --no-cpu-baseline --serial-steps 0 --alt-steps 0"
[ "${PART:-all}" = "a" ] && exit 0
# secondary configurations in the bench default (f16x3_all) and, suffix _f32, in the exact-fp32 mode
for P in f16x3_all f32; do
  X=""; [ $P = f32 ] && X="_f32"
  timeout -k 10 400 python bench.py $S --precision $P --arch UnetPlusPlus > gpurun_out/${T}_bench_unetplusplus$X.json 2>/dev/null; tail -c 200 gpurun_out/${T}_bench_unetplusplus$X.json
  timeout -k 10 400 python bench.py $S --precision $P --encoder resnet50 > gpurun_out/${T}_bench_resnet50$X.json 2>/dev/null; tail -c 200 gpurun_out/${T}_bench_resnet50$X.json
  timeout -k 10 400 python bench.py --precision $P --encoder efficientnet-b4 --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 > gpurun_out/${T}_bench_effb4$X.json 2>/dev/null; tail -c 200 gpurun_out/${T}_bench_effb4$X.json
  # configs/unet_watermark_large.yaml: UnetPlusPlus-resnet50, decoder (1024,512,256,128,64), 1024x1024, batch 8
  timeout -k 10 500 python bench.py --precision $P --arch UnetPlusPlus --encoder resnet50 --decoder-channels 1024,512,256,128,64 --size 1024 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 > gpurun_out/${T}_bench_large_yaml$X.json 2>/dev/null; tail -c 200 gpurun_out/${T}_bench_large_yaml$X.json
done
timeout -k 10 600 python bench.py --precision f32 --alt-steps 0 > gpurun_out/${T}_bench_1gpu_f32.json 2>/dev/null; tail -c 300 gpurun_out/${T}_bench_1gpu_f32.json
timeout -k 10 300 python bench_predict.py --precision f32 > gpurun_out/${T}_bench_predict_f32.json 2>/dev/null; tail -c 300 gpurun_out/${T}_bench_predict_f32.json
UWM_FORCE_DDP=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $S > gpurun_out/${T}_bench_ddp_1rank.json 2>/dev/null; tail -c 200 gpurun_out/${T}_bench_ddp_1rank.json
timeout -k 10 300 python scripts/cpu_enqueue_time.py > gpurun_out/${T}_cpu_enqueue_time.txt 2>&1 || true; tail -4 gpurun_out/${T}_cpu_enqueue_time.txt
timeout -k 10 300 python scripts/logit_error.py resnet34 2 256 256 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_logit_error.txt; cat gpurun_out/${T}_logit_error.txt
# the fp16x3 kernels one launch at a time against the fp32 Winograd kernels they replace (op entry: filter bank / partial reduce included)
timeout -k 10 200 python scripts/time_f16x3.py 2>&1 | grep "\^2:" > gpurun_out/${T}_time_conv_f16x3.txt; cat gpurun_out/${T}_time_conv_f16x3.txt
timeout -k 10 200 python scripts/time_wgrad_f16x3.py 2>&1 | grep "\^2:" > gpurun_out/${T}_time_wgrad_f16x3.txt; cat gpurun_out/${T}_time_wgrad_f16x3.txt
