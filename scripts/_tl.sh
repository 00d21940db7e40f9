export TMPDIR=/tmp
R=/root/repo; cd /tmp
UWM_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r50b -o x -- python $R/bench.py --encoder resnet50 --steps 6 --warmup 2 --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 > $R/gpurun_out/r50b.json 2>/dev/null || exit 1
python $R/scripts/kstat.py $R/gpurun_out/r50b > $R/gpurun_out/r50b_kstat.txt; head -14 $R/gpurun_out/r50b_kstat.txt
