export TMPDIR=/tmp
R=/root/repo; cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "wgrad_igemm_f16x3 or dgrad_and_wgrad" > gpurun_out/wi_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/wi_tests.log
cd /tmp
UWM_SIDE_STREAM=0 timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tl12 -o x -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 > /dev/null 2>&1 || exit 1
python $R/scripts/kstat.py $R/gpurun_out/tl12 wgrad_igemm
cd $R && bash scripts/_ab.sh
