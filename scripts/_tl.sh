export TMPDIR=/tmp
R=/root/repo; cd $R
bash scripts/_ab.sh
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -q -x -k "hipgraph or bit_reproducible or golden or full_size or predict" > gpurun_out/cd_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/cd_tests.log | cut -c1-200
