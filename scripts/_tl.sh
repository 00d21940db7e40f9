export TMPDIR=/tmp
R=/root/repo; cd $R
python scripts/_dbg_grad.py 2>&1 | grep -E "blocks.3.conv1.0|encoder.conv1"
timeout -k 10 800 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "full_size or bit_reproducible or staged" > gpurun_out/c32_model.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/c32_model.log | cut -c1-200
bash scripts/_ab.sh
