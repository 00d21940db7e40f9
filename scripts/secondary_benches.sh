cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S="--steps 20 --warmup 5 --no-cpu-baseline --serial-steps 0 --alt-steps 0 --prof-steps 0"
r() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j.get('ms_per_step', j.get('ms_per_batch')))"; }
for p in f16x3_all f32; do
timeout -k 10 300 python bench.py $S --precision $p --arch UnetPlusPlus 2>/dev/null | r "upp-$p"
timeout -k 10 300 python bench.py $S --precision $p --encoder resnet50 2>/dev/null | r "r50-$p"
timeout -k 10 300 python bench.py --encoder efficientnet-b4 --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 --precision $p 2>/dev/null | r "effb4-$p"
timeout -k 10 400 python bench.py --arch UnetPlusPlus --encoder resnet50 --decoder-channels 1024,512,256,128,64 --size 1024 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 --precision $p 2>/dev/null | r "largeyaml-$p"
done
timeout -k 10 300 python bench_predict.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('predict f16x3', j['value'], j['bitwise_equal_to_batch1_path'])"
timeout -k 10 300 python bench_predict.py --precision f32 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('predict f32', j['value'], j['bitwise_equal_to_batch1_path'])"
