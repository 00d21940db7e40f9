# (baseline library: scripts/build_baseline_lib.sh <rev>)
# same-box A/B: the committed build (abl/libuwm_base.so) against the working tree, alternating runs
export TMPDIR=/tmp
B="--steps ${AB_STEPS:-60} --warmup ${AB_WARM:-15} --no-cpu-baseline --alt-steps 0 --serial-steps 0 --prof-steps 0 ${AB_ARGS}"
for i in 1 2 3; do
  for v in new base; do
    if [ $v = base ]; then export UWM_LIB=$PWD/unet-watermark_amd/abl/libuwm_base.so; else unset UWM_LIB; fi
    timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['loss'])" || exit 1
  done
done
