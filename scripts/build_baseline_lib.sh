#!/bin/bash
# Builds unet-watermark_amd/abl/libuwm_base.so from the kernel sources of a git revision (default HEAD~1): the baseline of scripts/_ab.sh's
# same-box A/B (UWM_LIB=<that file> makes _lib.py load it instead of the working tree's libuwm.so).  usage: scripts/build_baseline_lib.sh [rev]
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD~1}
T=$(mktemp -d)
git archive "$REV" unet-watermark_amd/csrc include | tar -x -C "$T"
mkdir -p unet-watermark_amd/abl "$T/obj"
SRCS=$(git show "$REV:unet-watermark_amd/_lib.py" | python3 -c "import re,sys; m=re.search(r'SOURCES = \[(.*?)\]', sys.stdin.read(), re.S); print(' '.join(x.strip().strip('\"') for x in m.group(1).split(',')))")
cd "$T/unet-watermark_amd/csrc"
for f in $SRCS; do
  EX=""; case $f in conv_f16x3.hip|conv_f16x3v2.hip|wgrad_f16x3.hip|conv_stem_f16x3.hip) EX="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EX -c $f -o "$T/obj/${f%.hip}.o" 2>/dev/null &
  if (( $(jobs -r | wc -l) >= 8 )); then wait -n; fi
done
wait
cd - > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unet-watermark_amd/abl/libuwm_base.so "$T"/obj/*.o
rm -rf "$T"
echo "built unet-watermark_amd/abl/libuwm_base.so from $REV"
