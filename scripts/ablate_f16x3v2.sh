#!/bin/bash
# Timing-ablation builds of conv_f16x3v2.hip (compile-time bits UWM_F16V2_ABL: 1 no MFMA, 2 no filter-fragment loads, 4 no pixel-fragment
# LDS reads, 8 no patch staging) as unet-watermark_amd/abl/libuwm_v2_<bits>.so; on the GPU box: UWM_LIB=<that file> python scripts/time_f16x3.py 604.
# Results of such builds are garbage by construction.  Extra -D flags: EXTRA="-DFOO=1" scripts/ablate_f16x3v2.sh <bits> ...
set -e
cd "$(dirname "$0")/.."
mkdir -p unet-watermark_amd/abl
python -c "import sys; sys.path.insert(0,'.'); import __graft_entry__ as g; g.build()"
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DUWM_F16V2_ABL=$b $EXTRA -c unet-watermark_amd/csrc/conv_f16x3v2.hip -o unet-watermark_amd/abl/conv_f16x3v2_${TAG}$b.o
  objs=$(ls unet-watermark_amd/build/*.o | grep -v /conv_f16x3v2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unet-watermark_amd/abl/libuwm_v2_${TAG}$b.so $objs unet-watermark_amd/abl/conv_f16x3v2_${TAG}$b.o
done
