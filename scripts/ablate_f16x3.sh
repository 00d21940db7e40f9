#!/bin/bash
# Build timing-ablation variants of conv_f16x3.hip (compile-time bits, see UWM_F16_ABL in the source) as
# unet-watermark_amd/abl/libuwm_f16_<bits>.so; run on the GPU box with UWM_LIB=<that file> scripts/time_f16x3.py.
# bits: 1 no MFMA, 2 no filter-fragment loads, 4 no pixel-fragment LDS reads, 8 no patch global loads / LDS stores, 16 no epilogue.
# Results of such builds are garbage by construction.
set -e
cd "$(dirname "$0")/.."
mkdir -p unet-watermark_amd/abl
python -c "import sys; sys.path.insert(0,'.'); import __graft_entry__ as g; g.build()"
# `scripts/ablate_f16x3.sh wgrad <bits...>`: the same for wgrad_f16x3.hip (UWM_WG16_ABL: 1 no MFMA, 2 no X fragment reads, 4 no loader
# work after the first stage) -> libuwm_wg16_<bits>.so, for scripts/time_wgrad_f16x3.py
src=conv_f16x3; def=UWM_F16_ABL; tag=f16
if [ "$1" = wgrad ]; then src=wgrad_f16x3; def=UWM_WG16_ABL; tag=wg16; shift; fi
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -D$def=$b -c unet-watermark_amd/csrc/$src.hip -o unet-watermark_amd/abl/${src}_$b.o
  objs=$(ls unet-watermark_amd/build/*.o | grep -v /$src.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unet-watermark_amd/abl/libuwm_${tag}_$b.so $objs unet-watermark_amd/abl/${src}_$b.o
done
