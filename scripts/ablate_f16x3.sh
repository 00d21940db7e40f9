#!/bin/bash
# Build timing-ablation variants of conv_f16x3.hip (compile-time bits, see UWM_F16_ABL in the source) as
# unet-watermark_amd/abl/libuwm_f16_<bits>.so; run on the GPU box with UWM_LIB=<that file> scripts/time_f16x3.py.
# bits: 1 no MFMA, 2 no filter-fragment loads, 4 no pixel-fragment LDS reads, 8 no patch global loads / LDS stores, 16 no epilogue.
# Results of such builds are garbage by construction.
set -e
cd "$(dirname "$0")/.."
mkdir -p unet-watermark_amd/abl
python -c "import sys; sys.path.insert(0,'.'); import __graft_entry__ as g; g.build()"
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUWM_F16_ABL=$b -c unet-watermark_amd/csrc/conv_f16x3.hip -o unet-watermark_amd/abl/conv_f16x3_$b.o
  objs=$(ls unet-watermark_amd/build/*.o | grep -v conv_f16x3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unet-watermark_amd/abl/libuwm_f16_$b.so $objs unet-watermark_amd/abl/conv_f16x3_$b.o
done
