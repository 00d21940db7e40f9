#!/bin/bash
# Build timing-ablation variants of conv_wino.hip (compile-time bits, see UWM_WINO_ABL in the source) as
# unet-watermark_amd/abl/libuwm_<bits>.so; run on the GPU box with UWM_LIB=<that file> scripts/time_conv.py.
# bits: 1 no MFMA, 2 no U LDS-DMA, 4 no patch global loads/stores, 8 no V-transform LDS reads, 16 no U fragment
# reads, 32 no epilogue, 64 no barrier.  Results of such builds are garbage by construction.
set -e
cd "$(dirname "$0")/.."
mkdir -p unet-watermark_amd/abl
python -c "import sys; sys.path.insert(0,'.'); import __graft_entry__ as g; g.build()"
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUWM_WINO_ABL=$b -c unet-watermark_amd/csrc/conv_wino.hip -o unet-watermark_amd/abl/conv_wino_$b.o
  objs=$(ls unet-watermark_amd/build/*.o | grep -v conv_wino.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unet-watermark_amd/abl/libuwm_$b.so $objs unet-watermark_amd/abl/conv_wino_$b.o
done
