#!/bin/bash
# timing-only ablations of the PIPE 5 igemm main loop (tools/build_one_variant.sh p5d<bits> conv_igemm.hip -DVNQA_P5_DIAG=<bits>):
# bits 1 = no pixel DMA, 2 = no weight DMA, 4 = no fragment reads, 8 = no barrier, 16 = no MFMAs
export PYTHONPATH=. VNQA_NO_REBUILD=1
for L in conv21 conv12; do
for V in 0 1 2 3 4 8 7 15 16 19 23; do
  echo -n "diag=$V "; VNQA_LIB=$PWD/videonavqa_amd/lib/libvnqa_p5d$V.so python tools/bench_stem_layers.py --frames 280 --iters 10 --layers $L --tiles 19 2>&1 | grep '"layer"'
done; done
