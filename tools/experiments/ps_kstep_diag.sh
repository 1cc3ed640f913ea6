#!/bin/bash
# ON THE GPU BOX: timing-only builds of the patch-stationary kernel (WRONG results): what do the mid-K-step barrier and the DMA wait cost?
#   bash tools/build_one_variant.sh psnew2 conv_ps.hip; ... psd1 conv_ps.hip -DVNQA_PS_DIAG=1 (no barrier); ... psd2 conv_ps.hip -DVNQA_PS_DIAG=2 (no vmcnt wait)
R=$PWD; export PYTHONPATH=$R
for i in 1 2; do for v in psnew2 psd1 psd2; do
  echo "$v $(VNQA_LIB=$R/videonavqa_amd/lib/libvnqa_$v.so VNQA_NO_REBUILD=1 VNQA_HALF=bf16 timeout 120 python3 tools/stem_only.py --iters 20 --precision bf16 2>/dev/null < /dev/null | tail -1)"
done; done
