#!/bin/bash
# ON THE GPU BOX: which phase of the fused conv1 kernel owns its LDS bank conflicts?  Timing-only build (-DVNQA_DIAG_SKIP_DMA) with the
# phase switches of csrc/conv_c64.hip (relu flag 256: conv1_1 patch computed for the first tile only, 512: no epilogue, 1024: no
# MFMA loop), PMC pass SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per variant.
ROOT=$PWD; export PYTHONPATH=$ROOT
python tools/build_variant.py diag -DVNQA_DIAG_SKIP_DMA > /dev/null 2>&1
export VNQA_LIB=$ROOT/videonavqa_amd/lib/libvnqa_diag.so VNQA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp
for F in 1 257 513 1025 769 1281; do
  rm -rf /tmp/pc1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/pc1 -- python3 $ROOT/tools/bench_fused_first.py $F > /tmp/pc1.out 2>/dev/null
  echo "flags $F: $(grep wide /tmp/pc1.out | tail -1)"
  python3 $ROOT/tools/pmc_kernel.py /tmp/pc1 conv_first_c64_wide | tr '\n' ' '; echo
done
