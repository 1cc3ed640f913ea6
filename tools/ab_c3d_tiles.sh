#!/bin/bash
# ON THE GPU BOX: config 2 under different igemm tiles for the 128-cout (conv2 fwd, conv3a) and 64-cout (conv2 dgrad) 3-D convs
export PYTHONPATH=$PWD
q() { python tools/bench_cnn3d.py bf16 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%7.1f clips/s %6.2f ms" % (d["clips_per_s"], d["ms_per_step"]))'; }
echo "default: $(q)"
for t in 15 10 8 4 1; do echo "128-cout tile $t: $(VNQA_C3D_TILE_128=$t q)"; done
for t in 2 9 5 4; do echo "64-cout tile $t: $(VNQA_C3D_TILE_64=$t q)"; done
echo "default: $(q)"
