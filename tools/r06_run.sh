set -x
O=gpurun_out/r06; mkdir -p $O
timeout 900 python tools/parity_localize.py --low fp16h --height 160 --width 208 --seed 3 --batches 4 > $O/localize_160_s3.json 2>$O/localize_160_s3.err; cat $O/localize_160_s3.json; tail -3 $O/localize_160_s3.err
timeout 900 python tools/parity_localize.py --low fp16h --seed 3 --batches 4 > $O/localize_224_s3.json 2>/dev/null; cat $O/localize_224_s3.json
