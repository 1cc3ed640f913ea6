R=$PWD; export PYTHONPATH=$R
for i in 1 2 3; do for v in wkeep wm0; do
  echo "$v $(VNQA_LIB=$R/videonavqa_amd/lib/libvnqa_$v.so VNQA_NO_REBUILD=1 VNQA_HALF=bf16 timeout 120 python3 tools/stem_only.py --iters 20 --precision bf16 2>/dev/null < /dev/null | tail -1)"
done; done
