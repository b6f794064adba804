#!/bin/bash
# A/B of a variant library against the product one on ONE box: interleaved step benches (boxes differ by +-3 %, runs on one box by ~0.3 %).
#   tools/ab_step.sh walkgpt_amd/_abl/lib_<tag>.so [bench.py flags]   -> gpurun_out/ab_<tag>.log
lib=$1; shift
tag=$(basename $lib .so)
out=gpurun_out/ab_$tag.log; : > $out
for i in 1 2 3; do
  echo "product:" >> $out; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out || exit 1
  echo "$tag:" >> $out; WG_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out || exit 1
done
cat $out
