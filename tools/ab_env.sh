#!/bin/bash
# interleaved A/B of an environment setting on one box: ab_env.sh VAR=VALUE [bench flags]
kv="$1"; shift
for i in 1 2 3; do
  echo -n "[default] "; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  echo -n "[$kv] "; env "$kv" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
