#!/bin/bash
# interleaved A/B of bench flags on one box: ab_flags.sh "<flagsA>" "<flagsB>" [common flags]
A="$1"; B="$2"; shift; shift
for i in 1 2 3; do
  for F in "$A" "$B"; do
    echo -n "[$F] "; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
