#!/bin/bash
# A/B of the persistent GEMM's cache policy (output stores sc1 / nt, A loads nt) and tile order (WG_GEMM_COLBLOCK) on one box:
# end to end (bench.py --steps-only) and per shape (tools/bench_gemm.py).  Variants: tools/build_variant.py.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in base csc1 cnt csc1_ant; do
  for cb in h 0; do
    L=""; [ $v != base ] && L="walkgpt_amd/_abl/lib_$v.so"
    C=""; [ $cb != h ] && C=$cb
    r=$(WG_LIB=$L WG_GEMM_COLBLOCK=$C python bench.py --steps 20 --warmup 5 --steps-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "e2e rep $rep lib=$v colblock=$cb : $r"
  done
done
done
for v in base csc1; do
  for cb in h 0; do
    L=""; [ $v != base ] && L="walkgpt_amd/_abl/lib_$v.so"
    C=""; [ $cb != h ] && C=$cb
    echo "== gemm lib=$v colblock=$cb"
    WG_LIB=$L WG_GEMM_COLBLOCK=$C python tools/bench_gemm.py 2>/dev/null | grep "M=" | awk '{print $1,$2,$3,$4,$5,$6, $NF}' | head -8
  done
done
