for cb in "" -1; do
  echo "=== WG_GEMM_COLBLOCK=$cb"; QUICK=1 WG_GEMM_COLBLOCK=$cb timeout -k 10 200 python tools/bench_gemm_fr.py 2>&1 | grep "^sam\|^clip\|^8k\|^4k" | awk '{print $1,$2,$3,$4,$5,$6,"|",$8,$9,$10,$11,$12,"|",$20,$21,$22,$23,$24}'
done
