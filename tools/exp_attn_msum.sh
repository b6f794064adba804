#!/bin/bash
# A/B on one box: product library vs -DWG_ATTN_MFMA_SUM=1 (python tools/build_variant.py msum -DWG_ATTN_MFMA_SUM=1), attention kernels alone + the C2 step
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  echo "== base";   python tools/bench_attn.py 2>/dev/null | head -5
  echo "== msum"; WG_LIB=walkgpt_amd/_abl/lib_msum.so python tools/bench_attn.py 2>/dev/null | head -5
done
echo "== step base";   python bench.py --steps 20 --warmup 5 --steps-only 2>/dev/null | cut -c1-120
echo "== step msum"; WG_LIB=walkgpt_amd/_abl/lib_msum.so python bench.py --steps 20 --warmup 5 --steps-only 2>/dev/null | cut -c1-120
echo "== step base";   python bench.py --steps 20 --warmup 5 --steps-only 2>/dev/null | cut -c1-120
echo "== step msum"; WG_LIB=walkgpt_amd/_abl/lib_msum.so python bench.py --steps 20 --warmup 5 --steps-only 2>/dev/null | cut -c1-120
