#!/bin/bash
# The rocprofv3 passes behind profiles/r03_* (run on the GPU box from the repo root; outputs under gpurun_out/).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d gpurun_out/p3_stats -o s -- python3 bench.py --steps 3 --warmup 2 --steps-only > gpurun_out/p3_stats.log 2>&1
echo stats done
$P --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p3_fetch -o s -- python3 bench.py --steps 1 --warmup 1 --steps-only > gpurun_out/p3_fetch.log 2>&1
$P --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p3_write -o s -- python3 bench.py --steps 1 --warmup 1 --steps-only > gpurun_out/p3_write.log 2>&1
echo traffic done
$P --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/p3_sq -o s -- python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p3_sq.log 2>&1
echo sq done
$P --kernel-trace --stats -d gpurun_out/p3_ss -o s -- python3 bench.py --steps 3 --warmup 2 --steps-only --single-stream > gpurun_out/p3_ss.log 2>&1
$P --kernel-trace --stats -d gpurun_out/p3_c3 -o s -- python3 bench.py --config C3 --steps 2 --warmup 1 --steps-only > gpurun_out/p3_c3.log 2>&1
echo c3 done
$P --kernel-trace --stats -d gpurun_out/p3_c5 -o s -- python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only > gpurun_out/p3_c5.log 2>&1
$P --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/p3_c5sq -o s -- python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p3_c5sq.log 2>&1
$P --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p3_c5fetch -o s -- python3 bench.py --config C5 --dtype fp8 --steps 1 --warmup 1 --steps-only > gpurun_out/p3_c5fetch.log 2>&1
$P --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p3_c5write -o s -- python3 bench.py --config C5 --dtype fp8 --steps 1 --warmup 1 --steps-only > gpurun_out/p3_c5write.log 2>&1
echo c5 done
for f in gpurun_out/p3_stats.log gpurun_out/p3_c3.log gpurun_out/p3_c5.log; do tail -n 2 "$f"; done

# Back in the build container: gpurun_out/ -> the tracked summaries
#   python tools/summarize_prof.py r03 --stats gpurun_out/p3_stats --fetch gpurun_out/p3_fetch --write gpurun_out/p3_write --sq gpurun_out/p3_sq \
#       --steps 5 --cmd "python3 bench.py --steps 3 --warmup 2 --steps-only" --sq-cmd "python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream" \
#       --config '{"batch": 8, "dtype": "bf16", "sam": "vit_b", "seg_tokens": 1, "with_msqp": false, "world": 1}'
#   python tools/summarize_prof.py r03_single_stream --stats gpurun_out/p3_ss --steps 5 --cmd "python3 bench.py --steps 3 --warmup 2 --steps-only --single-stream"
#   python tools/summarize_prof.py r03_c3 --stats gpurun_out/p3_c3 --steps 3 --cmd "python3 bench.py --config C3 --steps 2 --warmup 1 --steps-only"
#   python tools/summarize_prof.py r03_c5_fp8 --stats gpurun_out/p3_c5 --fetch gpurun_out/p3_c5fetch --write gpurun_out/p3_c5write --sq gpurun_out/p3_c5sq --steps 3 --config '{"batch": 8, "dtype": "fp8", "sam": "vit_h", "seg_tokens": 14, "with_msqp": true, "world": 1}' --cmd "python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only" --sq-cmd "python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only --single-stream"
