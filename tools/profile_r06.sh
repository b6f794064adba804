#!/bin/bash
# The rocprofv3 passes behind profiles/r06_* (run on the GPU box from the repo root; outputs under gpurun_out/).  PMC passes are separate runs with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P="rocprofv3 --output-format csv"
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
$P --kernel-trace --stats -d gpurun_out/p6_stats -o s -- python3 bench.py --steps 3 --warmup 2 --steps-only > gpurun_out/p6_stats.log 2>&1
echo stats done
$P --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p6_fetch -o s -- python3 bench.py --steps 1 --warmup 1 --steps-only > gpurun_out/p6_fetch.log 2>&1
$P --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p6_write -o s -- python3 bench.py --steps 1 --warmup 1 --steps-only > gpurun_out/p6_write.log 2>&1
echo traffic done
$P --kernel-trace --pmc $SQ -d gpurun_out/p6_sq -o s -- python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p6_sq.log 2>&1
$P --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES -d gpurun_out/p6_lds -o s -- python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p6_lds.log 2>&1 || echo "lds pass failed (counter name?)"
$P --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/p6_l2 -o s -- python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p6_l2.log 2>&1 || echo "l2 pass failed (counter name?)"
echo sq done
$P --kernel-trace --stats -d gpurun_out/p6_ss -o s -- python3 bench.py --steps 3 --warmup 2 --steps-only --single-stream > gpurun_out/p6_ss.log 2>&1
$P --kernel-trace --stats -d gpurun_out/p6_c3 -o s -- python3 bench.py --config C3 --steps 2 --warmup 1 --steps-only > gpurun_out/p6_c3.log 2>&1
echo c3 done
$P --kernel-trace --stats -d gpurun_out/p6_c5 -o s -- python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only > gpurun_out/p6_c5.log 2>&1
$P --kernel-trace --pmc $SQ -d gpurun_out/p6_c5sq -o s -- python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only --single-stream > gpurun_out/p6_c5sq.log 2>&1
$P --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p6_c5fetch -o s -- python3 bench.py --config C5 --dtype fp8 --steps 1 --warmup 1 --steps-only > gpurun_out/p6_c5fetch.log 2>&1
$P --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p6_c5write -o s -- python3 bench.py --config C5 --dtype fp8 --steps 1 --warmup 1 --steps-only > gpurun_out/p6_c5write.log 2>&1
echo c5 done
for f in gpurun_out/p6_stats.log gpurun_out/p6_c3.log gpurun_out/p6_c5.log; do tail -n 1 "$f" | cut -c1-160; done

# Back in the build container: gpurun_out/ -> the tracked summaries
#   python tools/summarize_prof.py r06 --stats gpurun_out/p6_stats --fetch gpurun_out/p6_fetch --write gpurun_out/p6_write --sq gpurun_out/p6_sq \
#       --steps 5 --cmd "python3 bench.py --steps 3 --warmup 2 --steps-only" --sq-cmd "python3 bench.py --steps 2 --warmup 1 --steps-only --single-stream" \
#       --config '{"batch": 8, "dtype": "bf16", "sam": "vit_b", "seg_tokens": 1, "with_msqp": false, "world": 1, "fp8_clip": false}'
#   python tools/summarize_prof.py r06_single_stream --stats gpurun_out/p6_ss --steps 5 --cmd "python3 bench.py --steps 3 --warmup 2 --steps-only --single-stream"
#   python tools/summarize_prof.py r06_c3 --stats gpurun_out/p6_c3 --steps 3 --cmd "python3 bench.py --config C3 --steps 2 --warmup 1 --steps-only"
#   python tools/summarize_prof.py r06_c5_fp8 --stats gpurun_out/p6_c5 --fetch gpurun_out/p6_c5fetch --write gpurun_out/p6_c5write --sq gpurun_out/p6_c5sq --steps 3 \
#       --config '{"batch": 8, "dtype": "fp8", "sam": "vit_h", "seg_tokens": 14, "with_msqp": true, "world": 1, "fp8_clip": false}' \
#       --cmd "python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only" --sq-cmd "python3 bench.py --config C5 --dtype fp8 --steps 2 --warmup 1 --steps-only --single-stream"
