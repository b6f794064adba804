#!/bin/bash
# The bench line of every configuration the build reports, on one box, back to back (gpurun_out/r5_matrix.log -> profiles/r05_bench_matrix.md).
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; python bench.py "$@" --steps-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms/step |', d['config']['workload'][:90])"; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --single-stream
run --steps 20 --warmup 5 --single-stream --tail-tiles
run --steps 20 --warmup 5 --seg-tokens 14
run --steps 12 --warmup 3 --batch 16
run --steps 8 --warmup 2 --config C4
run --steps 20 --warmup 5 --clip-skip-unused-layer
run --steps 20 --warmup 5 --dtype fp8
run --steps 6 --warmup 2 --config C3
run --steps 6 --warmup 2 --config C3 --single-stream
run --steps 6 --warmup 2 --config C5
run --steps 6 --warmup 2 --config C5 --dtype fp8
run --steps 6 --warmup 2 --config C5 --dtype fp8 --fp8-clip
run --steps 20 --warmup 5
