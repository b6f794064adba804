run() { echo "=== $*"; env "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
timeout -k 10 400 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py -m gpu -x -q -k "clip or c2" 2>&1 | tail -3
for i in 1 2; do
run WG_PEEL_TAIL=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only
run WG_PEEL_TAIL=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only
done
run WG_PEEL_TAIL=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only --single-stream
run WG_PEEL_TAIL=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --steps-only --single-stream
