#!/bin/bash
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -w tools/micro/coexec_probe.hip -o /tmp/coexec && /tmp/coexec
for rep in 1 2; do
  echo "== attention, product library"; python tools/bench_attn.py 2>/dev/null | grep -v amdgpu
  echo "== attention, waves 4-7 at s_setprio 1"; WG_LIB=walkgpt_amd/_abl/lib_aprio.so python tools/bench_attn.py 2>/dev/null | grep -v amdgpu
done
python -m pytest tests/test_gpu_modules.py -q -x -k "grounding_pipeline" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mask_decode'])"
