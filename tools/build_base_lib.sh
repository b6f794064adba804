#!/bin/bash
# Build the library of a given commit (default HEAD) next to the working tree's as walkgpt_amd/_abl/lib_base.so, for same-box A/B runs
# (WG_LIB=walkgpt_amd/_abl/lib_base.so ...; tools/ab_step.sh).   tools/build_base_lib.sh [rev] [name] [extra hipcc flags]
rev=${1:-HEAD}
out=${2:-base}
shift; shift
extra="$@"      # e.g. -DWG_GEMM_STAMP -> lib_<out>.so
tmp=$(mktemp -d)
mkdir -p walkgpt_amd/_abl
git archive $rev walkgpt_amd/csrc | tar -x -C $tmp
objs=""
for f in $tmp/walkgpt_amd/csrc/*.hip; do
  o=$tmp/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffast-math -fno-finite-math-only -Wno-unused-result $extra -I $tmp/walkgpt_amd/csrc -c $f -o $o 2>/dev/null &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o walkgpt_amd/_abl/lib_$out.so $objs && echo walkgpt_amd/_abl/lib_$out.so
rm -rf $tmp
