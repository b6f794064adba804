timeout -k 10 200 python tools/bench_gemm_fr.py 2>&1 | grep -v amdgpu.ids
echo "=== stamps"; WG_LIB=walkgpt_amd/_abl/lib_gstamp.so timeout -k 10 200 python tools/gemm_fr_stamps.py 2>&1 | grep -v amdgpu.ids
