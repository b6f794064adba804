"""Diagnostic builds of the library next to the product one:  python tools/build_variant.py <tag> -DWG_GEMM_C_AUX=16 ...
-> walkgpt_amd/_abl/lib_<tag>.so (git-ignored, travels to the GPU box); run anything against it with WG_LIB=walkgpt_amd/_abl/lib_<tag>.so."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from walkgpt_amd import _build  # noqa: E402

tag, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "walkgpt_amd", "_abl")
obj_dir = os.path.join(out_dir, "obj_" + tag)
os.makedirs(obj_dir, exist_ok=True)


def cc(src):
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
    # only gemm.hip / attn.hip read the experiment macros: every other object is shared with the product build
    base = os.path.join(_build.OBJ, os.path.basename(src)[:-4] + ".o")
    if not any(m in open(src).read() for m in ("WG_GEMM_", "WG_ATTN_", "WG_DEC_")) and os.path.exists(base):
        return base
    subprocess.run([_build.HIPCC] + _build.FLAGS + extra + ["-c", src, "-o", obj], check=True, capture_output=True)
    return obj


_build.build_library()
srcs = _build.sources()
if "-DWG_GEMM_FR" in extra:      # the experimental one-barrier GEMM lives beside the other probes, not in the product library
    srcs = srcs + [os.path.join(ROOT, "tools", "micro", "gemm_fr.hip")]
with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(cc, srcs))
lib = os.path.join(out_dir, "lib_%s.so" % tag)
subprocess.run([_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
