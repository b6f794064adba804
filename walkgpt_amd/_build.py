"""Compile walkgpt_amd/csrc/*.hip for gfx950 into walkgpt_amd/libwalkgpt_hip.so (in-tree, so it travels with
the repo snapshot to the GPU box).  Plain hipcc, no torch extension machinery: the library is a C-ABI."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libwalkgpt_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math", "-fno-finite-math-only",
         "-Wno-unused-result", "-I", CSRC] + os.environ.get("WG_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DWG_ATTN_STAMP (diagnostics)


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())   # names, not paths: the tree is built here and loaded from another directory on the GPU box
            h.update(f.read())
    h.update(" ".join(f for f in FLAGS if f != CSRC).encode())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_library(force=False, verbose=True):
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp.txt")
    want = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == want:
        return LIB
    hdr_digest = _digest(hdrs)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        tag = obj + ".tag"
        d = _digest([src]) + hdr_digest
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == d:
            return obj
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[walkgpt_amd build]", " ".join(cmd), file=sys.stderr, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        with open(tag, "w") as f:
            f.write(d)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
