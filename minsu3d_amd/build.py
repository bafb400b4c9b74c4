"""Build libminsu3d_hip.so in-tree: plain `hipcc --offload-arch=gfx950` on the hand-written
.hip sources (no hipify, no torch extension machinery).  Cross-compiles without a GPU."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libminsu3d_hip.so")
OBJ_DIR = os.path.join(HERE, "..", "build", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Experiment builds (knock-out variants, tools/scripts/ko_build_run.sh): extra flags build into THEIR OWN object
# directory and library file (suffix = a digest of the flags), so that the product library and its objects are never
# overwritten by a variant and a later plain build() cannot pick a variant's objects up; the returned path
# goes into MS3D_LIB to run on the variant.
_EXTRA = os.environ.get("MS3D_EXTRA_HIPCC_FLAGS", "").split()
if _EXTRA:
    import hashlib
    _tag = hashlib.sha256(" ".join(_EXTRA).encode()).hexdigest()[:10]
    FLAGS += _EXTRA
    OBJ_DIR = os.path.join(HERE, "..", "build", "obj_variant_" + _tag)
    LIB_PATH = os.path.join(HERE, "..", "build", "variants", "libminsu3d_hip_" + _tag + ".so")
    LIB_DIR = os.path.dirname(LIB_PATH)


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "minsu3d_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hm = _headers_mtime()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))

    def cc(job):
        cmd = ["hipcc", "-c", job[0], "-o", job[1]] + FLAGS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB_PATH):
        subprocess.check_call(["hipcc", "-shared", "-o", LIB_PATH, "--offload-arch=gfx950"] + objs)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
