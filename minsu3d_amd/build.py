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


# ---------------------------------------------------------------------------------------------------------------
# Host-side PyTorch C++ extensions (no device code: g++ against the torch headers, linked to libminsu3d_hip.so):
#   lib/_ms3d_host.so            per-layer fast path of the engine (csrc_host/ms3d_host.cpp), used by backend.py
#   dropin_ext/COMMON_OPS.so     `import COMMON_OPS`: the reference's pybind module (csrc_host/common_ops_ext.cpp)
HOST_SRC = os.path.join(HERE, "csrc_host")
HOST_EXT_PATH = os.path.join(HERE, "lib", "_ms3d_host.so")
COMMON_OPS_EXT_DIR = os.path.join(HERE, "dropin_ext")
COMMON_OPS_EXT_PATH = os.path.join(COMMON_OPS_EXT_DIR, "COMMON_OPS.so")


def build_host(force=False, verbose=False):
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    import pybind11
    lib = build()                                   # the extensions link against the kernel library
    os.makedirs(COMMON_OPS_EXT_DIR, exist_ok=True)
    tlib = ce.library_paths()[0]
    inc = ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"],
                                                    "-I" + pybind11.get_include()]
    hdr = os.path.join(HERE, "..", "include", "minsu3d_hip.h")
    jobs = []
    for src, out, name, rpath in ((os.path.join(HOST_SRC, "ms3d_host.cpp"), HOST_EXT_PATH, "_ms3d_host", "$ORIGIN"),
                                  (os.path.join(HOST_SRC, "common_ops_ext.cpp"), COMMON_OPS_EXT_PATH, "COMMON_OPS",
                                   "$ORIGIN/../lib")):
        stale = force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr))
        if stale:
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
                         f"-DTORCH_EXTENSION_NAME={name}", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                         "-Wno-deprecated-declarations", *inc, src, "-o", out, "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch",
                         "-ltorch_cpu", "-ltorch_python", "-L" + LIB_DIR, "-lminsu3d_hip",
                         f"-Wl,-rpath,{rpath}", f"-Wl,-rpath,{tlib}"])

    def cc(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=2) as ex:
            list(ex.map(cc, jobs))
    return HOST_EXT_PATH, COMMON_OPS_EXT_PATH


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    if "--host" in sys.argv:
        print(build_host(force="--force" in sys.argv, verbose=True))
