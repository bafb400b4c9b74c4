"""Build oracle/_ref/libminsu3d_ref.so from the reference's OWN common_ops sources.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (this container); the GPU
box uses the prebuilt oracle/_ref/*.so that travels with the snapshot (oracle/_ref/ is
git-ignored, not gpurun-ignored).

Recipe (ours -- the reference's setup.py / build system is not used):
  1. copy minsu3d/common_ops/src to a throw-away temp dir OUTSIDE the repo,
  2. translate the CUDA spellings with the image's own torch.utils.hipify (the reference is
     CUDA; this image has no CUDA),
  3. compile its two unity translation units + oracle/ref_shim.cpp with hipcc for gfx950,
  4. keep only the .so; the temp dir is deleted.  No reference source enters the repo.
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/minsu3d/common_ops/src"
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "libminsu3d_ref.so")


def build(force=False):
    if not os.path.isdir(REF_SRC):
        return None
    shim = os.path.join(HERE, "ref_shim.cpp")
    if (not force and os.path.exists(OUT_SO)
            and os.path.getmtime(OUT_SO) >= max(os.path.getmtime(shim), os.path.getmtime(__file__))):
        return OUT_SO
    import torch
    from torch.utils.hipify import hipify_python

    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="minsu3d_ref_build_")
    try:
        src = os.path.join(tmp, "src")
        shutil.copytree(REF_SRC, src)
        files = [f for f in glob.glob(src + "/**/*", recursive=True) if os.path.isfile(f)]
        hipify_python.hipify(project_directory=src, output_directory=src, includes=[src + "/*"],
                             extra_files=files, show_detailed=False, show_progress=False,
                             is_pytorch_extension=True, hipify_extra_files_only=True)
        tdir = os.path.dirname(torch.__file__)
        import sysconfig
        inc = [f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", f"-I{src}",
               "-I" + sysconfig.get_paths()["include"]]
        common = ["-O2", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                  "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                  "-Wno-everything"] + inc
        objs = []
        arch = ["--offload-arch=gfx950"]
        units = [(os.path.join(src, "common_ops_hip.cpp"), arch),
                 (os.path.join(src, "hip.hip"), arch),
                 (shim, arch)]
        for path, extra in units:
            obj = os.path.join(tmp, os.path.basename(path) + ".o")
            if path.endswith("common_ops_hip.cpp"):
                # host-only unit: g++, as the reference's own build does (it relies on the GNU
                # "int visited[n] = {0}" VLA-initialiser extension, which clang rejects)
                subprocess.check_call(["g++", "-c", path, "-o", obj, "-I/opt/rocm/include"]
                                      + [c for c in common if c != "-Wno-everything"] + ["-w"])
            else:
                subprocess.check_call(["hipcc", "-c", path, "-o", obj] + common + extra)
            objs.append(obj)
        subprocess.check_call(["hipcc", "-shared", "-o", OUT_SO, "--offload-arch=gfx950"] + objs + [
            f"-L{tdir}/lib", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
            f"-Wl,-rpath,{tdir}/lib"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
