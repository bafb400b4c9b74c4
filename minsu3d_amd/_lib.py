"""ctypes binding of libminsu3d_hip.so (the C ABI declared in include/minsu3d_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MS3D_LIB: another build of the same library (A/B measurements of a kernel change on one box)
LIB_PATH = os.environ.get("MS3D_LIB") or os.path.join(_HERE, "lib", "libminsu3d_hip.so")
_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback in the product path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.ms3d_version.restype = C.c_char_p
        for name in ("ms3d_ballquery_workspace_bytes", "ms3d_bfs_workspace_bytes", "ms3d_hais_workspace_bytes",
                     "ms3d_coord_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_size_t
    return _lib


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed with code {rc}")


def ptr(t):
    """device (or host) pointer of a contiguous torch tensor, None -> NULL"""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


_raw_stream = None


def stream_handle():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
    (~10 us, called for every library call: 2 ms of host time per training step); the raw accessor is a plain lookup."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
