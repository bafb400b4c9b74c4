"""CPU: the C-ABI library loads and exports every symbol include/minsu3d_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "minsu3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ms3d_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("ms3d_ballquery_batch_p", "ms3d_pg_bfs_cluster", "ms3d_sg_bfs_cluster", "ms3d_hierarchical_aggregation",
                 "ms3d_sec_mean", "ms3d_sec_min", "ms3d_sec_max", "ms3d_roipool_fp", "ms3d_roipool_bp",
                 "ms3d_global_avg_pool_fp", "ms3d_global_avg_pool_bp", "ms3d_get_iou", "ms3d_get_mask_iou_on_cluster",
                 "ms3d_get_mask_iou_on_pred", "ms3d_get_mask_label", "ms3d_sparse_quantize", "ms3d_kmap_k3",
                 "ms3d_downsample", "ms3d_kmap_k2", "ms3d_spconv_forward", "ms3d_spconv_backward_weight", "ms3d_bn_stats"):
        assert must in syms, must


def test_library_exports_every_declared_symbol():
    from minsu3d_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.ms3d_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ms3d_version()
    lib.ms3d_bfs_workspace_bytes.restype = ctypes.c_size_t
    assert lib.ms3d_bfs_workspace_bytes(1000) > 1000 * 4 * 10      # pure host arithmetic, no device needed


def test_product_path_refuses_cpu_tensors_and_missing_library(monkeypatch):
    import pytest
    import torch
    from minsu3d_amd import _lib, backend
    be = backend.HipBackend()
    with pytest.raises(_lib.HipLibraryError):
        be.sec_mean(torch.zeros(4, 3), torch.tensor([0, 4], dtype=torch.int32))     # no CPU fallback
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libminsu3d_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.HipLibraryError):
        backend.HipBackend()


def test_host_extensions_build_and_export_the_reference_surface():
    """the PyTorch C++ extensions over the C ABI (built by __graft_entry__.build() with the host compiler): `COMMON_OPS`
    exports exactly the 15 functions of the reference's pybind module (common_ops_api.cpp:6-30) and the engine fast path
    its four entry points; loading them needs no GPU (no compute here)"""
    import torch  # noqa: F401
    from minsu3d_amd import build as hipbuild
    import minsu3d_amd.dropin as dropin
    hipbuild.build_host()
    ops = dropin.load_extension()
    want = {"sg_bfs_cluster", "global_avg_pool_fp", "global_avg_pool_bp", "ballquery_batch_p", "sec_mean", "sec_min",
            "sec_max", "roipool_fp", "roipool_bp", "get_iou", "get_mask_iou_on_cluster", "get_mask_iou_on_pred",
            "get_mask_label", "pg_bfs_cluster", "hierarchical_aggregation"}
    assert {n for n in dir(ops) if not n.startswith("_")} == want
    import minsu3d_amd.dropin.COMMON_OPS as shim
    assert set(shim.__all__) == want                     # the ctypes form of the module offers the same names
    from minsu3d_amd.backend import _load_host_ext
    ext = _load_host_ext()
    assert ext is not None and all(hasattr(ext, n) for n in ("conv_layer_forward", "conv_layer_backward", "bn_finalize",
                                                               "gather_rows"))
