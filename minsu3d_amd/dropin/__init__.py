"""Drop-in module names for the reference's two native dependencies on the hot path.

The reference imports its native layer under two top-level names:

    import COMMON_OPS                    minsu3d/common_ops/functions/*.py (the pybind module of common_ops_api.cpp:6-30)
    import MinkowskiEngine as ME         minsu3d/model/module/backbone.py:4, common.py:2, tiny_unet.py, general_model.py:4,
                                         data/data_module.py, data/dataset/general_dataset.py

Either put this directory on PYTHONPATH (it holds `COMMON_OPS.py` and `MinkowskiEngine.py`), or call `install()`
once before the reference's modules are imported; both names then resolve to the MI355X implementation
(libminsu3d_hip.so behind the C ABI of include/minsu3d_hip.h).  INTEGRATION.md section 1 shows both ways."""
import importlib
import sys


def load_extension():
    """the PyTorch-ROCm C++ extension form of COMMON_OPS (minsu3d_amd/csrc_host/common_ops_ext.cpp, built by
    `minsu3d_amd.build.build_host()` into minsu3d_amd/dropin_ext/COMMON_OPS.so; putting that directory on PYTHONPATH
    makes a plain `import COMMON_OPS` resolve to it).  Raises when it has not been built."""
    import importlib.util
    import os
    import torch  # noqa: F401  (the extension links against libtorch: it must be loaded first)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin_ext", "COMMON_OPS.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m minsu3d_amd.build --host`")
    spec = importlib.util.spec_from_file_location("COMMON_OPS", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install(native=None):
    """register `COMMON_OPS` and `MinkowskiEngine` (+ `MinkowskiEngine.utils`) in sys.modules.
    native=True (or MS3D_DROPIN_NATIVE=1): `COMMON_OPS` is the C++ extension (always the HIP library); default: the
    Python module over ctypes, which answers through `backend.get_backend()` -- what the CPU-side tests rely on to put
    the oracle test double behind the reference's own wrapper files."""
    import os
    if native is None:
        native = os.environ.get("MS3D_DROPIN_NATIVE", "0") == "1"
    if native:
        _ops = load_extension()
    else:
        from . import COMMON_OPS as _ops
    me = importlib.import_module("minsu3d_amd.MinkowskiEngine")
    sys.modules.setdefault("COMMON_OPS", _ops)
    sys.modules.setdefault("MinkowskiEngine", me)
    sys.modules.setdefault("MinkowskiEngine.utils", me.utils)
    return _ops, me
