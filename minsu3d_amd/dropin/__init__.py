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


def install():
    """register `COMMON_OPS` and `MinkowskiEngine` (+ `MinkowskiEngine.utils`) in sys.modules"""
    from . import COMMON_OPS as _ops
    me = importlib.import_module("minsu3d_amd.MinkowskiEngine")
    sys.modules.setdefault("COMMON_OPS", _ops)
    sys.modules.setdefault("MinkowskiEngine", me)
    sys.modules.setdefault("MinkowskiEngine.utils", me.utils)
    return _ops, me
