"""CPU: the drop-in module names (`COMMON_OPS`, `MinkowskiEngine`) exist with the reference's surface.
No compute (that needs the GPU: tests/test_dropin_gpu.py)."""
import inspect
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# positional parameter counts of the 15 functions, read off the reference's headers
# (minsu3d/common_ops/src/*/**.h; registered in common_ops_api.cpp:6-30)
ARITY = {"sg_bfs_cluster": 8, "global_avg_pool_fp": 5, "global_avg_pool_bp": 5, "ballquery_batch_p": 8, "sec_mean": 5,
         "sec_min": 5, "sec_max": 5, "roipool_fp": 6, "roipool_bp": 6, "get_iou": 7, "get_mask_iou_on_cluster": 7,
         "get_mask_iou_on_pred": 8, "get_mask_label": 11, "pg_bfs_cluster": 7, "hierarchical_aggregation": 21}
ME_SYMBOLS = ("SparseTensor", "MinkowskiConvolution", "MinkowskiConvolutionTranspose", "MinkowskiBatchNorm",
              "MinkowskiReLU", "cat", "utils")


def test_install_registers_both_names():
    import minsu3d_amd.dropin as dropin
    dropin.install()
    import COMMON_OPS
    import MinkowskiEngine as ME
    assert sorted(COMMON_OPS.__all__) == sorted(ARITY)
    for name, n in ARITY.items():
        params = inspect.signature(getattr(COMMON_OPS, name)).parameters
        assert len(params) == n, (name, len(params))
    for s in ME_SYMBOLS:
        assert hasattr(ME, s), s
    assert callable(ME.utils.sparse_quantize) and callable(ME.utils.sparse_collate)


def test_pythonpath_only():
    """the directory alone on PYTHONPATH is enough (no install() call)"""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "minsu3d_amd", "dropin"), ROOT])
    code = ("import COMMON_OPS, MinkowskiEngine as ME, MinkowskiEngine.utils as U;"
            "assert callable(COMMON_OPS.pg_bfs_cluster) and callable(U.sparse_quantize);"
            "conv = ME.MinkowskiConvolution(4, 8, kernel_size=3, dimension=3); assert tuple(conv.kernel.shape) == (27, 4, 8);"
            "print('ok')")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/minsu3d/common_ops/functions"),
                    reason="the reference tree only exists in the build container")
def test_reference_wrappers_import_against_the_shim():
    """the reference's own wrapper modules import (and bind their autograd Functions) with our COMMON_OPS in place of
    the pybind module -- run in a subprocess so the reference never enters this test session's module table"""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "minsu3d_amd", "dropin"), ROOT])
    code = ("import sys, importlib.util as u\n"
            "base = '/root/reference/minsu3d/common_ops/functions/'\n"
            "for n in ('common_ops', 'pointgroup_ops', 'hais_ops', 'softgroup_ops'):\n"
            "    spec = u.spec_from_file_location('ref_' + n, base + n + '.py'); m = u.module_from_spec(spec)\n"
            "    spec.loader.exec_module(m)\n"
            "    assert m.COMMON_OPS.__name__.endswith('COMMON_OPS')\n"
            "print('ok')")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
