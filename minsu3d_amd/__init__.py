"""minsu3d_amd -- MI355X (gfx950) native hot path for minsu3d-style sparse-voxel instance segmentation.

Sub-packages mirror the reference's import surface for the hot path only:
  minsu3d_amd.common_ops.functions.{common_ops,pointgroup_ops,hais_ops,softgroup_ops}
      <-> minsu3d/common_ops/functions/*.py   (same operator names / argument order)
  minsu3d_amd.MinkowskiEngine
      <-> the 9 MinkowskiEngine symbols the reference models import
"""
__version__ = "0.1.0"
