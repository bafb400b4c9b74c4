"""Networks assembled from the U-Net blocks: the `Backbone` (sparse U-Net + semantic / offset heads over points;
reference minsu3d/model/module/backbone.py:8-43) and the `TinyUnet` that refines voxelised proposals (reference
minsu3d/model/module/tiny_unet.py:7-19).  Sub-module names follow the reference so checkpoints keep their keys."""
import torch.nn as nn

from ... import MinkowskiEngine as ME
from ...MinkowskiEngine import functional as ME_F
from .common import ResidualBlock, UBlock


class PointLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose forward runs the tall-skinny [N_points, 16..32] matmul
    through the engine's K = 1 MFMA path instead of a library GEMM"""

    def forward(self, x):
        if x.dim() == 2 and x.size(0) >= 4096:
            return ME_F.dense_linear(x, self.weight, self.bias)
        return super().forward(x)


def _head(c_in, c_out):
    return nn.Sequential(PointLinear(c_in, c_in), nn.BatchNorm1d(c_in), nn.ReLU(inplace=True), PointLinear(c_in, c_out))


class Backbone(nn.Module):
    def __init__(self, input_channel, output_channel, block_channels, block_reps, sem_classes):
        super().__init__()
        m = output_channel
        self.unet = nn.Sequential(
            ME.MinkowskiConvolution(in_channels=input_channel, out_channels=m, kernel_size=3, dimension=3),
            UBlock([m * c for c in block_channels], ME.MinkowskiBatchNorm, block_reps, ResidualBlock),
            ME.MinkowskiBatchNorm(m),
            ME.MinkowskiReLU(inplace=True))
        self.semantic_branch = _head(m, sem_classes)
        self.offset_branch = _head(m, 3)
        self.n_levels = len(block_channels)
        self.level_channels = [m * c for c in block_channels]

    def forward(self, voxel_features, voxel_coordinates, v2p_map):
        x = ME.SparseTensor(features=voxel_features, coordinates=voxel_coordinates)
        x.coordinate_manager.prepare(self.n_levels)
        point_features = ME.gather_rows(self.unet(x).features, v2p_map)   # voxel -> point broadcast
        return {"point_features": point_features,
                "semantic_scores": self.semantic_branch(point_features),
                "point_offsets": self.offset_branch(point_features)}


class TinyUnet(nn.Module):
    """two-level U-Net (channel -> 2*channel -> channel) + BN + ReLU, run once on all proposals of a batch"""

    def __init__(self, channel):
        super().__init__()
        levels = [channel, 2 * channel]
        self.unet = nn.Sequential(UBlock(levels, ME.MinkowskiBatchNorm, 2, ResidualBlock),
                                  ME.MinkowskiBatchNorm(channel), ME.MinkowskiReLU(inplace=True))

    def forward(self, proposals_voxel_feats):
        proposals_voxel_feats.coordinate_manager.prepare(2)
        return self.unet(proposals_voxel_feats)
