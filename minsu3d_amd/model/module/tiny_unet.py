"""Two-level U-Net run on the voxelised proposals (reference minsu3d/model/module/tiny_unet.py:7-19)."""
import torch.nn as nn

from ... import MinkowskiEngine as ME
from .common import ResidualBlock, UBlock


class TinyUnet(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.unet = nn.Sequential(
            UBlock([channel, 2 * channel], ME.MinkowskiBatchNorm, 2, ResidualBlock),
            ME.MinkowskiBatchNorm(channel),
            ME.MinkowskiReLU(inplace=True))

    def forward(self, proposals_voxel_feats):
        return self.unet(proposals_voxel_feats)
