"""Networks assembled from the U-Net blocks: the `Backbone` (sparse U-Net + semantic / offset heads over points;
reference minsu3d/model/module/backbone.py:8-43) and the `TinyUnet` that refines voxelised proposals (reference
minsu3d/model/module/tiny_unet.py:7-19).  Sub-module names follow the reference so checkpoints keep their keys."""
import os

import torch
import torch.nn as nn

from ... import MinkowskiEngine as ME
from ...MinkowskiEngine import functional as ME_F
from ...backend import get_backend
from .common import ResidualBlock, UBlock


class PointLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose forward runs the tall-skinny [N_points, 16..32] matmul
    through the engine's K = 1 MFMA path instead of a library GEMM"""

    def forward(self, x):
        if x.dim() == 2 and x.size(0) >= 4096:
            return ME_F.dense_linear(x, self.weight, self.bias)
        return super().forward(x)


class PointBatchNormReLU(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and state_dict keys) followed by ReLU over [N_points, C] rows through the
    engine's statistics / apply / backward kernels (float4 streaming passes; torch's channels-last BatchNorm kernels need
    ~270 us forward + backward for 575k x 16 rows, these ~110 us).  The nn.ReLU that follows it in the reference's
    Sequential (backbone.py:21-30) is replaced by nn.Identity: no parameters, same keys."""

    def forward(self, x):
        if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) >= 4096 and self.affine):
            return torch.relu(super().forward(x))
        use_batch = self.training or not self.track_running_stats
        with torch.no_grad():
            if use_batch:
                track = self.training and self.track_running_stats
                rm, rv = (self.running_mean, self.running_var) if track else (None, None)
                if self.momentum is None:
                    mom = 1.0 / (int(self.num_batches_tracked) + 1) if track else 0.0
                else:
                    mom = self.momentum
                mean, invstd, scale, shift = get_backend().bn_stats(x.detach().contiguous(), self.eps, mom,
                                                                    self.weight.detach(), self.bias.detach(), rm, rv)
                if track and self.num_batches_tracked is not None:
                    self.num_batches_tracked += 1
            else:
                invstd = torch.rsqrt(self.running_var + self.eps)
                mean = self.running_mean
                scale = self.weight * invstd
                shift = self.bias - mean * scale
        pending = dict(gamma=self.weight, beta=self.bias, mean=mean.contiguous(), invstd=invstd.contiguous(),
                       scale=scale.contiguous(), shift=shift.contiguous(), relu=True, training=use_batch)
        return ME_F.bn_act(x.contiguous(), pending)


def _head(c_in, c_out):
    return nn.Sequential(PointLinear(c_in, c_in), PointBatchNormReLU(c_in), nn.Identity(), PointLinear(c_in, c_out))


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
        if (point_features.is_cuda and torch.is_grad_enabled() and point_features.requires_grad and self.training
                and os.environ.get("MS3D_EARLY_HEADS", "0") == "2"):
            # Scheduling only (GeneralModel._early_point_backward, mode 2): the two heads run on their own stream; the
            # caller's stream waits for their FORWARD (it needs the scores / offsets), while the losses and the heads'
            # backward -- queued on that stream right behind -- run beside the grouping window
            from ...backend import heads_stream
            main, side = torch.cuda.current_stream(), heads_stream(point_features.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                sem = self.semantic_branch(point_features)
                off = self.offset_branch(point_features)
                done = torch.cuda.Event()
                done.record(side)
            point_features.record_stream(side)
            main.wait_event(done)
            sem.record_stream(main); off.record_stream(main)
            return {"point_features": point_features, "semantic_scores": sem, "point_offsets": off, "_heads_stream": side}
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
