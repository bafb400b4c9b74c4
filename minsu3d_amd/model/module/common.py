"""U-Net building blocks over the sparse engine.  Topology and sub-module names follow the reference
(minsu3d/model/module/common.py:21-95) so state_dict keys are identical:
  ResidualBlock : conv_branch = [BN, ReLU, conv3, BN, ReLU, conv3]  (+ `downsample` = [1x1 conv] when widths differ)
  UBlock        : blocks -> (conv: [BN, ReLU, k2s2]) -> u -> (deconv: [BN, ReLU, k2s2^T]) -> cat -> blocks_tail
Every [BN, ReLU, conv] triple executes as ONE fused gather kernel (lazy BN/ReLU, see MinkowskiEngine/tensor.py)."""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from ... import MinkowskiEngine as ME
from ...MinkowskiEngine import functional as ME_F
from ...backend import get_backend

_FUSE_BLOCKS = os.environ.get("MS3D_FUSED_BLOCKS", "1") != "0"


def _bn_relu_conv(norm_fn, c_in, conv):
    return nn.Sequential(norm_fn(c_in), ME.MinkowskiReLU(inplace=True), conv)


class ResidualBlock(nn.Module):
    fuse_skip_grad = True     # the skip connection's gradient through the first convolution's backward pass (see forward)

    def __init__(self, in_channels, out_channels, dimension=3, norm_fn=None):
        super().__init__()
        norm_fn = norm_fn or ME.MinkowskiBatchNorm
        self.downsample = None
        if in_channels != out_channels:   # 1x1 projection of the skip path
            self.downsample = nn.Sequential(
                ME.MinkowskiConvolution(in_channels, out_channels, kernel_size=1, dimension=dimension))
        first = _bn_relu_conv(norm_fn, in_channels,
                              ME.MinkowskiConvolution(in_channels, out_channels, kernel_size=3, dimension=dimension))
        second = _bn_relu_conv(norm_fn, out_channels,
                               ME.MinkowskiConvolution(out_channels, out_channels, kernel_size=3, dimension=dimension))
        self.conv_branch = nn.Sequential(*first, *second)

    def _fused(self, x):
        """the whole block as one autograd node (ME.functional.ResBlockFn, or ResBlockDownFn when the skip path has its
        1x1 projection) when nothing but the plain training-mode chain is asked for; None = take the module chain"""
        if (not self.training or not torch.is_grad_enabled() or x._pending is not None or _FUSE_BLOCKS is False
                or not x._F.is_cuda):
            return None
        st = x._stats
        if isinstance(st, tuple):
            if self.downsample is None or sum(p.size(2) for p in st) != x._F.size(1):
                return None
        elif not torch.is_tensor(st) or st.numel() == 0:
            return None
        if getattr(get_backend(), "name", "") != "hip":
            return None
        cb = self.conv_branch
        plan = self.__dict__.get("_fuse_plan")
        if plan is None:
            ok = (len(cb) == 6 and isinstance(cb[0], ME.MinkowskiBatchNorm) and isinstance(cb[1], ME.MinkowskiReLU)
                  and isinstance(cb[2], ME.MinkowskiConvolution) and isinstance(cb[3], ME.MinkowskiBatchNorm)
                  and isinstance(cb[4], ME.MinkowskiReLU) and isinstance(cb[5], ME.MinkowskiConvolution)
                  and all(c.kernel_size == 3 and c.stride == 1 for c in (cb[2], cb[5]))
                  and all(b.bn.affine and b.bn.track_running_stats and b.bn.momentum is not None for b in (cb[0], cb[3]))
                  and cb[2].out_channels == cb[5].in_channels == cb[5].out_channels
                  and (self.downsample is None or (len(self.downsample) == 1
                                                   and isinstance(self.downsample[0], ME.MinkowskiConvolution)
                                                   and self.downsample[0].kernel_size == 1 and self.downsample[0].stride == 1
                                                   and self.downsample[0].kernel.dim() == 2)))
            plan = self.__dict__["_fuse_plan"] = (tuple(cb) + (tuple(self.downsample) if self.downsample is not None else ())) if ok else ()
        if not plan:
            return None
        for m in plan:            # somebody is watching an inner module: its hooks must fire
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
                return None
        bn0, _, conv1, bn1, _, conv2 = plan[:6]
        if not conv1.kernel.is_cuda:
            return None
        if not (bn0.training and bn1.training and conv2.training):
            # a BatchNorm frozen with .eval() inside a train()-mode block normalises with its running statistics (and a
            # convolution in eval mode collects none for the next one): the module chain honours that
            return None
        cm, ts = x.coordinate_manager, x.tensor_stride
        nbr, V = cm.k3(ts), cm.size(ts)
        spec = ME_F.ConvSpec(nbr, nbr, V, V, 27, conv1.in_channels, conv1.out_channels, True)
        if self.downsample is None:
            y, stats = ME_F.ResBlockFn.apply(x._F, conv1._kernel(), bn0.bn.weight, bn0.bn.bias, conv2._kernel(),
                                             bn1.bn.weight, bn1.bn.bias, spec, st, bn0.bn, bn1.bn, True)
        else:
            if self.downsample._forward_hooks or self.downsample._forward_pre_hooks:
                return None
            y, stats = ME_F.ResBlockDownFn.apply(x._F, plan[6]._kernel(), conv1._kernel(), bn0.bn.weight, bn0.bn.bias,
                                                 conv2._kernel(), bn1.bn.weight, bn1.bn.bias, spec, cm.identity(ts), st, bn0.bn,
                                                 bn1.bn)
        bn0._pending_batches += 1
        bn1._pending_batches += 1
        return x._like(y, stats=stats)

    def forward(self, x):
        fused = self._fused(x)
        if fused is not None:
            return fused
        skip = x if self.downsample is None else self.downsample(x)
        layers = self.__dict__.get("_layers")
        if layers is None or len(layers) != len(self.conv_branch):
            # (slicing an nn.Sequential builds a new module on every call: 32 blocks x ~20 us per step)
            layers = self.__dict__["_layers"] = tuple(self.conv_branch)
        # identity skip: out = x + branch(x) sends the output gradient to x twice.  Instead of autograd's elementwise add per
        # block, the last convolution hands its dy to the first one, whose BatchNorm-backward pass adds it to dx
        # (MinkowskiEngine/functional.py, SkipLink).  Only when the skipped tensor IS the first convolution's input rows.
        link = None
        if (self.fuse_skip_grad and self.downsample is None and torch.is_grad_enabled() and x._pending is None
                and isinstance(layers[-1], ME.MinkowskiConvolution)):
            link = ME.SkipLink()
        h = x
        for layer in layers[:-1]:
            if link is not None and not link.armed and isinstance(layer, ME.MinkowskiConvolution):
                if h._F is not x._F:
                    link = None          # something already touched the rows: leave the block to autograd
                    h = layer(h)
                else:
                    h = layer(h, skip=("head", link))
                continue
            h = layer(h)
        # `y = conv(h); y += skip` of the reference (common.py:43-49) as one kernel: the residual add and the batch
        # statistics the next BatchNorm needs ride in the last convolution's epilogue
        if link is not None and link.armed:
            return layers[-1](h, residual=skip, skip=("tail", link))
        return layers[-1](h, residual=skip)


def _bn_relu_conv_fused(seq, x):
    """Sequential(BatchNorm, ReLU, strided / transposed convolution) as one autograd node (ME.functional.BnReluConvFn) in
    training mode when the input still carries its producer's statistics; otherwise (and when somebody hooks an inner
    module) the module chain"""
    if (_FUSE_BLOCKS is False or not seq.training or not torch.is_grad_enabled() or x._pending is not None or not x._F.is_cuda
            or x._stats is None or (torch.is_tensor(x._stats) and x._stats.numel() == 0)):
        return seq(x)
    plan = seq.__dict__.get("_fuse_plan")
    if plan is None:
        ok = (len(seq) == 3 and isinstance(seq[0], ME.MinkowskiBatchNorm) and isinstance(seq[1], ME.MinkowskiReLU)
              and isinstance(seq[2], (ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose))
              and seq[2].kernel_size == 2 and seq[2].stride == 2
              and seq[0].bn.affine and seq[0].bn.track_running_stats and seq[0].bn.momentum is not None)
        plan = seq.__dict__["_fuse_plan"] = tuple(seq) if ok else ()
    if not plan or getattr(get_backend(), "name", "") != "hip" or seq._forward_hooks or seq._forward_pre_hooks:
        return seq(x)
    for m in plan:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            return seq(x)
    bn, _, conv = plan
    st = x._stats
    if isinstance(st, tuple) and sum(p.size(2) for p in st) != x._F.size(1):
        return seq(x)
    if not (bn.training and conv.training):     # a frozen BatchNorm inside a train()-mode network: the module chain
        return seq(x)
    cm, ts = x.coordinate_manager, x.tensor_stride
    if isinstance(conv, ME.MinkowskiConvolutionTranspose):
        fine = ts // 2
        down, up = cm.k2(fine)
        spec = ME_F.ConvSpec(up, down, cm.size(ts), cm.size(fine), 8, conv.in_channels, conv.out_channels, False)
        out_ts = fine
    else:
        down, up = cm.k2(ts)
        spec = ME_F.ConvSpec(down, up, cm.size(ts), cm.size(2 * ts), 8, conv.in_channels, conv.out_channels, False)
        out_ts = 2 * ts
    y, stats = ME_F.BnReluConvFn.apply(x._F, conv._kernel(), bn.bn.weight, bn.bn.bias, spec, st, bn.bn, True)
    bn._pending_batches += 1
    return x._like(y, tensor_stride=out_ts, stats=stats)


class UBlock(nn.Module):
    """recursive encoder/decoder level; n_planes = channel widths from this level down"""

    def __init__(self, n_planes, norm_fn, block_reps, block):
        super().__init__()
        self.nPlanes = list(n_planes)
        c = self.nPlanes[0]
        self.blocks = nn.Sequential(OrderedDict(
            (f"block{i}", block(c, c, 3, norm_fn)) for i in range(block_reps)))
        if len(self.nPlanes) > 1:
            c_next = self.nPlanes[1]
            self.conv = _bn_relu_conv(norm_fn, c, ME.MinkowskiConvolution(c, c_next, kernel_size=2, stride=2, dimension=3))
            self.u = UBlock(self.nPlanes[1:], norm_fn, block_reps, block)
            self.deconv = _bn_relu_conv(
                norm_fn, c_next, ME.MinkowskiConvolutionTranspose(c_next, c, kernel_size=2, stride=2, dimension=3))
            self.blocks_tail = nn.Sequential(OrderedDict(
                (f"block{i}", block(c * (2 - i), c, 3, norm_fn)) for i in range(block_reps)))

    def forward(self, x):
        skip = self.blocks(x)
        if len(self.nPlanes) == 1:
            return skip
        up = _bn_relu_conv_fused(self.deconv, self.u(_bn_relu_conv_fused(self.conv, skip)))
        return self.blocks_tail(ME.cat(skip, up))
