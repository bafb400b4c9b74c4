"""nn.Module surface of the MinkowskiEngine subset (same constructor arguments, parameter names and init as ME
v0.5.4): `kernel` of shape (K, in, out) -- (in, out) for kernel_size 1 -- no bias,
uniform(-1/sqrt(in*K), +1/sqrt(in*K)); MinkowskiBatchNorm wraps `self.bn = nn.BatchNorm1d`."""
import math
import os

import torch
import torch.nn as nn

from ..backend import get_backend
from . import functional as Fn
from .tensor import SparseTensor


class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not bias, "only what the reference uses is implemented"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        K = kernel_size ** 3
        self.kernel_volume = K
        shape = (in_channels, out_channels) if (kernel_size == 1 and stride == 1) else (K, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        s = 1.0 / math.sqrt(in_channels * K)
        with torch.no_grad():
            self.kernel.uniform_(-s, s)

    def _kernel(self):
        """the kernel this forward uses: the parameter, or -- inside a prepare_conv_weights window in training -- its
        alias behind the group's GroupFlushFn node (same storage; the gradient reaches the parameter through that node)"""
        eff = self.__dict__.get("_kernel_eff")
        if eff is not None and eff[1] == getattr(get_backend(), "weight_token", None):
            return eff[0]
        return self.kernel

    def extra_repr(self):
        return f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"


def _flush_group(name):
    """Which deferred-reduction group a convolution belongs to, from its module path.  A group's slab reductions run when
    its LAST layer has finished its backward pass, and only then do its gradients reach the parameters (and a
    data-parallel wrapper's bucket hooks), so the groups follow the order of the backward pass and keep the bulk of the
    bytes early: the proposal network first; then the decoder side of the three finest U-Net levels; then everything
    from level 3 down (~90 % of the parameter bytes, complete half way through the pass); last the encoder side of the
    finest levels and the input convolution (~1.5 MB at m = 16)."""
    parts = name.split(".")
    if "backbone" not in parts:
        return 0
    depth = parts.count("u")
    if depth >= 3:
        return 2
    return 1 if ("blocks_tail" in parts or "deconv" in parts) else 3


def prepare_conv_weights(root):
    """Lay out the kernel-side images of EVERY convolution weight under `root` in one launch (instead of one ~5 us
    launch per layer inside each convolution call; ~90 per U-Net step) and stamp the parameters.  The stamp is valid
    until `release_conv_weights()`: call the pair around a forward pass during which the weights do not change
    (GeneralModel.__call__ does).  Convolutions called outside such a window lay out their own weights as before.
    In training the convolutions are also handed their kernels through `GroupFlushFn` nodes (functional.py): the slab
    reductions behind the backward-weight kernels of a whole group of layers then run as one launch.
    Not part of ME's API."""
    be = get_backend()
    if not hasattr(be, "prep_weights_multi") or os.environ.get("MS3D_WEIGHT_MULTI", "1") == "0":
        return
    layers = []
    convs = root.__dict__.get("_ms3d_convs")
    if convs is None:
        # (walking the module tree costs 0.7 ms per step on the 400-module networks; the set of convolutions of a built
        # model does not change -- `del model._ms3d_convs` after surgery on it)
        named = [(n, m) for n, m in root.named_modules() if isinstance(m, _ConvBase)]
        convs = root.__dict__["_ms3d_convs"] = tuple(m for _, m in named)
        root.__dict__["_ms3d_conv_groups"] = tuple(_flush_group(n) for n, _ in named)
    for m, grp in zip(convs, root.__dict__["_ms3d_conv_groups"]):
        m.__dict__.pop("_kernel_eff", None)
        if m.kernel.is_cuda:
            K, cin, cout = m.kernel_volume, m.in_channels, m.out_channels
            if m.kernel.dim() == 2:
                K = 1
            buf = m.__dict__.get("_wf_buf")
            if buf is None or buf.device != m.kernel.device:
                buf = m.__dict__["_wf_buf"] = torch.empty(be.wf_floats(K, cin, cout), dtype=torch.float32, device=m.kernel.device)
            # same orientation rule as the forward() of the module: 3x3x3 maps mirror their offsets in backward-data
            layers.append((m.kernel, buf, K, cin, cout, m.kernel_size == 3 and m.stride == 1, m, grp))
    be.prep_weights_multi([l[:6] for l in layers])
    token = be.weight_token
    defer = torch.is_grad_enabled() and hasattr(be, "wgrad_queue")
    if defer:
        groups = {}
        for w, buf, *_, m, grp in layers:
            if w.requires_grad:
                groups.setdefault(grp, []).append((w, buf, m))
        for g, members in groups.items():
            queue = be.wgrad_queue()
            if queue is None:
                defer = False
                break
            eff = Fn.GroupFlushFn.apply(queue, *[w for w, _, _ in members])
            for e, (w, buf, m) in zip(eff, members):
                e._ms3d_wf = (buf, token)
                e._ms3d_defer = (queue, token, [0])      # [uses of this alias in the forward]: functional._defer_of
                m.__dict__["_kernel_eff"] = (e, token)
    for w, buf, *_ in layers:
        w._ms3d_wf = (buf, token)


def release_conv_weights():
    be = get_backend()
    if hasattr(be, "release_weights"):
        be.release_weights()


class MinkowskiConvolution(_ConvBase):
    """k3 s1 (submanifold, output coords = input coords), k2 s2 (downsample) and k1 s1"""

    def forward(self, x: SparseTensor, residual: SparseTensor = None, skip=None):
        """`residual` (same coordinate map as the output) is added in the kernel epilogue -- used by ResidualBlock
        instead of a separate `+=` pass; `skip` = ("head" | "tail", functional.SkipLink) routes the skip connection's
        gradient through the block's first convolution instead of an elementwise add.  Neither is part of ME's API."""
        cm, ts = x.coordinate_manager, x.tensor_stride
        cin, cout = self.in_channels, self.out_channels
        if self.kernel_size == 3 and self.stride == 1:
            nbr = cm.k3(ts)
            V = cm.size(ts)
            spec = Fn.ConvSpec(nbr, nbr, V, V, 27, cin, cout, True)
            out_ts = ts
        elif self.kernel_size == 2 and self.stride == 2:
            down, up = cm.k2(ts)
            spec = Fn.ConvSpec(down, up, cm.size(ts), cm.size(2 * ts), 8, cin, cout, False)
            out_ts = 2 * ts
        elif self.kernel_size == 1 and self.stride == 1:
            ident = cm.identity(ts)
            V = cm.size(ts)
            spec = Fn.ConvSpec(ident, ident, V, V, 1, cin, cout, False)
            out_ts = ts
        else:
            raise NotImplementedError((self.kernel_size, self.stride))
        feats, kernel, pending = x._F, self._kernel(), x._pending
        if cin % 16 != 0 and cin < 16 and self.kernel_volume > 1 and pending is None and spec.vout >= 30000:
            # the network's input convolution (6 channels) at full resolution: zero-padding rows and weights to one
            # 16-channel chunk lets it take the aligned pair-list kernels (16-byte row gathers) instead of the
            # scalar-load table walk; the padded weight rows see zeros, autograd slices their gradient away
            pad = 16 - cin
            feats = torch.nn.functional.pad(feats, (0, pad))
            kernel = torch.nn.functional.pad(kernel, (0, 0, 0, pad))
            spec = Fn.ConvSpec(spec.nbr_fwd, spec.nbr_bwd, spec.vin, spec.vout, spec.K, 16, cout, spec.mirror)
        if skip is not None and feats is not x._F:
            skip = None            # (padded input rows: not the skipped tensor any more)
        y, stats = Fn.conv(feats, kernel, spec, pending,
                           residual=None if residual is None else residual._raw(), want_stats=self.training, skip=skip)
        return x._like(y, tensor_stride=out_ts, stats=stats)


class MinkowskiConvolutionTranspose(_ConvBase):
    """k2 s2 transposed: output lives on the cached coordinate set of stride ts/2 (the encoder's)"""

    def forward(self, x: SparseTensor):
        cm, ts = x.coordinate_manager, x.tensor_stride
        assert self.kernel_size == 2 and self.stride == 2 and ts % 2 == 0
        fine = ts // 2
        down, up = cm.k2(fine)  # cached by the encoder's strided convolution
        spec = Fn.ConvSpec(up, down, cm.size(ts), cm.size(fine), 8, self.in_channels, self.out_channels, False)
        y, stats = Fn.conv(x._F, self._kernel(), spec, x._pending, want_stats=self.training)
        return x._like(y, tensor_stride=fine, stats=stats)


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over the rows of .F.  Statistics are reduced by a HIP kernel now; normalisation itself is
    deferred to the consumer (see tensor.py)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        # `num_batches_tracked += 1` is one tiny launch per BatchNorm per step (80 per step in the backbone); the
        # count is kept on the host and added to the buffer whenever somebody reads it (state_dict / checkpoint)
        self._pending_batches = 0
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_batches_tracked())
        # a loaded num_batches_tracked replaces the count, it is not added to what this instance had pending
        self.register_load_state_dict_pre_hook(lambda module, *a, **k: setattr(module, "_pending_batches", 0))

    def flush_batches_tracked(self):
        if self._pending_batches and self.bn.num_batches_tracked is not None:
            self.bn.num_batches_tracked += self._pending_batches
        self._pending_batches = 0

    def forward(self, x: SparseTensor):
        bn = self.bn
        feats = x._raw()  # materialise anything pending in front of this BN (engine row order)
        use_batch = self.training or not bn.track_running_stats
        if use_batch:
            with torch.no_grad():
                rm = bn.running_mean if (self.training and bn.track_running_stats) else None
                rv = bn.running_var if rm is not None else None
                if bn.momentum is None:     # cumulative moving average: 1 / (batches seen so far, this one included)
                    seen = self._pending_batches + (int(bn.num_batches_tracked) if rm is not None else 0)
                    mom = 1.0 / (seen + 1)
                else:
                    mom = bn.momentum
                g = bn.weight.detach() if bn.affine else None
                b = bn.bias.detach() if bn.affine else None
                st = x._stats
                if isinstance(st, tuple) and x._pending is None and hasattr(get_backend(), "bn_finalize_parts") \
                        and sum(p.size(2) for p in st) == feats.size(1):
                    # a concatenation of convolution outputs (ME.cat): finalize every part's partials into its slice
                    mean, invstd, scale, shift = get_backend().bn_finalize_parts(st, feats.size(0), bn.eps, mom, g, b,
                                                                                 rm, rv)
                elif torch.is_tensor(st) and st.numel() > 0 and x._pending is None:
                    # the convolution that produced these rows already summed them in its epilogue
                    mean, invstd, scale, shift = get_backend().bn_finalize(st, feats.size(0), bn.eps, mom, g, b,
                                                                           rm, rv)
                else:
                    mean, invstd, scale, shift = get_backend().bn_stats(feats.detach(), bn.eps, mom, g, b, rm, rv)
                if rm is not None:
                    self._pending_batches += 1
        else:
            with torch.no_grad():
                invstd = torch.rsqrt(bn.running_var + bn.eps)
                mean = bn.running_mean
                scale = bn.weight * invstd if bn.affine else invstd
                shift = (bn.bias if bn.affine else 0) - mean * scale
        pending = dict(gamma=bn.weight if bn.affine else torch.ones_like(scale),
                       beta=bn.bias if bn.affine else torch.zeros_like(scale), mean=mean.contiguous(),
                       invstd=invstd.contiguous(), scale=scale.contiguous(), shift=shift.contiguous(), relu=False,
                       training=use_batch)
        return x._like(feats, pending=pending)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x: SparseTensor):
        if x._pending is not None and x._pending.get("gamma") is not None and not x._pending["relu"]:
            p = dict(x._pending)
            p["relu"] = True
            return x._like(x._F, pending=p)
        return x._like(torch.relu(x._raw()))
