"""SparseTensor + coordinate manager.

A SparseTensor is (features [V, C], coordinate manager, tensor stride).  Row order of the input coordinates is
preserved (the reference indexes the U-Net output with the dataset's voxel_point_map, backbone.py:40).

BatchNorm / ReLU are LAZY: `MinkowskiBatchNorm` computes the batch statistics with a HIP reduction and returns
a tensor that only records (scale, shift[, relu]); the following convolution applies them while gathering its
input rows, so `Sequential(BN, ReLU, Conv)` is one gather kernel and never materialises the normalised
activations.  Touching `.features` of such a tensor materialises it (one elementwise kernel).
"""
import torch

from ..backend import get_backend
from . import functional as Fn


class CoordinateManager:
    """coordinates per tensor stride and kernel maps per (stride, kind), built once and shared by every
    convolution of a level and by the backward pass (ME's coordinate manager does the same caching)."""

    def __init__(self, coordinates, spatial_sort=False):
        assert coordinates.dtype == torch.int32 and coordinates.size(1) == 4
        # Engine-internal row order: voxels sorted by (batch, Morton code) so that the rows a convolution gathers
        # for neighbouring outputs are neighbours in HBM / the same XCD's L2.  `perm` maps internal row -> caller
        # row, `inv` the other way; callers never see the internal order (SparseTensor.features un-permutes).
        self.perm = self.inv = None
        if spatial_sort:
            perm = get_backend().spatial_order(coordinates.contiguous())
            if perm is not None:
                self.perm = perm
                self.inv = torch.empty_like(perm)
                self.inv[perm] = torch.arange(perm.numel(), device=perm.device)
                self.coords_external = coordinates
                coordinates = coordinates[perm]
        self.coords = {1: coordinates.contiguous()}
        self._k3 = {}
        self._k2 = {}       # fine stride -> (nbr_down [8,Vc], nbr_up [8,Vf])
        self._ident = {}

    def k3(self, ts):
        if ts not in self._k3:
            self._k3[ts] = get_backend().kmap_k3(self.coords[ts], ts)
        return self._k3[ts]

    def k2(self, ts):
        """stride-2 map from tensor stride ts to 2*ts; creates the coarse coordinate set on first use"""
        if ts not in self._k2:
            be = get_backend()
            oc, parent, koff = be.downsample(self.coords[ts], ts)
            if 2 * ts not in self.coords:
                self.coords[2 * ts] = oc.contiguous()
            self._k2[ts] = be.kmap_k2(parent, koff, oc.size(0))
        return self._k2[ts]

    def prepare(self, n_levels):
        """build the coordinate sets and kernel maps of `n_levels` U-Net levels now.  Each stride-2 map needs the
        coarse voxel count on the host (one sync); done up front the syncs hit an almost empty queue, done lazily
        inside the network each one drains the convolutions queued before it and the GPU then idles while the host
        catches up."""
        ts = 1
        for lvl in range(n_levels):
            self.k3(ts)
            if lvl + 1 < n_levels:
                self.k2(ts)
            ts *= 2

    def identity(self, ts):
        if ts not in self._ident:
            V = self.coords[ts].size(0)
            self._ident[ts] = torch.arange(V, dtype=torch.int32, device=self.coords[ts].device).view(1, V)
        return self._ident[ts]

    def size(self, ts):
        return self.coords[ts].size(0)


class SparseTensor:
    def __init__(self, features, coordinates=None, device=None, coordinate_manager=None, tensor_stride=1,
                 _pending=None, _stats=None):
        if coordinate_manager is None:
            if device is not None:
                features, coordinates = features.to(device), coordinates.to(device)
            coordinate_manager = CoordinateManager(coordinates.to(torch.int32), spatial_sort=True)
            if coordinate_manager.perm is not None:
                features = features[coordinate_manager.perm]
        self._F = features
        self._F_ext = None
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = tensor_stride
        self._pending = _pending  # None or dict(scale, shift, relu, bn ctx) not yet applied to _F
        self._stats = _stats      # per-block (sum, sum^2) partials of _F left by the producing conv's epilogue

    # ---- ME attribute surface
    @property
    def features(self):
        """rows in the CALLER's order (the order of the coordinates the tensor was built from)"""
        self._materialize()
        inv = self.coordinate_manager.inv if self.tensor_stride == 1 else None
        if inv is None:
            return self._F
        if self._F_ext is None:
            self._F_ext = Fn.PermuteRowsFn.apply(self._F, inv, self.coordinate_manager.perm)
        return self._F_ext

    F = features

    def _raw(self):
        """materialised rows in the engine's internal order"""
        self._materialize()
        return self._F

    @property
    def coordinates(self):
        cm = self.coordinate_manager
        if self.tensor_stride == 1 and cm.inv is not None:
            return cm.coords_external
        return cm.coords[self.tensor_stride]

    C = coordinates

    @property
    def device(self):
        return self._F.device

    def _materialize(self):
        if self._pending is not None:
            self._F = Fn.bn_act(self._F, self._pending)
            self._pending = None
            self._stats = None
            self._F_ext = None

    def _like(self, features, pending=None, tensor_stride=None, stats=None):
        return SparseTensor(features, coordinate_manager=self.coordinate_manager,
                            tensor_stride=self.tensor_stride if tensor_stride is None else tensor_stride,
                            _pending=pending, _stats=stats)

    def __iadd__(self, other):      # `x += identity` (common.py:48)
        self._F = self._raw() + other._raw()
        self._stats = None
        self._F_ext = None
        return self

    def __add__(self, other):
        return self._like(self._raw() + other._raw())


def cat(*tensors):
    """ME.cat: channel concat of tensors sharing one coordinate map (common.py:93)"""
    return tensors[0]._like(torch.cat([t._raw() for t in tensors], dim=1))
