"""SparseTensor + coordinate manager.

A SparseTensor is (features [V, C], coordinate manager, tensor stride).  Row order of the input coordinates is
preserved (the reference indexes the U-Net output with the dataset's voxel_point_map, backbone.py:40).

BatchNorm / ReLU are LAZY: `MinkowskiBatchNorm` computes the batch statistics with a HIP reduction and returns
a tensor that only records (scale, shift[, relu]); the following convolution applies them while gathering its
input rows, so `Sequential(BN, ReLU, Conv)` is one gather kernel and never materialises the normalised
activations.  Touching `.features` of such a tensor materialises it (one elementwise kernel).
"""
import os

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
            self.identity(ts)        # table of the 1x1 projections (ResidualBlock.downsample) of this level
            if lvl + 1 < n_levels:
                self.k2(ts)
            ts *= 2

    def identity(self, ts):
        if ts not in self._ident:
            V = self.coords[ts].size(0)
            self._ident[ts] = _iota(V, self.coords[ts].device).view(1, V)
        return self._ident[ts]

    def size(self, ts):
        return self.coords[ts].size(0)


_IOTA = {}


def _iota(n, device):
    """int32 0..n-1 as a view of one long, growing arange per device (the K = 1 tables of every level of every batch: 9
    arange launches per step otherwise)"""
    t = _IOTA.get(device)
    if t is None or t.numel() < n:
        t = _IOTA[device] = torch.arange(max(2 * n, 1 << 20), dtype=torch.int32, device=device)
        if t.is_cuda:
            # used from several streams (main, prefetch) without further ordering: complete before anybody sees it
            # (happens once, and again only when a larger batch makes it grow)
            torch.cuda.current_stream(device).synchronize()
    return t[:n]


_PREFETCHED = {}   # (data_ptr, shape) of a coordinate tensor -> (future of (manager, cuda event), the tensor itself)


def prefetch_coordinates(coordinates, n_levels, wait_current_stream=True, channels=None, point_map=None):
    """Input pipelining (not part of ME's API): build everything that depends on the COORDINATES of a batch the next
    forward will use -- engine row order, the coordinate sets and kernel maps of `n_levels` U-Net levels, their pair
    lists -- on the helper thread and a side stream, e.g. while the current step's backward pass keeps the GPU busy and
    the interpreter idle.  `SparseTensor(features, coordinates)` picks the result up when it is given the same tensor.
    wait_current_stream=False: the coordinates are known to be complete (a resident batch), the side stream need not
    wait for the work queued on the caller's stream.  channels: the channel width of every level (the pair list a
    convolution walks depends on it: 128-row tiles above 32 channels); without it the 16 / 32-channel lists are built and
    a wider level builds its own on first use.  point_map: the batch's voxel_point_map -- its stable sort (the fixed
    summation order of the voxel -> point broadcast's backward, backend.sorted_rows) is built here too."""
    if not coordinates.is_cuda or os.environ.get("MS3D_PREFETCH_COORDS", "1") == "0":
        return
    key = (coordinates.data_ptr(), tuple(coordinates.shape))
    if key in _PREFETCHED:
        return
    from ..backend import prefetch_stream, prefetch_worker as worker
    side = prefetch_stream(coordinates.device)
    if wait_current_stream:
        side.wait_stream(torch.cuda.current_stream())

    def build():
        with torch.cuda.stream(side), torch.no_grad():
            be = get_backend()
            cm = CoordinateManager(coordinates.to(torch.int32), spatial_sort=True)
            cm.prepare(n_levels)
            ts = 1
            for lvl in range(n_levels):          # the lists the convolutions of these levels will ask for
                nbr, v = cm.k3(ts), cm.size(ts)
                c = channels[lvl] if channels is not None else 16
                be.pairlist(nbr, 27, v, c, c)
                be.offsetlist(nbr, 27, v)
                if lvl + 1 < n_levels:
                    down, up = cm.k2(ts)
                    vc = cm.size(2 * ts)
                    c2 = channels[lvl + 1] if channels is not None else 16
                    be.pairlist(down, 8, vc, c, c2); be.pairlist(down, 8, vc, c2, c); be.offsetlist(down, 8, vc)
                    be.pairlist(up, 8, v, c2, c); be.pairlist(up, 8, v, c, c2); be.offsetlist(up, 8, v)
                ts *= 2
            if getattr(be, "kernel_timer", None) is not None:
                # bench.py's roofline wants the valid-pair count of every table (algorithmic bytes): counted here, on the
                # prefetch stream, instead of by two torch launches per table inside a sampled (timed) step
                for t in list(cm._k3.values()) + [x for pair in cm._k2.values() for x in pair]:
                    t._ms3d_pairs_dev = (t >= 0).sum()
            if point_map is not None and point_map.is_cuda and point_map.dtype == torch.int64 and be.deterministic():
                cm._aux_tensors = list(be.sorted_rows(point_map))
            ev = torch.cuda.Event()
            ev.record(side)
            return cm, ev

    _PREFETCHED[key] = (worker().submit(build), coordinates)
    while len(_PREFETCHED) > 2:      # prefetched but never used (end of an epoch, a skipped batch): do not pile up
        fut, coords = _PREFETCHED.pop(next(iter(_PREFETCHED)))
        # its kernels may still be reading `coords` on the side stream when the last reference goes away here
        fut.add_done_callback(lambda f, c=coords, s=side: c.record_stream(s))


def _take_prefetched(coordinates):
    pf = _PREFETCHED.pop((coordinates.data_ptr(), tuple(coordinates.shape)), None) if _PREFETCHED else None
    if pf is None or pf[1] is not coordinates:
        return None
    cm, ev = pf[0].result()
    cur = torch.cuda.current_stream()
    cur.wait_event(ev)
    # everything was allocated under the side stream and is used (and eventually freed) under this one
    held = [cm.perm, cm.inv] + list(cm.coords.values()) + list(cm._k3.values()) + [t for pair in cm._k2.values() for t in pair]
    held.extend(getattr(cm, "_aux_tensors", ()))
    ext = getattr(cm, "coords_external", None)
    if ext is not None and ext is not coordinates:
        held.append(ext)     # an int32 copy made on the side stream (the caller's coordinates had another dtype)
    for t in list(held):
        if t is not None:
            for pl in (getattr(t, "_ms3d_pairlist", None) or {}).values():      # {rows per tile: (tile_start, entries)}
                if isinstance(pl, tuple):                                       # (+ "dense": the table's density verdict)
                    held.extend(x for x in pl if x is not None)
            held.extend(x for x in (getattr(t, "_ms3d_offsetlist", None) or ()) if x is not None)
    for t in held:
        if t is not None and t.is_cuda:
            t.record_stream(cur)
    return cm


_SORT_MIN_ROWS = int(os.environ.get("MS3D_SORT_MIN_ROWS", "100000"))


class SparseTensor:
    def __init__(self, features, coordinates=None, device=None, coordinate_manager=None, tensor_stride=1,
                 _pending=None, _stats=None):
        if coordinate_manager is None:
            if device is not None:
                features, coordinates = features.to(device), coordinates.to(device)
            coordinate_manager = _take_prefetched(coordinates) if coordinates.is_cuda else None
            if coordinate_manager is None:
                # small tensors (the proposal grids of the score / refinement nets: tens of thousands of rows, a few MB
                # of features that live in L2 anyway) keep the caller's row order: the Morton sort, the permutation
                # and its inverse cost more than the locality buys
                coordinate_manager = CoordinateManager(coordinates.to(torch.int32),
                                                       spatial_sort=coordinates.size(0) >= _SORT_MIN_ROWS)
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
    parts = [t._stats if (t._pending is None and torch.is_tensor(t._stats) and t._stats.numel() > 0) else None
             for t in tensors]
    # every part still carries the (sum, sum of squares) partials its convolution left behind: the statistics of the
    # concatenation are the parts' statistics side by side -- the BatchNorm that follows (ResidualBlock after the
    # U-Net's skip concat) finalizes them per part instead of running a statistics pass over the 2c-wide rows
    stats = tuple(parts) if all(p is not None for p in parts) else None
    return tensors[0]._like(torch.cat([t._raw() for t in tensors], dim=1), stats=stats)
