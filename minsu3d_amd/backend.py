"""Backend selection for the hot-path operators.

The product backend is `HipBackend`: torch device tensors in, calls into libminsu3d_hip.so through
the C ABI (include/minsu3d_hip.h), torch device tensors out.  It raises if the library is missing --
there is no CPU fallback.  Tests and bench.py's cpu_baseline leg may install another object with
the same method surface via `set_backend` (e.g. the oracle-backed one in oracle/oracle_backend.py);
nothing in this package imports the oracle.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

_BACKEND = None


def get_backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(b):
    global _BACKEND
    prev = _BACKEND
    _BACKEND = b
    return prev


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


_HINT = object()   # placeholder in a C argument list: replaced by the capped/symmetric hint of the graph


class _HandleTable:
    """What the library knows about a tensor it produced (or was shown) that the reference's tensor-only operator
    signatures (bfs_cluster.h:15-19) have no argument for: an entry per tensor OBJECT -- keyed by id() and guarded by a weak
    reference (an address or an id may be recycled, a live object is not) and by the tensor's version counter (an in-place
    write voids it).  Round 6: replaces the `_ms3d_capped` / `_ms3d_max_scene` attributes that rode on the
    tensors themselves; a copy / slice / host round trip of the tensor is simply not in the table and takes the general
    route (BFS: the validated one; ball query: one device->host read)."""

    def __init__(self):
        self._entries = {}

    def put(self, tensor, value):
        import weakref
        key = id(tensor)
        entries = self._entries

        def _gone(_ref, key=key, entries=entries):
            e = entries.get(key)
            if e is not None and e[0] is _ref:
                del entries[key]
        self._entries[key] = (weakref.ref(tensor, _gone), tensor._version, value)

    def get(self, tensor, default=None):
        e = self._entries.get(id(tensor))
        if e is None or e[0]() is not tensor or e[1] != tensor._version:
            return default
        return e[2]


GRAPHS = _HandleTable()        # start_len tensor of a ball query -> its `capped` flag (0 / 1)
SCENE_BOUNDS = _HandleTable()  # batch_offsets tensor -> points of its largest scene


class _Workspace:
    """grow-only device scratch, one per purpose (avoids hipMalloc on the hot path)"""

    def __init__(self):
        self.buf = {}

    def get(self, key, nbytes, device):
        # one buffer per purpose AND stream: operators may run concurrently on different streams (PointGroup's two
        # independent groupings do), and a kernel's scratch must not be shared between them
        k = (key, device, _lib.stream_handle().value)
        b = self.buf.get(k)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self.buf[k] = b
        return b


_POOL = None
_SIDE = {}
_WGRAD = {}


def worker():
    """the one helper thread that runs host-blocking pieces of a step (operators that size their outputs on the host)
    beside the main thread; the library calls release the GIL"""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ms3d-side")
    return _POOL


_PF_POOL = None
_PF = {}


def prefetch_worker():
    """the helper thread of the input pipeline (MinkowskiEngine.prefetch_coordinates): its own thread and stream, because
    PointGroup's second grouping occupies `worker()` / `side_stream()` exactly when the next batch's coordinate work is
    meant to run -- inside the grouping window, where the chip is mostly idle"""
    global _PF_POOL
    if _PF_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _PF_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ms3d-prefetch")
    return _PF_POOL


def prefetch_stream(device):
    # under data parallelism the prefetch shares the second grouping stream (parallel.stream_plan: at most three streams
    # of this process beside the collective's -- inside the default four hardware queues)
    from .parallel import stream_plan
    if stream_plan()["prefetch_stream"] == "side":
        return side_stream(device)
    s = _PF.get(device)
    if s is None:
        s = _PF[device] = torch.cuda.Stream(device=device)
    return s


def wgrad_stream(device):
    """the stream the backward-weight kernels run on beside the backward-data chain (conv_layer_backward)"""
    s = _WGRAD.get(device)
    if s is None:
        s = _WGRAD[device] = torch.cuda.Stream(device=device)
    return s


_HEADS = {}


def heads_stream(device):
    """the stream the per-point heads (forward, losses, early backward) run on beside the grouping window
    (GeneralModel, MS3D_EARLY_HEADS=2)"""
    s = _HEADS.get(device)
    if s is None:
        s = _HEADS[device] = torch.cuda.Stream(device=device)
    return s


def side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


def _load_host_ext():
    """minsu3d_amd/lib/_ms3d_host.so (csrc_host/ms3d_host.cpp, built by build.build_host): the per-layer calls as one
    pybind call each instead of ctypes marshalling.  None when it is not built or MS3D_HOST_EXT=0 -- ctypes then serves
    every call (same library, same kernels)."""
    if os.environ.get("MS3D_HOST_EXT", "1") == "0":
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_ms3d_host.so")
    if not os.path.exists(path) or (os.environ.get("MS3D_LIB") and os.environ["MS3D_LIB"] != _lib.LIB_PATH):
        return None
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("_ms3d_host", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except (ImportError, OSError) as e:      # e.g. built against another torch: fall back, but say so once
        import warnings
        warnings.warn(f"minsu3d_amd: host extension {path} did not load ({e}); using the ctypes path")
        return None


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.lib()  # raises HipLibraryError when the .so is absent
        self.ext = _load_host_ext() if not os.environ.get("MS3D_LIB") else None
        self.ws = _Workspace()
        self.kernel_timer = None  # bench.py installs a KernelTimer to bracket chosen launches with HIP events
        self.weight_token = 0     # see prep_weights_multi
        self._prep_table = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _dev(t):
        """contiguous device tensor.  The reference's callers hand CPU tensors to the clustering operators
        (model/pointgroup.py:49-52, hais.py:52-56, softgroup.py:60-63): those are copied to the current GPU -- the
        computation itself always runs in the HIP library, there is no CPU fallback."""
        if not t.is_cuda:
            if not torch.cuda.is_available():
                raise _lib.HipLibraryError("HipBackend needs a GPU (got a CPU tensor and no device is present); "
                                           "the product path has no CPU fallback")
            t = t.cuda(non_blocking=True)
        return t.contiguous()

    # ------------------------------------------------------------------ ball query
    def ballquery_batch_p(self, coords, batch_idxs, batch_offsets, radius, meanActive, max_scene_points=0):
        """reference wrapper: functions/common_ops.py:11-47 (retry loop kept)"""
        coords = self._dev(coords); batch_idxs = self._dev(batch_idxs); batch_offsets = self._dev(batch_offsets)
        n = coords.size(0)
        dev = coords.device
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        if n == 0:
            return torch.zeros(0, dtype=torch.int32, device=dev), start_len
        if max_scene_points <= 0:
            # tight per-scene bound -> smaller LDS bitmap, more waves per CU; the models query the same scene layout
            # twice (shifted / original coordinates), so the value is remembered per offsets tensor (SCENE_BOUNDS)
            max_scene_points = SCENE_BOUNDS.get(batch_offsets)
            if max_scene_points is None:
                max_scene_points = int((batch_offsets[1:] - batch_offsets[:-1]).max().item())
                SCENE_BOUNDS.put(batch_offsets, max_scene_points)
        ws_bytes = self.lib.ms3d_ballquery_workspace_bytes(n)
        ws = self.ws.get("bq", ws_bytes, dev)
        n_active, capped = C.c_int(0), C.c_int(0)
        timer = self.kernel_timer
        ev0 = timer.op_begin() if timer is not None else None
        # The reference's retry loop (common_ops.py:31-38) starts from the configured meanActive on every call; a call
        # site whose lists outgrow it (SoftGroup's batched per-class query) would run the whole query twice per step.
        # The first guess therefore remembers what the same call site (radius, configured value) needed last time.
        hint_key = (round(float(radius), 6), int(meanActive))
        hints = self.__dict__.setdefault("_bq_mean_active", {})
        meanActive = max(int(meanActive), hints.get(hint_key, 0))
        while True:
            idx = torch.empty(n * meanActive, dtype=torch.int32, device=dev)
            rc = self.lib.ms3d_ballquery_batch_p(
                n, int(meanActive), C.c_float(radius), _lib.ptr(coords), _lib.ptr(batch_idxs),
                _lib.ptr(batch_offsets), int(batch_offsets.numel() - 1), int(max_scene_points), _lib.ptr(idx),
                _lib.ptr(start_len), C.byref(n_active), C.byref(capped), _lib.ptr(ws), C.c_size_t(ws.numel()),
                _lib.stream_handle())
            _lib.check(rc, "ms3d_ballquery_batch_p")
            if n_active.value <= n * meanActive:
                break
            meanActive = int(n_active.value // n + 1)
            hints[hint_key] = meanActive + meanActive // 8 + 1     # 12 % head room for the next batches
        if ev0 is not None:
            # n*12 + n*27*c*12 + nActive*4 + n*8; 27*c = candidates per query = hits * 27*1.01^3 / (4/3 pi)
            timer.op_end("ballquery_batch_p", ev0, n * 20 + n_active.value * (6.64 * 12 + 4))
        # remembered for the clustering call that consumes this graph: "no list reached the 1000 cap" means the graph
        # is symmetric, which lets the BFS skip a device->host check
        GRAPHS.put(start_len, int(capped.value))
        return idx[:n_active.value], start_len

    # ------------------------------------------------------------------ BFS
    def _bfs(self, fn_name, args_head, ball_idx, start_len, args_tail):
        hint = int(GRAPHS.get(start_len, -1))   # noted by ballquery_batch_p for the tensor it returned
        ball_idx = self._dev(ball_idx); start_len = self._dev(start_len)
        N = start_len.size(0)
        dev = start_len.device
        cluster_idxs = torch.empty((max(N, 1), 2), dtype=torch.int32, device=dev)
        cluster_offsets = torch.empty(N + 1, dtype=torch.int32, device=dev)
        ws_bytes = self.lib.ms3d_bfs_workspace_bytes(N)
        ws = self.ws.get("bfs", ws_bytes, dev)
        counts = (C.c_int * 2)(0, 0)
        args_tail = tuple(hint if a is _HINT else a for a in args_tail)
        timer = self.kernel_timer
        ev0 = timer.op_begin() if timer is not None else None
        rc = getattr(self.lib, fn_name)(*args_head, _lib.ptr(ball_idx), C.c_long(ball_idx.numel()), _lib.ptr(start_len), N,
                                        *args_tail,
                                        _lib.ptr(cluster_idxs), _lib.ptr(cluster_offsets), counts, _lib.ptr(ws),
                                        C.c_size_t(ws.numel()), _lib.stream_handle())
        _lib.check(rc, fn_name)
        if ev0 is not None:   # nActive*4 + n*(2+8+4) + S*8
            timer.op_end(fn_name[5:], ev0, ball_idx.numel() * 4 + N * 14 + counts[1] * 8)
        return cluster_idxs[:counts[1]], cluster_offsets[:counts[0] + 1]

    def pg_bfs_cluster(self, semantic_label, ball_query_idxs, start_len, threshold):
        sem = self._dev(semantic_label)
        assert sem.dtype == torch.int16
        return self._bfs("ms3d_pg_bfs_cluster", (_lib.ptr(sem),), ball_query_idxs, start_len, (int(threshold), _HINT))

    def sg_bfs_cluster(self, class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
        mean = (C.c_float * len(class_numpoint_mean))(*[float(x) for x in class_numpoint_mean])
        return self._bfs("ms3d_sg_bfs_cluster", (mean,), ball_query_idxs, start_len,
                         (C.c_float(threshold), _HINT, int(class_id)))

    def sg_bfs_cluster_batched(self, group_of_point, thr_per_group, ball_query_idxs, start_len):
        g = self._dev(group_of_point); t = self._dev(thr_per_group)
        assert g.dtype == torch.uint8 and t.dtype == torch.float32
        return self._bfs("ms3d_sg_bfs_cluster_batched", (_lib.ptr(g), _lib.ptr(t)), ball_query_idxs, start_len, (_HINT,))

    def hierarchical_aggregation(self, sem, coord_shift, ball_idx, start_len, batch_idxs, using_set_aggr,
                                 point_num_avg, radius_avg, ignored_label=-1):
        hint = int(GRAPHS.get(start_len, -1))   # noted by ballquery_batch_p for the tensor it returned
        sem = self._dev(sem); cs = self._dev(coord_shift); ball_idx = self._dev(ball_idx)
        start_len = self._dev(start_len); batch_idxs = self._dev(batch_idxs)
        assert sem.dtype == torch.int16 and batch_idxs.dtype == torch.uint8
        N, dev, ncls = start_len.size(0), start_len.device, len(point_num_avg)
        pna = (C.c_float * ncls)(*[float(x) for x in point_num_avg])
        ra = (C.c_float * ncls)(*[float(x) for x in radius_avg])
        out_idx = torch.empty((max(2 * N, 1), 2), dtype=torch.int32, device=dev)
        out_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.lib.ms3d_hais_workspace_bytes.restype = C.c_size_t
        ws = self.ws.get("hais", self.lib.ms3d_hais_workspace_bytes(N, ncls), dev)
        counts = (C.c_int * 2)(0, 0)
        timer = self.kernel_timer
        ev0 = timer.op_begin() if timer is not None else None
        _lib.check(self.lib.ms3d_hierarchical_aggregation(
            _lib.ptr(sem), _lib.ptr(cs), _lib.ptr(batch_idxs), _lib.ptr(ball_idx), C.c_long(ball_idx.numel()),
            _lib.ptr(start_len), N, hint, int(bool(using_set_aggr)), pna, ra, ncls, _lib.ptr(out_idx), _lib.ptr(out_off),
            counts, _lib.ptr(ws), C.c_size_t(ws.numel()), _lib.stream_handle()), "ms3d_hierarchical_aggregation")
        if ev0 is not None:   # the BFS byte model + the centre sums: 12 B per member
            timer.op_end("hierarchical_aggregation", ev0, ball_idx.numel() * 4 + N * 14 + counts[1] * 8 + N * 12)
        return out_idx[:counts[1]], out_off[:counts[0] + 1]

    def hierarchical_aggregation_parts(self, sem, coord_shift, ball_idx, start_len, batch_idxs, using_set_aggr,
                                       point_num_avg, radius_avg):
        """the reference's own output contract (hierarchical_aggregation.cpp:105-184) -> dict of device tensors:
        kept / primary / fragment = (idxs [rows,2], offsets [n+1], centers [n,5]), post = (idxs [F+P rows,2] with a
        zero tail, offsets [n_primary+1]); fragment and post are None without set aggregation"""
        hint = int(GRAPHS.get(start_len, -1))
        sem = self._dev(sem); cs = self._dev(coord_shift); ball_idx = self._dev(ball_idx)
        start_len = self._dev(start_len); batch_idxs = self._dev(batch_idxs)
        assert sem.dtype == torch.int16 and batch_idxs.dtype == torch.uint8
        N, dev, ncls = start_len.size(0), start_len.device, len(point_num_avg)
        pna = (C.c_float * ncls)(*[float(x) for x in point_num_avg])
        ra = (C.c_float * ncls)(*[float(x) for x in radius_avg])
        M = max(N, 1)
        ib = torch.empty((4, M, 2), dtype=torch.int32, device=dev)
        ob = torch.empty((4, M + 1), dtype=torch.int32, device=dev)
        cb = torch.empty((3, M, 5), dtype=torch.float32, device=dev)
        self.lib.ms3d_hais_workspace_bytes.restype = C.c_size_t
        ws = self.ws.get("hais", self.lib.ms3d_hais_workspace_bytes(N, ncls), dev)
        counts = (C.c_int * 8)()
        _lib.check(self.lib.ms3d_hierarchical_aggregation_parts(
            _lib.ptr(sem), _lib.ptr(cs), _lib.ptr(batch_idxs), _lib.ptr(ball_idx), C.c_long(ball_idx.numel()),
            _lib.ptr(start_len), N, hint, int(bool(using_set_aggr)), pna, ra, ncls,
            _lib.ptr(ib[0]), _lib.ptr(ob[0]), _lib.ptr(cb[0]), _lib.ptr(ib[1]), _lib.ptr(ob[1]), _lib.ptr(cb[1]),
            _lib.ptr(ib[2]), _lib.ptr(ob[2]), _lib.ptr(cb[2]), _lib.ptr(ib[3]), _lib.ptr(ob[3]), counts,
            _lib.ptr(ws), C.c_size_t(ws.numel()), _lib.stream_handle()), "ms3d_hierarchical_aggregation_parts")
        nk, rk, npr, _rpost, nf, rf, rp = [counts[i] for i in range(7)]
        out = {"kept": (ib[0, :rk], ob[0, :nk + 1], cb[0, :nk]), "primary": (ib[1, :rp], ob[1, :npr + 1], cb[1, :npr]),
               "fragment": None, "post": None}
        if using_set_aggr:
            out["fragment"] = (ib[2, :rf], ob[2, :nf + 1], cb[2, :nf])
            out["post"] = (ib[3, :rf + rp], ob[3, :npr + 1])
        return out

    # ------------------------------------------------------------------ segment ops / pools
    def _seg(self, fn_name, inp, offsets):
        inp = self._dev(inp); offsets = self._dev(offsets)
        P, Cc = offsets.numel() - 1, inp.size(1)
        out = torch.zeros((P, Cc), dtype=torch.float32, device=inp.device)
        rc = getattr(self.lib, fn_name)(P, Cc, _lib.ptr(inp), _lib.ptr(offsets), _lib.ptr(out), _lib.stream_handle())
        _lib.check(rc, fn_name)
        return out

    def sec_mean(self, inp, offsets): return self._seg("ms3d_sec_mean", inp, offsets)
    def sec_min(self, inp, offsets): return self._seg("ms3d_sec_min", inp, offsets)
    def sec_max(self, inp, offsets): return self._seg("ms3d_sec_max", inp, offsets)
    def global_avg_pool_fp(self, feats, offsets): return self._seg("ms3d_global_avg_pool_fp", feats, offsets)

    def proposal_voxel_coords(self, clusters_idx, clusters_offset, coords, scale, spatial_shape, rand6):
        """(proposal, x, y, z) i32 [S, 4]: member coordinates centred on the proposal mean, scaled into the
        `spatial_shape` cube and randomly placed in it (the arithmetic of the reference's clusters_voxelization,
        general_model.py:152-193, in its float operation order); rand6 = the two U(0,1)^3 draws (u1, u2) on the device"""
        ci = self._dev(clusters_idx); off = self._dev(clusters_offset); xyz = self._dev(coords); r6 = self._dev(rand6)
        assert ci.dtype == torch.int64 and ci.is_contiguous() and off.dtype == torch.int32
        assert xyz.dtype == torch.float32 and xyz.is_contiguous() and r6.dtype == torch.float32 and r6.numel() == 6
        S, P, dev = ci.size(0), off.numel() - 1, xyz.device
        out = torch.empty((S, 4), dtype=torch.int32, device=dev)
        ws = torch.empty(3 * S + 7 * max(P, 1), dtype=torch.float32, device=dev)
        base = ws.data_ptr()
        _lib.check(self.lib.ms3d_proposal_voxel_coords(
            _lib.ptr(ci), S, _lib.ptr(off), P, _lib.ptr(xyz), C.c_float(float(scale)), int(spatial_shape), _lib.ptr(r6),
            C.c_void_p(base), C.c_void_p(base + 12 * S), C.c_void_p(base + 12 * S + 12 * max(P, 1)), _lib.ptr(out),
            _lib.stream_handle()), "ms3d_proposal_voxel_coords")
        return out

    def roipool_fp(self, feats, offsets):
        feats = self._dev(feats); offsets = self._dev(offsets)
        P, Cc = offsets.numel() - 1, feats.size(1)
        out = torch.zeros((P, Cc), dtype=torch.float32, device=feats.device)
        maxidx = torch.zeros((P, Cc), dtype=torch.int32, device=feats.device)
        rc = self.lib.ms3d_roipool_fp(P, Cc, _lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(out), _lib.ptr(maxidx),
                                      _lib.stream_handle())
        _lib.check(rc, "ms3d_roipool_fp")
        return out, maxidx

    def roipool_bp(self, d_out, offsets, maxidx, sum_npoint):
        d_out = self._dev(d_out)
        P, Cc = d_out.shape
        d_feats = torch.zeros((sum_npoint, Cc), dtype=torch.float32, device=d_out.device)
        rc = self.lib.ms3d_roipool_bp(P, Cc, _lib.ptr(d_feats), _lib.ptr(offsets), _lib.ptr(maxidx), _lib.ptr(d_out),
                                      _lib.stream_handle())
        _lib.check(rc, "ms3d_roipool_bp")
        return d_feats

    def global_avg_pool_bp(self, d_out, offsets, sum_npoint):
        d_out = self._dev(d_out)
        P, Cc = d_out.shape
        d_feats = torch.zeros((sum_npoint, Cc), dtype=torch.float32, device=d_out.device)
        rc = self.lib.ms3d_global_avg_pool_bp_rows(P, Cc, C.c_long(int(sum_npoint)), _lib.ptr(d_feats),
                                                   _lib.ptr(self._dev(offsets)), _lib.ptr(d_out), _lib.stream_handle())
        _lib.check(rc, "ms3d_global_avg_pool_bp_rows")
        return d_feats

    def gather_rows(self, x, idx):
        """x[idx] for f32 [V, C] rows and an int64 index, at copy speed"""
        x = self._dev(x); idx = self._dev(idx)
        if self.ext is not None:
            return self.ext.gather_rows(x, idx)
        assert idx.dtype == torch.int64 and x.dtype == torch.float32 and x.dim() == 2
        x = x.contiguous(); idx = idx.contiguous()
        out = torch.empty((idx.numel(), x.size(1)), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.ms3d_gather_rows(_lib.ptr(x), _lib.ptr(idx), C.c_long(idx.numel()), int(x.size(1)), _lib.ptr(out),
                                             _lib.stream_handle()), "ms3d_gather_rows")
        return out

    def sorted_rows(self, idx):
        """(keys, order) of a STABLE ascending sort of an int64 row index, cached on the tensor: what the fixed-order
        scatter-add needs.  Callers that know the index ahead of its backward (the loader's voxel_point_map) call this
        early, off the critical path; torch's radix sort (rocPRIM) is plumbing, not a hot kernel."""
        return self._use_sorted(self._sorted_cache(idx))

    def _sorted_cache(self, idx):
        """the cache entry (keys, order, version of idx, event recorded behind the sort, the stream it ran on): the sort is
        often queued on the prefetch stream, so the entry carries its own ordering instead of relying on an unrelated
        wait of the consumer"""
        cached = getattr(idx, "_ms3d_sorted", None)
        if cached is None or cached[2] != idx._version:
            keys, order = torch.sort(idx, stable=True)
            ev, st = None, None
            if idx.is_cuda:
                st = torch.cuda.current_stream(idx.device)
                ev = torch.cuda.Event()
                ev.record(st)
            cached = idx._ms3d_sorted = (keys, order, idx._version, ev, st)
        return cached

    @staticmethod
    def _use_sorted(cached):
        """(keys, order) of a cache entry, ordered behind its sort on the CURRENT stream"""
        keys, order = cached[0], cached[1]
        if len(cached) >= 5 and cached[3] is not None:
            cur = torch.cuda.current_stream(keys.device)
            if cur != cached[4]:
                cur.wait_event(cached[3])
                keys.record_stream(cur); order.record_stream(cur)
        return keys, order

    def presort_rows(self, idx):
        """Scheduling only: the stable sort of `idx` queued on the helper thread and the side stream NOW (the forward of
        a row gather: ~200 us of radix sort for 800k rows that the backward would otherwise wait for on the critical
        path) -> a handle for scatter_add_rows(sorted_=...)"""
        main = torch.cuda.current_stream(idx.device)
        ready = torch.cuda.Event()
        ready.record(main)
        side = side_stream(idx.device)

        def job():
            with torch.cuda.stream(side), torch.no_grad():
                side.wait_event(ready)
                idx.record_stream(side)      # the sort may still be reading it when the last reference on the caller's side dies
                keys, order = torch.sort(idx, stable=True)
                done = torch.cuda.Event()
                done.record(side)
                return keys, order, done
        return (worker().submit(job), side)

    def scatter_add_rows(self, src, idx, n_rows, max_dup=None, sorted_=None):
        """dst[idx[i]] += src[i] -> dst [n_rows, C], bit-reproducible: rows that may collect more than two sources are
        summed in ascending source order over a stable sort of the index (ms3d_scatter_add_rows_sorted); max_dup <= 2
        (the caller knows no row has more than two sources: a + b = b + a exactly) keeps the one-launch float atomics,
        as does MS3D_DETERMINISTIC=0.  sorted_: (keys, order) of that sort, or a presort_rows handle."""
        src = self._dev(src); idx = self._dev(idx)
        assert idx.dtype == torch.int64
        dst = torch.zeros((n_rows, src.size(1)), dtype=torch.float32, device=src.device)
        if (max_dup is not None and max_dup <= 2) or not self.deterministic():
            _lib.check(self.lib.ms3d_scatter_add_rows(_lib.ptr(src), _lib.ptr(idx), C.c_long(src.size(0)), int(src.size(1)),
                                                      _lib.ptr(dst), _lib.stream_handle()), "ms3d_scatter_add_rows")
            return dst
        if sorted_ is not None and not torch.is_tensor(sorted_[0]):
            keys, order, done = sorted_[0].result()
            cur = torch.cuda.current_stream(src.device)
            cur.wait_event(done)
            keys.record_stream(cur); order.record_stream(cur); idx.record_stream(sorted_[1])
        elif sorted_ is not None:
            keys, order = self._use_sorted(sorted_)
        else:
            keys, order = self.sorted_rows(idx)
        _lib.check(self.lib.ms3d_scatter_add_rows_sorted(_lib.ptr(src), _lib.ptr(keys), _lib.ptr(order),
                                                         C.c_long(src.size(0)), int(src.size(1)), _lib.ptr(dst),
                                                         _lib.stream_handle()), "ms3d_scatter_add_rows_sorted")
        return dst

    def deterministic(self):
        """MS3D_DETERMINISTIC (default 1): every float sum of a training step in a fixed order"""
        on = self.__dict__.get("_deterministic")
        if on is None:
            on = self._deterministic = os.environ.get("MS3D_DETERMINISTIC", "1") != "0"
        return on

    # ------------------------------------------------------------------ per-point losses
    def point_losses_forward(self, scores, labels, pred_offsets, centre, xyz, instance_ids):
        """-> (out5 device f32 [5] = three losses + the two 1/count factors, d_scores [N,C], d_norm [N,3], d_dir [N,3]
        unnormalised gradients); see include/minsu3d_hip.h"""
        N, Cc = scores.shape
        dev = scores.device
        assert labels.dtype == torch.int16 and instance_ids.dtype == torch.int16 and scores.dtype == torch.float32
        out5 = torch.empty(5, dtype=torch.float32, device=dev)
        d_scores = torch.empty_like(scores)
        d_off = torch.empty((2, N, 3), dtype=torch.float32, device=dev)
        ws = self.ws.get("ploss", 40 * self.lib.ms3d_point_losses_blocks(C.c_long(N)), dev)
        _lib.check(self.lib.ms3d_point_losses_forward(
            _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(pred_offsets), _lib.ptr(centre), _lib.ptr(xyz),
            _lib.ptr(instance_ids), C.c_long(N), int(Cc), _lib.ptr(d_scores), _lib.ptr(d_off[0]), _lib.ptr(d_off[1]),
            _lib.ptr(ws), _lib.ptr(out5), _lib.stream_handle()), "ms3d_point_losses_forward")
        return out5, d_scores, d_off

    def point_losses_scale_grads(self, d_scores, d_off, out5, g_sem, g_norm, g_dir):
        _lib.check(self.lib.ms3d_point_losses_scale_grads(
            _lib.ptr(d_scores), C.c_long(d_scores.numel()), _lib.ptr(d_off[0]), _lib.ptr(d_off[1]),
            C.c_long(d_off[0].numel()), _lib.ptr(out5), _lib.ptr(g_sem), _lib.ptr(g_norm), _lib.ptr(g_dir),
            _lib.stream_handle()), "ms3d_point_losses_scale_grads")

    # ------------------------------------------------------------------ IoU family
    def _iou(self, fn_name, prop_idx, prop_off, inst_labels, inst_pointnum, sigmoid=None):
        prop_idx = self._dev(prop_idx); prop_off = self._dev(prop_off)
        inst_labels = self._dev(inst_labels); inst_pointnum = self._dev(inst_pointnum)
        assert prop_idx.dtype == torch.int32 and inst_labels.dtype == torch.int16
        I, P = inst_pointnum.numel(), prop_off.numel() - 1
        iou = torch.zeros((P, I), dtype=torch.float32, device=prop_idx.device)
        args = [I, P, _lib.ptr(prop_idx), _lib.ptr(prop_off), _lib.ptr(inst_labels), _lib.ptr(inst_pointnum),
                _lib.ptr(iou)]
        if sigmoid is not None:
            args.append(_lib.ptr(self._dev(sigmoid)))
        rc = getattr(self.lib, fn_name)(*args, _lib.stream_handle())
        _lib.check(rc, fn_name)
        return iou

    def get_iou(self, pi, po, il, pn): return self._iou("ms3d_get_iou", pi, po, il, pn)
    def get_mask_iou_on_cluster(self, pi, po, il, pn): return self._iou("ms3d_get_mask_iou_on_cluster", pi, po, il, pn)
    def get_mask_iou_on_pred(self, pi, po, il, pn, sg): return self._iou("ms3d_get_mask_iou_on_pred", pi, po, il, pn, sg)

    def get_mask_label(self, prop_idx, prop_off, inst_labels, inst_cls, iou, ignored_label, iou_thr):
        prop_idx = self._dev(prop_idx); prop_off = self._dev(prop_off); iou = self._dev(iou)
        P, I = iou.shape
        ml = torch.zeros(prop_idx.shape, dtype=torch.bool, device=prop_idx.device)
        mlm = torch.zeros(prop_idx.shape, dtype=torch.bool, device=prop_idx.device)
        rc = self.lib.ms3d_get_mask_label(I, P, int(ignored_label), C.c_float(iou_thr), _lib.ptr(prop_idx),
                                          _lib.ptr(prop_off), _lib.ptr(self._dev(inst_labels)),
                                          _lib.ptr(self._dev(inst_cls)), _lib.ptr(iou), _lib.ptr(ml), _lib.ptr(mlm),
                                          _lib.stream_handle())
        _lib.check(rc, "ms3d_get_mask_label")
        return ml, mlm


# ---------------------------------------------------------------------------------------------
# Sparse-voxel engine methods (coordinate maps, fused BN/ReLU/conv kernels).  Kept in a mixin-style
# block so the oracle-backed test double (oracle/oracle_backend.py) can mirror the same surface.
# ---------------------------------------------------------------------------------------------
def _f32(t):
    return None if t is None else t.contiguous()


def _p(t):
    """raw pointer (int) of a contiguous tensor or None -- for entry points whose argtypes are declared (no ctypes
    object per argument: the per-layer calls are made ~250 times per step)"""
    return None if t is None else t.data_ptr()


_VP, _I = C.c_void_p, C.c_int
_FAST_ARGTYPES = {
    "ms3d_spconv_layer_forward": [_VP] * 3 + [_I] * 5 + [_VP] * 2 + [_I] + [_VP] * 5 + [_VP] * 2 + [_VP] * 2 + [_VP],
    "ms3d_spconv_layer_backward": [_VP] * 5 + [_I] * 5 + [_VP] * 4 + [_I] * 3 + [_VP] * 5 + [_VP] * 4 + [_VP] * 4 +
                                  [_VP, _VP, _I, _VP, _VP, _VP, _VP],
    "ms3d_bn_finalize": [_VP, _I, C.c_long, _I, C.c_float, C.c_float] + [_VP] * 4 + [_VP] * 4 + [_VP],
}


class KernelTimer:
    """Live kernel timing for bench.py's roofline: HIP events recorded on the stream a kernel is launched on --
    INSIDE the library call, immediately before and after the kernel, for the sparse convolutions (forward,
    backward-data, backward-weight); around the library call for the grouping operators (they end with the
    host read of their output size, so the bracket holds exactly their kernels).  Every record carries the
    ALGORITHMIC byte / flop count of the launch (SURVEY 8d formulas with the launch's real pair / edge counts).
    Recording is switched on for chosen steps only (`sampling`): an event pair costs a few microseconds of host time."""

    def __init__(self, lib):
        self.lib = lib
        lib.ms3d_event_create.restype = C.c_void_p
        lib.ms3d_event_elapsed_ms.restype = C.c_float
        self.records = []               # (group key, start_event, stop_event, bytes or (table, per-pair bytes, const bytes), flops per pair)
        self.sampling = False
        self.steps_sampled = 0
        self._pool = []                 # events are created up front: hipEventCreate inside a timed step is host time

    def reserve(self, n_events):
        while len(self._pool) < n_events:
            self._pool.append(C.c_void_p(self.lib.ms3d_event_create()))

    def _event(self):
        return self._pool.pop() if self._pool else C.c_void_p(self.lib.ms3d_event_create())

    # ---- convolutions: events are handed to the library, which records them around the kernel
    def conv(self, kind, K, cin, cout, nbr, rows):
        """kind: 'fwd' (forward / backward-data, the same kernels) or 'wgrad' -> (ev_start, ev_stop) or None"""
        if not self.sampling:
            return None
        pairs = getattr(nbr, "_ms3d_pairs_dev", None)   # valid (in, out) pairs of the table: counted on the device, read
        if pairs is None:                                # back only in summary() (no host sync inside a step)
            pairs = (nbr >= 0).sum()
            nbr._ms3d_pairs_dev = pairs
        ev = (self._event(), self._event())
        self.records.append((("spconv_" + kind, K, cin, cout, rows), ev[0], ev[1],
                             (pairs, (cin + cout) * 4 + 8, K * cin * cout * 4), 2 * cin * cout))
        return ev

    def conv_group(self, kind, layers):
        """one event pair for a BATCHED launch over several layers; layers = [(K, cin, cout, rows, table)]: the record
        carries the layers' summed algorithmic bytes / flops -> (ev_start, ev_stop), recorded by the caller"""
        if not self.sampling or not layers:
            return None
        ev = (self._event(), self._event())
        parts = []
        for K, cin, cout, rows, nbr in layers:
            pairs = getattr(nbr, "_ms3d_pairs_dev", None)
            if pairs is None:
                pairs = (nbr >= 0).sum()
                nbr._ms3d_pairs_dev = pairs
            parts.append((pairs, (cin + cout) * 4 + 8, K * cin * cout * 4, 2 * cin * cout))
        K, cin, cout, rows, _ = layers[0]
        self.records.append((("spconv_" + kind, K, cin, cout, rows, len(layers)), ev[0], ev[1], parts, None))
        return ev

    # ---- grouping operators: bracket a library call on the current stream
    def op_begin(self):
        if not self.sampling:
            return None
        ev = self._event()
        self.lib.ms3d_event_record(ev, _lib.stream_handle())
        return ev

    def op_end(self, name, ev0, nbytes):
        ev1 = self._event()
        self.lib.ms3d_event_record(ev1, _lib.stream_handle())
        self.records.append(((name,), ev0, ev1, float(nbytes), 0))

    def summary(self):
        """-> {group key: dict(launches, ms, bytes, flops)} summed over the sampled steps"""
        out = {}
        for key, a, b, nb, fpp in self.records:
            ms = self.lib.ms3d_event_elapsed_ms(a, b)
            self.lib.ms3d_event_destroy(a); self.lib.ms3d_event_destroy(b)
            if ms < 0:
                continue
            flops = 0.0
            if isinstance(nb, list):          # a batched launch: the sum over its layers
                tot = 0.0
                for pairs_t, per_pair, const, fpp_l in nb:
                    pairs = float(pairs_t.item())
                    tot += pairs * per_pair + const
                    flops += pairs * fpp_l
                nb = tot
            elif isinstance(nb, tuple):
                pairs = float(nb[0].item())
                flops = pairs * fpp
                nb = pairs * nb[1] + nb[2]
            g = out.setdefault(key, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
            g["launches"] += 1; g["ms"] += ms; g["bytes"] += nb; g["flops"] += flops
        self.records = []
        return out


class WgradQueue:
    """Backward-weight slab reductions of a GROUP of layers, run as ONE launch (ms3d_wgrad_reduce_multi) when the group's
    last layer has finished its backward pass (MinkowskiEngine/functional.py, GroupFlushFn) instead of one ~5 us launch
    per layer behind every backward-weight kernel (~90 per PointGroup step, each with a dependent-launch gap).  The
    layers' dW tensors are handed to autograd unreduced; the flush fills them in before anybody downstream (gradient
    accumulation, a data-parallel wrapper's bucket hooks, the optimizer) sees them.  Summation order per element is the
    per-layer kernels': bit-identical gradients."""
    _host = _host_np = _dev = _copied = None   # descriptor staging, shared by all queues of the process (one device per process)
    _turn = 0
    RING = 16                           # pinned staging slots: a slot is rewritten 16 flushes (~4 steps) later

    side_mode = False            # MS3D_WGRAD_STREAM=3: see flush()

    def __init__(self, lib, timer=None):
        self.lib = lib
        self.items = []          # (slabs tensor, dW tensor, n floats per slab, slabs)
        self.launches = []       # (variant, 128-byte description, blocks, tensors it points to, timing record or None)
        self.timer = timer

    def add(self, slabs, dW, n, nblk):
        self.items.append((slabs, dW, int(n), int(nblk)))

    def add_launch(self, desc, keep, timing=None):
        """a backward-weight KERNEL left to a batched launch (ms3d_spconv_wgrad_multi): desc = the 128 bytes the library
        wrote (int32: first block, grid x / y / z, variant, slabs); keep = the tensors the description points to"""
        head = np.frombuffer(desc, dtype=np.int32, count=6)
        self.launches.append((int(head[4]), desc, int(head[1]) * int(head[2]) * int(head[3]), keep, timing))

    def flush(self):
        """ONE staging table and ONE host->device copy per flush: [launch descriptions, 128 bytes each, grouped by kernel
        shape class | reduction descriptions, 32 bytes each], then one launch per shape class and one reduction launch"""
        launches, self.launches = self.launches, []
        items, self.items = self.items, []
        if self.side_mode:
            # MS3D_WGRAD_STREAM=3: the group's backward-weight kernels ran on the second stream beside the backward-data
            # chain; their slab reduction follows them THERE, and the caller's stream waits for that stream once per group
            # -- here, in the group's GroupFlushFn node, i.e. in front of everything that reads a dW of the group (gradient
            # accumulation, a data-parallel wrapper's bucket hooks, the optimizer).
            dev_ = items[0][1].device if items else torch.device("cuda", torch.cuda.current_device())
            main, side = torch.cuda.current_stream(dev_), wgrad_stream(dev_)
            if items:
                with torch.cuda.stream(side):
                    self._flush(launches, items)
            main.wait_stream(side)
            return
        self._flush(launches, items)

    def _flush(self, launches, items):
        if not launches and not items:
            return
        dev = (launches[0][3][0] if launches else items[0][1]).device
        by_variant = {}
        for l in launches:
            by_variant.setdefault(l[0], []).append(l)
        rows = 4 * len(launches) + len(items)            # in 32-byte rows
        cls = WgradQueue
        if cls._host is None or cls._host[0].shape[0] < rows or cls._dev[0].device != dev:
            cap = (max(2 * rows, 512) + 3) & ~3          # whole 128-byte rows
            cls._host = [torch.empty((cap, 4), dtype=torch.int64).pin_memory() for _ in range(self.RING)]
            cls._host_np = [h.numpy().view(np.uint64) for h in cls._host]
            cls._dev = [torch.empty((cap, 4), dtype=torch.int64, device=dev) for _ in range(self.RING)]
            cls._copied = [None] * self.RING
        turn = cls._turn
        cls._turn = (turn + 1) % self.RING
        hn, tdev = cls._host_np[turn], cls._dev[turn]
        if cls._copied[turn] is not None:
            # the slot's previous host->device copy must have executed before the pinned buffer is rewritten: a host that
            # runs more than RING flushes (~4 steps) ahead of the GPU would otherwise corrupt a queued table
            cls._copied[turn].synchronize()
        h32 = hn.view(np.int32).reshape(-1, 32)          # the same bytes as 128-byte rows
        plan, row = [], 0
        for variant, group in by_variant.items():
            begin = 0
            for i, (_, desc, blocks, _, _) in enumerate(group):
                h32[row + i] = np.frombuffer(desc, dtype=np.int32, count=32)
                h32[row + i, 0] = begin
                begin += blocks
            plan.append((variant, row, len(group), begin, [g[4] for g in group if g[4] is not None]))
            row += len(group)
        r0 = 4 * row
        begin_r = 0
        for i, (slabs, dW, n, nblk) in enumerate(items):
            ps, pw = slabs.data_ptr(), dW.data_ptr()
            wide = n >= 32768 and (n & 3) == 0 and (ps & 15) == 0 and (pw & 15) == 0   # ms3d_wgrad_reduce_blocks
            blocks = -(-(n // 4) // 64) if wide else -(-n // 16)
            hn[r0 + i, 0], hn[r0 + i, 1], hn[r0 + i, 2] = ps, pw, n
            hn[r0 + i, 3] = (nblk & 0xffffffff) | (((begin_r | (0x80000000 if wide else 0)) & 0xffffffff) << 32)
            begin_r += blocks
        tdev[:rows].copy_(cls._host[turn][:rows], non_blocking=True)
        ev = cls._copied[turn]
        if ev is None:
            ev = cls._copied[turn] = torch.cuda.Event()
        ev.record()
        base = tdev.data_ptr()
        for variant, first, n_desc, total, timing in plan:
            ev = self.timer.conv_group("wgrad", timing) if (self.timer is not None and timing) else None
            if ev is not None:
                self.lib.ms3d_event_record(ev[0], _lib.stream_handle())
            _lib.check(self.lib.ms3d_spconv_wgrad_multi(C.c_void_p(base + 128 * first), n_desc, int(total), int(variant),
                                                        _lib.stream_handle()), "ms3d_spconv_wgrad_multi")
            if ev is not None:
                self.lib.ms3d_event_record(ev[1], _lib.stream_handle())
        if items:
            _lib.check(self.lib.ms3d_wgrad_reduce_multi(C.c_void_p(base + 32 * r0), len(items), int(begin_r),
                                                        _lib.stream_handle()), "ms3d_wgrad_reduce_multi")
        # `launches` / `items` (x, dy, the slab tensors) die here: the caching allocator reuses them stream-ordered,
        # behind the launches


class _HipEngine:
    PARTIAL_ROWS = 1024

    # ---- coordinates
    def _cws(self, n, dev):
        return self.ws.get("coord", self.lib.ms3d_coord_workspace_bytes(max(n, 1)), dev)

    def sparse_quantize(self, coords):
        coords = self._dev(coords)
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.size(1) == 4
        n, dev = coords.size(0), coords.device
        uniq = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ws = self._cws(n, dev)
        nu = C.c_int(0)
        _lib.check(self.lib.ms3d_sparse_quantize(_lib.ptr(coords), n, _lib.ptr(uniq), _lib.ptr(inv), C.byref(nu),
                                                 _lib.ptr(ws), C.c_size_t(ws.numel()), _lib.stream_handle()),
                   "ms3d_sparse_quantize")
        return uniq[:nu.value], inv[:n]

    def spatial_order(self, coords):
        """row permutation that sorts voxels by (batch, Morton code); None = keep the given order"""
        coords = self._dev(coords)
        V = coords.size(0)
        if V < 4096:
            return None
        keys = torch.empty(V, dtype=torch.int64, device=coords.device)
        _lib.check(self.lib.ms3d_morton_keys(_lib.ptr(coords), V, _lib.ptr(keys), _lib.stream_handle()), "ms3d_morton_keys")
        return torch.sort(keys).indices   # radix sort (rocPRIM through torch): plumbing, not a hot kernel

    def kmap_k3(self, coords, ts):
        coords = self._dev(coords)
        V, dev = coords.size(0), coords.device
        nbr = torch.empty((27, max(V, 1)), dtype=torch.int32, device=dev)
        ws = self._cws(V, dev)
        _lib.check(self.lib.ms3d_kmap_k3(_lib.ptr(coords), V, int(ts), _lib.ptr(nbr), _lib.ptr(ws),
                                         C.c_size_t(ws.numel()), _lib.stream_handle()), "ms3d_kmap_k3")
        return nbr

    def downsample(self, coords, ts):
        coords = self._dev(coords)
        V, dev = coords.size(0), coords.device
        oc = torch.empty((max(V, 1), 4), dtype=torch.int32, device=dev)
        parent = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
        koff = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
        ws = self._cws(V, dev)
        nc = C.c_int(0)
        _lib.check(self.lib.ms3d_downsample(_lib.ptr(coords), V, int(ts), _lib.ptr(oc), _lib.ptr(parent),
                                            _lib.ptr(koff), C.byref(nc), _lib.ptr(ws), C.c_size_t(ws.numel()),
                                            _lib.stream_handle()), "ms3d_downsample")
        return oc[:nc.value], parent[:V], koff[:V]

    def kmap_k2(self, parent, koff, vc):
        vf, dev = parent.numel(), parent.device
        down = torch.empty((8, max(vc, 1)), dtype=torch.int32, device=dev)
        up = torch.empty((8, max(vf, 1)), dtype=torch.int32, device=dev)
        _lib.check(self.lib.ms3d_kmap_k2(_lib.ptr(parent), _lib.ptr(koff), vf, int(vc), _lib.ptr(down), _lib.ptr(up),
                                         _lib.stream_handle()), "ms3d_kmap_k2")
        return down, up

    # ---- convolution
    def prep_weights(self, W, K, cin_e, cout_e, transpose=False, mirror=False):
        """-> weight image handle [2, n]: row 0 the fragment-major image, row 1 the same weights in streamed order"""
        W = self._dev(W)
        self.lib.ms3d_spconv_wf_floats.restype = C.c_size_t
        nfl = self.lib.ms3d_spconv_wf_floats(int(K), int(cin_e), int(cout_e))
        wf = torch.empty((2, nfl), dtype=torch.float32, device=W.device)
        _lib.check(self.lib.ms3d_spconv_prep_weights(_lib.ptr(W), int(K), int(cin_e), int(cout_e), int(transpose),
                                                     int(mirror), _lib.ptr(wf[0]), _lib.ptr(wf[1]), _lib.stream_handle()),
                   "ms3d_spconv_prep_weights")
        return wf

    def prep_weights_pair(self, W, K, cin, cout, mirror_bwd=False):
        """forward image and backward-data image (W^T, offsets mirrored for k3) in one launch"""
        W = self._dev(W)
        self.lib.ms3d_spconv_wf_floats.restype = C.c_size_t
        n = self.lib.ms3d_spconv_wf_floats(int(K), int(cin), int(cout))    # == wf_floats(K, cout, cin)
        buf = torch.empty((2, 2, n), dtype=torch.float32, device=W.device)
        wf, wft = buf[0], buf[1]                                             # each [2, n]: image, streamed image
        _lib.check(self.lib.ms3d_spconv_prep_weights_pair(_lib.ptr(W), int(K), int(cin), int(cout), int(mirror_bwd),
                                                          _lib.ptr(wf[0]), _lib.ptr(wft[0]), _lib.ptr(wf[1]),
                                                          _lib.ptr(wft[1]), _lib.stream_handle()),
                   "ms3d_spconv_prep_weights_pair")
        return wf, wft

    def identity_table(self, n, device):
        """[1, n] table for K = 1 (dense per-row matmul through the conv kernel)"""
        t = self._ident.get((n, device)) if hasattr(self, "_ident") else None
        if t is None:
            if not hasattr(self, "_ident"):
                self._ident = {}
            if len(self._ident) > 8:
                self._ident.clear()
            t = torch.arange(n, dtype=torch.int32, device=device).view(1, n)
            self._ident[(n, device)] = t
        return t

    def pairlist(self, nbr, K, vout, cin=16, cout=16):
        """tile-compacted pair list of an offset-major table for a convolution of this shape, built on first use and cached
        on the table tensor (every convolution of a level shares it) -> (tile_start, entries) or (None, None) when the
        shape takes a kernel that walks the table itself.  Layers with more than 32 channels on a side want 128-row
        tiles, the others 64-row tiles (ms3d_spconv_pairlist_rows)."""
        rows = self._geom("ms3d_spconv_pairlist_rows", vout, K, cin, cout)
        if rows == 0:
            return (None, None)
        cache = getattr(nbr, "_ms3d_pairlist", None)
        if cache is None:
            cache = nbr._ms3d_pairlist = {}
        dense_rows = self._geom("ms3d_spconv_pairlist_rows_dense", vout, K, cin, cout)
        if dense_rows != rows:
            # a layer shape with a second kernel variant for DENSE tables (32 -> 32: both column blocks per wave on 32-row
            # tiles).  The table's pair count decides, once per table: a 32-row tile of a full-resolution level (5.5 of 27
            # neighbours) pads 6.5 pairs per offset to 16 and loses 11 %, at 10 neighbours the variant saves 10-12 %.
            dense = cache.get("dense")
            if dense is None:
                import threading
                mode = os.environ.get("MS3D_PL_NARROW", "1")
                if mode == "2":
                    dense = True
                elif mode == "3" or threading.current_thread().name.startswith("ms3d-prefetch"):
                    # one device->host read per table -- ONLY on the input pipeline's thread (ME.prefetch_coordinates builds
                    # the lists there, on its own stream): on the calling thread the read would drain the queue the
                    # interpreter has run ahead of, which costs more than the variant saves.  (3 = read wherever it is asked)
                    pairs = getattr(nbr, "_ms3d_pairs_dev", None)
                    if pairs is None:
                        pairs = nbr._ms3d_pairs_dev = (nbr >= 0).sum()
                    dense = float(pairs) >= float(os.environ.get("MS3D_PL_NARROW_DENSITY", "8.0")) * max(int(vout), 1)
                else:
                    dense = False          # a table built on the calling thread: the 64-row list (deterministic per configuration)
                cache["dense"] = dense
            if dense:
                rows = dense_rows
        pl = cache.get(rows)
        if pl is None:
            self.lib.ms3d_kmap_pairlist_capacity_rows.restype = C.c_size_t
            cap = self.lib.ms3d_kmap_pairlist_capacity_rows(int(K), int(vout), rows)
            tile_start = torch.empty(self.lib.ms3d_kmap_pairlist_header_ints_rows(int(vout), rows), dtype=torch.int32,
                                     device=nbr.device)
            entries = torch.empty((cap, 2), dtype=torch.int32, device=nbr.device)
            ws = self._cws(1, nbr.device)
            _lib.check(self.lib.ms3d_kmap_pairlist_build_rows(_lib.ptr(nbr), int(K), int(vout), rows, _lib.ptr(tile_start),
                                                              _lib.ptr(entries), _lib.ptr(ws), C.c_size_t(ws.numel()),
                                                              _lib.stream_handle()), "ms3d_kmap_pairlist_build_rows")
            pl = cache[rows] = (tile_start, entries)
        return pl

    def _pl_rows(self, pl):
        """rows per tile of a pair list (0 = no list): what ms3d_spconv_partial_blocks wants to know about it"""
        return int(self.lib.ms3d_kmap_pairlist_rows_of(_lib.ptr(pl[0]))) if pl[0] is not None else 0

    def offsetlist(self, nbr, K, vout):
        """offset-major pair list of a table for the backward-weight kernel, cached on the table tensor
        -> (kt_start, entries) or (None, None) for small levels"""
        ol = getattr(nbr, "_ms3d_offsetlist", None)
        if ol is None:
            ol = (None, None)
            if self.lib.ms3d_kmap_pairlist_wanted(int(K), int(vout)):
                self.lib.ms3d_kmap_offsetlist_capacity.restype = C.c_size_t
                cap = self.lib.ms3d_kmap_offsetlist_capacity(int(K), int(vout))
                self.lib.ms3d_kmap_offsetlist_header_ints.restype = C.c_size_t
                kt_start = torch.empty(self.lib.ms3d_kmap_offsetlist_header_ints(int(K), int(vout)), dtype=torch.int32,
                                       device=nbr.device)
                entries = torch.empty((cap, 2), dtype=torch.int32, device=nbr.device)
                ws = self._cws(1, nbr.device)
                _lib.check(self.lib.ms3d_kmap_offsetlist_build(_lib.ptr(nbr), int(K), int(vout), _lib.ptr(kt_start),
                                                               _lib.ptr(entries), _lib.ptr(ws), C.c_size_t(ws.numel()),
                                                               _lib.stream_handle()), "ms3d_kmap_offsetlist_build")
                ol = (kt_start, entries)
            nbr._ms3d_offsetlist = ol
        return ol

    def conv_forward(self, x, wf, nbr, vout, K, cin, cout, pre=None, pre_relu=False, residual=None, bn_bwd=None,
                     out_stats=False, bias=None):
        """out = sum_k act(x[nbr[k]]) @ Weff[k] (+ residual).  pre = (scale, shift) fuses BN(+ReLU) on the input.
        bn_bwd = (bn_x, scale, shift, mean, invstd): backward-data epilogue of a fused BN+ReLU; returns (dz, s1s2).
        out_stats: also return the per-block (sum, sum of squares) partials of the output [nparts, 2, cout]."""
        x = self._dev(x)
        out = torch.empty((vout, cout), dtype=torch.float32, device=x.device)
        ps, pb = (pre if pre is not None else (None, None))
        partial = None
        bnargs = [None] * 5
        pl = self.pairlist(nbr, K, vout, cin, cout)
        if bn_bwd is not None or out_stats:
            nparts = self.lib.ms3d_spconv_partial_blocks(int(vout), int(K), int(cin), int(cout), self._pl_rows(pl))
            partial = torch.empty((nparts, 2, cout), dtype=torch.float32, device=x.device)
        if bn_bwd is not None:
            bnargs = [_f32(t) for t in bn_bwd]
        _lib.check(self.lib.ms3d_spconv_forward(
            _lib.ptr(x), _lib.ptr(wf[0] if wf.dim() == 2 else wf), _lib.ptr(nbr), int(vout), int(K), int(cin), int(cout), _lib.ptr(out),
            _lib.ptr(_f32(ps)), _lib.ptr(_f32(pb)), int(bool(pre_relu)), _lib.ptr(_f32(residual)),
            *[_lib.ptr(t) for t in bnargs], _lib.ptr(partial), int(bool(out_stats)), _lib.ptr(_f32(bias)),
            _lib.ptr(pl[0]), _lib.ptr(pl[1]), _lib.ptr(wf[1]) if wf.dim() == 2 else None,
            _lib.stream_handle()), "ms3d_spconv_forward")
        if bn_bwd is None:
            return (out, partial) if out_stats else out
        s1s2 = torch.empty((2, cout), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.ms3d_reduce_partials(_lib.ptr(partial), partial.size(0), 2 * cout, _lib.ptr(s1s2),
                                                 _lib.stream_handle()), "ms3d_reduce_partials")
        return out, s1s2

    # ---- augmentation
    def elastic(self, xyz, noise, gran, mag):
        """xyz [N,3] (voxel units), noise float32 [3,bx,by,bz] from the host RNG -> float64 [N,3] distorted points"""
        xyz = self._dev(xyz).to(torch.float64).contiguous()
        noise = self._dev(noise).to(torch.float32).contiguous().clone()
        tmp = torch.empty_like(noise)
        out = torch.empty_like(xyz)
        _, bx, by, bz = noise.shape
        _lib.check(self.lib.ms3d_elastic_distort(_lib.ptr(xyz), int(xyz.size(0)), _lib.ptr(noise), _lib.ptr(tmp), int(bx),
                                                 int(by), int(bz), C.c_double(gran), C.c_double(mag), _lib.ptr(out),
                                                 _lib.stream_handle()), "ms3d_elastic_distort")
        return out

    # ---- instance post-processing (validation / test)
    def proposal_cross_intersection(self, pair_point, pair_cluster, P):
        """unique (cluster, point) pairs sorted by point -> int32 [P, P] shared-point counts (sizes on the diagonal)"""
        pair_point = self._dev(pair_point).to(torch.int32).contiguous()
        pair_cluster = self._dev(pair_cluster).to(torch.int32).contiguous()
        inter = torch.empty((P, P), dtype=torch.int32, device=pair_point.device)
        _lib.check(self.lib.ms3d_proposal_cross_intersection(_lib.ptr(pair_point), _lib.ptr(pair_cluster),
                                                             int(pair_point.numel()), int(P), _lib.ptr(inter),
                                                             _lib.stream_handle()), "ms3d_proposal_cross_intersection")
        return inter

    def nms_greedy(self, inter, order, threshold):
        """greedy NMS over `order` (descending score) -> picked proposal ids in pick order (device int32)"""
        inter = self._dev(inter); order = self._dev(order).to(torch.int32).contiguous()
        P = inter.size(0)
        pick = torch.empty(max(P, 1), dtype=torch.int32, device=inter.device)
        n_pick = torch.zeros(1, dtype=torch.int32, device=inter.device)
        ws = torch.empty(max(P, 1), dtype=torch.uint8, device=inter.device)
        _lib.check(self.lib.ms3d_nms_greedy(_lib.ptr(inter), _lib.ptr(order), int(P), C.c_float(threshold), _lib.ptr(ws),
                                            _lib.ptr(pick), _lib.ptr(n_pick), _lib.stream_handle()), "ms3d_nms_greedy")
        return pick[:int(n_pick.item())]

    # ---- one library call per layer and direction (used by MinkowskiEngine/functional.py)
    def _fast(self, name):
        """library entry point with declared argtypes (plain ints / None for pointers)"""
        cache = self.__dict__.setdefault("_fast_cache", {})
        fn = cache.get(name)
        if fn is None:
            fn = getattr(self.lib, name)
            fn.argtypes = _FAST_ARGTYPES[name]
            fn.restype = C.c_int
            cache[name] = fn
        return fn

    def _geom(self, what, *key):
        """launch-geometry queries of the library, memoised (three ctypes calls per layer otherwise)"""
        cache = self.__dict__.setdefault("_geom_cache", {})
        k = (what,) + key
        v = cache.get(k)
        if v is None:
            fn = getattr(self.lib, what)
            if what in ("ms3d_spconv_wf_floats", "ms3d_spconv_layer_ws_floats", "ms3d_spconv_wgrad_ws_floats"):
                fn.restype = C.c_size_t
            v = cache[k] = fn(*[int(a) for a in key])
        return v

    def wf_floats(self, K, cin, cout):
        """floats of a layer's weight buffer: forward and backward-data image, each followed by its streamed form"""
        return 6 * self._geom("ms3d_spconv_wf_floats", K, cin, cout)    # [image | aux 2n | transposed image | aux 2n]

    def prep_weights_multi(self, layers):
        """layers: [(W [K,cin,cout] parameter, wf_buf, K, cin, cout, mirror_bwd)] -> both weight images of every layer in
        ONE launch; bumps `weight_token`, the stamp that says "the images in these buffers are the current weights"
        (release_weights() bumps it again).  The descriptor table lives on the device and is rebuilt only when the set
        of tensors changes."""
        if not layers:
            return
        key = tuple((w.data_ptr(), b.data_ptr(), m) for w, b, _, _, _, m in layers)
        cached = self._prep_table
        if cached is None or cached[0] != key:
            rec = np.zeros(len(layers), dtype=np.dtype([("W", "<u8"), ("wf", "<u8"), ("wft", "<u8"), ("K", "<i4"),
                                                         ("Cin", "<i4"), ("Cout", "<i4"), ("mirror", "<i4"),
                                                         ("begin", "<i4"), ("stream", "<i4")]))
            begin = 0
            for i, (w, b, K, cin, cout, m) in enumerate(layers):
                assert w.is_contiguous() and w.dtype == torch.float32 and b.numel() >= self.wf_floats(K, cin, cout)
                rec[i] = (w.data_ptr(), b.data_ptr(), b.data_ptr() + 4 * 3 * self._geom("ms3d_spconv_wf_floats", K, cin, cout),
                          K, cin, cout, int(bool(m)), begin,
                          int(self.lib.ms3d_spconv_aux_kind(int(K), int(cin), int(cout))))
                begin += self.lib.ms3d_spconv_prep_blocks(int(K), int(cin), int(cout))
            table = torch.from_numpy(rec.view(np.uint8).copy()).to(layers[0][0].device)
            cached = self._prep_table = (key, table, begin)
        _lib.check(self.lib.ms3d_spconv_prep_weights_multi(_lib.ptr(cached[1]), len(layers), int(cached[2]),
                                                           _lib.stream_handle()), "ms3d_spconv_prep_weights_multi")
        self.weight_token += 1
        if self.__dict__.get("_wgrad_join_queued"):
            # MS3D_WGRAD_STREAM=2: a backward pass that raised never ran its end-of-pass callback -- join here, or no
            # later pass would queue one
            self._wgrad_join_queued = False
            torch.cuda.current_stream().wait_stream(wgrad_stream(layers[0][0].device))

    def release_weights(self):
        self.weight_token += 1

    def wgrad_queue(self):
        """a fresh WgradQueue, or None when deferred slab reductions are off (MS3D_WGRAD_DEFER=0, or backward-weight on
        its own stream: the flush would have to run there)"""
        on = self.__dict__.get("_wgrad_defer")
        if on is None:
            on = self._wgrad_defer = os.environ.get("MS3D_WGRAD_DEFER", "1") != "0" and self.wgrad_stream_mode() in (0, 3)
        if not on:
            return None
        q = WgradQueue(self.lib, self.kernel_timer)
        q.side_mode = self.wgrad_stream_mode() == 3     # the group's backward-weight work ran on the second stream
        return q

    def wgrad_batch_enabled(self):
        """MS3D_WGRAD_BATCH (default 1): the backward-weight kernels of the small levels (f32 table walk) of a layer group run
        as ONE launch per kernel shape class when the group is flushed, instead of one launch per layer"""
        on = self.__dict__.get("_wgrad_batch")
        if on is None:
            on = self._wgrad_batch = os.environ.get("MS3D_WGRAD_BATCH", "1") != "0"
        return on

    def conv_layer_forward(self, x, W3, nbr_fwd, vout, K, cin, cout, mirror_bwd, pre, pre_relu, residual, bias,
                           want_stats, wf_ready=None):
        """-> (y, stats or None, wf_buf) ; wf_buf carries both weight images to the backward call.  wf_ready: a buffer
        that already holds the current images (prep_weights_multi) -- the per-layer re-lay launch is skipped."""
        x = self._dev(x)
        dev = x.device
        wf_buf = wf_ready if wf_ready is not None else torch.empty(self.wf_floats(K, cin, cout), dtype=torch.float32, device=dev)
        pl = self.pairlist(nbr_fwd, K, vout, cin, cout)
        nparts = self._geom("ms3d_spconv_partial_blocks", vout, K, cin, cout, self._pl_rows(pl)) if want_stats else 0
        ps, pb = (pre if pre is not None else (None, None))
        timer = self.kernel_timer
        tok = timer.conv("fwd", K, cin, cout, nbr_fwd, vout) if timer is not None else None
        ev0, ev1 = tok if tok is not None else (None, None)
        ext = self.ext
        if ext is not None:
            # one pybind call: outputs allocated and the C ABI entered from native code (csrc_host/ms3d_host.cpp)
            y, stats = ext.conv_layer_forward(
                x, None if wf_ready is not None else self._dev(W3), nbr_fwd, vout, K, cin, cout, bool(mirror_bwd),
                _f32(ps), _f32(pb), bool(pre_relu), _f32(residual), _f32(bias), wf_buf, nparts, pl[0], pl[1],
                (ev0.value or 0) if ev0 is not None else 0, (ev1.value or 0) if ev1 is not None else 0)
            return y, stats, wf_buf
        y = torch.empty((vout, cout), dtype=torch.float32, device=dev)
        stats = torch.empty((nparts, 2, cout), dtype=torch.float32, device=dev) if want_stats else None
        _lib.check(self._fast("ms3d_spconv_layer_forward")(
            _p(x), _p(None if wf_ready is not None else self._dev(W3)), _p(nbr_fwd), int(vout), int(K), int(cin), int(cout),
            int(bool(mirror_bwd)), _p(_f32(ps)), _p(_f32(pb)), int(bool(pre_relu)), _p(_f32(residual)),
            _p(_f32(bias)), _p(wf_buf), _p(y), _p(stats), _p(pl[0]), _p(pl[1]),
            ev0, ev1, _lib.stream_handle()), "ms3d_spconv_layer_forward")
        return y, stats, wf_buf

    def res_block_forward(self, x, stats_in, wf1, wf2, nbr, V, c, bn0, g0, b0, bn1, g1, b1, want_stats):
        """the four library calls of an identity-skip residual block's forward from ONE host call (csrc_host/ms3d_host.cpp:
        res_block_forward) -> (y1, y2, stats of y2 or None, bn0 = (mean, invstd, scale, shift), bn1 = ...), or None when the
        fast path does not apply (no host extension, a sampled step of the kernel timer, an unusual BatchNorm)"""
        ext = self.ext
        timer = self.kernel_timer
        if (ext is None or (timer is not None and getattr(timer, "sampling", True)) or not hasattr(ext, "res_block_forward")
                or os.environ.get("MS3D_RESBLOCK_EXT", "1") == "0"):
            return None
        pl = self.pairlist(nbr, 27, V, c, c)
        nparts = self._geom("ms3d_spconv_partial_blocks", V, 27, c, c, self._pl_rows(pl))
        y1, y2, st2, o0, o1 = ext.res_block_forward(
            x, stats_in, wf1, wf2, nbr, V, c, pl[0], pl[1], nparts, bool(want_stats), _f32(g0), _f32(b0), bn0.running_mean,
            bn0.running_var, bn0.eps, bn0.momentum, _f32(g1), _f32(b1), bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum)
        return y1, y2, st2, o0.unbind(0), o1.unbind(0)

    def conv_layer_backward(self, x, dy, wf_buf, nbr_fwd, nbr_bwd, vin, vout, K, cin, cout, bn, need_dx, dx_add=None,
                            defer=None, join_now=False):
        """-> (dx or None, dgb [2,cin] = (dbeta, dgamma) or None, dW [K,cin,cout]).  dx_add [vin, cin]: a gradient that
        reaches x over a skip connection, added to dx inside the BatchNorm-backward pass / the residual epilogue (needs
        training-mode statistics when a BatchNorm is fused: `fuses_dx_add`).  defer: a WgradQueue -- dW is returned
        UNREDUCED and completed by the queue's flush.  join_now: with MS3D_WGRAD_STREAM=2, make the caller's stream wait
        for this layer's backward-weight (somebody consumes dW on it right after this call)."""
        x = self._dev(x); dy = self._dev(dy)
        dev = x.device
        ws = self.ws.get("layer", 4 * self._geom("ms3d_spconv_layer_ws_floats", vin, vout, K, cin, cout), dev)
        has_bn = bn is not None
        want_dx = need_dx or has_bn
        plf = self.offsetlist(nbr_fwd, K, vout)
        plb = self.pairlist(nbr_bwd, K, vin, cout, cin) if want_dx else (None, None)
        timer = self.kernel_timer
        tok = timer.conv("fwd", K, cout, cin, nbr_bwd, vin) if (timer is not None and want_dx) else None
        ev0, ev1 = tok if tok is not None else (None, None)
        tok = timer.conv("wgrad", K, cin, cout, nbr_fwd, vout) if timer is not None else None
        ev2, ev3 = tok if tok is not None else (None, None)
        mode = self.wgrad_stream_mode()
        # the backward-weight KERNEL of a small-level layer (f32 table walk) is left to the queue's batched launch too
        batch = defer is not None and mode == 0 and self.wgrad_batch_enabled() and \
            self._geom("ms3d_spconv_wgrad_is_table_walk", vout, K, cin, cout, int(plf[0] is not None)) == 1
        timing = None
        if batch and ev2 is not None:
            # a sampled step of bench.py: the batched launch is timed as a whole, this layer contributes its byte count
            timing = timer.records.pop()[0][1:] + (nbr_fwd,)
            timer._pool.extend((ev2, ev3))
            ev2 = ev3 = None
        if self.ext is not None and mode == 0:
            evs = [(e.value or 0) if e is not None else 0 for e in (ev0, ev1, ev2, ev3)]
            dx, dgb, dW, slabs, nblk, desc = self.ext.conv_layer_backward(
                x, dy, wf_buf, nbr_fwd, nbr_bwd, vin, vout, K, cin, cout,
                bn["scale"] if has_bn else None, bn["shift"] if has_bn else None, bn["mean"] if has_bn else None,
                bn["invstd"] if has_bn else None, bool(has_bn and bn["relu"]), bool(has_bn and bn["training"]), bool(need_dx),
                _f32(dx_add) if (dx_add is not None and need_dx) else None, ws, plf[0], plf[1], plb[0], plb[1], *evs,
                self._geom("ms3d_spconv_wgrad_ws_floats", vout, K, cin, cout) if defer is not None else 0, batch)
            if desc:
                defer.add_launch(desc, (x, dy, nbr_fwd, bn, slabs, dW), timing)
            if nblk > 0:
                defer.add(slabs, dW, K * cin * cout, nblk)
            return dx, dgb, dW
        dx = torch.empty((vin, cin), dtype=torch.float32, device=dev) if want_dx else None
        dgb = torch.empty((2, cin), dtype=torch.float32, device=dev) if has_bn else None
        dW = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
        side = wgrad_stream(dev) if mode else None
        ws2 = None
        if side is not None:
            # the second stream's slab area is allocated UNDER that stream: when a larger layer makes it grow, the old
            # block goes back to the side stream's pool and is reused behind the kernels still writing it
            with torch.cuda.stream(side):
                ws2 = self.ws.get("layer_wgrad", 4 * self._geom("ms3d_spconv_layer_ws_floats", vin, vout, K, cin, cout), dev)
            if mode != 3:
                defer = None       # (mode 3 keeps the deferral: the group's flush runs on the second stream and joins there)
        slabs, n_defer = None, None
        if defer is not None:
            if "_defer_n" not in self.__dict__:
                self._defer_n = C.c_int(0)
                self._defer_n_addr = C.addressof(self._defer_n)
            slabs = torch.empty(self._geom("ms3d_spconv_wgrad_ws_floats", vout, K, cin, cout), dtype=torch.float32, device=dev)
            n_defer = self._defer_n_addr
        launch = (C.c_char * 128)() if (batch and slabs is not None) else None
        _lib.check(self._fast("ms3d_spconv_layer_backward")(
            _p(x), _p(dy), _p(wf_buf), _p(nbr_fwd), _p(nbr_bwd), int(vin), int(vout), int(K),
            int(cin), int(cout), _p(bn["scale"] if has_bn else None), _p(bn["shift"] if has_bn else None),
            _p(bn["mean"] if has_bn else None), _p(bn["invstd"] if has_bn else None),
            int(bool(has_bn and bn["relu"])), int(bool(has_bn and bn["training"])), int(bool(need_dx)), _p(dx),
            _p(_f32(dx_add) if (dx_add is not None and need_dx) else None), _p(dgb), _p(dW), _p(ws), _p(plf[0]), _p(plf[1]), _p(plb[0]),
            _p(plb[1]), ev0, ev1, ev2, ev3, _p(ws2), side.cuda_stream if side is not None else None,
            int(mode == 1 or (mode == 2 and join_now) or (mode == 3 and defer is None)), _p(slabs), n_defer,
            C.addressof(launch) if launch is not None else None, _lib.stream_handle()),
            "ms3d_spconv_layer_backward")
        if launch is not None and np.frombuffer(launch, dtype=np.int32, count=6)[4] != 0:
            defer.add_launch(bytes(launch), (x, dy, nbr_fwd, bn, slabs, dW), timing)
        if slabs is not None and self._defer_n.value > 0:
            defer.add(slabs, dW, K * cin * cout, self._defer_n.value)
        if mode == 2:
            # the second stream runs free until the end of the backward pass: the allocator must not hand x / dy to
            # somebody else while it still reads them, and the pass ends with the join
            x.record_stream(side); dy.record_stream(side); dW.record_stream(side)
            if not join_now:
                self._queue_wgrad_join(side)
        elif mode == 3 and defer is not None:
            # joined by the layer group's flush (WgradQueue.flush, behind which every consumer of dW sits): until then
            # the second stream reads x / dy and writes the slabs / dW, all of them blocks of the caller's stream's pool
            x.record_stream(side); dy.record_stream(side); dW.record_stream(side)
            if slabs is not None:
                slabs.record_stream(side)
        return (dx if need_dx else None), dgb, dW

    @staticmethod
    def fuses_dx_add(bn):
        """can conv_layer_backward add a skip gradient itself for this fused BatchNorm description?"""
        return bn is None or (bool(bn["relu"]) and bool(bn["training"]))

    def wgrad_stream_mode(self):
        """MS3D_WGRAD_STREAM: 0 (default) backward-weight on the caller's stream; 1: on a second stream, joined at the
        end of every layer's backward (safe under DistributedDataParallel, whose gradient hooks order only against the
        caller's stream); 2: joined once, at the end of the backward pass (single process); 3 (round 5): joined once
        per layer GROUP, inside the group's deferred-reduction node (functional.GroupFlushFn) -- safe under
        DistributedDataParallel like 1 (no gradient reaches a hook in front of the join), overlapping like 2.
        Measured (profiles/r03_wgrad_stream_sweep.txt, one MI355X, 40 steps): HAIS 81.5 -> 83.7 (mode 1) -> 85.0 scenes/s
        (mode 2); PointGroup unchanged within its run-to-run noise (169-175 in every mode).  The two streams' kernels
        slow each other down (summed kernel time of the backward pass 11.3 -> 15.2 ms on PointGroup for a span that
        shrinks 12.8 -> 12.6 ms), so the per-kernel durations the roofline is computed from get worse while the step
        gets slightly better: off by default, a throughput knob for the m = 32 models."""
        m = self.__dict__.get("_wgrad_mode")
        if m is None:
            m = self._wgrad_mode = int(os.environ.get("MS3D_WGRAD_STREAM", "0"))
        return m

    def _queue_wgrad_join(self, side):
        if self.__dict__.get("_wgrad_join_queued"):
            return

        def join():
            self._wgrad_join_queued = False
            torch.cuda.current_stream().wait_stream(side)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join)
            self._wgrad_join_queued = True
        except RuntimeError:            # not inside a backward pass (a direct call): join now
            join()

    def conv_backward_weight(self, x, dout, nbr, vout, K, cin, cout, pre=None, pre_relu=False):
        x = self._dev(x); dout = self._dev(dout)
        dW = torch.empty((K, cin, cout), dtype=torch.float32, device=x.device)
        ps, pb = (pre if pre is not None else (None, None))
        self.lib.ms3d_spconv_wgrad_ws_floats.restype = C.c_size_t
        ws = self.ws.get("wgrad", 4 * self.lib.ms3d_spconv_wgrad_ws_floats(int(vout), int(K), int(cin), int(cout)), x.device)
        ol = self.offsetlist(nbr, K, vout)
        _lib.check(self.lib.ms3d_spconv_backward_weight(
            _lib.ptr(x), _lib.ptr(dout), _lib.ptr(nbr), int(vout), int(K), int(cin), int(cout), _lib.ptr(dW),
            _lib.ptr(_f32(ps)), _lib.ptr(_f32(pb)), int(bool(pre_relu)), _lib.ptr(ws), _lib.ptr(ol[0]), _lib.ptr(ol[1]),
            _lib.stream_handle()), "ms3d_spconv_backward_weight")
        return dW

    # ---- batch norm pieces
    def _partial_ws(self, C_, dev):
        return self.ws.get("bnpart", self.PARTIAL_ROWS * 2 * C_ * 4, dev)

    def bn_stats(self, x, eps, momentum, gamma, beta, running_mean, running_var):
        """training-mode statistics -> (mean, invstd, scale, shift); running stats updated in place"""
        x = self._dev(x)
        V, C_ = x.shape
        dev = x.device
        outs = torch.empty((4, C_), dtype=torch.float32, device=dev)
        ws = self._partial_ws(C_, dev)
        _lib.check(self.lib.ms3d_bn_stats(_lib.ptr(x), C.c_long(V), int(C_), C.c_float(eps), C.c_float(momentum),
                                          _lib.ptr(_f32(gamma)), _lib.ptr(_f32(beta)), _lib.ptr(running_mean),
                                          _lib.ptr(running_var), _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]),
                                          _lib.ptr(outs[3]), _lib.ptr(ws), self.PARTIAL_ROWS, _lib.stream_handle()),
                   "ms3d_bn_stats")
        return outs[0], outs[1], outs[2], outs[3]

    def column_sum(self, x):
        """x.sum(0) for a tall f32 [V, C] matrix (bias gradients of the per-point Linear layers)"""
        x = self._dev(x)
        V, C_ = x.shape
        out = torch.empty((2, C_), dtype=torch.float32, device=x.device)
        ws = self._partial_ws(C_, x.device)
        _lib.check(self.lib.ms3d_column_sum(_lib.ptr(x), C.c_long(V), int(C_), _lib.ptr(ws), self.PARTIAL_ROWS, _lib.ptr(out),
                                            _lib.stream_handle()), "ms3d_column_sum")
        return out[0]

    def bn_finalize(self, partial, V, eps, momentum, gamma, beta, running_mean, running_var):
        """statistics from the (sum, sum of squares) partials a conv epilogue left behind -> (mean, invstd, scale, shift)"""
        if self.ext is not None:
            return self.ext.bn_finalize(partial, V, eps, momentum, _f32(gamma), _f32(beta), running_mean, running_var).unbind(0)
        C_ = partial.size(2)
        outs = torch.empty((4, C_), dtype=torch.float32, device=partial.device)
        base, row = outs.data_ptr(), 4 * C_
        _lib.check(self._fast("ms3d_bn_finalize")(_p(partial), int(partial.size(0)), int(V), int(C_),
                                                  float(eps), float(momentum), _p(_f32(gamma)),
                                                  _p(_f32(beta)), _p(running_mean), _p(running_var),
                                                  base, base + row, base + 2 * row, base + 3 * row,
                                                  _lib.stream_handle()), "ms3d_bn_finalize")
        return outs.unbind(0)

    def bn_finalize_parts(self, partials, V, eps, momentum, gamma, beta, running_mean, running_var):
        """bn_finalize over the channel-wise concatenation of several partial buffers (ME.cat of convolution outputs):
        one finalize launch per part, each on its slice of the parameters, running statistics and outputs"""
        C_ = sum(int(p.size(2)) for p in partials)
        outs = torch.empty((4, C_), dtype=torch.float32, device=partials[0].device)
        base, row = outs.data_ptr(), 4 * C_
        gamma, beta = _f32(gamma), _f32(beta)
        fn, c0 = self._fast("ms3d_bn_finalize"), 0
        for part in partials:
            c = int(part.size(2))
            o = 4 * c0
            at = lambda t: (t.data_ptr() + o) if t is not None else None
            _lib.check(fn(_p(part), int(part.size(0)), int(V), c, float(eps), float(momentum), at(gamma), at(beta),
                          at(running_mean), at(running_var), base + o, base + row + o, base + 2 * row + o,
                          base + 3 * row + o, _lib.stream_handle()), "ms3d_bn_finalize")
            c0 += c
        return outs.unbind(0)

    def bn_apply(self, x, scale, shift, relu):
        x = self._dev(x)
        y = torch.empty_like(x)
        _lib.check(self.lib.ms3d_bn_apply(_lib.ptr(x), C.c_long(x.size(0)), int(x.size(1)), _lib.ptr(scale),
                                          _lib.ptr(shift), int(bool(relu)), _lib.ptr(y), _lib.stream_handle()),
                   "ms3d_bn_apply")
        return y

    def bn_bwd_reduce(self, dy, x, scale, shift, mean, invstd, relu):
        """stand-alone BN(+ReLU) backward stage 1 -> (dz, s1s2)"""
        dy = self._dev(dy); x = self._dev(x)
        V, C_ = x.shape
        dz = torch.empty_like(x)
        ws = self._partial_ws(C_, x.device)
        nparts = C.c_int(0)
        _lib.check(self.lib.ms3d_bn_bwd_partial(_lib.ptr(dy), _lib.ptr(x), C.c_long(V), int(C_), _lib.ptr(scale),
                                                _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), int(bool(relu)),
                                                _lib.ptr(dz), _lib.ptr(ws), self.PARTIAL_ROWS, C.byref(nparts),
                                                _lib.stream_handle()), "ms3d_bn_bwd_partial")
        s1s2 = torch.empty((2, C_), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.ms3d_reduce_partials(_lib.ptr(ws), nparts.value, 2 * C_, _lib.ptr(s1s2),
                                                 _lib.stream_handle()), "ms3d_reduce_partials")
        return dz, s1s2

    def bn_bwd_apply(self, dz, x, scale, mean, invstd, s1s2):
        dx = torch.empty_like(dz)
        _lib.check(self.lib.ms3d_bn_bwd_apply(_lib.ptr(dz), _lib.ptr(x), C.c_long(x.size(0)), int(x.size(1)),
                                              _lib.ptr(scale), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(s1s2),
                                              _lib.ptr(dx), _lib.stream_handle()), "ms3d_bn_bwd_apply")
        return dx


for _name, _fn in list(vars(_HipEngine).items()):
    if not _name.startswith("__"):
        setattr(HipBackend, _name, _fn)
