"""Backend selection for the hot-path operators.

The product backend is `HipBackend`: torch device tensors in, calls into libminsu3d_hip.so through
the C ABI (include/minsu3d_hip.h), torch device tensors out.  It raises if the library is missing --
there is no CPU fallback.  Tests and bench.py's cpu_baseline leg may install another object with
the same method surface via `set_backend` (e.g. the oracle-backed one in oracle/oracle_backend.py);
nothing in this package imports the oracle.
"""
import ctypes as C

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


class _Workspace:
    """grow-only device scratch, one per purpose (avoids hipMalloc on the hot path)"""

    def __init__(self):
        self.buf = {}

    def get(self, key, nbytes, device):
        b = self.buf.get((key, device))
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self.buf[(key, device)] = b
        return b


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.lib()  # raises HipLibraryError when the .so is absent
        self.ws = _Workspace()

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _dev(t):
        if not t.is_cuda:
            raise _lib.HipLibraryError("HipBackend needs device tensors (got a CPU tensor); "
                                       "the product path has no CPU fallback")
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
        ws_bytes = self.lib.ms3d_ballquery_workspace_bytes(n)
        ws = self.ws.get("bq", ws_bytes, dev)
        n_active = C.c_int(0)
        while True:
            idx = torch.empty(n * meanActive, dtype=torch.int32, device=dev)
            rc = self.lib.ms3d_ballquery_batch_p(
                n, int(meanActive), C.c_float(radius), _lib.ptr(coords), _lib.ptr(batch_idxs),
                _lib.ptr(batch_offsets), int(batch_offsets.numel() - 1), int(max_scene_points), _lib.ptr(idx),
                _lib.ptr(start_len), C.byref(n_active), _lib.ptr(ws), C.c_size_t(ws.numel()), _lib.stream_handle())
            _lib.check(rc, "ms3d_ballquery_batch_p")
            if n_active.value <= n * meanActive:
                break
            meanActive = int(n_active.value // n + 1)
        return idx[:n_active.value], start_len

    # ------------------------------------------------------------------ BFS
    def _bfs(self, fn_name, args_head, ball_idx, start_len, args_tail):
        ball_idx = self._dev(ball_idx); start_len = self._dev(start_len)
        N = start_len.size(0)
        dev = start_len.device
        cluster_idxs = torch.empty((max(N, 1), 2), dtype=torch.int32, device=dev)
        cluster_offsets = torch.empty(N + 1, dtype=torch.int32, device=dev)
        ws_bytes = self.lib.ms3d_bfs_workspace_bytes(N)
        ws = self.ws.get("bfs", ws_bytes, dev)
        counts = (C.c_int * 2)(0, 0)
        rc = getattr(self.lib, fn_name)(*args_head, _lib.ptr(ball_idx), _lib.ptr(start_len), N, *args_tail,
                                        _lib.ptr(cluster_idxs), _lib.ptr(cluster_offsets), counts, _lib.ptr(ws),
                                        C.c_size_t(ws.numel()), _lib.stream_handle())
        _lib.check(rc, fn_name)
        return cluster_idxs[:counts[1]], cluster_offsets[:counts[0] + 1]

    def pg_bfs_cluster(self, semantic_label, ball_query_idxs, start_len, threshold):
        sem = self._dev(semantic_label)
        assert sem.dtype == torch.int16
        return self._bfs("ms3d_pg_bfs_cluster", (_lib.ptr(sem),), ball_query_idxs, start_len, (int(threshold),))

    def sg_bfs_cluster(self, class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
        mean = (C.c_float * len(class_numpoint_mean))(*[float(x) for x in class_numpoint_mean])
        return self._bfs("ms3d_sg_bfs_cluster", (mean,), ball_query_idxs, start_len,
                         (C.c_float(threshold), int(class_id)))

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
        rc = self.lib.ms3d_global_avg_pool_bp(P, Cc, _lib.ptr(d_feats), _lib.ptr(offsets), _lib.ptr(d_out),
                                              _lib.stream_handle())
        _lib.check(rc, "ms3d_global_avg_pool_bp")
        return d_feats

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
