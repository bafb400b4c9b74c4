"""`import COMMON_OPS` -- the 15 functions of the reference's pybind module
(minsu3d/common_ops/src/common_ops_api.cpp:6-30) with the reference's own positional signatures and ownership
rules, served by libminsu3d_hip.so through the C ABI (include/minsu3d_hip.h).  With this module on the path the
reference's wrappers (minsu3d/common_ops/functions/{common_ops,pointgroup_ops,hais_ops,softgroup_ops}.py) run
unchanged:

  * the caller allocates every output and passes it in; the callee writes in place (sec_mean.cpp:9-15, roipool.cpp,
    get_iou.cpp, cal_iou_and_masklabel.cpp);
  * the clustering functions `resize_()` the caller's (possibly empty) output tensors to the data-dependent sizes
    (bfs_cluster.cpp:157-160, hierarchical_aggregation.cpp:133-175) -- and accept CPU tensors, which is what
    model/pointgroup.py:49-52, hais.py:52-56 and softgroup.py:60-63 pass: inputs are copied to the GPU, the
    clustering runs there (there is no CPU implementation in the product), results are copied into the caller's tensors;
  * `ballquery_batch_p` returns the total hit count and leaves positions >= n*meanActive unwritten, so the wrapper's
    retry loop (functions/common_ops.py:31-38) behaves as with the reference (bfs_cluster.cu:51-58).

Kernels run on torch's current stream (the reference: ball query on the current stream, the rest on the legacy
default stream followed by device synchronisation)."""
import ctypes as C

import torch

from minsu3d_amd import _lib
from minsu3d_amd import backend as _backend
from minsu3d_amd.backend import get_backend

__all__ = ["sg_bfs_cluster", "global_avg_pool_fp", "global_avg_pool_bp", "ballquery_batch_p", "sec_mean", "sec_min",
           "sec_max", "roipool_fp", "roipool_bp", "get_iou", "get_mask_iou_on_cluster", "get_mask_iou_on_pred",
           "get_mask_label", "pg_bfs_cluster", "hierarchical_aggregation"]


def _be():
    return get_backend()     # raises HipLibraryError when libminsu3d_hip.so is missing: no fallback


def _assign(dst, src):
    """the callee-side `resize_` + fill of a caller-provided tensor (bfs_cluster.cpp:157-160)"""
    dst.resize_(src.shape)
    dst.copy_(src)


def _run(name, *args):
    _lib.check(getattr(_be().lib, name)(*args, _lib.stream_handle()), name)


def _d(t):
    return _be()._dev(t)


# ---------------------------------------------------------------------------------------------- common
def ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, meanActive, radius):
    """bfs_cluster.cpp:15-25 -> cumsum (int).  idx i32[n*meanActive], start_len i32[n,2] are written in place."""
    be = _be()
    assert idx.is_cuda and start_len.is_cuda and idx.is_contiguous() and start_len.is_contiguous()
    n = int(n)
    if n == 0:
        return 0
    xyz = _d(xyz); batch_idxs = _d(batch_idxs); batch_offsets = _d(batch_offsets)
    ws = be.ws.get("bq", be.lib.ms3d_ballquery_workspace_bytes(n), xyz.device)
    n_active, capped = C.c_int(0), C.c_int(0)
    _lib.check(be.lib.ms3d_ballquery_batch_p(
        n, int(meanActive), C.c_float(radius), _lib.ptr(xyz), _lib.ptr(batch_idxs), _lib.ptr(batch_offsets),
        int(batch_offsets.numel() - 1), 0, _lib.ptr(idx), _lib.ptr(start_len), C.byref(n_active), C.byref(capped),
        _lib.ptr(ws), C.c_size_t(ws.numel()), _lib.stream_handle()), "ms3d_ballquery_batch_p")
    _backend.GRAPHS.put(start_len, int(capped.value))
    _remember_graph(idx, start_len, int(n_active.value), int(capped.value))
    return int(n_active.value)


# ---- the reference's host round trip without the upload (OPT-IN: MS3D_DROPIN_REUSE=1) ----------------------------
# model/pointgroup.py:43-55 (hais.py:45-56, softgroup.py:54-63) moves the ball-query result to the host
# (`idx.cpu()`, `start_len.cpu()`: up to n * 300 * 4 B ~ 150 MB per call), clusters there and moves the clusters
# back.  Run unchanged against this module, the clustering call uploads that list again for a BFS that runs on the GPU
# anyway.  With MS3D_DROPIN_REUSE=1 the last two ball-query results are remembered ON THE DEVICE with a checksum over
# EVERY entry of both tensors (column sums weighted by position: any single edited entry changes it), computed on the
# device when the graph is made; a clustering call that is handed HOST tensors of the same sizes computes the same
# checksum over the host tensors and takes the device copies only when it agrees.  Off by default: the boundary's
# contract is "the callee clusters the tensors it is given", and what the shortcut saves (the 2.7 ms upload; the
# reference's own `.cpu()` of the list costs 28-40 ms and stays) does not justify deciding identity from anything
# less than the whole tensor.  A remembered graph is dropped once it has been used.
_GRAPHS = []          # newest first: dict(n_active, n, sum_idx, sum_sl, idx, start_len, capped, v_idx, v_sl)
_REUSE_HITS = [0, 0]  # (device copies taken, host tensors uploaded) -- read by tests / tools
_CK_COLS = 4096


def _reuse_enabled():
    import os
    return os.environ.get("MS3D_DROPIN_REUSE", "0") == "1"


def _checksum(t):
    """position-weighted 64-bit checksum over every entry of an integer tensor (same arithmetic on host and device;
    int64 wraps): entry i contributes t[i] * w[i mod 4096] with odd weights, so no single edit leaves it unchanged"""
    flat = t.reshape(-1)
    w = (torch.arange(1, _CK_COLS + 1, dtype=torch.int64, device=flat.device) * 2654435761) | 1
    m = flat.numel() // _CK_COLS
    total = torch.zeros((), dtype=torch.int64, device=flat.device)
    if m:
        total = total + (flat[:m * _CK_COLS].view(m, _CK_COLS).sum(0, dtype=torch.int64) * w).sum()
    tail = flat[m * _CK_COLS:]
    if tail.numel():
        total = total + (tail.to(torch.int64) * w[:tail.numel()]).sum()
    return total + flat.numel()


def _remember_graph(idx, start_len, n_active, capped):
    if not _reuse_enabled() or n_active > idx.numel():
        return
    sums = torch.stack((_checksum(idx[:n_active]), _checksum(start_len))).cpu()     # one 16-byte device->host read
    _GRAPHS.insert(0, dict(n_active=n_active, n=start_len.size(0), sum_idx=int(sums[0]), sum_sl=int(sums[1]), idx=idx,
                           start_len=start_len, capped=capped, v_idx=idx._version, v_sl=start_len._version))
    del _GRAPHS[2:]


def _device_graph(ball_query_idxs, start_len):
    """the device copies of a ball-query result handed in as HOST tensors, or the arguments themselves"""
    if ball_query_idxs.is_cuda or start_len.is_cuda or not _GRAPHS or not _reuse_enabled():
        return ball_query_idxs, start_len
    host_sums = None
    for k, g in enumerate(_GRAPHS):
        if (g["n_active"] == ball_query_idxs.numel() and g["n"] == start_len.size(0)
                and g["idx"]._version == g["v_idx"] and g["start_len"]._version == g["v_sl"]):   # not written since
            if host_sums is None:
                host_sums = (int(_checksum(ball_query_idxs)), int(_checksum(start_len)))
            if host_sums == (g["sum_idx"], g["sum_sl"]):
                _REUSE_HITS[0] += 1
                sl = g["start_len"]
                _backend.GRAPHS.put(sl, g["capped"])
                del _GRAPHS[k]                     # used once: the 150 MB list is not kept alive for another step
                return g["idx"][:g["n_active"]], sl
    _REUSE_HITS[1] += 1
    return ball_query_idxs, start_len


def _seg(name, inp, offsets, out, nProposal, C_):
    assert out.is_cuda and out.is_contiguous()
    _run(name, int(nProposal), int(C_), _lib.ptr(_d(inp)), _lib.ptr(_d(offsets)), _lib.ptr(out))


def sec_mean(inp, offsets, out, nProposal, C_):
    """sec_mean.cpp:9-15"""
    _seg("ms3d_sec_mean", inp, offsets, out, nProposal, C_)


def sec_min(inp, offsets, out, nProposal, C_):
    _seg("ms3d_sec_min", inp, offsets, out, nProposal, C_)


def sec_max(inp, offsets, out, nProposal, C_):
    _seg("ms3d_sec_max", inp, offsets, out, nProposal, C_)


def roipool_fp(feats, proposals_offset, output_feats, output_maxidx, nProposal, C_):
    """roipool.cpp: segment max + argmax"""
    _run("ms3d_roipool_fp", int(nProposal), int(C_), _lib.ptr(_d(feats)), _lib.ptr(_d(proposals_offset)),
         _lib.ptr(output_feats), _lib.ptr(output_maxidx))


def roipool_bp(d_feats, proposals_offset, output_maxidx, d_output_feats, nProposal, C_):
    _run("ms3d_roipool_bp", int(nProposal), int(C_), _lib.ptr(d_feats), _lib.ptr(_d(proposals_offset)),
         _lib.ptr(_d(output_maxidx)), _lib.ptr(_d(d_output_feats)))


def global_avg_pool_fp(feats, proposals_offset, output_feats, nProposal, C_):
    _run("ms3d_global_avg_pool_fp", int(nProposal), int(C_), _lib.ptr(_d(feats)), _lib.ptr(_d(proposals_offset)),
         _lib.ptr(output_feats))


def global_avg_pool_bp(d_feats, proposals_offset, d_output_feats, nProposal, C_):
    _run("ms3d_global_avg_pool_bp", int(nProposal), int(C_), _lib.ptr(d_feats), _lib.ptr(_d(proposals_offset)),
         _lib.ptr(_d(d_output_feats)))


def _iou(name, proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance,
         nProposal, *extra):
    _run(name, int(nInstance), int(nProposal), _lib.ptr(_d(proposals_idx)), _lib.ptr(_d(proposals_offset)),
         _lib.ptr(_d(instance_labels)), _lib.ptr(_d(instance_pointnum)), _lib.ptr(proposals_iou),
         *[_lib.ptr(_d(e)) for e in extra])


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance, nProposal):
    """get_iou.cpp"""
    _iou("ms3d_get_iou", proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou,
         nInstance, nProposal)


def get_mask_iou_on_cluster(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou,
                            nInstance, nProposal):
    _iou("ms3d_get_mask_iou_on_cluster", proposals_idx, proposals_offset, instance_labels, instance_pointnum,
         proposals_iou, nInstance, nProposal)


def get_mask_iou_on_pred(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance,
                         nProposal, mask_scores_sigmoid):
    _iou("ms3d_get_mask_iou_on_pred", proposals_idx, proposals_offset, instance_labels, instance_pointnum,
         proposals_iou, nInstance, nProposal, mask_scores_sigmoid)


def get_mask_label(proposals_idx, proposals_offset, instance_labels, instance_cls, proposals_iou, nInstance, nProposal,
                   ignored_label, iou_thr, mask_labels, mask_labels_mask):
    """cal_iou_and_masklabel.cpp: mask_labels / mask_labels_mask (bool[sumNPoint]) written in place"""
    _run("ms3d_get_mask_label", int(nInstance), int(nProposal), int(ignored_label), C.c_float(iou_thr),
         _lib.ptr(_d(proposals_idx)), _lib.ptr(_d(proposals_offset)), _lib.ptr(_d(instance_labels)),
         _lib.ptr(_d(instance_cls)), _lib.ptr(_d(proposals_iou)), _lib.ptr(mask_labels), _lib.ptr(mask_labels_mask))


# ---------------------------------------------------------------------------------------------- clustering
def pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N, threshold):
    """bfs_cluster.cpp:140-166: cluster_idxs -> i32[sumNPoint,2], cluster_offsets -> i32[nCluster+1] (resized)"""
    assert int(N) == start_len.size(0)
    ball_query_idxs, start_len = _device_graph(ball_query_idxs, start_len)
    idxs, offs = _be().pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, int(threshold))
    _assign(cluster_idxs, idxs)
    _assign(cluster_offsets, offs)


def sg_bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N, threshold,
                   class_id):
    """bfs_cluster.cpp:168-187; class_numpoint_mean is a float32 CPU tensor (functions/softgroup_ops.py:23)"""
    assert int(N) == start_len.size(0)
    ball_query_idxs, start_len = _device_graph(ball_query_idxs, start_len)
    idxs, offs = _be().sg_bfs_cluster([float(v) for v in class_numpoint_mean.tolist()], ball_query_idxs, start_len,
                                      float(threshold), int(class_id))
    _assign(cluster_idxs, idxs)
    _assign(cluster_offsets, offs)


def hierarchical_aggregation(semantic_label, coord_shift, batch_idxs, ball_query_idxs, start_len,
                             fragment_idxs, fragment_offsets, fragment_centers,
                             cluster_idxs_kept, cluster_offsets_kept, cluster_centers_kept,
                             primary_idxs, primary_offsets, primary_centers,
                             primary_idxs_post, primary_offsets_post,
                             point_num_avg, radius_avg, N, using_set_aggr_, ignored_label):
    """hierarchical_aggregation.cpp:105-184.  Without set aggregation only the kept / primary lists are produced
    (the early return at .cpp:146-148); with it also all fragments and the post-aggregation primaries (zero tail
    beyond primary_offsets_post[-1], cut by the wrapper, functions/hais_ops.py:60-63).  Absorbed fragments are appended
    in ascending fragment index (the reference's order depends on atomics)."""
    assert int(N) == start_len.size(0)
    del ignored_label   # only the initial value of cc.cls_label, overwritten by the seed's label (.cpp:12-18)
    ball_query_idxs, start_len = _device_graph(ball_query_idxs, start_len)
    parts = _be().hierarchical_aggregation_parts(semantic_label, coord_shift, ball_query_idxs, start_len, batch_idxs,
                                                 bool(using_set_aggr_), point_num_avg.tolist(), radius_avg.tolist())
    for dst, src in zip((cluster_idxs_kept, cluster_offsets_kept, cluster_centers_kept), parts["kept"]):
        _assign(dst, src)
    for dst, src in zip((primary_idxs, primary_offsets, primary_centers), parts["primary"]):
        _assign(dst, src)
    if not using_set_aggr_:
        return
    for dst, src in zip((fragment_idxs, fragment_offsets, fragment_centers), parts["fragment"]):
        _assign(dst, src)
    for dst, src in zip((primary_idxs_post, primary_offsets_post), parts["post"]):
        _assign(dst, src)
