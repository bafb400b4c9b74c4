"""`minsu3d/common_ops/functions/hais_ops.py:6-79` counterpart."""
import torch

from ...backend import get_backend


def hierarchical_aggregation(semantic_label, coord_shift, ball_query_idxs, start_len, batch_idxs, using_set_aggr,
                             point_num_avg, radius_avg, ignored_label):
    """-> (cluster_idxs i32[S,2], cluster_offsets i32[P+1]); kept fragments first, then primaries"""
    with torch.no_grad():
        out = get_backend().hierarchical_aggregation(semantic_label, coord_shift.detach(), ball_query_idxs,
                                                     start_len, batch_idxs, bool(using_set_aggr), point_num_avg,
                                                     radius_avg, int(ignored_label))
    # CPU tensors in (the reference's call, model/hais.py:52-56) -> results on the CPU like the reference's
    return out if start_len.is_cuda else tuple(t.cpu() for t in out)
