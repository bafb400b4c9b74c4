"""`minsu3d/common_ops/functions/pointgroup_ops.py:6-36` counterpart.  Unlike the reference (CPU tensors,
serial host BFS) the clustering runs on the device.  Device tensors in -> device tensors out (no PCIe trip);
CPU tensors in (what the reference's model code passes, model/pointgroup.py:49-52) -> copied to the GPU, clustered
there, results returned on the CPU like the reference's."""
import torch

from ...backend import get_backend


def pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """-> (cluster_idxs i32[sumNPoint,2] (cluster_id, point), cluster_offsets i32[nCluster+1])"""
    with torch.no_grad():
        out = get_backend().pg_bfs_cluster(semantic_label, ball_query_idxs, start_len, int(threshold))
    return out if start_len.is_cuda else tuple(t.cpu() for t in out)
