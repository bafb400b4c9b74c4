"""Predicted instances of a scan from the network's proposals (validation / test time).

Same outputs as the reference's `_get_pred_instances` (minsu3d/model/pointgroup.py:197-265, hais.py:210-247): a list of
{'scan_id', 'label_id', 'conf', 'pred_mask': RLE, 'pred_bbox': xyzxyz}.  The reference copies every proposal to the host
and works on dense [P, N] boolean masks (O(P*N) memory, a [P, P] float matrix product for the NMS); here the proposals
stay (cluster, point) pair lists on the device -- thresholds by bincount, cross intersections and the greedy NMS by
the two kernels of csrc/postprocess.hip -- and only the surviving instances' run boundaries and boxes reach the host.
"""
import numpy as np
import torch

from ..backend import get_backend


def _unique_pairs(proposals_idx, n_points, keep_rows=None):
    """(cluster, point) rows -> unique pairs sorted by (cluster, point) (a repeated pair is one mask bit)"""
    idx = proposals_idx.long()
    if keep_rows is not None:
        idx = idx[keep_rows]
    key = torch.unique(idx[:, 0] * n_points + idx[:, 1])     # sorted
    return key // n_points, key % n_points


def _instances(scan_id, xyz, sem_pred, cluster, point, order_ids, conf, num_ignored):
    """emit the instances `order_ids` (cluster ids, output order); cluster/point sorted by (cluster, point)"""
    if order_ids.numel() == 0:
        return []
    n_clusters = int(cluster.max().item()) + 1 if cluster.numel() else 0
    rank = torch.full((max(n_clusters, int(order_ids.max().item()) + 1),), -1, dtype=torch.long, device=cluster.device)
    rank[order_ids.long()] = torch.arange(order_ids.numel(), device=cluster.device)
    sel = rank[cluster] >= 0
    c, p = rank[cluster[sel]], point[sel]                      # output slot of every surviving pair
    o = torch.argsort(c * (int(point.max().item()) + 1) + p)  # by (slot, point)
    c, p = c[o], p[o]
    K = order_ids.numel()
    first = torch.ones_like(c, dtype=torch.bool)
    first[1:] = c[1:] != c[:-1]
    run_start = first.clone()
    run_start[1:] |= p[1:] != p[:-1] + 1                        # a run ends where the point index jumps
    starts = torch.nonzero(run_start).view(-1)
    run_len = torch.diff(torch.cat([starts, starts.new_tensor([c.numel()])]))
    pts = torch.as_tensor(xyz, device=c.device, dtype=torch.float32)[p]
    lo = torch.full((K, 3), float("inf"), device=c.device).scatter_reduce_(0, c[:, None].expand(-1, 3), pts, "amin")
    hi = torch.full((K, 3), float("-inf"), device=c.device).scatter_reduce_(0, c[:, None].expand(-1, 3), pts, "amax")
    label = sem_pred[p[first]]                                  # label of the lowest point index of each instance
    # host side: a handful of small arrays
    run_slot, run_first, run_len = c[starts].cpu().numpy(), p[starts].cpu().numpy() + 1, run_len.cpu().numpy()
    boxes = torch.cat([lo, hi], 1).cpu().numpy()
    label, conf = label.cpu().numpy(), conf.cpu().numpy()
    n_points = int(sem_pred.numel())
    cuts = np.searchsorted(run_slot, np.arange(K + 1))
    out = []
    for k in range(K):
        seg = slice(cuts[k], cuts[k + 1])
        counts = " ".join(f"{a} {b}" for a, b in zip(run_first[seg], run_len[seg]))
        out.append({"scan_id": scan_id, "label_id": int(label[k]) - num_ignored + 1, "conf": conf[k],
                    "pred_mask": {"length": n_points, "counts": counts}, "pred_bbox": boxes[k]})
    return out


def pointgroup_instances(scan_id, gt_xyz, proposals_scores, proposals_idx, num_proposals, semantic_scores,
                         num_ignored_classes, score_thresh, npoint_thresh, nms_thresh):
    """reference pointgroup.py:197-265 (TEST_SCORE_THRESH, TEST_NPOINT_THRESH, TEST_NMS_THRESH)"""
    be = get_backend()
    dev = semantic_scores.device
    n_points = semantic_scores.size(0)
    sem_pred = semantic_scores.argmax(1)
    conf = torch.sigmoid(proposals_scores.reshape(-1).float())
    cluster, point = _unique_pairs(proposals_idx.to(dev), n_points)
    npoint = torch.bincount(cluster, minlength=num_proposals)
    keep = (conf > score_thresh) & (npoint > npoint_thresh)
    kept_ids = torch.nonzero(keep).view(-1)
    if kept_ids.numel() == 0:
        return []
    remap = torch.full((num_proposals,), -1, dtype=torch.long, device=dev)
    remap[kept_ids] = torch.arange(kept_ids.numel(), device=dev)
    sel = remap[cluster] >= 0
    cluster, point = remap[cluster[sel]], point[sel]
    conf = conf[kept_ids]
    by_point = torch.argsort(point * kept_ids.numel() + cluster)
    inter = be.proposal_cross_intersection(point[by_point], cluster[by_point], kept_ids.numel())
    order = torch.sort(conf, descending=True, stable=True)[1]
    pick = be.nms_greedy(inter, order, float(nms_thresh)).long()
    return _instances(scan_id, gt_xyz, sem_pred, cluster, point, pick, conf[pick], num_ignored_classes)


def hais_instances(scan_id, gt_xyz, scores, proposals_idx, num_proposals, mask_scores, semantic_scores,
                   num_ignored_classes, mask_score_thresh, score_thresh, npoint_thresh):
    """reference hais.py:210-247 (test_mask_score_thre, TEST_SCORE_THRESH, TEST_NPOINT_THRESH; no NMS)"""
    dev = semantic_scores.device
    n_points = semantic_scores.size(0)
    sem_pred = semantic_scores.argmax(1)
    conf = torch.sigmoid(scores.reshape(-1).float())
    cluster, point = _unique_pairs(proposals_idx.to(dev), n_points, mask_scores.reshape(-1) > mask_score_thresh)
    npoint = torch.bincount(cluster, minlength=num_proposals)
    kept_ids = torch.nonzero((conf > score_thresh) & (npoint >= npoint_thresh)).view(-1)
    return _instances(scan_id, gt_xyz, sem_pred, cluster, point, kept_ids, conf[kept_ids], num_ignored_classes)


def softgroup_instances(scan_id, gt_xyz, proposals_idx, num_points, cls_scores, iou_scores, mask_scores,
                        num_ignored_classes, instance_classes, cls_score_thr, mask_score_thr, min_npoint):
    """reference softgroup.py:269-313: every proposal is scored for every instance class; class i keeps the proposals
    with softmax(cls)[:, i] > cls_score_thr whose class-i mask (mask_scores[:, i] > mask_score_thr) has >= min_npoint
    points; score = cls prob * clamp(iou score, 0, 1); output is class-major.  (num_ignored_classes is unused there too.)"""
    dev = cls_scores.device
    num_inst = cls_scores.size(0)
    probs = cls_scores.softmax(1)
    idx = proposals_idx.to(dev).long()
    out = []
    sem_dummy = torch.zeros(num_points, dtype=torch.long, device=dev)
    for i in range(instance_classes):
        conf = probs[:, i] * iou_scores[:, i].clamp(0, 1)
        cluster, point = _unique_pairs(idx, num_points, mask_scores[:, i] > mask_score_thr)
        npoint = torch.bincount(cluster, minlength=num_inst)
        kept = torch.nonzero((probs[:, i] > cls_score_thr) & (npoint >= min_npoint)).view(-1)
        inst = _instances(scan_id, gt_xyz, sem_dummy, cluster, point, kept, conf[kept], 0)
        for d in inst:
            d["label_id"] = i + 1
        out += inst
    return out
