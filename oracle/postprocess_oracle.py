"""CPU restatement of the reference's instance post-processing -- TEST INFRASTRUCTURE ONLY.

Follows minsu3d/model/pointgroup.py:177-265 (_get_nms_instances, _get_pred_instances), hais.py:210-247 and
softgroup.py:269-313 with dense [P, N] boolean masks, exactly the formulation the reference uses.  PINNED: the reference's
own methods were run on seeded proposal sets in the build container (tests/golden/make_golden_model.py registers
container-only stand-ins for pytorch_lightning / hydra and this repository's MinkowskiEngine / COMMON_OPS modules so that
minsu3d.model imports) and their instance lists are the fixtures tests/golden/model_tree.json + model_cases.npz;
tests/test_reference_pins_cpu.py::test_postprocess_oracle_is_pinned compares this file with them."""
import numpy as np


def rle(mask):
    m = np.concatenate([[0], np.asarray(mask, np.int64), [0]])
    runs = np.where(m[1:] != m[:-1])[0] + 1
    runs[1::2] -= runs[::2]
    return {"length": int(len(mask)), "counts": " ".join(str(x) for x in runs)}


def cross_intersection(pair_point, pair_cluster, P):
    """dense restatement: masks @ masks.T (pointgroup.py:237-238)"""
    n = int(pair_point.max()) + 1 if pair_point.size else 1
    masks = np.zeros((P, n), np.float32)
    masks[pair_cluster, pair_point] = 1
    return (masks @ masks.T).astype(np.int32)


def nms_from_counts(inter, order, threshold):
    """pointgroup.py:177-195 on the IoU matrix of :239-243, walking `order` instead of argsort(-scores)"""
    inter = inter.astype(np.float32)
    n = np.diag(inter)
    iou = inter / (n[:, None] + n[None, :] - inter)
    ixs = np.array(order, np.int64)
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        remove = np.where(iou[i, ixs[1:]] > threshold)[0] + 1
        ixs = np.delete(ixs, remove)
        ixs = np.delete(ixs, 0)
    return np.array(pick, np.int32)


def pointgroup_instances(scan_id, xyz, scores, proposals_idx, num_proposals, semantic_scores, num_ignored, score_thr,
                         npoint_thr, nms_thr):
    """pointgroup.py:197-265 with dense masks (numpy)"""
    sem = semantic_scores.argmax(1)
    conf = (1.0 / (1.0 + np.exp(-scores.reshape(-1).astype(np.float64)))).astype(np.float32)
    N = semantic_scores.shape[0]
    masks = np.zeros((num_proposals, N), bool)
    masks[proposals_idx[:, 0], proposals_idx[:, 1]] = True
    keep = (conf > score_thr) & (masks.sum(1) > npoint_thr)
    conf, masks = conf[keep], masks[keep]
    if conf.shape[0] == 0:
        return []
    f = masks.astype(np.float32)
    inter = f @ f.T
    pick = nms_from_counts(inter, np.argsort(-conf, kind="stable"), nms_thr)
    out = []
    for i in pick:
        m = masks[i]
        pts = xyz[m]
        out.append({"scan_id": scan_id, "label_id": int(sem[m][0]) - num_ignored + 1, "conf": conf[i], "pred_mask": rle(m),
                    "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
    return out


def hais_instances(scan_id, xyz, scores, proposals_idx, num_proposals, mask_scores, semantic_scores, num_ignored,
                   mask_thr, score_thr, npoint_thr):
    """hais.py:210-247"""
    sem = semantic_scores.argmax(1)
    conf = (1.0 / (1.0 + np.exp(-scores.reshape(-1).astype(np.float64)))).astype(np.float32)
    N = semantic_scores.shape[0]
    masks = np.zeros((num_proposals, N), bool)
    ok = mask_scores.reshape(-1) > mask_thr
    masks[proposals_idx[ok, 0], proposals_idx[ok, 1]] = True
    keep = conf > score_thr
    conf, masks = conf[keep], masks[keep]
    keep = masks.sum(1) >= npoint_thr
    conf, masks = conf[keep], masks[keep]
    out = []
    for i in range(conf.shape[0]):
        m = masks[i]
        pts = xyz[m]
        out.append({"scan_id": scan_id, "label_id": int(sem[m][0]) - num_ignored + 1, "conf": conf[i], "pred_mask": rle(m),
                    "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
    return out


def softgroup_instances(scan_id, xyz, proposals_idx, num_points, cls_scores, iou_scores, mask_scores, instance_classes,
                        cls_score_thr, mask_score_thr, min_npoint):
    """softgroup.py:269-313"""
    e = np.exp(cls_scores - cls_scores.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    num_inst = cls_scores.shape[0]
    out = []
    for i in range(instance_classes):
        conf = probs[:, i] * np.clip(iou_scores[:, i], 0, 1)
        masks = np.zeros((num_inst, num_points), bool)
        ok = mask_scores[:, i] > mask_score_thr
        masks[proposals_idx[ok, 0], proposals_idx[ok, 1]] = True
        keep = probs[:, i] > cls_score_thr
        conf_k, masks_k = conf[keep], masks[keep]
        keep2 = masks_k.sum(1) >= min_npoint
        conf_k, masks_k = conf_k[keep2], masks_k[keep2]
        for j in range(conf_k.shape[0]):
            m = masks_k[j]
            pts = xyz[m]
            out.append({"scan_id": scan_id, "label_id": i + 1, "conf": conf_k[j], "pred_mask": rle(m),
                        "pred_bbox": np.concatenate((pts.min(0), pts.max(0)))})
    return out
