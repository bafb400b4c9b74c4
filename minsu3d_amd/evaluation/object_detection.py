"""Axis-aligned box AP at IoU 0.25 / 0.5 (VoteNet protocol) -- same interface and results as the reference's
`minsu3d/evaluation/object_detection.py` (voc_ap :5-37, get_iou :40-64, eval_det_cls :71-150, eval_sphere :206-254,
get_gt_bbox :257-276, evaluate_bbox_acc :279-298).  Box IoUs of a detection against all boxes of its scan are
computed as one vectorised expression; the greedy first-match rule is the reference's.

One deliberate difference: the reference indexes its per-class results by the position of the class among ALL
ground-truth classes (:246-252), which raises IndexError as soon as a ground-truth class without predictions precedes
one with predictions; here such classes simply score 0."""
import numpy as np


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:
        return float(sum((np.max(prec[rec >= t]) if np.any(rec >= t) else 0.0) / 11.0 for t in np.arange(0.0, 1.1, 0.1)))
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]          # precision envelope
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def get_iou(box_a, box_b):
    """xyzxyz boxes -> IoU (0 unless the overlap is strictly positive along every axis)"""
    lo = np.maximum(box_a[:3], box_b[:3])
    hi = np.minimum(box_a[3:6], box_b[3:6])
    if not (hi > lo).all():
        return 0.0
    inter = (hi - lo).prod()
    return 1.0 * inter / ((box_a[3:6] - box_a[:3]).prod() + (box_b[3:6] - box_b[:3]).prod() - inter)


def _iou_one_to_many(box, boxes):
    lo = np.maximum(box[:3], boxes[:, :3])
    hi = np.minimum(box[3:6], boxes[:, 3:6])
    ok = (hi > lo).all(1)
    inter = np.where(ok, (hi - lo).prod(1), 0.0)
    union = (box[3:6] - box[:3]).prod() + (boxes[:, 3:6] - boxes[:, :3]).prod(1) - inter
    return np.where(ok, 1.0 * inter / np.where(ok, union, 1.0), 0.0)


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, get_iou_func=get_iou):
    """one class: pred {scan: [(box, score)]}, gt {scan: [box]} -> (rec, prec, ap)"""
    boxes = {k: np.array(v, dtype=np.float32).astype(float).reshape(-1, 6) for k, v in gt.items()}
    taken = {k: np.zeros(len(b), bool) for k, b in boxes.items()}
    npos = sum(len(b) for b in boxes.values())
    det_scan, det_conf, det_box = [], [], []
    for scan, items in pred.items():
        for box, score in items:
            det_scan.append(scan); det_conf.append(score); det_box.append(box)
    order = np.argsort(-np.array(det_conf), kind="stable")
    nd = len(det_scan)
    tp, fp = np.zeros(nd, bool), np.zeros(nd, bool)
    for d, i in enumerate(order):
        cand = boxes.get(det_scan[i])
        best, j = -np.inf, -1
        if cand is not None and len(cand):
            if get_iou_func is get_iou:
                ious = _iou_one_to_many(np.asarray(det_box[i]).astype(float), cand)
            else:
                ious = np.array([get_iou_func(np.asarray(det_box[i]).astype(float), c) for c in cand])
            j = int(np.argmax(ious))           # first maximum, as the strict `>` scan of the reference (:125-128)
            best = ious[j]
        if best > ovthresh and not taken[det_scan[i]][j]:
            tp[d] = True
            taken[det_scan[i]][j] = True
        else:
            fp[d] = True
    fp = np.cumsum(fp, dtype=np.uint32)
    tp = np.cumsum(tp, dtype=np.uint32)
    rec = tp.astype(np.float32) / npos
    prec = tp / np.maximum(tp + fp, np.finfo(np.float32).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def _by_class(pred_all, gt_all):
    pred, gt = {}, {}
    for scan, items in pred_all.items():
        for cls, box, score in items:
            pred.setdefault(cls, {}).setdefault(scan, []).append((box, score))
            gt.setdefault(cls, {}).setdefault(scan, [])
    for scan, items in gt_all.items():
        for cls, box in items:
            gt.setdefault(cls, {}).setdefault(scan, []).append(box)
    return pred, gt


def eval_sphere(pred_all, gt_all, ovthresh, use_07_metric=False, get_iou_func=get_iou):
    """pred_all {scan: [(class, box, score)]}, gt_all {scan: [(class, box)]} -> (rec, prec, ap) dicts by class"""
    pred, gt = _by_class(pred_all, gt_all)
    rec, prec, ap = {}, {}, {}
    for cls in gt:
        if cls in pred:
            rec[cls], prec[cls], ap[cls] = eval_det_cls(pred[cls], gt[cls], ovthresh, use_07_metric, get_iou_func)
        else:
            rec[cls] = prec[cls] = ap[cls] = 0
    return rec, prec, ap


eval_det = eval_sphere


def get_gt_bbox(xyz, instance_ids, sem_labels, ignored_label, ignore_classes):
    """[(class index after dropping the ignored classes, xyzxyz box)] per ground-truth instance, ascending instance id"""
    out = []
    order = np.argsort(instance_ids, kind="stable")
    ids = instance_ids[order]
    cuts = np.flatnonzero(np.diff(ids)) + 1
    for seg in np.split(order, cuts):
        if seg.size == 0 or instance_ids[seg[0]] == ignored_label:
            continue
        sem = sem_labels[seg[0]]                         # label of the instance's first point (reference :265)
        if sem + 1 in ignore_classes or sem == ignored_label:
            continue
        pts = xyz[seg]
        out.append((sem - len(ignore_classes), np.concatenate((pts.min(0), pts.max(0)))))
    return out


def evaluate_bbox_acc(all_preds, all_gts, class_names, ignored_classes_indicies, print_result):
    pred_all, gt_all = {}, {}
    for preds, gts in zip(all_preds, all_gts):
        scan = preds[0]["scan_id"]
        pred_all[scan] = [(p["label_id"] - 1, p["pred_bbox"], p["conf"]) for p in preds]
        gt_all[scan] = gts
    out = {}
    for th in (0.25, 0.5):
        ap = eval_sphere(pred_all, gt_all, ovthresh=th)[-1]
        ap["avg"] = np.mean(list(ap.values()))
        out[f"all_bbox_ap_{th}"] = ap
    if print_result:
        print_results(out, class_names, ignored_classes_indicies)
    return out


def print_results(bbox_aps, class_names, ignored_classes_indices):
    width = 46
    names = [c for i, c in enumerate(class_names) if i + 1 not in ignored_classes_indices]
    print()
    print("#" * width)
    print("{:<15}:".format("what") + "{:>15}{:>15}".format("BBox_AP_50%", "BBOX_AP_25%"))
    print("#" * width)
    for li, name in enumerate(names):
        print("{:<15}:".format(name) + "{:>15.3f}{:>15.3f}".format(bbox_aps["all_bbox_ap_0.5"][li], bbox_aps["all_bbox_ap_0.25"][li]))
    print("-" * width)
    print("{:<15}:".format("average") + "{:>15.3f}{:>15.3f}".format(bbox_aps["all_bbox_ap_0.5"]["avg"], bbox_aps["all_bbox_ap_0.25"]["avg"]))
    print("#" * width)
    print()
