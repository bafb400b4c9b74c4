"""ScanNet-protocol instance-segmentation evaluation (AP / AP50 / AP25 and recalls).

Same interface and results as the reference's `minsu3d/evaluation/instance_segmentation.py` (rle_encode :10-24,
rle_decode :27-44, get_gt_instances :62-75, GeneralDatasetEvaluator :103-422), restated on arrays: per scan ONE
[predictions x ground-truth instances] intersection matrix from a single bincount over the mask points replaces the
reference's Python loop of `count_nonzero(logical_and(...))` per (prediction, instance) pair, and the matching /
precision-recall code works on flat arrays instead of nested dictionaries.  The greedy matching rules, the "ignore"
bookkeeping and the AP integration are the reference's (file:line cited at each step).
"""
import numpy as np


def rle_encode(mask):
    """1-D binary mask -> {'length', 'counts': 'start run start run ...'} with 1-based starts (reference :10-24)"""
    m = np.asarray(mask).astype(np.int8).ravel()
    edges = np.flatnonzero(np.diff(np.concatenate(([0], m, [0])))) + 1   # 1-based positions where the value flips
    starts, ends = edges[0::2], edges[1::2]
    pairs = np.stack([starts, ends - starts], 1).ravel()
    return {"length": int(m.shape[0]), "counts": " ".join(str(int(v)) for v in pairs)}


def rle_decode(rle):
    """inverse of rle_encode -> uint8 mask (reference :27-44)"""
    vals = np.array(rle["counts"].split(), dtype=np.int64)
    mask = np.zeros(int(rle["length"]) + 1, dtype=np.int32)
    if vals.size:
        starts, runs = vals[0::2] - 1, vals[1::2]
        np.add.at(mask, starts, 1)
        np.add.at(mask, starts + runs, -1)
    return (np.cumsum(mask[:-1]) > 0).astype(np.uint8)


def rle_indices(rle):
    """sorted member indices of an RLE mask without materialising the dense mask"""
    vals = np.array(rle["counts"].split(), dtype=np.int64)
    if vals.size == 0:
        return np.zeros(0, np.int64)
    starts, runs = vals[0::2] - 1, vals[1::2]
    base = np.repeat(starts - np.concatenate(([0], np.cumsum(runs)[:-1])), runs)
    return base + np.arange(int(runs.sum()))


def get_gt_instances(semantic_labels, instance_labels, ignored_classes):
    """per-point ground-truth id = class * 1000 + instance (0 = ignore), classes renumbered 1..C after dropping the
    ignored ones (reference :62-75).  Like the reference, `instance_labels` is shifted by one IN PLACE."""
    shift = len(ignored_classes)
    sem = semantic_labels - shift + 1
    sem[sem < 0] = 0
    instance_labels += 1
    ids = sem * 1000 + instance_labels
    ids[instance_labels <= 0] = 0
    return ids


class GeneralDatasetEvaluator:
    """evaluate(pred_list, gt_list, print_result) -> {'all_ap', 'all_ap_50%', 'all_ap_25%', 'all_rc', ..., 'classes': {...}}

    pred_list[i]: list of {'scan_id', 'label_id', 'conf', 'pred_mask': rle}; gt_list[i]: per-point ids of get_gt_instances.
    """

    def __init__(self, class_labels, ignored_label, ignored_classes_indices, iou_type=None, use_label=True):
        self.valid_class_labels = [c for i, c in enumerate(class_labels) if i + 1 not in ignored_classes_indices]
        self.ignored_label = ignored_label
        self.valid_class_ids = np.arange(len(self.valid_class_labels)) + 1
        self.id2label = {int(i): c for i, c in zip(self.valid_class_ids, self.valid_class_labels)}
        self.label2id = {c: int(i) for i, c in zip(self.valid_class_ids, self.valid_class_labels)}
        self.ious = np.append(np.arange(0.5, 0.95, 0.05), 0.25)      # reference :115
        self.min_region_sizes = np.array([100])
        self.distance_threshes = np.array([float("inf")])
        self.distance_confs = np.array([-float("inf")])
        self.iou_type = iou_type
        self.use_label = use_label
        self.eval_class_labels = self.valid_class_labels if use_label else ["class_agnostic"]

    # ------------------------------------------------------------------ per scan: predictions x instances
    def _scan_tables(self, preds, gts):
        """-> dict of arrays for one scan (reference assign_instances_for_scan :306-383):
        gt_id, gt_class (index into eval_class_labels), gt_count;  pred_class, pred_conf, pred_count, pred_void;
        inter [P, G] restricted to same-class pairs (the reference only looks at instances of the prediction's class)."""
        gts = np.asarray(gts).astype(np.int64).ravel()
        n_cls = len(self.eval_class_labels)
        ids, counts = np.unique(gts, return_counts=True)
        ok = (ids != 0) & np.isin(ids // 1000, self.valid_class_ids)           # :47-59 (id 0 skipped, valid classes only)
        gt_id, gt_count = ids[ok], counts[ok]
        gt_class = (gt_id // 1000 - 1) if self.use_label else np.zeros(gt_id.size, np.int64)
        gt_index = np.full(int(gts.max(initial=0)) + 1, -1, np.int64)          # instance id -> column
        gt_index[gt_id] = np.arange(gt_id.size)
        void = ~np.isin(gts // 1000, self.valid_class_ids)                     # :333
        pc, conf, cnt, pvoid, rows, cols = [], [], [], [], [], []
        for pred in preds:
            if self.use_label:
                if int(pred["label_id"]) not in self.id2label:                 # :338-339
                    continue
                cls = int(pred["label_id"]) - 1
            else:
                cls = 0
            members = rle_indices(pred["pred_mask"])
            if members.size < self.min_region_sizes[0]:                        # :350-351
                continue
            g = gt_index[gts[members]]
            g = g[g >= 0]
            g = g[gt_class[g] == cls]
            rows.append(np.full(g.size, len(pc), np.int64))
            cols.append(g)
            pc.append(cls); conf.append(pred["conf"]); cnt.append(members.size)
            pvoid.append(int(np.count_nonzero(void[members])))
        P, G = len(pc), gt_id.size
        inter = np.zeros((P, G), np.int64)
        if P and G and rows:
            flat = np.concatenate(rows) * G + np.concatenate(cols)
            inter = np.bincount(flat, minlength=P * G).reshape(P, G)
        return dict(gt_id=gt_id, gt_class=gt_class, gt_count=gt_count, pred_class=np.array(pc, np.int64),
                    pred_conf=np.array(conf, np.float64), pred_count=np.array(cnt, np.int64),
                    pred_void=np.array(pvoid, np.int64), inter=inter, n_cls=n_cls)

    # ------------------------------------------------------------------ AP of one (class, IoU threshold)
    @staticmethod
    def _average_precision(y_true, y_score, hard_fn):
        """precision-recall integration of the reference (:239-285) -> (ap, recall at the lowest threshold)"""
        order = np.argsort(y_score, kind="stable")
        score, true = y_score[order], y_true[order]
        cum = np.cumsum(true)
        _, first = np.unique(score, return_index=True)
        n_ex = score.size
        n_true = cum[-1] if cum.size else 0
        cum = np.append(cum, 0)                    # index -1 -> 0 (":258 deal with the first point")
        below = cum[first - 1]
        tp = n_true - below
        fp = n_ex - first - tp
        fn = below + hard_fn
        precision = np.append(tp / (tp + fp), 1.0)
        recall = np.append(tp / (tp + fn), 0.0)
        rc_current = recall[0]
        r = np.concatenate(([recall[0]], recall, [0.0]))
        widths = np.convolve(r, [-0.5, 0, 0.5], "valid")
        return float(np.dot(precision, widths)), float(rc_current)

    def evaluate_matches(self, scans):
        """scans: list of _scan_tables dicts -> (ap, rc) float32 [1, classes, ious] (reference :127-292)"""
        n_cls = len(self.eval_class_labels)
        ap = np.zeros((1, n_cls, len(self.ious)), np.float32)
        rc = np.zeros((1, n_cls, len(self.ious)), np.float32)
        min_size = self.min_region_sizes[0]
        for oi, th in enumerate(self.ious):
            visited = [np.zeros(s["pred_class"].size, bool) for s in scans]     # reset per threshold (:139-145)
            for li in range(n_cls):
                y_true, y_score = [], []
                hard_fn, has_gt, has_pred = 0, False, False
                for s, seen in zip(scans, visited):
                    gsel = np.flatnonzero((s["gt_class"] == li) & (s["gt_id"] >= 1000) & (s["gt_count"] >= min_size))
                    psel = np.flatnonzero(s["pred_class"] == li)
                    has_gt |= gsel.size > 0
                    has_pred |= psel.size > 0
                    inter = s["inter"]
                    union = s["pred_count"][:, None] + s["gt_count"][None, :] - inter
                    iou = np.where(inter > 0, inter / np.maximum(union, 1), 0.0)
                    cur_true, cur_score = [], []
                    for g in gsel:                                               # greedy assignment (:169-194)
                        best, matched = -np.inf, False
                        for p in psel[inter[psel, g] > 0]:                       # predictions in list order
                            if seen[p] or not iou[p, g] > th:
                                continue
                            c = s["pred_conf"][p]
                            if matched:                                          # second hit on the same instance:
                                lo, best = min(best, c), max(best, c)            # the lower score becomes a false positive
                                cur_true.append(0); cur_score.append(lo)
                            else:
                                matched, best = True, c
                                seen[p] = True
                        if matched:
                            cur_true.append(1); cur_score.append(best)
                        else:
                            hard_fn += 1
                    # predictions without any instance above the threshold (:200-222)
                    small = s["gt_count"] < min_size
                    group = s["gt_id"] < 1000
                    for p in psel:
                        row_i = inter[p]
                        hit = (row_i > 0) & (s["gt_class"] == li)
                        if np.any(iou[p, hit] > th):
                            continue
                        ignore = s["pred_void"][p] + row_i[hit & group].sum() + row_i[hit & small].sum()
                        if float(ignore) / s["pred_count"][p] <= th:
                            cur_true.append(0); cur_score.append(s["pred_conf"][p])
                    y_true += cur_true
                    y_score += cur_score
                if has_gt and has_pred:
                    a, r = self._average_precision(np.array(y_true, np.float64), np.array(y_score, np.float64), hard_fn)
                elif has_gt:
                    a, r = 0.0, 0.0
                else:
                    a, r = float("nan"), float("nan")
                ap[0, li, oi], rc[0, li, oi] = a, r
        return ap, rc

    def compute_averages(self, aps, rcs):
        """(reference :294-316)"""
        o50 = np.where(np.isclose(self.ious, 0.5))
        o25 = np.where(np.isclose(self.ious, 0.25))
        rest = np.where(np.logical_not(np.isclose(self.ious, 0.25)))
        out = {"all_ap": np.nanmean(aps[0, :, rest]), "all_ap_50%": np.nanmean(aps[0, :, o50]),
               "all_ap_25%": np.nanmean(aps[0, :, o25]), "all_rc": np.nanmean(rcs[0, :, rest]),
               "all_rc_50%": np.nanmean(rcs[0, :, o50]), "all_rc_25%": np.nanmean(rcs[0, :, o25]), "classes": {}}
        for li, name in enumerate(self.eval_class_labels):
            out["classes"][name] = {"ap": np.average(aps[0, li, rest]), "ap50%": np.average(aps[0, li, o50]),
                                    "ap25%": np.average(aps[0, li, o25]), "rc": np.average(rcs[0, li, rest]),
                                    "rc50%": np.average(rcs[0, li, o50]), "rc25%": np.average(rcs[0, li, o25])}
        return out

    def evaluate(self, pred_list, gt_list, print_result):
        assert len(pred_list) == len(gt_list)
        scans = [self._scan_tables(p, g) for p, g in zip(pred_list, gt_list)]
        ap, rc = self.evaluate_matches(scans)
        avgs = self.compute_averages(ap, rc)
        if print_result:
            self.print_results(avgs)
        return avgs

    def print_results(self, avgs):
        """the reference's table (:424-476): per class and average AP / AP_50% / AP_25% / AR / RC_50% / RC_25%"""
        keys = ("ap", "ap50%", "ap25%", "rc", "rc50%", "rc25%")
        width = 64
        print()
        print("#" * width)
        print("{:<15}:".format("what") + "".join("{:>8}".format(h) for h in ("AP", "AP_50%", "AP_25%", "AR", "RC_50%", "RC_25%")))
        print("#" * width)
        for name in self.eval_class_labels:
            print("{:<15}:".format(name) + "".join("{:>8.3f}".format(avgs["classes"][name][k]) for k in keys))
        print("-" * width)
        print("{:<15}:".format("average") + "".join("{:>8.3f}".format(avgs[k]) for k in
                                                    ("all_ap", "all_ap_50%", "all_ap_25%", "all_rc", "all_rc_50%", "all_rc_25%")))
        print("#" * width)
        print()
