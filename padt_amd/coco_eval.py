"""COCO bbox mAP for OVD (BASELINE configs[3]) — what eval/evaluation_scripts/eval_coco.py:78-93 gets from pycocotools' COCOeval, written
from the published definition of the metric because pycocotools is not part of this environment:

  * per image and category, detections sorted by score (stable), at most maxDets of them; IoU of xywh boxes as the COCO mask API's bbIou
    does it (a CROWD ground truth absorbs detections with IoU = intersection / detection area, and may be matched any number of times);
  * greedy matching per IoU threshold .50:.05:.95 — a detection takes the best still-free ground truth with IoU >= threshold, preferring
    regular over ignored ones (ignored = crowd or outside the area range; ground truths are visited regular-first);
    unmatched detections outside the area range are ignored as well;
  * per category / area range / maxDets: detections of all images merged by score (stable), cumulative TP / FP, precision made monotone
    from the right, sampled at the 101 recall thresholds 0:.01:1 (searchsorted side='left'); -1 where a cell has no regular ground truth;
  * the 12 summary numbers of COCOeval.summarize(): AP, AP50, AP75, AP small / medium / large (maxDets 100), AR@1, AR@10, AR@100,
    AR small / medium / large — each the mean over the cells that are not -1.

`score_coco` is the assembly of eval_coco.py:21-76 around it: category NAME → id (predictions with an unknown name are dropped, :69-76),
ground-truth boxes from the dataset's normalised (x1, y1, x2, y2) rounded to pixel xywh (:55), annotation ids from 1.
"""
from collections import defaultdict
from typing import Dict, Iterable, List, Sequence

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))      # all, small, medium, large
AREA_LBL = ("all", "small", "medium", "large")


def bbox_iou_matrix(dt: np.ndarray, gt: np.ndarray, iscrowd: Sequence[int]) -> np.ndarray:
    """(D, 4), (G, 4) xywh → (D, G) IoU; column g of a crowd ground truth uses the detection's area as the union."""
    D, G = len(dt), len(gt)
    out = np.zeros((D, G), dtype=np.float64)
    if D == 0 or G == 0:
        return out
    dt = np.asarray(dt, dtype=np.float64).reshape(D, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(G, 4)
    da = dt[:, 2] * dt[:, 3]
    ga = gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    crowd = np.asarray(iscrowd, dtype=bool)[None, :]
    union = np.where(crowd, da[:, None], da[:, None] + ga[None, :] - inter)
    np.divide(inter, union, out=out, where=inter > 0)
    return out


def _evaluate_img(gts: List[Dict], dts: List[Dict], ious_sorted_dt: np.ndarray, area_rng, max_det: int):
    """One (image, category, area range, maxDets) cell → matches per IoU threshold.  `dts` are already sorted by score (desc, stable)
    and `ious_sorted_dt` rows follow that order; its columns follow `gts`."""
    if not gts and not dts:
        return None
    g_ign = np.array([1 if (g["ignore"] or g["area"] < area_rng[0] or g["area"] > area_rng[1]) else 0 for g in gts], dtype=np.int64)
    gtind = np.argsort(g_ign, kind="mergesort")                       # regular ground truths first
    g_ign = g_ign[gtind]
    iscrowd = [int(gts[i]["iscrowd"]) for i in gtind]
    dts = dts[:max_det]
    D, G, T = len(dts), len(gtind), len(IOU_THRS)
    ious = ious_sorted_dt[:D][:, gtind] if G and D else np.zeros((D, G))
    gtm = np.zeros((T, G), dtype=bool)
    dtm = np.zeros((T, D), dtype=bool)
    dt_ign = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(IOU_THRS):
        for di in range(D):
            best = min(t, 1 - 1e-10)
            m = -1
            for gi in range(G):
                if gtm[ti, gi] and not iscrowd[gi]:
                    continue
                if m > -1 and g_ign[m] == 0 and g_ign[gi] == 1:       # a regular match exists and only ignored ones remain
                    break
                if ious[di, gi] < best:
                    continue
                best = ious[di, gi]
                m = gi
            if m == -1:
                continue
            dt_ign[ti, di] = bool(g_ign[m])
            dtm[ti, di] = True
            gtm[ti, m] = True
    out_of_range = np.array([d["area"] < area_rng[0] or d["area"] > area_rng[1] for d in dts], dtype=bool).reshape(1, D)
    dt_ign = dt_ign | (~dtm & np.repeat(out_of_range, T, 0))
    return {"dtm": dtm, "dt_ign": dt_ign, "g_ign": g_ign, "scores": np.array([d["score"] for d in dts], dtype=np.float64)}


def coco_eval_bbox(gt_anns: Iterable[Dict], dt_anns: Iterable[Dict], img_ids: Sequence[int], cat_ids: Sequence[int]) -> Dict:
    """gt_anns: {"image_id", "category_id", "bbox" xywh, "area", "iscrowd"}; dt_anns: {"image_id", "category_id", "bbox" xywh, "score"}
    (detection area = w * h, as COCO.loadRes sets it).  → {"stats": 12 floats, "precision": (T, R, K, A, M), "recall": (T, K, A, M)}."""
    img_ids, cat_ids = sorted(set(img_ids)), sorted(set(cat_ids))
    G = defaultdict(list)
    Dd = defaultdict(list)
    for g in gt_anns:
        g = dict(g)
        g["iscrowd"] = int(g.get("iscrowd", 0))
        g["ignore"] = g["iscrowd"]
        G[g["image_id"], g["category_id"]].append(g)
    for d in dt_anns:
        d = dict(d)
        d["area"] = float(d["bbox"][2]) * float(d["bbox"][3])
        Dd[d["image_id"], d["category_id"]].append(d)
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(cat_ids), len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for ki, cat in enumerate(cat_ids):
        cells = []                                                    # per image: (gts, dts sorted, ious)
        for img in img_ids:
            gts, dts = G.get((img, cat), []), Dd.get((img, cat), [])
            if not gts and not dts:
                continue
            order = np.argsort([-d["score"] for d in dts], kind="mergesort")
            dts = [dts[i] for i in order][: MAX_DETS[-1]]
            ious = bbox_iou_matrix(np.array([d["bbox"] for d in dts], dtype=np.float64).reshape(-1, 4),
                                   np.array([g["bbox"] for g in gts], dtype=np.float64).reshape(-1, 4), [g["iscrowd"] for g in gts])
            cells.append((gts, dts, ious))
        for ai, rng in enumerate(AREA_RNG):
            for mi, max_det in enumerate(MAX_DETS):
                E = [e for e in (_evaluate_img(g, d, i, rng, max_det) for g, d, i in cells) if e is not None]
                if not E:
                    continue
                scores = np.concatenate([e["scores"] for e in E])
                inds = np.argsort(-scores, kind="mergesort")
                dtm = np.concatenate([e["dtm"] for e in E], axis=1)[:, inds]
                dig = np.concatenate([e["dt_ign"] for e in E], axis=1)[:, inds]
                npig = int(np.count_nonzero(np.concatenate([e["g_ign"] for e in E]) == 0))
                if npig == 0:
                    continue
                tp_sum = np.cumsum(dtm & ~dig, axis=1).astype(np.float64)
                fp_sum = np.cumsum(~dtm & ~dig, axis=1).astype(np.float64)
                for ti in range(T):
                    tp, fp = tp_sum[ti], fp_sum[ti]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[ti, ki, ai, mi] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):                    # precision envelope
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros(R)
                    pos = np.searchsorted(rc, REC_THRS, side="left")
                    for ri, pi in enumerate(pos):
                        if pi >= nd:
                            break                                     # recall level never reached: precision 0 from here on
                        q[ri] = pr[pi]
                    precision[:, :, ki, ai, mi][ti] = q

    def summarize(ap, iou_thr=None, area="all", max_dets=100):
        ai, mi = AREA_LBL.index(area), MAX_DETS.index(max_dets)
        s = precision if ap else recall
        if iou_thr is not None:
            s = s[np.where(np.isclose(IOU_THRS, iou_thr))[0]]
        s = s[:, :, :, ai, mi] if ap else s[:, :, ai, mi]
        return float(np.mean(s[s > -1])) if (s > -1).any() else -1.0

    stats = [summarize(1), summarize(1, 0.5), summarize(1, 0.75), summarize(1, area="small"), summarize(1, area="medium"),
             summarize(1, area="large"), summarize(0, max_dets=1), summarize(0, max_dets=10), summarize(0, max_dets=100),
             summarize(0, area="small"), summarize(0, area="medium"), summarize(0, area="large")]
    return {"stats": stats, "precision": precision, "recall": recall, "img_ids": img_ids, "cat_ids": cat_ids}


STAT_NAMES = ("AP", "AP50", "AP75", "AP_small", "AP_medium", "AP_large", "AR@1", "AR@10", "AR@100", "AR_small", "AR_medium", "AR_large")


def score_coco(preds: Iterable[Dict], data_items: Sequence[Dict], categories: Sequence[Dict], images: Sequence[Dict]) -> Dict:
    """eval_coco.py:21-93.  preds: lines of coco_*_pred_results_*.json ({"image_id", "score", "category" NAME, "bbox" xywh pixels, ...});
    data_items: the dataset json lines ({"id", "objects": [{"label", "bbox" normalised x1 y1 x2 y2, "iscrowd", "area"}]});
    categories / images: instances_val2017.json's lists ({"id", "name"} / {"id", "height", "width"}).  → {"mAP": stats[0], names...}."""
    name_to_cat = {c["name"]: c["id"] for c in categories}
    size = {im["id"]: (im["height"], im["width"]) for im in images}
    gts, ann_id = [], 1
    for item in data_items:
        img_h, img_w = size[item["id"]]
        for o in item["objects"]:
            x1, y1, x2, y2 = o["bbox"]
            gts.append({"id": ann_id, "image_id": item["id"], "category_id": name_to_cat[o["label"]], "iscrowd": o["iscrowd"], "area": o["area"],
                        "bbox": [round(x1 * img_w), round(y1 * img_h), round((x2 - x1) * img_w), round((y2 - y1) * img_h)]})
            ann_id += 1
    dts = []
    for p in preds:
        cid = name_to_cat.get(str(p.get("category", "")).lower())
        if cid is None:                                               # eval_coco.py:70-75: bare except → prediction dropped
            continue
        dts.append({"image_id": p["image_id"], "category_id": cid, "bbox": list(p["bbox"]), "score": p["score"]})
    res = coco_eval_bbox(gts, dts, [im["id"] for im in images], [c["id"] for c in categories])
    out = {"mAP": res["stats"][0], "n_gt": len(gts), "n_dt": len(dts)}
    out.update(dict(zip(STAT_NAMES, res["stats"])))
    return out
