"""Score aggregation of the reference's RefCOCO evaluation (eval/evaluation_scripts/eval_refcoco.py:82-119), on the JSONL records the
harness writes (`padt_amd/harness.py` = utils.py:176-266): REC AP@0.5 over the best box IoU per referring expression, RES mean cIoU over
the best mask IoU — including the reference's conventions (keys "%d_%s" % (id, label); ground-truth boxes rounded to pixels with Python's
round(); the cIoU mean runs over the expressions that received at least one prediction, eval_refcoco.py:104-116).

COCO RLE is decoded here (rleFrString + decode of the COCO maskApi, the inverse of postprocess.rle_string / rle_counts): pycocotools is
not in the image.  The OVD score — COCO bbox mAP, eval_coco.py:21-93 — is `score_coco` (padt_amd/coco_eval.py, COCOeval's bbox protocol
written from its published definition, re-exported here).
"""
from collections import defaultdict
from typing import Dict, Iterable, List, Sequence

import numpy as np

from .coco_eval import coco_eval_bbox, score_coco  # noqa: F401  (OVD: eval_coco.py)
from .postprocess import box_iou_xywh


def rle_counts_from_string(s: str) -> List[int]:
    """COCO maskApi rleFrString: 5 payload bits per character (+48), bit 5 = continuation, sign-extended, counts beyond the second
    are stored as differences to the count two places back."""
    counts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(rle: Dict) -> np.ndarray:
    """{"size": [h, w], "counts": str | list} → uint8 (h, w) mask (column-major runs, zero run first), as cocomask.decode."""
    h, w = rle["size"]
    counts = rle["counts"]
    if isinstance(counts, (str, bytes)):
        counts = rle_counts_from_string(counts.decode() if isinstance(counts, bytes) else counts)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos: pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


def mask_ciou(pred: np.ndarray, gt: np.ndarray) -> float:
    """calculate_ciou, eval_refcoco.py:44-47."""
    i = np.logical_and(pred, gt).sum()
    u = np.logical_or(pred, gt).sum()
    return float(i / u) if u > 0 else 0.0


def score_refcoco(preds: Iterable[Dict], gts: Sequence[Dict]) -> Dict[str, float]:
    """preds: records of *_pred_results_*.json ({"image_id", "category", "bbox" (x, y, w, h pixels), "mask" RLE or None});
    gts: {"id", "label", "bbox" normalised (x1, y1, x2, y2), "width", "height", "rle" or "mask"} — the fields eval_refcoco.py:82-97 reads
    from the dataset json and the image.  → {"rec_ap50", "res_ciou", "n_expressions", "n_scored_masks"}."""
    gt_dict = {}
    accuracy = {}
    mask_cious = defaultdict(float)
    for item in gts:
        name = "%d_%s" % (item["id"], item["label"])
        x1, y1, x2, y2 = item["bbox"]
        w, h = item["width"], item["height"]
        gt_bbox = [round(x1 * w), round(y1 * h), round((x2 - x1) * w), round((y2 - y1) * h)]
        gt_mask = item["mask"] if item.get("mask") is not None else (rle_decode(item["rle"]) if item.get("rle") is not None else None)
        gt_dict[name] = (gt_bbox, gt_mask)
        accuracy[name] = 0.0
    for pred in preds:
        name = "%d_%s" % (pred["image_id"], pred["category"])
        if name not in gt_dict:
            continue
        gt_bbox, gt_mask = gt_dict[name]
        if pred.get("mask") is not None and gt_mask is not None:
            pm = pred["mask"] if isinstance(pred["mask"], np.ndarray) else rle_decode(pred["mask"])
            mask_cious[name] = max(mask_ciou(pm > 0, gt_mask > 0), mask_cious[name])
        accuracy[name] = max(box_iou_xywh(gt_bbox, pred["bbox"]), accuracy[name])
    ious = np.array(list(accuracy.values()), dtype=np.float64)
    cious = np.array(list(mask_cious.values()), dtype=np.float64)
    return {"rec_ap50": float((ious >= 0.5).mean()) if ious.size else 0.0, "res_ciou": float(cious.mean()) if cious.size else 0.0,
            "n_expressions": int(ious.size), "n_scored_masks": int(cious.size)}
