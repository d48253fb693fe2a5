"""COCO bbox mAP (padt_amd/coco_eval.py = what eval/evaluation_scripts/eval_coco.py:78-93 gets from pycocotools' COCOeval) against
HAND-COMPUTED expectations: pycocotools is not in the container, so the evaluator is pinned by cases small enough to evaluate on paper
(the arithmetic is in each docstring).  Unpinned against pycocotools itself — stated in coco_eval.py / DESIGN.md."""
import numpy as np
import pytest

from padt_amd.coco_eval import IOU_THRS, REC_THRS, bbox_iou_matrix, coco_eval_bbox, score_coco


def G(img, cat, box, area=None, crowd=0):
    return {"image_id": img, "category_id": cat, "bbox": box, "area": box[2] * box[3] if area is None else area, "iscrowd": crowd}


def D(img, cat, box, score):
    return {"image_id": img, "category_id": cat, "bbox": box, "score": score}


def test_thresholds_are_the_published_ones():
    assert len(IOU_THRS) == 10 and IOU_THRS[0] == 0.5 and abs(IOU_THRS[-1] - 0.95) < 1e-12
    assert len(REC_THRS) == 101 and REC_THRS[0] == 0.0 and REC_THRS[-1] == 1.0


def test_iou_matrix_with_a_crowd_column():
    """det (0,0,10,10) vs gt (5,0,10,10): inter 50, union 150 → 1/3; same gt as a crowd: inter / det area = 0.5; touching boxes → 0."""
    m = bbox_iou_matrix(np.array([[0, 0, 10, 10]]), np.array([[5, 0, 10, 10], [5, 0, 10, 10], [10, 0, 5, 5]]), [0, 1, 0])
    assert np.allclose(m, [[1 / 3, 0.5, 0.0]])


def test_one_image_two_objects_hand_computed():
    """gts A (0,0,10,10), B (20,20,10,10); dets 0.9 on A (IoU 1), 0.8 far away, 0.7 = (20,20,10,8) on B (IoU 0.8).
    IoU thr <= 0.8 (7 of 10): TP FP TP → recall .5 .5 1, precision 1 .5 .667 → envelope 1 .667 .667 → 51 recall levels (0 … .50) at 1,
    50 at 2/3 → AP = (51 + 50 * 2/3) / 101 = 0.834983.  thr > 0.8 (3 of 10): TP FP FP → AP = 51/101 = 0.504950.
    AP = (7 * 0.834983 + 3 * 0.504950) / 10 = 0.735974; AP50 = AP75 = 0.834983; objects are small (area 100): AP_small = AP,
    medium / large have no ground truth → -1.  AR@1 = 0.5; AR@10 = AR@100 = (7 * 1 + 3 * 0.5) / 10 = 0.85."""
    gts = [G(1, 1, [0, 0, 10, 10]), G(1, 1, [20, 20, 10, 10])]
    dts = [D(1, 1, [0, 0, 10, 10], 0.9), D(1, 1, [50, 50, 10, 10], 0.8), D(1, 1, [20, 20, 10, 8], 0.7)]
    s = coco_eval_bbox(gts, dts, [1], [1])["stats"]
    ap_lo, ap_hi = (51 + 50 * 2 / 3) / 101, 51 / 101
    assert s[0] == pytest.approx((7 * ap_lo + 3 * ap_hi) / 10, abs=1e-12)
    assert s[1] == pytest.approx(ap_lo, abs=1e-12) and s[2] == pytest.approx(ap_lo, abs=1e-12)
    assert s[3] == pytest.approx(s[0]) and s[4] == -1 and s[5] == -1
    assert s[6] == pytest.approx(0.5) and s[7] == pytest.approx(0.85) and s[8] == pytest.approx(0.85)


def test_detections_are_merged_across_images_by_score():
    """image 1: gt + det 0.6 on it; image 2: gt + det 0.9 elsewhere + det 0.5 on it.  Merged: 0.9 FP, 0.6 TP, 0.5 TP → recall 0 .5 1,
    precision 0 .5 .667 → envelope .667 everywhere → AP = 2/3 at every threshold (the matches are exact boxes)."""
    gts = [G(1, 1, [0, 0, 40, 40]), G(2, 1, [0, 0, 40, 40])]
    dts = [D(1, 1, [0, 0, 40, 40], 0.6), D(2, 1, [100, 100, 40, 40], 0.9), D(2, 1, [0, 0, 40, 40], 0.5)]
    r = coco_eval_bbox(gts, dts, [1, 2], [1])
    assert r["stats"][0] == pytest.approx(2 / 3, abs=1e-9)
    assert r["stats"][4] == pytest.approx(2 / 3, abs=1e-9) and r["stats"][3] == -1      # area 1600: medium
    assert r["stats"][6] == pytest.approx(0.5)                                             # AR@1: image 1's det hits, image 2's top det misses


def test_crowd_region_absorbs_detections_and_is_not_a_target():
    """gt R regular (0,0,10,10), gt C crowd (100,100,100,100).  dets: 0.9 on R (TP); 0.8 and 0.7 both inside C (each matched to the
    crowd, which stays available → ignored, neither TP nor FP); 0.6 nowhere (FP).  Regular ground truths: 1.
    Counted sequence: TP, FP → recall 1 1, precision 1 .5 → every recall level reads index 0 → AP = 1."""
    gts = [G(1, 1, [0, 0, 10, 10]), G(1, 1, [100, 100, 100, 100], crowd=1)]
    dts = [D(1, 1, [0, 0, 10, 10], 0.9), D(1, 1, [110, 110, 20, 20], 0.8), D(1, 1, [150, 150, 30, 30], 0.7), D(1, 1, [300, 300, 10, 10], 0.6)]
    r = coco_eval_bbox(gts, dts, [1], [1])
    assert r["stats"][0] == pytest.approx(1.0) and r["stats"][8] == pytest.approx(1.0)
    # the same detections with C as a REGULAR object: 0.8 has IoU 400/10000 with it → FP, so does 0.7 → TP FP FP FP, recall .5 → AP = 51/101
    gts[1]["iscrowd"] = 0
    assert coco_eval_bbox(gts, dts, [1], [1])["stats"][0] == pytest.approx(51 / 101)


def test_area_ranges_ignore_out_of_range_objects_and_their_matches():
    """one medium object (area 2500) detected exactly, one unmatched small detection (area 100).  'all': TP, FP → AP 1 (TP first).
    'small': the object is ignored (no regular gt → -1).  'medium': the small FP is out of range and ignored → AP 1.  'large': -1."""
    gts = [G(1, 1, [0, 0, 50, 50])]
    dts = [D(1, 1, [0, 0, 50, 50], 0.9), D(1, 1, [200, 200, 10, 10], 0.95)]
    s = coco_eval_bbox(gts, dts, [1], [1])["stats"]
    # 'all': FP (0.95) comes first → precision .5 at recall 1 → AP .5;  'medium': the FP is ignored → 1
    assert s[0] == pytest.approx(0.5) and s[4] == pytest.approx(1.0) and s[3] == -1 and s[5] == -1


def test_max_dets_and_categories_are_averaged_over_cells_with_ground_truth():
    """category 1: perfect (AP 1); category 2: its only object is missed (AP 0); category 3: detections but no ground truth (-1, not
    averaged) → mAP = 0.5.  maxDets 1 keeps only the best-scored detection per image AND category."""
    gts = [G(1, 1, [0, 0, 40, 40]), G(1, 2, [100, 0, 40, 40])]
    dts = [D(1, 1, [0, 0, 40, 40], 0.9), D(1, 2, [300, 300, 40, 40], 0.8), D(1, 3, [0, 0, 40, 40], 0.99)]
    r = coco_eval_bbox(gts, dts, [1], [1, 2, 3])
    assert r["stats"][0] == pytest.approx(0.5) and r["stats"][6] == pytest.approx(0.5)
    assert (r["precision"][:, :, 2] == -1).all()


def test_score_coco_assembly_follows_eval_coco_py():
    """name → id (lower-cased prediction names, unknown names dropped: eval_coco.py:69-76), ground-truth boxes from normalised
    x1 y1 x2 y2 rounded to pixel xywh (:55), all of val's image ids evaluated (images without anything do not count)."""
    cats = [{"id": 1, "name": "person"}, {"id": 18, "name": "dog"}]
    images = [{"id": 7, "height": 200, "width": 400}, {"id": 8, "height": 100, "width": 100}]
    data = [{"id": 7, "objects": [{"label": "person", "bbox": [0.1, 0.1, 0.35, 0.6], "iscrowd": 0, "area": 9000.0},
                                  {"label": "dog", "bbox": [0.5, 0.5, 0.75, 1.0], "iscrowd": 0, "area": 9000.0}]}]
    preds = [{"image_id": 7, "score": 0.9, "category": "Person", "bbox": [40, 20, 100, 100], "mask": None},
             {"image_id": 7, "score": 0.8, "category": "unicorn", "bbox": [0, 0, 10, 10], "mask": None},
             {"image_id": 7, "score": 0.7, "category": "dog", "bbox": [200, 100, 100, 100], "mask": None}]
    out = score_coco(preds, data, cats, images)
    assert out["n_gt"] == 2 and out["n_dt"] == 2                    # the unicorn is gone
    assert out["mAP"] == pytest.approx(1.0) and out["AP50"] == pytest.approx(1.0) and out["AR@1"] == pytest.approx(1.0)
    # shift the dog box by 10 % of its width: IoU = 90 / 110 = 0.818 → matched at thresholds <= 0.8 only → dog AP = 0.7, mAP = 0.85
    preds[2]["bbox"] = [210, 100, 100, 100]
    assert score_coco(preds, data, cats, images)["mAP"] == pytest.approx(0.85)


def test_random_scenes_against_a_direct_statement_of_ap_at_one_threshold():
    """Randomised cross-check of the cell the other 11 numbers are built from: one category, no crowds, area range 'all', maxDets 100.
    AP at IoU t stated directly — per image the detections by falling score take the best still-free ground truth with IoU >= t; all
    detections merged by score; precision made monotone from the right and read at the first position whose recall reaches 0, .01, ..., 1
    (0 where it never does) — against precision[t, :, 0, 0, 2] of coco_eval_bbox, for t = .5, .75, .95 on 20 random scenes."""
    from padt_amd.coco_eval import IOU_THRS, REC_THRS, coco_eval_bbox
    rng = np.random.default_rng(7)

    def iou(a, b):
        w = min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0])
        h = min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1])
        inter = w * h if w > 0 and h > 0 else 0.0
        return inter / (a[2] * a[3] + b[2] * b[3] - inter)

    for scene in range(20):
        gts, dts = [], []
        for img in range(1, 5):
            for _ in range(rng.integers(0, 5)):
                x, y, w, h = rng.uniform(0, 200), rng.uniform(0, 200), rng.uniform(20, 120), rng.uniform(20, 120)
                gts.append({"id": len(gts) + 1, "image_id": img, "category_id": 1, "bbox": [x, y, w, h], "area": w * h, "iscrowd": 0})
                for _ in range(rng.integers(0, 3)):                  # detections near a ground truth, of varying quality
                    j = rng.normal(0, 12, 4)
                    dts.append({"image_id": img, "category_id": 1, "bbox": [x + j[0], y + j[1], max(5.0, w + j[2]), max(5.0, h + j[3])],
                                "score": float(rng.uniform(0, 1))})
            for _ in range(rng.integers(0, 3)):                      # stray detections
                dts.append({"image_id": img, "category_id": 1, "bbox": [rng.uniform(0, 250), rng.uniform(0, 250), 40.0, 40.0],
                            "score": float(rng.uniform(0, 1))})
        if not gts:
            continue
        res = coco_eval_bbox(gts, dts, [1, 2, 3, 4], [1])
        for t in (0.5, 0.75, 0.95):
            flagged = []                                             # (score, is true positive)
            for img in range(1, 5):
                g = [x for x in gts if x["image_id"] == img]
                taken = [False] * len(g)
                for d in sorted((x for x in dts if x["image_id"] == img), key=lambda x: -x["score"]):
                    cand = [(iou(d["bbox"], g[k]["bbox"]), k) for k in range(len(g)) if not taken[k]]
                    cand = [c for c in cand if c[0] >= t]
                    if cand:
                        taken[max(cand)[1]] = True
                    flagged.append((d["score"], bool(cand)))
            flagged.sort(key=lambda x: -x[0])
            tp = np.cumsum([f[1] for f in flagged]) if flagged else np.zeros(0)
            fp = np.cumsum([not f[1] for f in flagged]) if flagged else np.zeros(0)
            rc = tp / len(gts)
            pr = tp / np.maximum(tp + fp, 1)
            for i in range(len(pr) - 2, -1, -1):
                pr[i] = max(pr[i], pr[i + 1])
            want = [next((pr[i] for i in range(len(rc)) if rc[i] >= r), 0.0) for r in REC_THRS]
            ti = int(np.where(np.isclose(IOU_THRS, t))[0][0])
            got = res["precision"][ti, :, 0, 0, 2]
            assert np.allclose(got, want, atol=1e-12), (scene, t)
