"""Post-processing: oracle vs the golden fixture, host integer logic of the product module."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "postprocess.npz"))
    decoded = {"pred_boxes": torch.from_numpy(z["boxes"]), "pred_score": torch.from_numpy(z["scores"]),
               "pred_mask": torch.from_numpy(z["masks"]),
               "pred_mask_valid_hw": (torch.from_numpy(z["valid_h"]), torch.from_numpy(z["valid_w"])),
               "sample_idx": z["sample_idx"].tolist()}
    labels = [s.split(";") for s in z["labels"].tolist()]
    sizes = [tuple(int(v) for v in r) for r in z["image_sizes"]]
    return z, decoded, labels, sizes


def split(flat, lens):
    out, o = [], 0
    for l in lens:
        out.append(flat[o:o + l])
        o += l
    return out


def rle_decode_string(s):
    """Inverse of COCO's rleToString (rleFrString), written independently of the encoder: checks the pair is consistent."""
    counts, p = [], 0
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


def test_oracle_postprocess_matches_golden():
    import padt_oracle as O
    z, decoded, labels, sizes = load()
    res = O.postprocess_results(decoded, labels, sizes)
    assert [list(r["bbox"]) for r in res] == z["exp_box"].tolist()
    assert np.allclose([r["score"] for r in res], z["exp_score"], rtol=0, atol=1e-7)
    assert [r["category"] for r in res] == ["person", "dog", "cat", "bus"]
    bits = split(z["exp_bits"], z["exp_bits_len"])
    counts = split(z["exp_counts"], z["exp_counts_len"])
    for r, b, c, s in zip(res, bits, counts, z["exp_str"].tolist()):
        assert np.array_equal(np.packbits(r["mask"].flatten()), b)
        assert r["rle_counts"] == c.tolist() and r["rle"]["counts"] == s
        assert sum(r["rle_counts"]) == r["mask"].size
        assert rle_decode_string(s) == c.tolist()                   # encoder/decoder pair consistent


def test_host_box_and_rle_logic_matches_golden():
    from padt_amd import postprocess as P
    z, decoded, labels, sizes = load()
    for box, si, exp in zip(z["boxes"], z["sample_idx"], z["exp_box"].tolist()):
        assert list(P.box_to_pixels(box.tolist(), *sizes[si])) == exp
    assert P.box_to_pixels([0.5, 0.5, 0.25, 0.5], 10, 10) == (4, 2, 2, 5)          # 3.75→4, 2.5→2 (half-even), 2.5→2, 5.0→5
    bits = split(z["exp_bits"], z["exp_bits_len"])
    counts = split(z["exp_counts"], z["exp_counts_len"])
    for b, c, s, si in zip(bits, counts, z["exp_str"].tolist(), z["sample_idx"]):
        w, h = sizes[si]
        m = np.unpackbits(b)[: h * w].reshape(h, w)
        assert P.rle_counts(m) == c.tolist() and P.rle_string(c.tolist()) == s
    assert P.rle_counts(np.ones((2, 2), np.uint8)) == [0, 4] and P.rle_counts(np.zeros((0, 0), np.uint8)) == []
    assert P.rle_counts(np.array([[0, 1], [1, 1]], np.uint8)) == [1, 3]            # column-major: 0,1 | 1,1
    assert P.postprocess_results({"pred_boxes": torch.zeros(0, 4)}, [[]], []) == []


def test_box_iou_matches_reference_definition():
    import padt_oracle as O
    from padt_amd import postprocess as P
    cases = [((0, 0, 10, 10), (5, 5, 10, 10), 25 / 175), ((1, 2, 3, 4), (1, 2, 3, 4), 1.0), ((0, 0, 2, 2), (3, 3, 1, 1), 0.0),
             ((0, 0, 0, 0), (0, 0, 0, 0), 0.0), ((36, 19, 24, 38), (30, 20, 24, 30), None)]
    for b1, b2, exp in cases:
        a, b = O.box_iou_xywh(b1, b2), P.box_iou_xywh(b1, b2)
        assert a == b and (exp is None or abs(a - exp) < 1e-12)


def test_rle_round_trip_and_refcoco_score_aggregation():
    """scoring.py: COCO RLE string decode is the inverse of the encoder (random masks incl. empty / full / long runs, which exercise the
    difference coding and negative deltas), and the AP@0.5 / cIoU aggregation follows eval_refcoco.py:82-119 on a hand-checked case."""
    import numpy as np
    from padt_amd import postprocess as P
    from padt_amd import scoring as S
    rng = np.random.default_rng(3)
    for shape in [(7, 5), (64, 48), (1, 1), (480, 640)]:
        for dens in (0.0, 0.03, 0.5, 1.0):
            m = (rng.random(shape) < dens).astype(np.uint8)
            if shape == (480, 640):
                m[100:300, 200:500] = 1                     # long runs → multi-character counts
            counts = P.rle_counts(m)
            assert S.rle_counts_from_string(P.rle_string(counts)) == counts
            assert np.array_equal(S.rle_decode({"size": list(shape), "counts": P.rle_string(counts)}), m)
    H, W = 100, 200
    gm = np.zeros((H, W), np.uint8)
    gm[20:60, 40:120] = 1
    gts = [{"id": 1, "label": "dog", "bbox": [0.2, 0.2, 0.6, 0.6], "width": W, "height": H, "mask": gm},
           {"id": 2, "label": "cat", "bbox": [0.0, 0.0, 0.5, 0.5], "width": W, "height": H, "mask": gm},
           {"id": 3, "label": "bird", "bbox": [0.1, 0.1, 0.2, 0.2], "width": W, "height": H, "mask": gm}]
    pm = np.zeros((H, W), np.uint8)
    pm[20:60, 40:80] = 1                                     # half of the gt mask → cIoU 0.5
    enc = lambda m: {"size": [H, W], "counts": P.rle_string(P.rle_counts(m))}
    preds = [{"image_id": 1, "category": "dog", "bbox": [40, 20, 80, 40], "mask": enc(pm)},          # exact box → IoU 1
             {"image_id": 1, "category": "dog", "bbox": [0, 0, 10, 10], "mask": enc(np.zeros((H, W), np.uint8))},   # worse duplicate: max kept
             {"image_id": 2, "category": "cat", "bbox": [90, 40, 100, 50], "mask": enc(gm)},           # box IoU < 0.5, mask perfect
             {"image_id": 9, "category": "none", "bbox": [0, 0, 1, 1], "mask": None}]                  # unknown expression: ignored
    r = S.score_refcoco(preds, gts)
    assert r["n_expressions"] == 3 and r["n_scored_masks"] == 2          # "bird" got no prediction: counts in AP, not in the cIoU mean
    assert abs(r["rec_ap50"] - 1 / 3) < 1e-12 and abs(r["res_ciou"] - 0.75) < 1e-12
