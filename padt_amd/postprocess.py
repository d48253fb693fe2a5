"""Caller-side post-processing of vl_decode's output: what the reference's eval loop does inline at
eval/evaluation_scripts/utils.py:252-266 (and eval/test_demo.py:145-161) — score sigmoid, box cxcywh → clamped, image-scaled,
rounded xywh, mask bilinear up-sampling + sigmoid > 0.5, COCO RLE.

The 4-number box arithmetic stays on the host in Python floats with Python's round() — exactly the reference's expressions, so
the integers are bit-identical.  The mask resize + threshold (h*w pixels per object) is one HIP kernel
(padt_mask_upsample_binarize).  RLE run lengths are integer work on the binary mask (column-major, zero run first, as
pycocotools' `encode(np.asfortranarray(mask))`); the compressed `counts` string follows COCO maskApi's rleToString.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops


def rle_counts(mask_u8: np.ndarray) -> List[int]:
    flat = np.asarray(mask_u8, dtype=np.uint8).flatten(order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    runs = np.diff(np.concatenate([[0], change, [flat.size]])).tolist()
    return runs if flat[0] == 0 else [0] + runs


def rle_string(counts: Sequence[int]) -> str:
    """COCO maskApi rleToString: every count (from the fourth on: its difference to the count two places earlier) as little-endian 5-bit
    groups, bit 5 = "more follows", sign-extended, + 48.  Vectorised: a value x needs g = (bit_length(x >= 0 ? x : ~x) + 5) // 5 groups
    (the smallest g with -2^(5g-1) <= x < 2^(5g-1)), so the output offsets are a prefix sum and group j of all values that have one is
    one numpy pass.  (A noisy 640 x 640 mask has ~50 000 runs; the per-count Python loop — kept below as the statement the tests compare
    against — cost 25 ms per mask and was the bottleneck of bench.py's to_rle leg.)"""
    c = np.asarray(counts, dtype=np.int64)
    if c.size == 0:
        return ""
    x = c.copy()
    x[3:] -= c[1:-2]
    ax = np.where(x >= 0, x, ~x)
    bits = np.frexp(ax.astype(np.float64))[1].astype(np.int64)       # bit length (exact below 2^53)
    g = (bits + 5) // 5
    off = np.cumsum(g) - g
    out = np.empty(int(g.sum()), dtype=np.uint8)
    sel = np.arange(c.size)
    for j in range(int(g.max())):
        sel = sel[g[sel] > j]
        ch = (x[sel] >> (5 * j)) & 0x1F
        out[off[sel] + j] = (ch | ((g[sel] - 1 > j) << 5)) + 48
    return out.tobytes().decode("ascii")


def rle_string_loop(counts: Sequence[int]) -> str:
    """rleToString, count by count."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def box_to_pixels(box: Sequence[float], w: int, h: int) -> Tuple[int, int, int, int]:
    """utils.py:258-260 verbatim semantics (Python floats, Python round)."""
    b0, b1, b2, b3 = (float(v) for v in box)
    e = (max(b0 - b2 / 2, 0), max(b1 - b3 / 2, 0), min(b2, 1), min(b3, 1))
    return (round(e[0] * w), round(e[1] * h), round(e[2] * w), round(e[3] * h))


def box_iou_xywh(b1: Sequence[float], b2: Sequence[float]) -> float:
    """calculate_iou of eval/evaluation_scripts/eval_refcoco.py:15-41 on (x, y, w, h) boxes."""
    x1, y1, w1, h1 = b1
    x2, y2, w2, h2 = b2
    iw = max(0, min(x1 + w1, x2 + w2) - max(x1, x2))
    ih = max(0, min(y1 + h1, y2 + h2) - max(y1, y2))
    inter = iw * ih
    union = w1 * h1 + w2 * h2 - inter
    return 0.0 if union == 0 else inter / union


def postprocess_results(decoded: Dict, labels: List[List[str]], image_sizes: Sequence[Tuple[int, int]], rle: bool = True,
                        want_mask: bool = True, device_rle: bool = True) -> List[Dict]:
    """decoded: vl_decode's dict; labels: parseVRTintoCompletion's per-sample label lists; image_sizes: (w, h) per sample
    (PIL order, as `images[sample_idx].size`).  → one dict per object: sample_idx, score, category, bbox, mask (uint8 numpy
    (h, w); want_mask), rle {size, counts}.
    device_rle (default): the run lengths and the COCO counts string are computed on the GPU from the binarised masks (padt_mask_rle);
    only the strings cross PCIe — with want_mask=False, which is the end state of the reference's eval loop (utils.py:262-265 writes the
    RLE, not the mask), the 640 x 640 masks never leave the device.  device_rle=False: the host statement (rle_counts + rle_string)."""
    n = decoded["pred_boxes"].shape[0]
    if n == 0:
        return []
    sidx = list(decoded["sample_idx"])
    sizes = [image_sizes[s] for s in sidx]
    masks = decoded.get("pred_mask")
    bin_dev = rle_h = None
    if masks is not None:
        # device work first — mask up-sampling + threshold, run lengths + counts strings — so that the host waits ONCE (at the first copy below)
        dev = masks.device
        src = decoded.get("pred_mask_src_hw")                          # vl_decode's int32 device copy of 4 x valid_hw (no ATen kernels on this path)
        if src is not None and src[0].numel() == n:
            hs, ws = src
        else:                                                          # a dict that did not come from vl_decode
            hs = (decoded["pred_mask_valid_hw"][0].to(torch.int32) * 4).to(dev)
            ws = (decoded["pred_mask_valid_hw"][1].to(torch.int32) * 4).to(dev)
        dh = torch.tensor([s[1] for s in sizes], dtype=torch.int32, device=dev)
        dw = torch.tensor([s[0] for s in sizes], dtype=torch.int32, device=dev)
        mh, mw = max(s[1] for s in sizes), max(s[0] for s in sizes)
        bin_dev = ops.mask_upsample_binarize(masks.float().contiguous(), hs, ws, dh, dw, mh, mw)
        if rle and device_rle:
            rle_h = ops.mask_rle_launch(bin_dev, dh, dw)
    boxes = decoded["pred_boxes"].float().cpu()
    scores = decoded["pred_score"].float().reshape(-1).cpu().sigmoid()     # n numbers: on the host (fp32, what the oracle does)
    flat_labels = sum(labels, [])
    res = [{"sample_idx": int(s), "score": scores[i].item(), "category": flat_labels[i],
            "bbox": box_to_pixels(boxes[i].tolist(), sizes[i][0], sizes[i][1])} for i, s in enumerate(sidx)]
    if masks is not None:
        strs = ops.mask_rle_fetch(rle_h, on_overflow="none") if rle_h is not None else None     # None: beyond the bounded device scratch → host statement below
        binm = bin_dev.cpu().numpy() if (want_mask or (rle and strs is None)) else None
        for i, r in enumerate(res):
            if binm is not None:
                m = binm[i, : sizes[i][1], : sizes[i][0]]
                if want_mask:
                    r["mask"] = m
            if rle:
                r["rle"] = {"size": [sizes[i][1], sizes[i][0]], "counts": strs[i] if strs is not None else rle_string(rle_counts(m))}
    return res
