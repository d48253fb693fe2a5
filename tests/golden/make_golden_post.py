"""Golden vectors for the caller-side post-processing.

The reference has no function for this step: eval/evaluation_scripts/utils.py:252-266 (and eval/test_demo.py:145-161) write it
inline in the eval loop with torch / Python built-ins.  This script evaluates those expressions with the installed torch on
seeded inputs (no reference import needed — nothing of the reference's package is involved) and stores inputs + outputs.
pycocotools (setup.py:31) is not in the container, so the RLE entries are the run lengths of `np.asfortranarray(mask)`
computed with numpy, plus the COCO string form produced by the restated encoder (string form: parity unpinned).

run:  python tests/golden/make_golden_post.py      (writes tests/golden/postprocess.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import padt_oracle as O  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20260927)
    sample_idx = [0, 0, 1, 2]
    image_sizes = [(97, 75), (64, 64), (130, 51)]                  # (w, h), PIL order
    valid_h = torch.tensor([6, 6, 4, 8])
    valid_w = torch.tensor([8, 8, 4, 5])
    Hm, Wm = 4 * int(valid_h.max()), 4 * int(valid_w.max())
    # smooth-ish logits so the zero level set is a curve, plus noise
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, Hm), torch.linspace(-1, 1, Wm), indexing="ij")
    masks = torch.stack([2.5 * torch.sin(3 * xx + i) * torch.cos(2 * yy - i) + 0.7 * torch.randn(Hm, Wm, generator=g) for i in range(4)])
    boxes = torch.rand(4, 4, generator=g)
    boxes[0] = torch.tensor([0.10, 0.90, 0.40, 0.30])             # cx - w/2 < 0 → clamp
    boxes[1] = torch.tensor([0.5, 0.5, 0.25, 0.5])                # 0.25*w = x.5 ties for round-half-even
    scores = torch.randn(4, 1, generator=g) * 2
    decoded = {"pred_boxes": boxes, "pred_score": scores, "pred_mask": masks, "pred_mask_valid_hw": (valid_h, valid_w),
               "sample_idx": sample_idx}
    labels = [["person", "dog"], ["cat"], ["bus"]]
    # the reference's expressions, utils.py:256-266
    exp_box, exp_score, exp_bits, exp_counts, exp_str, exp_up = [], [], [], [], [], []
    hw = torch.stack([valid_h, valid_w], dim=-1)
    for box, score, mask, mask_hw, si in zip(boxes, scores.sigmoid(), masks, hw, sample_idx):
        eb = (max(box[0].item() - box[2].item() / 2, 0), max(box[1].item() - box[3].item() / 2, 0), min(box[2].item(), 1), min(box[3].item(), 1))
        w, h = image_sizes[si]
        exp_box.append((round(eb[0] * w), round(eb[1] * h), round(eb[2] * w), round(eb[3] * h)))
        exp_score.append(score.item())
        up = torch.nn.functional.interpolate(mask[None, None, :mask_hw[0] * 4, :mask_hw[1] * 4], size=(h, w), mode="bilinear")[0, 0]
        m = (up.sigmoid() > 0.5).cpu().numpy().astype(np.uint8)
        exp_up.append(up.numpy().flatten())
        exp_bits.append(np.packbits(m.flatten()))
        c = O.mask_rle_counts(m)
        exp_counts.append(np.array(c, dtype=np.int64))
        exp_str.append(O.rle_counts_to_string(c))
    got = O.postprocess_results(decoded, labels, image_sizes)
    assert [r["bbox"] for r in got] == exp_box and [r["rle_counts"] for r in got] == [c.tolist() for c in exp_counts]
    np.savez_compressed(
        os.path.join(HERE, "postprocess.npz"), boxes=boxes.numpy(), scores=scores.numpy(), masks=masks.numpy(),
        valid_h=valid_h.numpy(), valid_w=valid_w.numpy(), sample_idx=np.array(sample_idx), image_sizes=np.array(image_sizes),
        labels=np.array(["person;dog", "cat", "bus"]), exp_box=np.array(exp_box), exp_score=np.array(exp_score),
        exp_bits=np.concatenate(exp_bits), exp_bits_len=np.array([len(b) for b in exp_bits]),
        exp_counts=np.concatenate(exp_counts), exp_counts_len=np.array([len(c) for c in exp_counts]),
        exp_str=np.array(exp_str), exp_up=np.concatenate(exp_up))
    print("wrote postprocess.npz:", exp_box, [len(c) for c in exp_counts])


if __name__ == "__main__":
    main()
