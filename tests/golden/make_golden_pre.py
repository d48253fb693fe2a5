"""Golden vectors for the image front-end, generated with the INSTALLED transformers' Qwen2-VL PIL image
processor (the class AutoProcessor resolves to without torchvision) — run with HF_HUB_OFFLINE=1 TRANSFORMERS_OFFLINE=1.
The reference pins transformers==4.50.0 (not in the container): parity with 4.50's processor is unpinned.

run:  HF_HUB_OFFLINE=1 TRANSFORMERS_OFFLINE=1 python tests/golden/make_golden_pre.py      (writes tests/golden/preprocess.npz)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil, smart_resize
    p = Qwen2VLImageProcessorPil()
    rs = np.random.RandomState(20260927)
    imgs = [(rs.rand(56, 84, 3) * 255).astype(np.uint8), (rs.rand(84, 112, 3) * 255).astype(np.uint8)]
    imgs[1][:4, :4] = np.array([[0, 255, 128]], dtype=np.uint8)          # extremes
    outs = [p(images=[im], return_tensors="np") for im in imgs]
    both = p(images=imgs, return_tensors="np")
    assert np.array_equal(both["pixel_values"], np.concatenate([o["pixel_values"] for o in outs]))
    # images that are NOT at their smart_resize size: the processor resizes them itself (PIL bicubic) before patchifying
    raws = [(rs.rand(61, 97, 3) * 255).astype(np.uint8), (rs.rand(150, 70, 3) * 255).astype(np.uint8)]
    raw_out = p(images=raws, return_tensors="np")
    sizes = [(640, 640), (100, 37), (37, 100), (1, 1), (28, 28), (2000, 3000), (4096, 64), (333, 555), (27, 5400), (644, 644)]
    sr = [smart_resize(h, w, factor=28, min_pixels=p.size.shortest_edge, max_pixels=p.size.longest_edge) for h, w in sizes]
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), img0=imgs[0], img1=imgs[1], pix=both["pixel_values"],
                        grid=both["image_grid_thw"], mean=np.array(p.image_mean), std=np.array(p.image_std),
                        rescale=np.array(p.rescale_factor), sizes=np.array(sizes), smart=np.array(sr),
                        min_max=np.array([p.size.shortest_edge, p.size.longest_edge]), raw0=raws[0], raw1=raws[1],
                        pix_raw=raw_out["pixel_values"], grid_raw=raw_out["image_grid_thw"])
    print("wrote preprocess.npz", both["pixel_values"].shape, both["image_grid_thw"].tolist(), sr)


if __name__ == "__main__":
    main()
