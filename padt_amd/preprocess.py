"""Image front-end: what the reference gets from `AutoProcessor` — the HF Qwen2-VL image processor
(transformers/models/qwen2_vl/image_processing_pil_qwen2_vl.py: smart_resize → bicubic resize → rescale → normalize → patchify).

smart_resize is integer/host logic (restated).  The bicubic resize stays PIL's on the host (it IS the reference's resize; a GPU
resampler would not be bit-identical to PIL's antialiased uint8 passes).  Everything after the resize — rescale, normalize and
the block-major patch layout, 2116 x 1176 values per 644 x 644 image — is one HIP kernel over the uint8 image
(padt_patchify_normalize), bit-exact against the processor's float32 output by construction (a 3 x 256 table computed with the
processor's own arithmetic).
"""
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import ops

IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)
RESCALE = 0.00392156862745098


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """HF smart_resize: both sides multiples of `factor`, pixel count inside [min_pixels, max_pixels], aspect kept."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def normalize_lut(mean: Sequence[float] = IMAGE_MEAN, std: Sequence[float] = IMAGE_STD, rescale: float = RESCALE) -> np.ndarray:
    """lut[c][u] = (float32(float64(u) * rescale) - float32(mean[c])) / float32(std[c]) — image_transforms.rescale / normalize."""
    r = (np.arange(256, dtype=np.float64) * rescale).astype(np.float32)
    m = np.array(mean, dtype=np.float32)[:, None]
    s = np.array(std, dtype=np.float32)[:, None]
    return ((r[None, :] - m) / s).astype(np.float32)


class ImageFrontEnd:
    def __init__(self, device, patch: int = 14, merge: int = 2, temporal: int = 2, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280, dtype=torch.bfloat16):
        self.device, self.patch, self.merge, self.temporal = device, patch, merge, temporal
        self.min_pixels, self.max_pixels, self.dtype = min_pixels, max_pixels, dtype
        self.lut = torch.from_numpy(normalize_lut()).to(device)

    def resize_host(self, image):
        """PIL image or (H, W, 3) uint8 array → uint8 array at the smart_resize size (PIL bicubic, as the processor does)."""
        if hasattr(image, "convert"):                              # PIL: the processor's do_convert_rgb (modes L / P / RGBA / CMYK ...)
            image = image.convert("RGB")
        arr = np.asarray(image)
        if arr.ndim == 2:                                          # gray ndarray → 3 equal channels
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        if arr.ndim != 3 or arr.shape[2] < 3 or arr.dtype != np.uint8:
            raise ValueError(f"image must be PIL or a (H, W[, >=3]) uint8 array, got shape {arr.shape} dtype {arr.dtype}")
        h, w = arr.shape[:2]
        rh, rw = smart_resize(h, w, self.patch * self.merge, self.min_pixels, self.max_pixels)
        if (rh, rw) == (h, w):
            return np.ascontiguousarray(arr[..., :3])
        from PIL import Image
        return np.asarray(Image.fromarray(arr[..., :3]).resize((rw, rh), resample=Image.BICUBIC))

    def __call__(self, images: List) -> Tuple[torch.Tensor, torch.Tensor]:
        """→ (pixel_values (ΣP, 1176) on the device, image_grid_thw (B, 3) int64 on the host)."""
        arrs = [self.resize_host(im) for im in images]
        grids = [[1, a.shape[0] // self.patch, a.shape[1] // self.patch] for a in arrs]
        total = sum(g[1] * g[2] for g in grids)
        row = 3 * self.temporal * self.patch * self.patch
        out = torch.empty((total, row), device=self.device, dtype=self.dtype)
        o = 0
        for a, g in zip(arrs, grids):
            n = g[1] * g[2]
            img = torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True)
            ops.patchify_normalize(img, self.lut, out[o:o + n], self.patch, self.merge, self.temporal)
            o += n
        return out, torch.tensor(grids, dtype=torch.int64)
