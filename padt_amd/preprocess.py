"""Image front-end: what the reference gets from `AutoProcessor` — the HF Qwen2-VL image processor
(transformers/models/qwen2_vl/image_processing_pil_qwen2_vl.py: smart_resize → bicubic resize → rescale → normalize → patchify) —
plus the callers' own LANCZOS pre-resize (eval/test_demo.py:67-73 max side 644; eval/evaluation_scripts/utils.py:205-218 min side 28).

All of it runs on the GPU, bit-exact:
  * smart_resize / the callers' size rules are integer host logic (restated);
  * the resize is Pillow's ImagingResample (libImaging/Resample.c) — two separable passes over 8-bit data with 22-bit fixed-point
    coefficients: integer arithmetic, so csrc/resize.hip reproduces PIL.Image.resize byte for byte once it is handed the same
    coefficient tables, which `pil_resample_coeffs` builds exactly as precompute_coeffs / normalize_coeffs_8bpc do (same double
    operations in the same order; cached per (in, out, filter));
  * rescale, normalize and the block-major patch layout (2116 x 1176 values per 644 x 644 image) are one kernel over the uint8 image
    (padt_patchify_normalize), bit-exact against the processor's float32 output through a 3 x 256 table computed with the
    processor's own arithmetic.
The host only decodes the file and uploads the raw uint8 pixels.
"""
import functools
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


# ------------------------------------------------------------------------------------------------ Pillow resample coefficients
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _lanczos(x):
    def sinc(v):                                                  # libm sin through math.sin — the function Pillow's C code calls
        out = np.ones_like(v)
        nz = v != 0.0
        vv = v[nz] * math.pi
        out[nz] = np.array([math.sin(t) for t in vv.tolist()], dtype=np.float64) / vv
        return out
    inside = (x >= -3.0) & (x < 3.0)
    xs = np.where(inside, x, 0.0)
    return np.where(inside, sinc(xs) * sinc(xs / 3), 0.0)


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


@functools.lru_cache(maxsize=4096)
def pil_resample_coeffs(in_size: int, out_size: int, filter_name: str = "bicubic"):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (libImaging/Resample.c) for one axis: → (bounds int32 [out][2] =
    (first source index, tap count), kk int32 [out][ksize] fixed-point taps).  Same double arithmetic in the same order: the taps
    are summed sequentially, divided by the sum, scaled by 2^22 and truncated towards zero after ±0.5."""
    f, sup = _FILTERS[filter_name]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = sup * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation (values >= -support + 0.5)
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    t = np.arange(ksize, dtype=np.int64)[None, :]
    valid = t < xmax[:, None]
    w = f((t + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                                          # sequential sum, tap by tap, as the C loop does
        ww = ww + w[:, j]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.where(w < 0, np.trunc(-0.5 + w * (1 << _PRECISION_BITS)), np.trunc(0.5 + w * (1 << _PRECISION_BITS))).astype(np.int32)
    kk = np.where(valid, kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return np.ascontiguousarray(bounds), np.ascontiguousarray(kk)


def demo_max_side_size(w: int, h: int, max_side: int = 644):
    """eval/test_demo.py:67-73: scale so that the longer side becomes `max_side` (LANCZOS), sizes truncated like int()."""
    scale = max_side / max(w, h)
    return int(w * scale), int(h * scale)


def fetch_image_size(w: int, h: int, factor: int = 28, min_pixels: int = 4 * 28 * 28, max_pixels: int = 16384 * 28 * 28):
    """eval/test_demo.py:61 `process_vision_info(message)` → qwen_vl_utils.vision_process.fetch_image: the decoded RGB image is resized
    (PIL's default filter for RGB: BICUBIC) to the package's OWN smart_resize(height, width, factor=28, min_pixels=4 * 28 * 28,
    max_pixels=16384 * 28 * 28) BEFORE the demo's LANCZOS pass.  qwen_vl_utils is a third-party dependency the reference lists WITHOUT a version
    (setup.py:28) and that is not in the build container: this restates the package's published rule — which is NOT the HF processor's
    smart_resize: it clamps each rounded side to at least `factor` BEFORE the pixel-budget branches (`max(factor, round_by_factor(side))`) and
    refuses aspect ratios above 200 — so a 10 x 400 image becomes 392 x 28 here and would be 364 x 28 under the HF rule.  Parity
    for this one step is unpinned against the package itself.  → (new_w, new_h)."""
    if max(h, w) / min(h, w) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(h, w) / min(h, w)}")
    h_bar = max(factor, round(h / factor) * factor)
    w_bar = max(factor, round(w / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((h * w) / max_pixels)
        h_bar = max(factor, math.floor(h / beta / factor) * factor)
        w_bar = max(factor, math.floor(w / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (h * w))
        h_bar = math.ceil(h * beta / factor) * factor
        w_bar = math.ceil(w * beta / factor) * factor
    return w_bar, h_bar


def eval_min_side_size(w: int, h: int, min_side: int = 28):
    """eval/evaluation_scripts/utils.py:205-218: images with a side below 28 px are enlarged (LANCZOS) so the short side is 28."""
    if w >= min_side and h >= min_side:
        return w, h
    if w < h:
        return min_side, int(h * (min_side / w))
    return int(w * (min_side / h)), min_side


class ImageFrontEnd:
    def __init__(self, device, patch: int = 14, merge: int = 2, temporal: int = 2, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280, dtype=torch.bfloat16, resize: str = "gpu", pre_resize=None):
        """resize: "gpu" (default: Pillow's resampler on the device, bit-exact) or "pil" (host PIL, the reference's own call).
        pre_resize: None, "demo644" (eval/test_demo.py:67-73) or "min28" (eval/evaluation_scripts/utils.py:205-218) — the callers'
        LANCZOS pass in front of the processor, applied to the image as handed to this class — or "demo" = the demo's WHOLE front end
        from the decoded file on: qwen_vl_utils' fetch_image resize (test_demo.py:61; BICUBIC to a multiple of 28, `fetch_image_size`: the
        package is absent and un-versioned in the reference, its published rule is restated, unpinned) → the LANCZOS max-side-644 pass →
        the processor's own smart_resize.  Every pass runs through the same byte-exact Pillow resampler."""
        self.device, self.patch, self.merge, self.temporal = device, patch, merge, temporal
        self.min_pixels, self.max_pixels, self.dtype = min_pixels, max_pixels, dtype
        self.resize, self.pre_resize = resize, pre_resize
        self.lut = torch.from_numpy(normalize_lut()).to(device)
        self._tables = {}

    # ---- size rules (host integers)
    def _plan_sizes(self, w, h):
        """→ list of (out_w, out_h, filter) passes applied in order."""
        steps = []
        if self.pre_resize == "demo":
            nw, nh = fetch_image_size(w, h)
            if (nw, nh) != (w, h):
                steps.append((nw, nh, "bicubic"))
                w, h = nw, nh
        if self.pre_resize in ("demo644", "demo"):
            nw, nh = demo_max_side_size(w, h)
            steps.append((nw, nh, "lanczos"))
            w, h = nw, nh
        elif self.pre_resize == "min28":
            nw, nh = eval_min_side_size(w, h)
            if (nw, nh) != (w, h):
                steps.append((nw, nh, "lanczos"))
                w, h = nw, nh
        rh, rw = smart_resize(h, w, self.patch * self.merge, self.min_pixels, self.max_pixels)
        if (rh, rw) != (h, w):
            steps.append((rw, rh, "bicubic"))
        return steps

    def _table(self, in_size, out_size, flt):
        key = (in_size, out_size, flt)
        t = self._tables.get(key)
        if t is None:
            b, k = pil_resample_coeffs(in_size, out_size, flt)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
            if len(self._tables) >= 1024:
                self._tables.pop(next(iter(self._tables)))
            self._tables[key] = t
        return t

    def resize_device(self, img: torch.Tensor, out_w: int, out_h: int, flt: str) -> torch.Tensor:
        """Pillow's two-pass resample of a (H, W, 3) uint8 device image: horizontal pass, then vertical (each skipped if the size
        along it does not change, as ImagingResample does)."""
        H, W, C = img.shape
        cur = img
        if out_w != W:
            b, k = self._table(W, out_w, flt)
            nxt = torch.empty((H, out_w, C), dtype=torch.uint8, device=img.device)
            cur = ops.resample_pass_u8(cur, nxt, b, k, horizontal=True)
        if out_h != H:
            b, k = self._table(H, out_h, flt)
            nxt = torch.empty((out_h, cur.shape[1], C), dtype=torch.uint8, device=img.device)
            cur = ops.resample_pass_u8(cur, nxt, b, k, horizontal=False)
        return cur

    def resize_host(self, image):
        """PIL image or (H, W, 3) uint8 array → uint8 array at the smart_resize size (PIL bicubic, as the processor does)."""
        if hasattr(image, "convert"):                              # PIL: the processor's do_convert_rgb (modes L / P / RGBA / CMYK ...)
            image = image.convert("RGB")
        arr = np.asarray(image)
        if arr.ndim == 2:                                          # gray ndarray → 3 equal channels
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        if arr.ndim != 3 or arr.shape[2] < 3 or arr.dtype != np.uint8:
            raise ValueError(f"image must be PIL or a (H, W[, >=3]) uint8 array, got shape {arr.shape} dtype {arr.dtype}")
        from PIL import Image
        pil = Image.fromarray(np.ascontiguousarray(arr[..., :3]))
        for (ow, oh, flt) in self._plan_sizes(arr.shape[1], arr.shape[0]):
            pil = pil.resize((ow, oh), resample=Image.LANCZOS if flt == "lanczos" else Image.BICUBIC)
        return np.asarray(pil)

    def _to_rgb_array(self, image):
        if hasattr(image, "convert"):                              # PIL: the processor's do_convert_rgb (modes L / P / RGBA / CMYK ...)
            image = image.convert("RGB")
        arr = np.asarray(image)
        if arr.ndim == 2:
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        if arr.ndim != 3 or arr.shape[2] < 3 or arr.dtype != np.uint8:
            raise ValueError(f"image must be PIL or a (H, W[, >=3]) uint8 array, got shape {arr.shape} dtype {arr.dtype}")
        return np.array(arr[..., :3], dtype=np.uint8, order="C")    # writable copy (torch.from_numpy)

    def __call__(self, images: List) -> Tuple[torch.Tensor, torch.Tensor]:
        """→ (pixel_values (ΣP, 1176) on the device, image_grid_thw (B, 3) int64 on the host)."""
        if self.resize == "pil":
            imgs = [torch.from_numpy(np.array(self.resize_host(im), dtype=np.uint8, order="C")).to(self.device, non_blocking=True) for im in images]
        else:
            imgs = []
            for im in images:
                if isinstance(im, torch.Tensor):                   # (H, W, 3) uint8, host (ideally pinned: the copy is then asynchronous) or device
                    if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                        raise ValueError(f"tensor images must be (H, W, 3) uint8, got {tuple(im.shape)} {im.dtype}")
                    cur = im.to(self.device, non_blocking=True).contiguous()
                    h0, w0 = int(im.shape[0]), int(im.shape[1])
                else:
                    arr = self._to_rgb_array(im)
                    cur = torch.from_numpy(arr).to(self.device, non_blocking=True)
                    h0, w0 = arr.shape[0], arr.shape[1]
                for (ow, oh, flt) in self._plan_sizes(w0, h0):
                    cur = self.resize_device(cur, ow, oh, flt)
                imgs.append(cur)
        grids = [[1, a.shape[0] // self.patch, a.shape[1] // self.patch] for a in imgs]
        total = sum(g[1] * g[2] for g in grids)
        row = 3 * self.temporal * self.patch * self.patch
        out = torch.empty((total, row), device=self.device, dtype=self.dtype)
        o = 0
        for img, g in zip(imgs, grids):
            n = g[1] * g[2]
            ops.patchify_normalize(img, self.lut, out[o:o + n], self.patch, self.merge, self.temporal)
            o += n
        return out, torch.tensor(grids, dtype=torch.int64)
