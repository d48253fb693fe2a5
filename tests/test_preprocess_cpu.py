"""Image front-end: oracle restatement and host logic vs fixtures generated with the HF Qwen2-VL PIL
image processor (tests/golden/make_golden_pre.py)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return np.load(os.path.join(HERE, "golden", "preprocess.npz"))


def test_oracle_patchify_normalize_bit_exact():
    import padt_oracle as O
    z = load()
    assert np.allclose(O.IMAGE_MEAN, z["mean"], rtol=0, atol=0) and np.allclose(O.IMAGE_STD, z["std"], rtol=0, atol=0)
    assert O.RESCALE == float(z["rescale"])
    rows, grids = [], []
    for k in ("img0", "img1"):
        r, gh, gw = O.patchify_normalize(z[k])
        rows.append(r)
        grids.append([1, gh, gw])
    assert grids == z["grid"].tolist()
    assert np.array_equal(np.concatenate(rows), z["pix"])            # float32, bit for bit


def test_smart_resize_and_lut_match_processor():
    import padt_oracle as O
    from padt_amd import preprocess as P
    z = load()
    mn, mx = (int(v) for v in z["min_max"])
    for (h, w), exp in zip(z["sizes"].tolist(), z["smart"].tolist()):
        assert list(O.smart_resize(h, w, 28, mn, mx)) == exp and list(P.smart_resize(h, w, 28, mn, mx)) == exp
    lut = P.normalize_lut()
    img = z["img1"]
    via_lut = lut[np.arange(3)[:, None, None], img.transpose(2, 0, 1)]
    ref, _, _ = O.patchify_normalize(img)
    C, H, W = via_lut.shape
    pt = via_lut.reshape(C, H // 28, 2, 14, W // 28, 2, 14).transpose(1, 4, 2, 5, 0, 3, 6)
    pt = np.broadcast_to(pt[:, :, :, :, :, None], (*pt.shape[:5], 2, 14, 14)).reshape(-1, 1176)
    assert np.array_equal(pt, ref)
    try:
        P.smart_resize(1, 500)
        assert False
    except ValueError:
        pass


RESIZE_CASES = [(480, 640, 476, 644), (333, 500, 644, 448), (100, 37, 28, 56), (640, 640, 644, 644), (1200, 900, 644, 476), (20, 300, 28, 420),
                (427, 640, 420, 644)]


def test_pil_resample_restatement_and_coefficient_tables_are_byte_exact_with_pillow():
    """The resize of the front-end is Pillow's ImagingResample.  Three statements of it must agree byte for byte with PIL.Image.resize
    on random images (down- and up-scaling, both filters the reference uses): the oracle's loop restatement, and the product's
    vectorised coefficient tables driven through the same integer pass the HIP kernel runs (csrc/resize.hip)."""
    import padt_oracle as O
    from PIL import Image
    from padt_amd.preprocess import pil_resample_coeffs
    rng = np.random.default_rng(0)
    for (H, W, oh, ow) in RESIZE_CASES:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for name, pf in (("bicubic", Image.BICUBIC), ("lanczos", Image.LANCZOS)):
            ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=pf))
            assert np.array_equal(O.pil_resample(img, ow, oh, name), ref), (H, W, oh, ow, name)
            cur = img
            for axis, (n_in, n_out) in ((1, (W, ow)), (0, (H, oh))):          # horizontal pass, then vertical
                if n_in == n_out:
                    continue
                b, kk = pil_resample_coeffs(n_in, n_out, name)
                assert b.dtype == np.int32 and kk.dtype == np.int32 and b.shape == (n_out, 2)
                src = np.moveaxis(cur, axis, 0).astype(np.int64)
                out = np.stack([np.clip(((1 << 21) + np.tensordot(kk[i, :b[i, 1]].astype(np.int64), src[b[i, 0]: b[i, 0] + b[i, 1]], axes=(0, 0))) >> 22, 0, 255)
                                for i in range(n_out)]).astype(np.uint8)
                cur = np.moveaxis(out, 0, axis)
            assert np.array_equal(cur, ref), (H, W, oh, ow, name, "coefficient tables")


def test_caller_size_rules():
    from padt_amd import preprocess as P
    assert P.demo_max_side_size(1280, 720) == (644, 362) and P.demo_max_side_size(333, 500) == (428, 644)      # test_demo.py:67-73
    assert P.eval_min_side_size(640, 480) == (640, 480) and P.eval_min_side_size(20, 300) == (28, 420)        # utils.py:205-218
    assert P.eval_min_side_size(300, 20) == (420, 28)


def test_oracle_front_end_with_resize_matches_hf_processor_bit_exact():
    """Images that are not at their smart_resize size: smart_resize → pil_resample (bicubic) → patchify_normalize == the HF processor's
    own pixel_values (fixture keys raw0/raw1 → pix_raw), float32 bit for bit."""
    import padt_oracle as O
    z = load()
    mn, mx = (int(v) for v in z["min_max"])
    rows, grids = [], []
    for k in ("raw0", "raw1"):
        im = z[k]
        rh, rw = O.smart_resize(im.shape[0], im.shape[1], 28, mn, mx)
        r, gh, gw = O.patchify_normalize(O.pil_resample(im, rw, rh, "bicubic"))
        rows.append(r)
        grids.append([1, gh, gw])
    assert grids == z["grid_raw"].tolist() and np.array_equal(np.concatenate(rows), z["pix_raw"])


def test_demo_front_end_plan_is_the_callers_pil_sequence():
    """pre_resize="demo" = eval/test_demo.py from the decoded file on: qwen_vl_utils.fetch_image's resize (BICUBIC to smart_resize(h, w, 28,
    4 * 28^2, 16384 * 28^2): restated, the package is absent) → LANCZOS to max side 644 (test_demo.py:67-73) → the processor's own
    smart_resize (BICUBIC).  The size plan, and the host-PIL path of the front end against the same three PIL calls written out."""
    from PIL import Image
    from padt_amd import preprocess as P
    assert P.fetch_image_size(640, 427) == (644, 420) and P.fetch_image_size(333, 500) == (336, 504) and P.fetch_image_size(20, 96) == (28, 140)
    assert P.fetch_image_size(644, 644) == (644, 644) and P.fetch_image_size(5000, 4000) == (4004, 3192)      # 16384 * 28^2 pixels cap
    # the package's OWN rule, not the HF processor's: a side below 14 px is clamped to 28 BEFORE the pixel-budget branches —
    # 10 x 400: round(10 / 28) * 28 = 0 → 28, 400 → 392, 28 * 392 >= 4 * 28^2: done; the HF smart_resize scales both sides instead
    assert P.fetch_image_size(400, 10) == (392, 28) and P.fetch_image_size(10, 400) == (28, 392)
    assert P.smart_resize(10, 400, 28, 4 * 28 * 28, 16384 * 28 * 28) == (28, 364)
    with pytest.raises(ValueError, match="aspect ratio"):
        P.fetch_image_size(4030, 20)
    fe = P.ImageFrontEnd("cpu", dtype=torch.float32, resize="pil", pre_resize="demo")
    assert fe._plan_sizes(640, 427) == [(644, 420, "bicubic"), (644, 420, "lanczos")]          # already a multiple of 28 afterwards: no third pass
    assert fe._plan_sizes(1700, 1100) == [(1708, 1092, "bicubic"), (644, 411, "lanczos"), (644, 420, "bicubic")]
    rng = np.random.default_rng(3)
    for (h, w) in [(427, 640), (1100, 1700), (96, 20)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pil = Image.fromarray(img)
        nw, nh = P.fetch_image_size(w, h)
        pil = pil.resize((nw, nh))                                                             # PIL's default for RGB: BICUBIC (fetch_image)
        mw, mh = P.demo_max_side_size(nw, nh)
        pil = pil.resize((mw, mh), Image.Resampling.LANCZOS)                                   # test_demo.py:73
        rh, rw = P.smart_resize(mh, mw, 28, fe.min_pixels, fe.max_pixels)
        pil = pil.resize((rw, rh), resample=Image.BICUBIC)                                     # the HF processor
        assert np.array_equal(fe.resize_host(img), np.asarray(pil)), (h, w)
