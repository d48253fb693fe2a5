"""Image front-end: oracle restatement and host logic vs fixtures generated with the HF Qwen2-VL PIL
image processor (tests/golden/make_golden_pre.py)."""
import os

import numpy as np

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
