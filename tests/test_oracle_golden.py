"""Pin the CPU oracle against vectors produced by the reference itself (tests/golden/make_golden.py).

Tolerances: integer tables bit-exact; floating point 2e-5 absolute (both sides are fp32 CPU, only the order of a few
reductions differs — e.g. softmax over a padded row vs over a packed segment).
"""
import numpy as np
import pytest
import torch

import padt_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


def tiny_cfg():
    return O.OracleConfig(
        vocab_size=512, hidden_size=64, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=16, intermediate_size=96,
        mrope_section=(2, 3, 3), vit_hidden=32, vit_depth=4, vit_heads=2, vit_intermediate=48, patch_size=2,
        temporal_patch_size=2, in_channels=3, window_size=16, fullatt_block_indexes=(1, 3), lora_r=8,
        dec_hidden=32, dec_heads=2, dec_intermediate=48, image_token_id=500, vision_start_token_id=501,
        eos_token_id=1, pad_token_id=0)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    z = np.load(f"{golden_dir}/tiny_e2e.npz", allow_pickle=False)
    w = {k[3:]: _t(z[k]) for k in z.files if k.startswith("w::")}
    return z, w


def test_index_tables_bit_exact(golden_dir):
    z = np.load(f"{golden_dir}/index_tables.npz")
    for name in ("46x46", "46x30", "10x12", "batch"):
        grid = _t(z[f"{name}.grid"])
        wi, cu = O.window_index(grid, 2, 112, 14)
        assert torch.equal(wi, _t(z[f"{name}.window_index"])), name
        assert cu == z[f"{name}.cu_window"].tolist(), name
        assert torch.equal(O.vision_position_ids(grid, 2), _t(z[f"{name}.pos_ids"])), name
    # 46x46 → 36 windows of 64x25, 48x10, 36x1 patches
    _, cu = O.window_index(torch.tensor([[1, 46, 46]]), 2, 112, 14)
    lens = np.diff(np.array(cu))
    assert sorted(lens.tolist()) == sorted([64] * 25 + [48] * 10 + [36])


def test_vit_and_prototypes(tiny):
    z, w = tiny
    cfg = tiny_cfg()
    low, high, (cos, sin) = O.vit_forward(w, cfg, _t(z["pixel_values"]), _t(z["grid"]))
    assert torch.allclose(low, _t(z["image_embeds"]), atol=2e-5)
    assert torch.allclose(high, _t(z["high_res"]), atol=2e-5)
    assert torch.allclose(cos, _t(z["cos"]), atol=1e-6) and torch.allclose(sin, _t(z["sin"]), atol=1e-6)
    proto = O.prototypes(w, cfg, low)
    assert torch.allclose(proto, _t(z["proto"]), atol=2e-5)
    lm = O.logit_mask(cfg, _t(z["grid"]), cfg.vocab_size + proto.shape[0])
    assert torch.equal(lm, _t(z["logit_mask"]))


def test_prefill_and_teacher_forced_decode(tiny):
    z, w = tiny
    cfg = tiny_cfg()
    ids, am = _t(z["input_ids_global"]), _t(z["attention_mask"])
    logits, hidden, st = O.prefill(w, cfg, ids, am, _t(z["pixel_values"]), _t(z["grid"]))
    valid = am.bool()
    assert torch.equal(st.rope_deltas, _t(z["rope_deltas"]))
    assert torch.allclose(hidden[valid], _t(z["prefill_hidden"])[valid], atol=2e-5)
    ref_last = _t(z["prefill_logits_last"])
    fin = torch.isfinite(ref_last)
    assert torch.equal(torch.isfinite(logits[:, -1]), fin)
    assert torch.allclose(logits[:, -1][fin], ref_last[fin], atol=5e-5)
    comp = _t(z["comp_global"])
    ref_steps, ref_hidden = _t(z["step_logits"]), _t(z["last_hidden"])
    assert torch.equal(logits[:, -1].argmax(-1), ref_steps[:, 0].argmax(-1))
    assert torch.allclose(hidden[:, -1], ref_hidden[:, 0], atol=2e-5)
    for s in range(comp.shape[1] - 1):
        lg, hd = O.decode_step(w, cfg, st, comp[:, s:s + 1])
        r = ref_steps[:, s + 1]
        f = torch.isfinite(r)
        assert torch.allclose(lg[:, -1][f], r[f], atol=5e-5), s
        assert torch.equal(lg[:, -1].argmax(-1), r.argmax(-1)), s
        assert torch.allclose(hd[:, -1], ref_hidden[:, s + 1], atol=2e-5), s


def test_vl_decode(tiny):
    z, w = tiny
    cfg = tiny_cfg()
    feats = [[_t(z["feat_0_0"])], [_t(z["feat_1_0"]), _t(z["feat_1_1"])]]
    out = O.vl_decode(w, cfg, feats, _t(z["proto"]), _t(z["high_res"]), _t(z["grid"]), (_t(z["cos"]), _t(z["sin"])))
    assert torch.allclose(out["pred_boxes"], _t(z["pred_boxes"]), atol=2e-5)
    assert torch.allclose(out["pred_score"], _t(z["pred_score"]), atol=2e-5)
    assert torch.allclose(out["pred_mask"], _t(z["pred_mask"]), atol=5e-5)
    assert torch.equal(out["pred_mask_valid_hw"][0], _t(z["valid_h"]))
    assert torch.equal(out["pred_mask_valid_hw"][1], _t(z["valid_w"]))
    assert out["sample_idx"] == z["sample_idx"].tolist()
    empty = O.vl_decode(w, cfg, [[], []], _t(z["proto"]), _t(z["high_res"]), _t(z["grid"]), (_t(z["cos"]), _t(z["sin"])))
    assert empty["pred_boxes"].shape == (0, 4) and empty["pred_mask"].shape == (0, 8, 8) and empty["pred_mask_valid_hw"] == ()


def _seeded(shape, name, scale, jitter_one=False):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = scale * torch.randn(shape, generator=g)
    return 1 + t if jitter_one else t


def test_real_shape_vit_block(golden_dir):
    z = np.load(f"{golden_dir}/real_vit_block.npz")
    cfg = O.OracleConfig()
    bw = {}
    for k, shp in O.weight_shapes(cfg).items():
        if not k.startswith("visual.blocks.0."):
            continue
        if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            bw[k] = _seeded(shp, k, 0.1, True)
        else:
            bw[k] = _seeded(shp, k, 0.02)
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    grid = torch.tensor([[1, 46, 46]])
    x = torch.randn(2116, 1280, generator=g)
    wi, cu_win = O.window_index(grid, 2, 112, 14)
    c, s = O.vit_rotary(cfg, grid, wi)
    rows = _t(z["rows"])
    yw = O.vit_block(bw, "visual.blocks.0.", cfg, x, cu_win, c, s)
    yf = O.vit_block(bw, "visual.blocks.0.", cfg, x, [0, 2116], c, s)
    assert torch.allclose(yw[rows], _t(z["y_win"]), atol=1e-5)
    assert torch.allclose(yf[rows], _t(z["y_full"]), atol=1e-5)


def test_real_shape_decoder(golden_dir):
    z = np.load(f"{golden_dir}/real_decoder.npz")
    cfg = O.OracleConfig()
    dw = {}
    for k, shp in O.weight_shapes(cfg).items():
        if not k.startswith("vl_decoder."):
            continue
        if O._is_norm_weight(k):
            dw[k] = _seeded(shp, k, 0.1, True)
        elif k.endswith("bias"):
            dw[k] = _seeded(shp, k, 0.02)
        else:
            dw[k] = _seeded(shp, k, 0.03)
    g = torch.Generator().manual_seed(123)
    _ = torch.randn(2116, 1280, generator=g)                      # same stream position as make_golden.py
    _ = torch.randint(0, 2116, (48,), generator=g)
    grids = torch.tensor([[1, 46, 30], [1, 8, 8]])
    Ps = [46 * 30, 64]
    low = torch.randn(sum(Ps) // 4, 2048, generator=g)
    high = torch.randn(sum(Ps), 1280, generator=g)
    wi, _ = O.window_index(grids, 2, 112, 14)
    c, s = O.vit_rotary(cfg, grids, wi)
    feats = [[torch.randn(5, 2048, generator=g), torch.randn(2, 2048, generator=g)], [torch.randn(4, 2048, generator=g)]]
    out = O.vl_decode(dw, cfg, feats, low, high, grids, (c, s))
    assert torch.allclose(out["pred_boxes"], _t(z["pred_boxes"]), atol=1e-5)
    assert torch.allclose(out["pred_score"], _t(z["pred_score"]), atol=1e-5)
    assert list(out["pred_mask"].shape) == z["mask_shape"].tolist()
    assert torch.allclose(out["pred_mask"].flatten()[_t(z["mask_idx"])], _t(z["mask_vals"]), atol=5e-5)
    assert out["sample_idx"] == z["sample_idx"].tolist()


def test_real_width_llm_layer(golden_dir):
    """One LLM layer at PaDT_Pro_3B width + final norm, prefill over the real 577-token prompt layout and two cached decode steps,
    against HF's Qwen2_5_VLTextModel (fixture: tests/golden/make_golden_llm.py)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mk_llm", os.path.join(golden_dir, "make_golden_llm.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(f"{golden_dir}/real_llm_layer.npz")
    cfg = mk.cfg_1layer()
    w = mk.layer_weights(cfg)
    ids, grid = mk.prompt(cfg)
    pos, deltas = O.rope_index(cfg, ids, grid, torch.ones_like(ids))
    assert torch.equal(deltas, _t(z["deltas"]))
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x = torch.randn(1, 577, cfg.hidden_size, generator=g)
    xd = torch.randn(2, 1, 1, cfg.hidden_size, generator=g)
    kv = O.KVCache(1)
    h = O.llm_forward(w, cfg, x, pos, torch.ones(1, 577, dtype=torch.long), kv)
    assert torch.allclose(h[0, _t(z["rows"])], _t(z["h_rows"]), atol=2e-5) and torch.allclose(h[0, -1], _t(z["h_last"]), atol=2e-5)
    for t_, key in enumerate(("step0", "step1")):
        p = (torch.tensor([[577 + t_]]) + deltas).view(1, 1, 1).expand(3, 1, 1)
        o = O.llm_forward(w, cfg, xd[t_], p, torch.ones(1, 578 + t_, dtype=torch.long), kv)
        assert torch.allclose(o[0, 0], _t(z[key]), atol=2e-5), key
