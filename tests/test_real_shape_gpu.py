"""HIP path at REAL PaDT_Pro_3B shapes against outputs of the reference itself (tests/golden/real_*.npz were produced by running
the reference's own `custom_visual_forward` block and `vl_decode` / `PaDTDecoder`, tests/golden/make_golden.py): one ViT block at
2116 x 1280 (window and full attention) and the 98 M-parameter PaDT decoder with 3 objects over 2 images.

The reference ran in fp32; the HIP path multiplies 16-bit operands (fp16 by default since round 4, bf16 as the A/B variant) → tolerances
are sized by the operand type and written at each assert."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _seeded(shape, name, scale, jitter_one=False):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = scale * torch.randn(shape, generator=g)
    return 1 + t if jitter_one else t


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


@pytest.mark.parametrize("operands", ["fp16", "bf16"])
def test_vit_block_real_shape_against_reference_output(golden_dir, operands):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import padt_oracle as O
    from padt_amd.config import VisionConfig
    from padt_amd.vision import VisionEncoder
    from padt_amd.weights import prepare_weights, synthetic_state_dict
    z = np.load(f"{golden_dir}/real_vit_block.npz")
    base = padt_amd.small_test_config()
    cfg = dataclasses.replace(base, vision_config=VisionConfig(hidden_size=1280, depth=1, num_heads=16, intermediate_size=3420,
                                                               fullatt_block_indexes=(), out_hidden_size=base.hidden_size))
    sd = synthetic_state_dict(cfg, seed=0, device="cpu")
    ocfg = O.OracleConfig()
    for k, shp in O.weight_shapes(ocfg).items():                    # the seeded block weights the reference ran with
        if k.startswith("visual.blocks.0."):
            sd[k] = _seeded(shp, k, 0.1, True) if (k.endswith("norm1.weight") or k.endswith("norm2.weight")) else _seeded(shp, k, 0.02)
    W = prepare_weights(sd, cfg, device="cuda", operands=operands)
    enc = VisionEncoder(cfg, W, "cuda")
    grid = torch.tensor([[1, 46, 46]])
    plan = enc.plan(grid)
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x0 = torch.randn(2116, 1280, generator=g).to(W.op16).cuda()
    rows = _t(z["rows"])
    for full, key in ((False, "y_win"), (True, "y_full")):
        x = x0.clone()
        bufs = (torch.empty(2116, device="cuda", dtype=torch.float32), torch.empty(2116, 3 * 1280, device="cuda", dtype=W.op16),
                torch.empty_like(x), torch.empty(2116, W.vit_ipad, device="cuda", dtype=W.op16))
        enc.block(0, x, plan, *bufs, force_full=full)
        mx, rms = rel(x[rows.cuda()], _t(z[key]))
        print(f"\\n[real ViT block {key}, {operands} operands and stream] vs reference: rel max {mx:.3e} rms {rms:.3e}")
        # 16-bit weights + activations + residual stream (this call runs the block without the fp32 stream), 7 kernels deep: bf16 measured
        # 3.2e-3 rms; fp16 carries 3 more mantissa bits
        lim = (1.5e-2, 6e-2) if operands == "bf16" else (3e-3, 1.2e-2)
        assert rms < lim[0] and mx < lim[1], f"{key}: rel err max {mx:.3e} rms {rms:.3e}"


@pytest.mark.parametrize("hp", [True, False])
def test_padt_decoder_real_shape_against_reference_output(golden_dir, hp, monkeypatch):
    """hp=True (default build of the model): split-precision decoder — the north star's 1e-3 on box coordinates / mask logits.
    hp=False (PADT_DECODER_HP=0): plain bf16 activation storage, kept as the fast variant with its measured, documented distance."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import padt_oracle as O
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    monkeypatch.setenv("PADT_DECODER_HP", "1" if hp else "0")
    z = np.load(f"{golden_dir}/real_decoder.npz")
    base = padt_amd.small_test_config(layers=1, vit_depth=1)
    cfg = dataclasses.replace(base, hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, intermediate_size=256,
                              vision_config=dataclasses.replace(base.vision_config, out_hidden_size=2048),
                              vl_decoder={"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "use_mask_loss": True})
    sd = synthetic_state_dict(cfg, seed=0, device="cpu")
    ocfg = O.OracleConfig()
    for k, shp in O.weight_shapes(ocfg).items():
        if k.startswith("vl_decoder."):
            sd[k] = _seeded(shp, k, 0.1, True) if O._is_norm_weight(k) else _seeded(shp, k, 0.02 if k.endswith("bias") else 0.03)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    assert model.W.dec_hp == hp
    g = torch.Generator().manual_seed(123)
    _ = torch.randn(2116, 1280, generator=g)                       # same stream position as make_golden.py
    _ = torch.randint(0, 2116, (48,), generator=g)
    grids = torch.tensor([[1, 46, 30], [1, 8, 8]])
    Ps = [46 * 30, 64]
    low = torch.randn(sum(Ps) // 4, 2048, generator=g)
    high = torch.randn(sum(Ps), 1280, generator=g)
    wi, _ = O.window_index(grids, 2, 112, 14)
    c, s = O.vit_rotary(ocfg, grids, wi)
    feats = [[torch.randn(5, 2048, generator=g), torch.randn(2, 2048, generator=g)], [torch.randn(4, 2048, generator=g)]]
    bf = torch.bfloat16
    out = model.vl_decode([[f.to(bf).cuda() for f in fs] for fs in feats], low.to(bf).cuda(), high.to(bf).cuda(), grids,
                          (c.cuda(), s.cuda()))
    assert out["sample_idx"] == z["sample_idx"].tolist() and list(out["pred_mask"].shape) == z["mask_shape"].tolist()
    assert torch.equal(out["pred_mask_valid_hw"][0].cpu(), _t(z["valid_h"])) and torch.equal(out["pred_mask_valid_hw"][1].cpu(), _t(z["valid_w"]))
    # (1) against the reference's fp32 run on its fp32 weights / inputs: dominated by rounding those operands to bf16 — which the
    #     reference's own GPU path (torch_dtype=bfloat16, test_demo.py:22) does too; ~1.4e-2 on box coordinates with these
    #     random N(0, 0.03^2) weights, independent of the activation precision
    db = (out["pred_boxes"].float().cpu() - _t(z["pred_boxes"])).abs().max().item()
    ds = (out["pred_score"].float().cpu() - _t(z["pred_score"])).abs().max().item()
    mx, rms = rel(out["pred_mask"].flatten()[_t(z["mask_idx"]).cuda()], _t(z["mask_vals"]))
    print(f"\n[real PaDT decoder hp={hp}] vs reference fp32 (fp32 operands): box |d|max {db:.3e} score |d|max {ds:.3e} mask rel max {mx:.3e} rms {rms:.3e}")
    assert db < 4e-2 and ds < 0.15 * (abs(z["pred_score"]).max() + 1) and rms < 6e-2
    # (2) against the oracle (which test_oracle_golden pins to that same reference output to 1e-5) fed the SAME bf16-representable
    #     weights and inputs — "the reference CPU path on the same inputs" of the north star: isolates the kernels' own arithmetic
    r = lambda t_: t_.to(bf).float()
    dwb = {k: r(v) for k, v in sd.items() if k.startswith("vl_decoder.")}
    odec = O.vl_decode(dwb, ocfg, [[r(f) for f in fs] for fs in feats], r(low), r(high), grids, (c, s))
    db2 = (out["pred_boxes"].float().cpu() - odec["pred_boxes"]).abs().max().item()
    ds2 = (out["pred_score"].float().cpu() - odec["pred_score"]).abs().max().item()
    mx2, rms2 = rel(out["pred_mask"], odec["pred_mask"])
    dm2 = (out["pred_mask"].float().cpu() - odec["pred_mask"]).abs().max().item()
    print(f"[real PaDT decoder hp={hp}] vs oracle on the same operands: box |d|max {db2:.3e} score |d|max {ds2:.3e} mask |d|max {dm2:.3e} "
          f"(logit |max| {odec['pred_mask'].abs().max().item():.2f}) rel max {mx2:.3e} rms {rms2:.3e}")
    if hp:
        # north star: box coords / mask logits within 1e-3 of the reference CPU path on the same inputs
        assert db2 < 1e-3, f"boxes differ from the oracle by {db2:.3e}"                    # boxes in [0, 1]
        assert ds2 < 1e-3 * (odec["pred_score"].abs().max().item() + 1)
        assert rms2 < 1e-3 and mx2 < 1e-3, f"mask logits: rel max {mx2:.3e} rms {rms2:.3e}"   # relative to the largest logit
    else:
        # ~35 kernel outputs are rounded to bf16 on the way (2^-9 relative each, random N(0, 0.03^2) weights of gain > 1): measured
        # 7.6e-3 on box coordinates, 1.0e-2 rms on mask logits — the distance the reference's own bf16 GPU path sits at
        assert db2 < 2e-2, f"boxes differ from the oracle by {db2:.3e}"
        assert ds2 < 5e-2 * (odec["pred_score"].abs().max().item() + 1)
        assert rms2 < 2e-2 and mx2 < 1e-1


def test_llm_layer_real_width_against_hf_text_model(golden_dir):
    """One LLM layer at PaDT_Pro_3B width (16/2 heads x 128, MLP 11008, mRoPE [16,24,24]) + final norm: packed prefill over the
    real 577-token prompt layout, then two decode steps through the fused decode kernels — against outputs of HF's
    Qwen2_5_VLTextModel (fixture from tests/golden/make_golden_llm.py).  The fixture feeds inputs_embeds directly: every prompt
    position has a unique token id (or is an image token), so the embedding table / image rows carry those vectors."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib.util
    import os
    import padt_amd
    from padt_amd import ops
    from padt_amd.llm import plan_prompt
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    spec = importlib.util.spec_from_file_location("mk_llm", os.path.join(golden_dir, "make_golden_llm.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(f"{golden_dir}/real_llm_layer.npz")
    ocfg = mk.cfg_1layer()
    w = mk.layer_weights(ocfg)
    ids, grid = mk.prompt(ocfg)
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x = torch.randn(1, 577, ocfg.hidden_size, generator=g)
    xd = torch.randn(2, 1, 1, ocfg.hidden_size, generator=g)
    base = padt_amd.small_test_config(layers=1, vit_depth=1)
    cfg = dataclasses.replace(base, vocab_size=ocfg.vocab_size, hidden_size=2048, num_attention_heads=16, num_key_value_heads=2,
                              intermediate_size=11008, image_token_id=ocfg.image_token_id,
                              vision_start_token_id=ocfg.vision_start_token_id, eos_token_id=ocfg.eos_token_id,
                              pad_token_id=ocfg.pad_token_id,
                              vision_config=dataclasses.replace(base.vision_config, out_hidden_size=2048))
    sd = synthetic_state_dict(cfg, seed=0, device="cpu")
    for k, v in w.items():
        sd[k] = v
    emb = torch.zeros(cfg.vocab_size, 2048)
    is_img = ids[0] == cfg.image_token_id
    emb[ids[0][~is_img]] = x[0][~is_img]                            # unique text ids → their input vectors
    emb[900], emb[901] = xd[0, 0, 0], xd[1, 0, 0]                   # the two decode-step inputs
    sd["model.embed_tokens.weight"] = emb
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    lm = model.lm
    plan = plan_prompt(cfg, ids, torch.ones_like(ids), grid, "cuda")
    sess = lm.session(1, 577 + 4, 529, 4)
    img = x[0][is_img].to(model.W.op16).cuda()
    hn = lm.prefill(plan, img, sess)
    rows = _t(z["rows"]).cuda()
    mx, rms = rel(hn[rows], _t(z["h_rows"]))
    print(f"\\n[real-width LLM layer, prefill] vs HF text model: rel max {mx:.3e} rms {rms:.3e}")
    assert rms < 1e-2 and mx < 5e-2
    # decode steps: position = 577 + t + rope_delta on all three axes, append slot = 577 + t
    delta = int(z["deltas"][0, 0])
    sess.vrt_off.zero_()
    for t_, key in enumerate(("step0", "step1")):
        sess.cur_tok.fill_(900 + t_)
        sess.slot.fill_(577 + t_)
        sess.lens.fill_(578 + t_)
        sess.pos3.fill_(577 + t_ + delta)
        sess.unfinished.fill_(1)
        sess.step.zero_()
        sess.step_kernels()
        mx, rms = rel(sess.hn[0], _t(z[key]))
        print(f"[real-width LLM layer, decode step {t_}] vs HF text model: rel max {mx:.3e} rms {rms:.3e}")
        assert rms < 1e-2 and mx < 5e-2
    assert int(sess.err) == 0


def test_vrt_head_real_vocabulary():
    """The logit head at the real table size (151936 text rows + 8 x 529 prototype rows, D = 2048, 8 samples), row-major and
    packed: logits vs a plain fp32 matmul, arg-max identical wherever the top-2 margin exceeds the bf16-input noise floor."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops
    V, NP, D, B = 151936, 8 * 529, 2048, 8
    g = torch.Generator(device="cuda").manual_seed(11)
    E = (torch.randn(V, D, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    P = (torch.randn(NP, D, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    h = torch.randn(B, D, device="cuda", generator=g).to(torch.bfloat16)
    off = torch.arange(0, NP + 1, 529, dtype=torch.int32, device="cuda")
    nblk = ops.vrt_head_nblk(V, NP)
    ref = h.float() @ torch.cat([E, P]).float().T
    allow = torch.zeros(B, V + NP, dtype=torch.bool, device="cuda")
    allow[:, :V] = True
    for b in range(B):
        allow[b, V + b * 529: V + (b + 1) * 529] = True
    ref = ref.masked_fill(~allow, float("-inf"))
    Ep = ops.pack_weight(E)
    hp = torch.zeros(16, D, device="cuda", dtype=torch.bfloat16)
    ops.pack_rows(h, hp, B, to_packed=True)
    for packed in (False, True):
        pv = torch.zeros(nblk * B, device="cuda")
        pi = torch.zeros(nblk * B, dtype=torch.int32, device="cuda")
        lg = torch.zeros(B, V + NP, device="cuda")
        if packed:
            ops.vrt_head(hp, E, P, off, pv, pi, 151645, logits=lg, table_packed=Ep, rows=B)
        else:
            ops.vrt_head(h, E, P, off, pv, pi, 151645, logits=lg)
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(lg), fin)
        err = (lg[fin] - ref[fin]).abs().max().item()
        assert err <= 2e-5 * ref[fin].abs().max().item() + 1e-5, f"logits differ by {err:.3e}"   # same bf16 operands, fp32 accumulate
        best = pv.view(nblk, B).max(0)
        tok = pi.view(nblk, B).gather(0, best.indices[None])[0].long()
        top2 = ref.topk(2, dim=-1)
        for b in range(B):
            if (top2.values[b, 0] - top2.values[b, 1]).item() > 1e-4:
                assert int(tok[b]) == int(top2.indices[b, 0])


def test_vrt_head_logits_do_not_depend_on_the_batch():
    """A sample's logit row (real table, packed path of the decode step) is bit-identical whether 8, 32, 64 or 128 rows share the launch
    (vrt_head_kernel<1 / 2 / 4 / 8>): merged decode groups pick the same tokens as a batch decoding alone."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops
    V, D = 151936, 2048
    g = torch.Generator(device="cuda").manual_seed(12)
    E = (torch.randn(V, D, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    Ep = ops.pack_weight(E)
    h64 = torch.randn(128, D, device="cuda", generator=g).to(torch.bfloat16)
    first = None
    for B in (8, 32, 64, 128):
        NP = B * 529
        P = (torch.randn(128 * 529, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(13)) * 0.02).to(torch.bfloat16)[:NP]
        off = torch.arange(0, NP + 1, 529, dtype=torch.int32, device="cuda")
        nblk = ops.vrt_head_nblk(V, NP)
        hp = torch.zeros((B + 15) // 16 * 16, D, device="cuda", dtype=torch.bfloat16)
        ops.pack_rows(h64[:B].contiguous(), hp, B, to_packed=True)
        pv = torch.zeros(nblk * ((B + 15) // 16 * 16), device="cuda")
        pi = torch.zeros(nblk * ((B + 15) // 16 * 16), dtype=torch.int32, device="cuda")
        lg = torch.zeros(B, V + NP, device="cuda")
        ops.vrt_head(hp, E, P, off, pv, pi, 151645, logits=lg, table_packed=Ep, rows=B)
        rows = torch.cat([lg[:8, :V], torch.stack([lg[b, V + b * 529: V + (b + 1) * 529] for b in range(8)])], 1)   # text + own VRT columns
        if first is None:
            first = rows.clone()
        assert torch.equal(rows, first), f"logit rows 0..7 change when {B} rows share the launch"


def test_full_depth_3b_teacher_forced_against_oracle():
    """The WHOLE PaDT_Pro_3B geometry (32 ViT blocks at 2116 x 1280, 36 LLM layers at D = 2048 / 16:2 heads / MLP 11008,
    151 936 + 529 table rows, 98 M-parameter decoder) for one 46 x 46 image, seeded random weights (bf16-representable, biases
    and norm jitter on), against the fp32 CPU oracle teacher-forced on the HIP tokens.  Two bars.  FLAT (the north star / the round-3
    verdict): box coordinates <= 1e-3, mask logits <= 5e-3 of their range or 1.5x the attributed floor.  DERIVED: the oracle is run a second
    time inside parity_util.operand_floor(the model's operand type) on the model's norm-folded weight images (every matmul operand on the
    activation side rounded to fp16, the folded q/k/v/gate/up matrices rounded like weights.py rounds them, everything else fp32 — the
    distance ANY implementation on these MFMA operands has; tests/studies/operand_attribution.py attributes it class by class) and
    every float quantity of the HIP path must be within 2x that floor on the same inputs.  (≈1 min of host CPU.)
      * generated ids: margin rule — a HIP token must be the oracle's arg-max unless the oracle's own top-2 margin is inside the
        logit noise the floor run shows at that step (2 x max |logit_floor - logit_fp32|); then it must be within that noise of the max;
      * ViT outputs, prototypes, per-step last-layer hidden rows: relative rms <= 2 x floor;
      * boxes <= 1e-3 and <= 2 x floor + 2e-4, score / mask logits <= 2 x floor end to end; the decoder alone on identical inputs 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    O = U.O
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    w = {k: v.float().cpu() for k, v in sd.items()}                 # the oracle's fp32 copy of the SAME bf16 values
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=77)
    assert ids.shape == (1, 577)
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 6))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid,
                         max_new_tokens=T, schedule=sched, do_sample=False)
    seq = out.sequences.cpu()
    toks = seq[:, 577:]
    assert toks.shape[1] == T and int(toks[0, -1]) == cfg.eos_token_id
    V = cfg.vocab_size
    torch.set_num_threads(min(32, os.cpu_count() or 8))         # 32: the fastest count for the oracle on the 256-core GPU box (bench.py calibrates the same)
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        op = model.W.op16
        wf = U.folded_weight_images(w, cfg, op)
        with U.operand_floor(op):
            fres = O.generate(wf, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        del wf
    t_or = time.perf_counter() - t0
    assert torch.equal(ores["sequences"], seq)
    st, fst = ores["state"], fres["state"]
    stream = ("fp32" if model.W.resid_f32 else "bf16") + " streams, " + str(op).replace("torch.", "") + " operands"
    K = 2.0 if model.W.resid_f32 else 6.0                           # PADT_RESID_F32=0 (round-2 arithmetic, kept for A/B runs) sits 2.2-3x above the floor
    # ---- ViT (32 blocks) and prototypes
    mx, rms_h = rel(out.past_high_res_image_embeds, st.high_res)
    mxp, rms_p = rel(out.past_image_embeds, st.proto)
    _, frms_h = rel(fst.high_res, st.high_res)
    _, frms_p = rel(fst.proto, st.proto)
    print(f"\n[full 3B, {stream}] oracle fp32 + operand floor {t_or:.1f} s on {torch.get_num_threads()} threads")
    print(f"[full 3B] ViT high_res rel rms {rms_h:.3e} (floor {frms_h:.3e}, x{rms_h / frms_h:.2f}); prototypes rel rms {rms_p:.3e} (floor {frms_p:.3e}, x{rms_p / frms_p:.2f})")
    assert rms_h < K * frms_h and rms_p < K * frms_p
    # ---- ids: margin rule with the measured logit noise
    n_tie = 0
    for t in range(T):
        lg, lf = ores["logits"][t][0], fres["logits"][t][0]
        fin = torch.isfinite(lg)
        noise = K * (lf[fin] - lg[fin]).abs().max().item()
        top2 = lg.topk(2).values
        chosen = lg[toks[0, t]].item()
        margin = (top2[0] - (top2[1] if torch.isfinite(top2[1]) else top2[0] - 1)).item()
        gap = top2[0].item() - chosen
        print(f"[full 3B] step {t} mode {sched[t]}: token {int(toks[0, t])}  oracle top-2 margin {margin:.3e}  logit noise bound (2 x floor) {noise:.3e} "
              f"= {noise / (lg[fin].abs().max().item() + 1e-30):.2%} of |logit|max  gap to oracle max {gap:.3e}")
        if margin > noise:
            assert gap == 0.0, f"step {t}: HIP token is not the oracle argmax (margin {margin:.3e} > noise {noise:.3e})"
        else:
            n_tie += 1
            assert gap <= noise
        if sched[t] == "v":
            assert V <= toks[0, t] < V + 529
    # ---- last-layer hidden rows (36 layers deep) that predicted each token
    hid = out.hidden_states.last_layer_rows().cpu().float()          # (T, 1, D)
    for t in range(T):
        mx, rms = rel(hid[t], ores["hidden"][t][:, -1])
        _, frms = rel(fres["hidden"][t][:, -1], ores["hidden"][t][:, -1])
        print(f"[full 3B] hidden step {t}: rel rms {rms:.3e} (floor {frms:.3e}, x{rms / frms:.2f})")
        assert rms < K * frms, f"hidden step {t}: rel rms {rms:.3e} vs floor {frms:.3e}"
    # ---- parse + decoder
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = V
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, 577:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False]))
    assert len(feats[0]) == 1 and feats[0][0].shape == (4, cfg.hidden_size)
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    with torch.no_grad():
        vf = lambda r: [[torch.cat([r["hidden"][t][0:1, -1] for t in range(2, 6)], 0)]]
        odec = O.vl_decode(w, oc, vf(ores), st.proto, st.high_res, grid, st.visual_pe)
        fdec = O.vl_decode(w, oc, vf(fres), fst.proto, fst.high_res, grid, fst.visual_pe)
        odec2 = O.vl_decode(w, oc, [[feats[0][0].cpu().float()]], out.past_image_embeds.cpu().float(),
                            out.past_high_res_image_embeds.cpu().float(), grid,
                            (out.past_visual_pe[0].cpu(), out.past_visual_pe[1].cpu()))
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    ds = (dec["pred_score"].cpu().float() - odec["pred_score"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    fdb = (fdec["pred_boxes"] - odec["pred_boxes"]).abs().max().item()
    fds = (fdec["pred_score"] - odec["pred_score"]).abs().max().item()
    fmx, frms = rel(fdec["pred_mask"], odec["pred_mask"])
    db2 = (dec["pred_boxes"].cpu().float() - odec2["pred_boxes"]).abs().max().item()
    mx2, rms2 = rel(dec["pred_mask"], odec2["pred_mask"])
    iou = O.box_iou_xywh(*[[float(b[0] - b[2] / 2), float(b[1] - b[3] / 2), float(b[2]), float(b[3])]
                           for b in (dec["pred_boxes"][0].cpu(), odec["pred_boxes"][0])])
    print(f"[full 3B] end to end: box |d|max {db:.3e} (floor {fdb:.3e}; IoU vs oracle {iou:.4f}) score |d|max {ds:.3e} (floor {fds:.3e}) "
          f"mask logits rel max {mx:.3e} (floor {fmx:.3e}, x{mx / fmx:.2f}) rms {rms:.3e} (floor {frms:.3e}); "
          f"decoder alone on identical inputs: box |d|max {db2:.3e} mask rel max {mx2:.3e} rms {rms2:.3e}; ties {n_tie}/{T}")
    assert db2 < 1e-3 and mx2 < 1e-3                                 # north star: decoder kernels on the same inputs
    assert db < K * fdb + 2e-4 and ds < K * fds + 1e-3 and mx < K * fmx and rms < K * frms
    if model.W.resid_f32:
        assert db < 1e-3 and iou > 0.99                              # north star on the box coordinates, end to end at full depth
    if op == torch.float16:
        # round 5: sized to what is MEASURED (3.7e-3 = x1.04 of the floor re-run above), not to a round number
        assert mx < 1.25 * fmx and mx < 4.6e-3, f"mask logits {mx:.3e} of their range (attributed floor {fmx:.3e})"


def test_3b_batch8_merged_runner_is_what_the_oracle_computes():
    """What bench.py times, asserted: PaDT_Pro_3B, batches of 8 different 46 x 46 images through PipelinedRunner(depth=2, merge=8)
    — 64-row decode steps, 8 x 529 prototypes per batch in one table, 16 REC tokens per image (VRT run of 5) — against
      (a) the un-merged path (rec_batch, one batch at a time): tokens, boxes, scores, mask logits BIT-identical for every batch;
      (b) the fp32 CPU oracle teacher-forced on the HIP tokens for all 8 samples of one batch (≈2 min of host CPU), and the oracle's
          operand-floor run (parity_util.operand_floor on the model's operand type and folded weight images) on the first 2 of those
          images (samples are independent; ≈1 min): every token by the margin rule with 2x the logit noise the floor run shows at that
          step (relative to the largest |logit|, worst of the 2 floor samples); FLAT: EVERY sample's box coordinates within the north
          star's 1e-3 (round 3 on bf16 operands: 1 of 8), every IoU > 0.995, mask logits within 5e-3 of their range or 1.5x the floor's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    O = U.O
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    B, T, NB = 8, 16, 10                                            # 10 batches: one full decode group of 8 + a partially filled one
    sched = U.rec_schedule(T, vrt_at=range(6, 11))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = [U.synthetic_batch(cfg, [[1, 46, 46]] * B, n_pre=15, n_post=33, seed=500 + i) for i in range(NB)]
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=8)
    res = []
    for grid, pix, ids, am in batches:
        res += runner.submit(ids.clone().cuda(), am.cuda(), pix.cuda().to(model.dtype), grid, max_new_tokens=T, schedule=sched)
    res += runner.flush()
    assert len(res) == NB
    # ---- (a) merged == un-merged, bit for bit
    for i in (0, 3, 7, 9):
        grid, pix, ids, am = batches[i]
        dec1, comp1, lab1, vrt1 = pipeline.rec_batch(model, proc, ids.clone().cuda(), am.cuda(), pix.cuda().to(model.dtype), grid,
                                                     max_new_tokens=T, schedule=sched)
        decm, compm, labm, vrtm = res[i]
        assert compm == comp1 and vrtm == vrt1, f"batch {i}: tokens differ between merged and un-merged decode"
        for k in ("pred_boxes", "pred_score", "pred_mask"):
            assert torch.equal(decm[k], dec1[k]), f"batch {i}: {k} differs between merged and un-merged decode"
    # ---- (b) one batch, all 8 samples, against the oracle
    grid, pix, ids, am = batches[0]
    out = model.generate(input_ids=proc.assign_to_global_vrt_id(ids.clone(), grid).cuda(), attention_mask=am.cuda(),
                         pixel_values=pix.cuda().to(model.dtype), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    toks = out.sequences[:, L:].cpu()
    decm = res[0][0]
    torch.set_num_threads(min(32, os.cpu_count() or 8))         # 32: the fastest count for the oracle on the 256-core GPU box (bench.py calibrates the same)
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        NF = 2                                                      # floor run on the first NF images (samples are independent)
        P1 = 46 * 46
        op = model.W.op16
        wf = U.folded_weight_images(w, cfg, op)
        with U.operand_floor(op):
            fres = O.generate(wf, oc, ids[:NF], am[:NF], pix[: NF * P1], grid[:NF], T, schedule=sched, collect_logits=True, force_tokens=toks[:NF])
        del wf
        vf = lambda r, nb: [[torch.cat([r["hidden"][t][b:b + 1, -1] for t in range(6, 11)], 0)] for b in range(nb)]
        ost, fst = ores["state"], fres["state"]
        odec = O.vl_decode(w, oc, vf(ores, B), ost.proto, ost.high_res, grid, ost.visual_pe)
        fdec = O.vl_decode(w, oc, vf(fres, NF), fst.proto, fst.high_res, grid[:NF], fst.visual_pe)
    print(f"\n[3B batch 8, {str(op).replace('torch.', '')} operands] oracle fp32 on 8 images + operand floor on {NF}, {T} tokens: {time.perf_counter() - t0:.1f} s")
    # logit noise of the floor run per step, as a fraction of the largest |logit| of that row (worst of the NF floor samples)
    frac = []
    for t in range(T):
        f_t = 0.0
        for b in range(NF):
            lf = fres["logits"][t][b]
            lg = ores["logits"][t][b][: lf.numel()]                # the floor run's table holds the first NF images' prototypes only
            fin = torch.isfinite(lg)
            f_t = max(f_t, (lf[fin] - lg[fin]).abs().max().item() / (lg[fin].abs().max().item() + 1e-30))
        frac.append(f_t)
    n_arg, n_tie = 0, 0
    for b in range(B):
        for t in range(T):
            lg = ores["logits"][t][b]
            fin = torch.isfinite(lg)
            noise = 2 * frac[t] * lg[fin].abs().max().item()       # 2 x the logit noise bf16 operands alone cause at this step
            top2 = lg.topk(2).values
            margin = (top2[0] - (top2[1] if torch.isfinite(top2[1]) else top2[0] - 1)).item()
            gap = top2[0].item() - lg[toks[b, t]].item()
            if gap == 0.0:
                n_arg += 1
            elif margin <= noise:
                n_tie += 1
                assert gap <= noise, f"sample {b} step {t}: gap {gap:.3e} beyond the noise bound {noise:.3e}"
            else:
                raise AssertionError(f"sample {b} step {t}: HIP token is not the oracle arg-max (margin {margin:.3e} > noise {noise:.3e})")
    db = (decm["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().amax(dim=1)
    ious = [O.box_iou_xywh(*[[float(x[0] - x[2] / 2), float(x[1] - x[3] / 2), float(x[2]), float(x[3])] for x in (decm["pred_boxes"][b].cpu(), odec["pred_boxes"][b])])
            for b in range(B)]
    fdb = (fdec["pred_boxes"] - odec["pred_boxes"][:NF]).abs().amax(dim=1)
    mx, rms = rel(decm["pred_mask"], odec["pred_mask"])
    fmx, frms = rel(fdec["pred_mask"], odec["pred_mask"][:NF])
    print(f"[3B batch 8] tokens: {n_arg}/{B * T} the oracle's arg-max, {n_tie} inside the logit noise; box |d|max per sample "
          f"{[f'{x:.1e}' for x in db.tolist()]} (operand floor {[f'{x:.1e}' for x in fdb.tolist()]}); IoU min {min(ious):.4f}; "
          f"mask logits rel max {mx:.3e} (floor {fmx:.3e}) rms {rms:.3e} (floor {frms:.3e})")
    if op == torch.float16:
        assert n_arg == B * T, f"{n_arg}/{B * T} tokens are the oracle's arg-max (measured since round 4: all of them)"
    else:
        assert n_arg >= int(0.85 * B * T)                              # bf16 operands (PADT_OPERANDS=bf16 hand runs): 123 / 128, the rest inside the noise bound
    assert float(db.max()) < 3 * float(fdb.max()) + 2e-4 and mx < 3 * fmx and rms < 3 * frms
    if op == torch.float16:
        assert float(db.max()) < 1e-3, f"box coordinates of {int((db >= 1e-3).sum())} of {B} samples beyond 1e-3"
        assert min(ious) > 0.995 and mx < 1.3 * fmx and mx < 6e-3          # measured 5.1e-3 = x1.13 of the floor of these samples
    else:
        assert min(ious) > 0.98


@pytest.mark.parametrize("llm_weights", ["bf16", "fp8"])          # "fp8+act" and full depth: test_7b_full_depth_single_image_against_oracle
def test_7b_geometry_ric_schedule_against_oracle(llm_weights):
    """BASELINE configs[4] at ITS geometry: padt_pro_7b() — D = 3584, 28 q / 4 kv heads (GQA group 7), MLP 18944 (padded to 18944 = 296 x 64),
    untied 152 064-row lm_head next to the embedding table, real 1280-wide ViT blocks and the real 98 M-parameter decoder — with the depth cut
    to 2 LLM layers / 2 ViT blocks so that the fp32 oracle runs in seconds.  Two ragged images, a RIC-shaped completion (caption text with
    3 interleaved runs of 5 VRTs, src/preprocess/process_ric.py:147,150 templates) through generate → parse → vl_decode:
    ids by the margin rule, per-step hidden rows, object grouping of the parser, boxes.  16-bit weights; "fp8": e4m3 WEIGHTS (the decode
    steps stream the fp8 image, the prompt pass multiplies the exactly dequantised 16-bit image) against the oracle on the dequantised
    matrices (parity_util.effective_llm_weights) — same bounds as the 16-bit weights; "fp8+act": the prompt pass additionally runs fp8 x fp8
    MFMA GEMMs over activation rows quantised to e4m3 on the fly, and the oracle quantises the same rows at prompt length
    (parity_util.fp8_prefill_hooks).  The token margin is DERIVED in the test: 2x the logit noise of the oracle's operand-floor run
    (parity_util.operand_floor on the model's operand type) at that step; e4m3 activations get the flat bound their noise needs (stated)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.synthetic import multi_object_schedule
    O = U.O
    base = padt_amd.padt_pro_7b()
    cfg = dataclasses.replace(base, num_hidden_layers=2,
                              vision_config=dataclasses.replace(base.vision_config, depth=2, fullatt_block_indexes=(1,)))
    assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.vocab_size, cfg.tie_word_embeddings) == \
        (3584, 28, 4, 18944, 152064, False)
    w = U.bf16_weights(cfg, seed=31, std=0.02)
    model = PaDTForConditionalGeneration(cfg, w, device="cuda", llm_weights=llm_weights)
    fp8, act8 = llm_weights != "bf16", llm_weights == "fp8+act"
    assert ("llm.0.gu.wq" in model.W) == fp8 and model.W.fp8_prefill == act8 and model.W["llm.head"].data_ptr() != model.W["llm.embed"].data_ptr()
    wo = U.effective_llm_weights(model, w) if fp8 else w
    oc = U.oracle_config(cfg)
    grids = [[1, 16, 20], [1, 12, 12]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=9, n_post=20, ragged=True, seed=88)
    T, n_obj, n_vrt = 28, 3, 5
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    seq = out.sequences.cpu()
    toks = seq[:, L:]
    assert toks.shape == (2, T) and (toks[:, -1] == cfg.eos_token_id).all()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    if act8:
        assert all(f"llm.0.{nm}.w8" in model.W for nm in ("qkv", "o", "gu", "down"))     # every 7B projection takes padt_gemm_fp8
    with torch.no_grad(), U.fp8_prefill_hooks(model):              # (no-op without e4m3 activations) prompt pass with e4m3 activation rows
        ores = O.generate(wo, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    with torch.no_grad(), U.operand_floor(model.W.op16):           # the distance the model's MFMA operand type alone imposes, on these inputs
        fres = O.generate(U.folded_weight_images(wo, cfg, model.W.op16), oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    n_tie = 0
    for t in range(T):
        lg = ores["logits"][t]
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        for b in range(2):
            fin = torch.isfinite(lg[b])
            # 2 x the floor run's logit noise at this step; e4m3 activation rows (3-bit mantissa, not in the floor run): 2.5 % of |logit|max
            floor = 2 * (fres["logits"][t][b][fin] - lg[b][fin]).abs().max().item()
            if act8:
                floor = max(floor, 2.5e-2 * lg[b][fin].abs().max().item())
            second = top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1
            if (top2[b, 0] - second).item() > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: not the oracle argmax"
            else:
                n_tie += 1
                assert (top2[b, 0] - chosen[b]).item() <= floor
    assert n_tie <= T // 2
    hid = out.hidden_states.last_layer_rows().cpu().float()
    worst = 0.0
    for t in range(T):
        mx, rms = rel(hid[t], ores["hidden"][t][:, -1])
        worst = max(worst, rms)
        # fp8: e4m3 activation rows at prompt length make the pass discontinuous (a one-bf16-ulp upstream difference flips e4m3 codes by a full
        # 12.5 % step), so after the first fp8 GEMM the two sides' quantisation errors (3.6 % rms per GEMM each) decorrelate: agreement with the
        # oracle that quantises the same rows is the fp8 noise level itself — measured 9.4e-2 on the prompt's last row after 2 layers x 4 GEMMs
        # at 7B width, 1-1.5e-2 on the decode-step rows (bf16 activations over the fp8-prefilled KV)
        _, frms = rel(fres["hidden"][t][:, -1], ores["hidden"][t][:, -1])
        assert rms < (1.5e-1 if act8 else 3 * frms + 1e-4), f"hidden step {t}: rel rms {rms:.3e} (operand floor {frms:.3e})"
    # ---- parser: 3 interleaved VRT runs per sample → 3 objects of 5 VRT features each; decoder on both sides
    n_m = [g[1] * g[2] // 4 for g in grids]
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, max(n_m)), 2)
    proc.model_embed_token_size = cfg.vocab_size
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False] * 2))
    assert [len(f) for f in feats] == [n_obj, n_obj] and all(o.shape == (n_vrt, cfg.hidden_size) for f in feats for o in f)
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
    with torch.no_grad():
        st = ores["state"]
        ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in r], 0) for r in runs] for b in range(2)]
        odec = O.vl_decode(wo, oc, ofeats, st.proto, st.high_res, grid, st.visual_pe)
    assert dec["sample_idx"] == odec["sample_idx"] == [0] * n_obj + [1] * n_obj
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    print(f"\n[7B geometry, {llm_weights}] ties {n_tie}/{2 * T}; hidden rel rms worst {worst:.3e}; {2 * n_obj} objects: box |d|max {db:.3e}, mask rel max {mx:.3e} rms {rms:.3e}")
    # e4m3 activations (measured in round 3: boxes 1.6e-3, mask logits 1.3e-2 — the price of a 3-bit mantissa at prompt length); 16-bit
    # activations: the north star's 1e-3 on the boxes, mask logits 5e-3 of their range
    assert (db < 5e-3 and mx < 5e-2) if act8 else (db < 1e-3 and mx < 5e-3)


def test_padt_decoder_ovd_shape_seven_objects_per_image():
    """BASELINE configs[3] (OVD COCO: ≈7 objects x 5 VRTs per image, eval/evaluation_scripts/inference_coco.py:101-110): vl_decode at the
    real decoder shape with 7 objects on each of two images (14 objects, 56 + 42 query rows → the tile-GEMM path on the query side,
    29 624 + 19 320 image rows replicated per object as padt.py:362-376 does) against the oracle on the same operands, at the north
    star's 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import padt_oracle as O
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    base = padt_amd.small_test_config(layers=1, vit_depth=1)
    cfg = dataclasses.replace(base, hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, intermediate_size=256,
                              vision_config=dataclasses.replace(base.vision_config, out_hidden_size=2048),
                              vl_decoder={"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "use_mask_loss": True})
    sd = synthetic_state_dict(cfg, seed=0, device="cpu")
    ocfg = O.OracleConfig()
    for k, shp in O.weight_shapes(ocfg).items():
        if k.startswith("vl_decoder."):
            sd[k] = _seeded(shp, k, 0.1, True) if O._is_norm_weight(k) else _seeded(shp, k, 0.02 if k.endswith("bias") else 0.03)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    g = torch.Generator().manual_seed(321)
    grids = torch.tensor([[1, 46, 46], [1, 46, 30]])
    Ps = [46 * 46, 46 * 30]
    bf = torch.bfloat16
    r = lambda t_: t_.to(bf).float()
    low = r(torch.randn(sum(Ps) // 4, 2048, generator=g))
    high = r(torch.randn(sum(Ps), 1280, generator=g))
    wi, _ = O.window_index(grids, 2, 112, 14)
    c, s = O.vit_rotary(ocfg, grids, wi)
    feats = [[r(torch.randn(5, 2048, generator=g)) for _ in range(7)] for _ in range(2)]
    out = model.vl_decode([[f.to(bf).cuda() for f in fs] for fs in feats], low.to(bf).cuda(), high.to(bf).cuda(), grids, (c.cuda(), s.cuda()))
    dwb = {k: r(v) for k, v in sd.items() if k.startswith("vl_decoder.")}
    odec = O.vl_decode(dwb, ocfg, feats, low, high, grids, (c, s))
    assert out["sample_idx"] == odec["sample_idx"] == [0] * 7 + [1] * 7 and out["pred_mask"].shape == odec["pred_mask"].shape
    db = (out["pred_boxes"].float().cpu() - odec["pred_boxes"]).abs().max().item()
    ds = (out["pred_score"].float().cpu() - odec["pred_score"]).abs().max().item()
    mx, rms = rel(out["pred_mask"], odec["pred_mask"])
    print(f"\n[OVD-shape decoder, 14 objects] vs oracle: box |d|max {db:.3e} score |d|max {ds:.3e} mask rel max {mx:.3e} rms {rms:.3e}")
    assert db < 1e-3 and ds < 1e-3 * (odec["pred_score"].abs().max().item() + 1) and mx < 1e-3 and rms < 1e-3


def _margin_rule(lg, tok, noise, what):
    """A HIP token must be the oracle's arg-max unless the oracle's own top-2 margin is inside `noise`; then within `noise` of the max.
    → 1 if the token was decided by the arg-max, 0 if it was a near-tie inside the bound."""
    top2 = lg.topk(2).values
    margin = (top2[0] - (top2[1] if torch.isfinite(top2[1]) else top2[0] - 1)).item()
    gap = top2[0].item() - lg[tok].item()
    if gap == 0.0:
        return 1
    assert margin <= noise and gap <= noise, f"{what}: HIP token is not the oracle arg-max (margin {margin:.3e}, gap {gap:.3e}, noise bound {noise:.3e})"
    return 0


def test_3b_ovd_geometry_merged_runner_against_oracle():
    """BASELINE configs[3] at ITS geometry and depth: PaDT_Pro_3B (32 ViT blocks, 36 layers), OVD batches of 8 images — the 80-class
    prompt (L = 890), T = 120 new tokens, 7 objects x 5 VRT per image (eval/evaluation_scripts/inference_coco.py:101-110) — through
    PipelinedRunner(depth 2, merge 16), nine batches so that the group's decode steps run the 128-row launch shapes (72 rows):
      (a) all 8 samples of the first and of the last batch bit-identical to the un-merged path (tokens, 56 boxes, scores, mask logits);
      (b) the fp32 CPU oracle teacher-forced on the HIP tokens of 2 samples (≈950 cached keys by the last step): every one of the 2 x 120
          tokens by the margin rule — noise bound 0.8 % of the largest |logit| = 1.5x what fp16 operands + folded weight images cost at THIS
          geometry (0.52 %: profiles/r04_ovd_length_floor.md, the oracle's operand floor on the first of these images, tests/studies/
          ovd_length_floor.py) —, the parser's 7 objects per sample, 14 boxes within the north star's 1e-3 (floor 2.7e-4), mask logits within
          8e-3 of their range = 2x that floor (4.0e-3; measured 6.5-6.8e-3 over the 14 objects (the value moved by 5 % when round 4 changed the rounding of ONE q element in ~2^13); the floor is not re-run inside the test: it
          would double its 3 minutes of host CPU)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.synthetic import multi_object_schedule
    from padt_amd.weights import synthetic_state_dict
    O = U.O
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    B, T, NB, n_obj, n_vrt = 8, 120, 9, 7, 5
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = [U.synthetic_batch(cfg, [[1, 46, 46]] * B, n_pre=15, n_post=346, seed=700 + i) for i in range(NB)]
    assert batches[0][2].shape == (B, 890)
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=16)
    res = []
    for grid, pix, ids, am in batches:
        res += runner.submit(ids.clone().cuda(), am.cuda(), pix.cuda().to(model.dtype), grid, max_new_tokens=T, schedule=sched)
    res += runner.flush()
    assert len(res) == NB
    for i in (0, NB - 1):                                            # ---- (a)
        grid, pix, ids, am = batches[i]
        dec1, comp1, lab1, vrt1 = pipeline.rec_batch(model, proc, ids.clone().cuda(), am.cuda(), pix.cuda().to(model.dtype), grid,
                                                     max_new_tokens=T, schedule=sched)
        decm, compm, labm, vrtm = res[i]
        assert decm["pred_boxes"].shape == (B * n_obj, 4) and all(len(v) == n_obj for v in vrtm)
        assert compm == comp1 and vrtm == vrt1, f"batch {i}: tokens differ between merged (128-row steps) and un-merged decode"
        for k in ("pred_boxes", "pred_score", "pred_mask"):
            assert torch.equal(decm[k], dec1[k]), f"batch {i}: {k} differs between merged and un-merged decode"
    # ---- (b) two samples of batch 0 against the oracle
    NS = 2
    grid, pix, ids, am = batches[0]
    P1 = 46 * 46
    gids = proc.assign_to_global_vrt_id(ids.clone(), grid)
    out = model.generate(input_ids=gids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda().to(model.dtype), image_grid_thw=grid,
                         max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    toks = out.sequences[:, L:].cpu()
    # sample b's global VRT ids are offset by b x 529 in the batch of 8; alone (rows 0..NS-1 of the batch) the offsets are the same
    torch.set_num_threads(min(32, os.cpu_count() or 8))         # 32: the fastest count for the oracle on the 256-core GPU box (bench.py calibrates the same)
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, gids[:NS], am[:NS], pix[: NS * P1], grid[:NS], T, schedule=sched, collect_logits=True, force_tokens=toks[:NS])
        runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
        st = ores["state"]
        ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in r], 0) for r in runs] for b in range(NS)]
        odec = O.vl_decode(w, oc, ofeats, st.proto, st.high_res, grid[:NS], st.visual_pe)
    t_or = time.perf_counter() - t0
    n_arg = 0
    for b in range(NS):
        for t in range(T):
            lg = ores["logits"][t][b]
            fin = torch.isfinite(lg)
            n_arg += _margin_rule(lg, int(toks[b, t]), 8e-3 * lg[fin].abs().max().item(), f"sample {b} step {t}")
    decm = res[0][0]
    sel = [i for i, s_ in enumerate(decm["sample_idx"]) if s_ < NS]
    assert [decm["sample_idx"][i] for i in sel] == odec["sample_idx"] == [0] * n_obj + [1] * n_obj
    db = (decm["pred_boxes"][sel].cpu().float() - odec["pred_boxes"]).abs().amax(dim=1)
    mx, rms = rel(decm["pred_mask"][sel], odec["pred_mask"])
    print(f"\n[3B OVD geometry, {model.dtype}] L = {L}, T = {T}, {NB} batches in a 16-batch decode group; oracle on {NS} samples {t_or:.1f} s; "
          f"tokens {n_arg}/{NS * T} the oracle's arg-max (rest inside the bound); 14 boxes |d|max {float(db.max()):.3e}; mask logits rel max {mx:.3e} rms {rms:.3e}")
    assert n_arg == NS * T, f"{n_arg}/{NS * T} tokens are the oracle's arg-max (measured: 240/240)"
    assert float(db.max()) < 1e-3 and mx < 7.5e-3                        # measured 6.5-6.8e-3 behind 950 cached keys


@pytest.mark.parametrize("llm_weights", ["bf16", "fp8+act"])
def test_7b_full_depth_single_image_against_oracle(llm_weights):
    """BASELINE configs[4] at FULL depth: padt_pro_7b() — 28 layers at D = 3584 / 28:4 heads / MLP 18944, untied 152 064-row head, 32 ViT
    blocks — one 46 x 46 image, a RIC-shaped completion (2 VRT runs of 3) through generate → parse → vl_decode, against the fp32 CPU oracle
    teacher-forced on the HIP tokens (33 GB of fp32 weights on the host).  16-bit weights: every float quantity within 2x the oracle's
    operand floor (parity_util.operand_floor on the model's operand type and folded weight images) and boxes within the north star's 1e-3.
    "fp8+act" (fp8 weight streaming in the decode steps + fp8 x fp8 MFMA prompt pass over e4m3 activation rows), against the oracle on the
    dequantised matrices quantising the same rows: the measured full-depth price of e4m3 activations is PRINTED and bounded (boxes 1e-2,
    IoU > 0.95) — it is what bench.py's ric_7b_fp8 workload string quotes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.synthetic import multi_object_schedule
    from padt_amd.weights import synthetic_state_dict
    O = U.O
    cfg = padt_amd.padt_pro_7b()
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.vision_config.depth) == (28, 3584, 32)
    sd = synthetic_state_dict(cfg, seed=41, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda", llm_weights=llm_weights)
    act8 = llm_weights == "fp8+act"
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    if act8:
        w = U.effective_llm_weights(model, w)
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=78)
    T, n_obj, n_vrt = 14, 2, 3
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    seq = out.sequences.cpu()
    toks = seq[:, L:]
    assert toks.shape == (1, T) and int(toks[0, -1]) == cfg.eos_token_id
    torch.set_num_threads(min(32, os.cpu_count() or 8))         # 32: the fastest count for the oracle on the 256-core GPU box (bench.py calibrates the same)
    t0 = time.perf_counter()
    with torch.no_grad():
        with U.fp8_prefill_hooks(model):
            ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        fres = None
        if not act8:
            wf = U.folded_weight_images(w, cfg, model.W.op16)
            with U.operand_floor(model.W.op16):
                fres = O.generate(wf, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
            del wf
    t_or = time.perf_counter() - t0
    n_arg, n_out, worst, worst_floor = 0, 0, 0.0, 0.0
    hid = out.hidden_states.last_layer_rows().cpu().float()
    for t in range(T):
        lg = ores["logits"][t][0]
        fin = torch.isfinite(lg)
        if act8:
            # e4m3 activation rows at prompt length, 28 layers deep: a flat (stated) 5 % bound, and a token outside it is COUNTED, not fatal —
            # two implementations of the same discontinuous quantised function decorrelate (DESIGN.md §4 numerics)
            try:
                n_arg += _margin_rule(lg, int(toks[0, t]), 5e-2 * lg[fin].abs().max().item(), f"step {t}")
            except AssertionError:
                n_out += 1
        else:
            n_arg += _margin_rule(lg, int(toks[0, t]), 2 * (fres["logits"][t][0][fin] - lg[fin]).abs().max().item(), f"step {t}")
        _, rms = rel(hid[t], ores["hidden"][t][:, -1])
        worst = max(worst, rms)
        if not act8:
            _, frms = rel(fres["hidden"][t][:, -1], ores["hidden"][t][:, -1])
            worst_floor = max(worst_floor, frms)
            assert rms < 2 * frms + 1e-4, f"hidden step {t}: rel rms {rms:.3e} vs operand floor {frms:.3e}"
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False]))
    assert [len(f) for f in feats] == [n_obj] and all(o.shape == (n_vrt, cfg.hidden_size) for o in feats[0])
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
    with torch.no_grad():
        st = ores["state"]
        odec = O.vl_decode(w, oc, [[torch.cat([ores["hidden"][t][0:1, -1] for t in r], 0) for r in runs]], st.proto, st.high_res, grid, st.visual_pe)
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    ious = [O.box_iou_xywh(*[[float(x[0] - x[2] / 2), float(x[1] - x[3] / 2), float(x[2]), float(x[3])] for x in (dec["pred_boxes"][k].cpu(), odec["pred_boxes"][k])])
            for k in range(n_obj)]
    print(f"\n[7B full depth, {llm_weights}, {model.dtype} operands] oracle {t_or:.1f} s; tokens {n_arg}/{T} the oracle's arg-max, {n_out} outside the bound; hidden rel rms worst "
          f"{worst:.3e} (operand floor {worst_floor:.3e}); {n_obj} boxes |d|max {db:.3e} IoU min {min(ious):.4f}; mask logits rel max {mx:.3e} rms {rms:.3e}")
    if act8:
        assert db < 1e-2 and min(ious) > 0.95 and n_out <= T // 4
    else:
        assert db < 1e-3 and min(ious) > 0.995 and mx < 4.8e-3           # measured 4.1e-3
