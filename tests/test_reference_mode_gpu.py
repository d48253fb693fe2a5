"""``precision="reference"`` (padt_amd/reference.py): ViT / LLM on the split-precision machinery of the PaDT decoder — fp32 streams, (hi, lo)
bf16 GEMM operands, fp32 attention everywhere (ViT windows / full layers, the LLM's causal GQA prompt pass, its decode steps over an fp32 KV cache).  The north star's letter: VRT token ids EQUAL to the fp32
reference's, box coordinates AND mask logits within 1e-3 — asserted here at the full PaDT_Pro_3B depth, where the default (fp16-operand) path
sits at 3.7e-3 on the mask logits (its operand type's floor, tests/test_real_shape_gpu.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def test_reference_precision_small_config_end_to_end():
    """Small config (real head widths, ragged batch of two images): ViT rows / prototypes at fp32-class distance, per-step hidden rows at the
    distance of the LLM's fp16 attention internals, tokens equal, boxes / score / mask logits far inside 1e-3; and the merged runner gives the
    same bits as rec_batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import parity_util as U
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    O = U.O
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=5, std=0.05)
    oc = U.oracle_config(cfg)
    model = PaDTForConditionalGeneration(cfg, w, device="cuda", precision="reference")
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 10, 12], [1, 8, 8]], n_pre=6, n_post=9, ragged=True)
    low, high, (cos, sin) = model.ref.visual(pix.cuda(), grid)
    olow, ohigh, _ = O.vit_forward(w, oc, pix, grid)
    mxh, rmsh = rel(high, ohigh)
    mxl, rmsl = rel(low, olow)
    print(f"\n[reference precision, small ViT] high_res rel max {mxh:.3e} rms {rmsh:.3e}; image_embeds rel max {mxl:.3e} rms {rmsl:.3e}")
    assert high.dtype == torch.float32 and low.dtype == torch.float32
    assert rmsh < 2e-5 and mxh < 5e-5 and rmsl < 2e-5 and mxl < 5e-5              # default path: 1.3e-4 (fp16) / 1e-3 (bf16)
    T = 10
    sched = U.rec_schedule(T, range(3, 7))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    toks = out.sequences.cpu()[:, L:]
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    for t in range(T):
        lg = ores["logits"][t]
        for b in range(2):
            assert int(torch.argmax(lg[b])) == int(toks[b, t]), f"step {t} sample {b}: not the oracle's arg-max"
    st = ores["state"]
    mxp, rmsp = rel(out.past_image_embeds, st.proto)
    hid = out.hidden_states.last_layer_rows()
    assert hid.dtype == torch.float32 and out.past_image_embeds.dtype == torch.float32
    worst = max(rel(hid[t], ores["hidden"][t][:, -1])[1] for t in range(T))
    print(f"[reference precision, small e2e] prototypes rel max {mxp:.3e} rms {rmsp:.3e}; hidden rows rel rms worst {worst:.3e}")
    assert rmsp < 2e-5 and worst < 5e-5                                           # measured 8.2e-6 (6.0e-4 while the LLM's attention still ran on the fp16 MFMA kernels)
    feats = [[hid[3:7, b]] for b in range(2)]
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in range(3, 7)], 0)] for b in range(2)]
    odec = O.vl_decode(w, oc, ofeats, st.proto, st.high_res, grid, st.visual_pe)
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    ds = (dec["pred_score"].cpu().float() - odec["pred_score"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    print(f"[reference precision, small e2e] box |d|max {db:.3e} score |d| {ds:.3e} mask logits rel max {mx:.3e} rms {rms:.3e}")
    assert db < 1e-5 and mx < 1e-4 and rms < 1e-4                                 # measured 1.8e-7 / 1.3e-5 / 1.2e-5
    # the runner (two batches in one decode session) on the same model: per-sample results bit-identical to batch-at-a-time
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 30), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s in range(2):
        g2, p2, i2, a2 = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5 + s, n_post=7, seed=500 + s, ragged=True)
        batches.append((i2.cuda(), a2.cuda(), p2.cuda(), g2))
    ref = [pipeline.rec_batch(model, proc, b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched) for b in batches]
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=2)
    got = []
    for b in batches:
        got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
    got += runner.flush()
    for (d0, c0, l0, v0), (d1, c1, l1, v1) in zip(ref, got):
        assert c0 == c1 and v0 == v1
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"])
    # caller hooks on this mode (round 6): the hooked loop launches the same split-precision steps kernel by kernel — no-op hooks give the captured run's bits,
    # a stopping criterion ends the sequences where HF's loop would
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    same = model.generate(logits_processor=LogitsProcessorList([lambda i, sc: sc]), **kw)
    assert torch.equal(same.sequences, out.sequences) and torch.equal(same.hidden_states.last_layer_rows(), hid)
    cut = model.generate(stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=L + 4)]), **kw)
    assert torch.equal(cut.sequences, out.sequences[:, : L + 4])


def test_reference_precision_full_depth_3b_meets_the_north_star_on_every_float_output():
    """The inputs of test_full_depth_3b_teacher_forced_against_oracle (one 46 x 46 image, 8 steps, a run of 4 VRT; seeded bf16-representable
    weights with biases and norm jitter) through precision="reference": token ids EQUAL the fp32 oracle's arg-max at every step, box
    coordinates and mask logits <= 1e-3 (of [0, 1] / of the logit range), ViT rows at fp32-class distance.  ≈40 s of host CPU for the oracle."""
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
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda", precision="reference")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=77)
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 6))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    seq = out.sequences.cpu()
    toks = seq[:, 577:]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    t_or = time.perf_counter() - t0
    st = ores["state"]
    n_eq = 0
    for t in range(T):
        lg = ores["logits"][t][0]
        top2 = lg.topk(2).values
        eq = int(torch.argmax(lg)) == int(toks[0, t])
        n_eq += int(eq)
        print(f"[reference precision, full 3B] step {t} mode {sched[t]}: token {int(toks[0, t])} {'==' if eq else '!='} oracle arg-max; oracle top-2 margin "
              f"{(top2[0] - (top2[1] if torch.isfinite(top2[1]) else top2[0] - 1)).item():.3e}")
    assert n_eq == T, f"{n_eq}/{T} tokens equal the oracle's arg-max"
    mxh, rmsh = rel(out.past_high_res_image_embeds, st.high_res)
    mxp, rmsp = rel(out.past_image_embeds, st.proto)
    hid = out.hidden_states.last_layer_rows()
    worst = max(rel(hid[t], ores["hidden"][t][:, -1])[1] for t in range(T))
    print(f"\n[reference precision, full 3B] generate {t_hip:.2f} s (first call), oracle {t_or:.1f} s; ViT high_res rel rms {rmsh:.3e} max {mxh:.3e}; prototypes rms {rmsp:.3e}; "
          f"hidden rows rel rms worst {worst:.3e}")
    assert rmsh < 5e-5 and rmsp < 5e-5                                            # default fp16 path: 1.09e-3
    assert worst < 3e-4                                                           # measured 4.1e-5 (1.13e-3 with fp16 attention internals in the LLM; default path: 3-4.7e-3)
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, 577:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False]))
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    with torch.no_grad():
        odec = O.vl_decode(w, oc, [[torch.cat([ores["hidden"][t][0:1, -1] for t in range(2, 6)], 0)]], st.proto, st.high_res, grid, st.visual_pe)
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    ds = (dec["pred_score"].cpu().float() - odec["pred_score"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    print(f"[reference precision, full 3B] end to end: box |d|max {db:.3e}  score |d| {ds:.3e}  mask logits max / range {mx:.3e} rms {rms:.3e}  "
          f"(default fp16 path on the same inputs: 1.3e-4 / 3.7e-3; this mode with the LLM's attention still on the fp16 MFMA kernels: 2.3e-5 / 3.3e-4)")
    assert db < 1e-3 and mx < 1e-3, f"north star missed: boxes {db:.3e}, mask logits {mx:.3e}"
    assert db < 2e-5 and mx < 2e-4                                                # sized to what is measured: 1.8e-6 / 2.7e-5


def test_reference_precision_batch8_every_sample_within_1e3():
    """What bench.py's `reference_precision` leg times — PaDT_Pro_3B, batches of 8 different 46 x 46 images, 16 REC tokens with a run of 5 VRT,
    two batches sharing one decode session through PipelinedRunner(depth=2, merge=2) — against the fp32 oracle for ALL 8 samples of a batch
    (≈1.5 min of host CPU): every token the oracle's arg-max, EVERY sample's box within 1e-3 and the mask logits within 1e-3 of their range
    (the default path on the same batch: boxes 0.7-3.4e-4, mask logits 5.1e-3); merged == un-merged bit for bit."""
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
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda", precision="reference")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    B, T, NB = 8, 16, 3
    sched = U.rec_schedule(T, vrt_at=range(6, 11))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = [U.synthetic_batch(cfg, [[1, 46, 46]] * B, n_pre=15, n_post=33, seed=500 + i) for i in range(NB)]
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=2)
    res = []
    for grid, pix, ids, am in batches:
        res += runner.submit(ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
    res += runner.flush()
    assert len(res) == NB
    grid, pix, ids, am = batches[1]                                   # the second batch of a merged group: row offset 8, prototype offset 8 x 529
    dec1, comp1, lab1, vrt1 = pipeline.rec_batch(model, proc, ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
    decm, compm, labm, vrtm = res[1]
    assert compm == comp1 and vrtm == vrt1
    for k in ("pred_boxes", "pred_score", "pred_mask"):
        assert torch.equal(decm[k], dec1[k]), f"{k} differs between merged and un-merged decode"
    out = model.generate(input_ids=proc.assign_to_global_vrt_id(ids.clone(), grid).cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(),
                         image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    toks = out.sequences[:, L:].cpu()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.perf_counter()
    with torch.no_grad():
        gids = proc.assign_to_global_vrt_id(ids.clone(), grid)
        ores = O.generate(w, oc, gids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        ost = ores["state"]
        odec = O.vl_decode(w, oc, [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in range(6, 11)], 0)] for b in range(B)], ost.proto, ost.high_res, grid, ost.visual_pe)
    n_eq = sum(int(torch.argmax(ores["logits"][t][b])) == int(toks[b, t]) for b in range(B) for t in range(T))
    db = (decm["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().amax(dim=1)
    rng = odec["pred_mask"].abs().max().item()
    dm = (decm["pred_mask"].cpu().float() - odec["pred_mask"]).abs().flatten(1).amax(dim=1) / rng
    print(f"\n[reference precision, 3B batch 8] oracle {time.perf_counter() - t0:.1f} s; tokens {n_eq}/{B * T} the oracle's arg-max; box |d|max per sample "
          f"{[f'{x:.1e}' for x in db.tolist()]}; mask logits max / range per sample {[f'{x:.1e}' for x in dm.tolist()]}")
    assert n_eq == B * T
    assert float(db.max()) < 1e-3 and float(dm.max()) < 1e-3


def test_reference_precision_ovd_geometry_mask_logits_within_1e3():
    """BASELINE configs[3]'s geometry — 80-class prompt (L = 890), 120 new tokens, 7 objects x 5 VRT per image, ≈950 cached keys by the last
    step — where the default path's mask logits are furthest from the oracle (6.5-6.8e-3 of their range,
    test_3b_ovd_geometry_merged_runner_against_oracle): with precision="reference" every one of the 2 x 120 tokens of the two samples checked is
    the oracle's arg-max, their 14 boxes and mask logits are within 1e-3.  (≈2 min of host CPU for the oracle.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    from padt_amd.synthetic import multi_object_schedule
    O = U.O
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda", precision="reference")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    B, T, n_obj, n_vrt, NS = 8, 120, 7, 5, 2
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]] * B, n_pre=15, n_post=346, seed=700)
    assert ids.shape == (B, 890)
    gids = proc.assign_to_global_vrt_id(ids.clone(), grid)
    out = model.generate(input_ids=gids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    seq = out.sequences.cpu()
    toks = seq[:, L:]
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False] * B))
    assert all(len(f) == n_obj for f in feats)
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    P1 = 46 * 46
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, gids[:NS], am[:NS], pix[: NS * P1], grid[:NS], T, schedule=sched, collect_logits=True, force_tokens=toks[:NS])
        runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
        st = ores["state"]
        ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in r], 0) for r in runs] for b in range(NS)]
        odec = O.vl_decode(w, oc, ofeats, st.proto, st.high_res, grid[:NS], st.visual_pe)
    n_eq = sum(int(torch.argmax(ores["logits"][t][b])) == int(toks[b, t]) for b in range(NS) for t in range(T))
    sel = [i for i, s_ in enumerate(dec["sample_idx"]) if s_ < NS]
    db = (dec["pred_boxes"][sel].cpu().float() - odec["pred_boxes"]).abs().amax(dim=1)
    mx, rms = rel(dec["pred_mask"][sel], odec["pred_mask"])
    print(f"\n[reference precision, 3B OVD geometry] L = {L}, T = {T}; oracle on {NS} samples {time.perf_counter() - t0:.1f} s; tokens {n_eq}/{NS * T} the oracle's "
          f"arg-max; 14 boxes |d|max {float(db.max()):.3e}; mask logits max / range {mx:.3e} rms {rms:.3e}  (default path: 4.1e-4 / 6.8e-3)")
    assert n_eq == NS * T
    assert float(db.max()) < 1e-3 and mx < 1e-3


def test_reference_precision_7b_full_depth_within_1e3():
    """BASELINE configs[4]'s model at FULL depth — padt_pro_7b(): 28 layers at D = 3584 / 28:4 heads / MLP 18944, untied 152 064-row head, 32
    ViT blocks — one 46 x 46 image, a RIC-shaped completion (2 VRT runs of 3), precision="reference": tokens equal, boxes and mask logits
    within 1e-3 (the default path on the same inputs: boxes 3.8e-4, mask logits 4.1e-3).  33 GB of fp32 oracle weights on the host, ≈40 s."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import synthetic_state_dict
    from padt_amd.synthetic import multi_object_schedule
    O = U.O
    cfg = padt_amd.padt_pro_7b()
    sd = synthetic_state_dict(cfg, seed=41, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cuda", dtype=torch.bfloat16)
    model = PaDTForConditionalGeneration(cfg, sd, device="cuda", precision="reference")
    w = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=78)
    T, n_obj, n_vrt = 14, 2, 3
    sched = multi_object_schedule(T, n_obj=n_obj, n_vrt=n_vrt)
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    seq = out.sequences.cpu()
    toks = seq[:, L:]
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 529), 2)
    proc.model_embed_token_size = cfg.vocab_size
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False]))
    assert len(feats[0]) == n_obj
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.perf_counter()
    with torch.no_grad():
        ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
        runs = [[t for t in range(T) if sched[t] == "v"][k * n_vrt: (k + 1) * n_vrt] for k in range(n_obj)]
        st = ores["state"]
        odec = O.vl_decode(w, oc, [[torch.cat([ores["hidden"][t][0:1, -1] for t in r], 0) for r in runs]], st.proto, st.high_res, grid, st.visual_pe)
    n_eq = sum(int(torch.argmax(ores["logits"][t][0])) == int(toks[0, t]) for t in range(T))
    hid = out.hidden_states.last_layer_rows()
    worst = max(rel(hid[t], ores["hidden"][t][:, -1])[1] for t in range(T))
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    mx, rms = rel(dec["pred_mask"], odec["pred_mask"])
    print(f"\n[reference precision, 7B full depth] oracle {time.perf_counter() - t0:.1f} s; tokens {n_eq}/{T} the oracle's arg-max; hidden rows rel rms worst {worst:.3e}; "
          f"boxes |d|max {db:.3e}; mask logits max / range {mx:.3e} rms {rms:.3e}  (default path: 3.8e-4 / 4.1e-3)")
    assert n_eq == T
    assert db < 1e-3 and mx < 1e-3
    assert db < 2e-5 and mx < 3e-4                                                # measured 1.3e-6 / 4.6e-5 (9.8e-4 while the LLM's attention ran on the fp16 MFMA kernels)
