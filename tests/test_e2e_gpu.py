"""End-to-end parity of the HIP path against the CPU oracle on the same seeded inputs (small config with the REAL head
dims: ViT/decoder 80, LLM 128, so the same kernel instantiations as PaDT_Pro_3B run).

Both sides see identical numbers going in (weights and pixels are bf16-representable); the oracle computes in fp32, the
HIP path multiplies 16-bit MFMA operands — fp16 (the default) and bf16 (the A/B variant): every test of the `setup` fixture runs for
both — over fp32 residual streams and accumulates in fp32.  Token ids follow the margin rule of SURVEY.md §7: the
oracle is teacher-forced on the HIP tokens and every HIP token must be the oracle's argmax unless the oracle's own
top-2 margin is below the operand-type noise floor (then it must still be within that floor of the max).
Tolerances are stated where they are asserted, per operand type: tol(model, bf16 bound, fp16 bound), each ≈3x the measured value.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp16", "bf16"])
def setup(request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    from padt_amd.modeling import PaDTForConditionalGeneration
    import parity_util as U
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=5, std=0.05)
    model = PaDTForConditionalGeneration(cfg, w, device="cuda", operands=request.param)
    assert model.dtype == (torch.float16 if request.param == "fp16" else torch.bfloat16)
    return cfg, w, model, U, U.oracle_config(cfg)


def tol(model, bf16, fp16):
    """The bound for the model's MFMA operand type."""
    return fp16 if model.dtype == torch.float16 else bf16


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def test_vit_prototypes_against_oracle(setup):
    cfg, w, model, U, oc = setup
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 10, 12], [1, 8, 8]], ragged=True)
    low, high, (cos, sin) = model.visual(pix.cuda(), grid)
    olow, ohigh, (ocos, osin) = U.O.vit_forward(w, oc, pix, grid)
    assert torch.equal(cos.cpu(), ocos) and torch.equal(sin.cpu(), osin)              # host-built tables: bit-exact
    # 4 ViT blocks, 16-bit operands over the fp32 stream; measured high_res: bf16 1.0e-3 rms / 9.6e-4 max, fp16 1.3e-4 / 1.3e-4 (the merger
    # output and the prototypes behind it sit higher: two more GEMMs on 4-patch rows)
    lim_rms, lim_mx = tol(model, 1e-2, 6e-4), tol(model, 2.5e-2, 1e-3)
    mx, rms = rel_err(high, ohigh)
    print(f"\n[small ViT, {model.dtype}] high_res rel max {mx:.3e} rms {rms:.3e}")
    assert rms < lim_rms and mx < lim_mx, f"high_res rel err max {mx:.3e} rms {rms:.3e}"
    mx, rms = rel_err(low, olow)
    print(f"[small ViT, {model.dtype}] image_embeds rel max {mx:.3e} rms {rms:.3e}")
    assert rms < lim_rms and mx < lim_mx, f"image_embeds rel err max {mx:.3e} rms {rms:.3e}"
    proto = model.lm.prototypes(low)
    mx, rms = rel_err(proto, U.O.prototypes(w, oc, olow))
    print(f"[small ViT, {model.dtype}] prototypes rel max {mx:.3e} rms {rms:.3e}")
    assert rms < lim_rms and mx < lim_mx, f"prototypes rel err max {mx:.3e} rms {rms:.3e}"


def test_generate_tokens_hidden_and_vl_decode(setup):
    cfg, w, model, U, oc = setup
    import padt_amd
    O = U.O
    grids = [[1, 10, 12], [1, 8, 8], [1, 6, 10]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=6, n_post=9, ragged=True)
    T = 12
    sched = U.rec_schedule(T, vrt_at=range(4, 8))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid,
                         max_new_tokens=T, schedule=sched, do_sample=False, output_hidden_states=True,
                         return_dict_in_generate=True)
    seq = out.sequences.cpu()
    L = ids.shape[1]
    toks = seq[:, L:]
    assert toks.shape[1] == T and (toks[:, -1] == cfg.eos_token_id).all()
    V = cfg.vocab_size
    n_m = [g[1] * g[2] // 4 for g in grids]
    off = [0, n_m[0], n_m[0] + n_m[1]]
    for b in range(3):                                              # schedule honoured, VRT ids global and in the sample's own range
        for t in range(T - 1):
            if sched[t] == "v":
                assert V + off[b] <= toks[b, t] < V + off[b] + n_m[b], (b, t, int(toks[b, t]))
            else:
                assert toks[b, t] < V
    # ---- oracle teacher-forced on the HIP tokens: margin rule
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    assert torch.equal(ores["sequences"], seq)
    noise = 0.0
    n_tie = 0
    for t in range(T):
        lg = ores["logits"][t]                                      # (B, V+N) after the schedule's processor
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        floor = tol(model, 2e-2, 4e-3) * lg[torch.isfinite(lg)].abs().max().item()     # operand noise floor on a logit: 2 % (bf16) / 0.4 % (fp16) of the logit scale
        for b in range(3):
            margin = (top2[b, 0] - (top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1)).item()
            if margin > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: HIP token is not the oracle argmax (margin {margin:.3e})"
            else:
                n_tie += 1
                assert (top2[b, 0] - chosen[b]).item() <= floor
            noise = max(noise, (top2[b, 0] - chosen[b]).item())
    assert n_tie <= T * 3 // 4, "too many near-ties: test has no power"
    # ---- hidden rows that predicted each token (what parseVRTintoCompletion gathers)
    hid = out.hidden_states.last_layer_rows().cpu().float()          # (T,B,D)
    for t in range(T):
        oh = ores["hidden"][t][:, -1].float()
        mx, rms = rel_err(hid[t], oh)
        assert rms < tol(model, 2e-2, 3e-3) and mx < tol(model, 8e-2, 1.2e-2), f"hidden step {t}: rel err max {mx:.3e} rms {rms:.3e}"
    # ---- parser on HIP output (local ids) → feats; vl_decode both sides
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, max(n_m)), 2)
    proc.model_embed_token_size = V
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False] * 3))
    assert [len(f) for f in feats] == [1, 1, 1] and all(f[0].shape == (4, cfg.hidden_size) for f in feats)
    for b in range(3):
        assert torch.equal(feats[b][0].cpu().float(), hid[4:8, b])
        assert vrts[b][0] == "".join("<|VRT_%d|>" % int(i - V) for i in local[b, 4:8])
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    # oracle decoder on the oracle's own (teacher-forced) features and image tensors = full-pipeline parity
    st = ores["state"]
    ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in range(4, 8)], 0)] for b in range(3)]
    odec = O.vl_decode(w, oc, ofeats, st.proto, st.high_res, grid, st.visual_pe)
    assert dec["sample_idx"] == odec["sample_idx"] == [0, 1, 2]
    assert torch.equal(dec["pred_mask_valid_hw"][0].cpu(), odec["pred_mask_valid_hw"][0])
    assert torch.equal(dec["pred_mask_valid_hw"][1].cpu(), odec["pred_mask_valid_hw"][1])
    assert dec["pred_mask"].shape == odec["pred_mask"].shape
    db = (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
    ds = (dec["pred_score"].cpu().float() - odec["pred_score"]).abs().max().item()
    mx, rms = rel_err(dec["pred_mask"], odec["pred_mask"])
    print(f"\n[e2e parity] box |d|max {db:.3e}  score |d|max {ds:.3e}  mask rel max {mx:.3e} rms {rms:.3e}  token noise {noise:.3e}")
    # measured (round 4): bf16 operands 8.4e-5 / 1.6e-3 / 2.8e-3 max, 3.9e-3 rms; fp16 operands 9.8e-6 / 2.5e-4 / 3.6e-4 max, 5.0e-4 rms — bounds ≈3x
    assert db < tol(model, 3e-4, 3e-5), f"box coords differ by {db:.3e}"               # boxes in [0,1]
    assert ds < tol(model, 5e-3, 8e-4) * (odec["pred_score"].abs().max().item() + 1), f"score logit differs by {ds:.3e}"
    assert rms < tol(model, 1.2e-2, 1.5e-3) and mx < tol(model, 9e-3, 1.1e-3), f"mask logits rel err max {mx:.3e} rms {rms:.3e}"
    # decoder alone on IDENTICAL inputs (HIP features fed to the oracle): isolates the decoder kernels
    odec2 = O.vl_decode(w, oc, [[f[0].cpu().float()] for f in feats], out.past_image_embeds.cpu().float(),
                        out.past_high_res_image_embeds.cpu().float(), grid,
                        (out.past_visual_pe[0].cpu(), out.past_visual_pe[1].cpu()))
    db2 = (dec["pred_boxes"].cpu().float() - odec2["pred_boxes"]).abs().max().item()
    mx2, rms2 = rel_err(dec["pred_mask"], odec2["pred_mask"])
    print(f"[decoder-only parity] box |d|max {db2:.3e}  mask rel max {mx2:.3e} rms {rms2:.3e}")
    assert db2 < 1e-3 and mx2 < 1e-3 and rms2 < 1e-3                  # split-precision decoder: the north star's 1e-3 on identical inputs


def test_massive_activation_channel_beyond_fp16_range():
    """Range safety of fp16 operands end to end.  Real checkpoints carry "massive activations": a few residual-stream channels at 1e3-1e4 (and
    more), the rest O(1), produced by ordinary-sized weights.  Here one row of layer 0's o_proj is set to 8192 (a weight fp16 holds), which
    drives channel 7 of the residual stream beyond fp16's 65504 from layer 0 on (checked on the oracle's layer outputs) — and the run must stay
    finite and on the oracle: the fp32 stream holds the value, its 16-bit mirror holds fp16(2^-4 x), the folded RMSNorm divides the factor out
    again, q / k / v come out O(1).  (A plain fp16 copy of the stream would be +inf in that channel and NaN one kernel later.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    from padt_amd.modeling import PaDTForConditionalGeneration
    import parity_util as U
    O = U.O
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=21, std=0.05)
    w["model.layers.0.self_attn.o_proj.weight"] = w["model.layers.0.self_attn.o_proj.weight"].clone()
    w["model.layers.0.self_attn.o_proj.weight"][7, :] = 8192.0
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=8, ragged=True, seed=41)
    _, (h_last, hs), _ = O.prefill(w, oc, ids, am, pix, grid, all_hidden=True)
    peak = max(float(h[..., 7].abs().max()) for h in hs[1:-1])
    assert peak > 65504.0, f"the test has no power: stream channel 7 peaks at {peak:.3e}"
    model = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="fp16")
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 5))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    L = ids.shape[1]
    toks = out.sequences.cpu()[:, L:]
    hid = out.hidden_states.last_layer_rows().cpu().float()
    assert torch.isfinite(hid).all(), "fp16 operands overflowed"
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    worst = 0.0
    for t in range(T):
        mx, rms = rel_err(hid[t], ores["hidden"][t][:, -1].float())
        worst = max(worst, rms)
        lg = ores["logits"][t]
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        floor = 1e-2 * lg[torch.isfinite(lg)].abs().max().item()
        for b in range(2):
            second = top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1
            if (top2[b, 0] - second).item() > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: not the oracle argmax"
    print(f"\n[stream channel at {peak:.3e} (> fp16 max), fp16 operands] hidden rel rms worst {worst:.3e}")
    assert worst < 1e-3                                               # measured 2.2e-5 (the massive channel dominates the norm)


def test_graph_replay_equals_eager_and_is_repeatable(setup):
    cfg, w, model, U, oc = setup
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 8, 8]], n_pre=5, n_post=7)
    T = 10
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
              schedule=sched)
    a = model.generate(use_graph=False, **kw)
    b = model.generate(use_graph=True, **kw)
    c = model.generate(use_graph=True, **kw)                         # second call replays the captured graph from step 1
    assert torch.equal(a.sequences, b.sequences) and torch.equal(b.sequences, c.sequences)
    assert torch.equal(a.hidden_states.last_layer_rows(), b.hidden_states.last_layer_rows())
    assert torch.equal(b.hidden_states.last_layer_rows(), c.hidden_states.last_layer_rows())


def test_empty_vl_decode_and_error_conventions(setup):
    cfg, w, model, U, oc = setup
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8]], n_pre=5, n_post=7)
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid,
                         max_new_tokens=3, schedule=["t", "t", "e"])
    r = model.vl_decode([[]], out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    assert r["pred_boxes"].shape == (0, 4) and r["pred_mask"].shape == (0, 8, 8) and r["pred_mask_valid_hw"] == () and r["sample_idx"] == []
    bad = ids.clone()
    bad[0, -3] = cfg.image_token_id                                  # one image token too many
    with pytest.raises(ValueError, match="Image features and image tokens do not match"):
        model.generate(input_ids=bad.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=2)
    assert out["sequences"] is out.sequences and model.config.vision_config.spatial_merge_size == 2
    assert model.model.embed_tokens.weight.shape[0] == cfg.vocab_size


@pytest.mark.parametrize("depth,merge", [(2, 1), (2, 2), (1, 3), (2, 4)])
def test_pipelined_runner_matches_sequential(setup, depth, merge):
    """Batches in flight on separate HIP streams / decode sessions — and, with merge > 1, decode steps of several batches
    sharing one session (in-flight batching) — give results bit-identical to one rec_batch call per batch.  Five ragged
    batches: groups of `merge` do not divide them, so partially filled groups and every row offset are exercised."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    T = 9
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    shapes = [[[1, 8, 8], [1, 10, 12]], [[1, 6, 10], [1, 8, 8]], [[1, 10, 12], [1, 10, 12]], [[1, 8, 8], [1, 6, 10]], [[1, 8, 8], [1, 10, 12]]]
    for s, g in enumerate(shapes):
        grid, pix, ids, am = U.synthetic_batch(cfg, g, n_pre=5 + s % 2, n_post=7, seed=100 + s, ragged=True)
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    ref = [pipeline.rec_batch(model, proc, b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched) for b in batches]
    runner = pipeline.PipelinedRunner(model, proc, depth=depth, merge=merge)
    got = []
    for b in batches:
        got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
    got += runner.flush()
    assert len(got) == len(batches)
    for (d0, c0, l0, v0), (d1, c1, l1, v1) in zip(ref, got):
        assert c0 == c1 and v0 == v1
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"])
        assert torch.equal(d0["pred_score"], d1["pred_score"])


def test_pipelined_runner_with_a_vit_stream_matches_sequential(setup):
    """PipelinedRunner(vit_stream=True): the ViT of batch b + 1 runs on its own stream, concurrent with the LLM prefill of batch b (only
    the first ViT of a decode group is ordered after the stream's earlier work) — results bit-identical to one rec_batch call per batch."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    T = 9
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s_ in range(7):
        g = [[1, 8, 8] if (s_ + i) % 3 else [1, 10, 12] for i in range(4)]
        grid, pix, ids, am = U.synthetic_batch(cfg, g, n_pre=5 + s_ % 2, n_post=7, seed=1300 + s_, ragged=True)
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    ref = [pipeline.rec_batch(model, proc, b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched) for b in batches]
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=3, vit_stream=True)
    got = []
    for rep in range(2):                                            # second pass: sessions and graphs already exist
        got = []
        for b in batches:
            got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
        got += runner.flush()
        assert len(got) == len(batches)
        for (d0, c0, l0, v0), (d1, c1, l1, v1) in zip(ref, got):
            assert c0 == c1 and v0 == v1
            assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"]) and torch.equal(d0["pred_score"], d1["pred_score"])


def test_pipelined_runner_128_row_decode_groups(setup):
    """Decode groups of 16 batches of 8 (128-row decode steps: what bench.py runs for the decode-heavy OVD / RIC shapes) — 18 batches, so one
    full group and a partially filled one — bit-identical to one rec_batch call per batch."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    T = 12
    sched = U.rec_schedule(T, vrt_at=range(4, 8))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s_ in range(18):
        g = [[1, 8, 8] if (s_ + i) % 3 else [1, 6, 10] for i in range(8)]
        grid, pix, ids, am = U.synthetic_batch(cfg, g, n_pre=5 + s_ % 2, n_post=7, seed=300 + s_, ragged=True)
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=16)
    got = []
    for b in batches:
        got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
    got += runner.flush()
    assert len(got) == len(batches)
    for i in (0, 7, 15, 16, 17):
        b = batches[i]
        d0, c0, l0, v0 = pipeline.rec_batch(model, proc, b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
        d1, c1, l1, v1 = got[i]
        assert c0 == c1 and v0 == v1, f"batch {i}: tokens differ in a 128-row decode group"
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"]) and torch.equal(d0["pred_score"], d1["pred_score"])


@pytest.mark.parametrize("merge", [5, 12])
def test_pipelined_runner_group_sizes_that_do_not_fill_the_row_blocks(setup, merge):
    """Decode groups of 5 / 12 batches of 8: 40- / 96-row steps, i.e. 3 of 4 / 6 of 8 sixteen-row blocks of the decode kernels' launch shape.
    The fragment-packed activation buffers hold exactly ceil(rows / 16) blocks; the kernels must not touch the blocks past them (a 96-row
    step of PaDT_Pro_3B read 2 x 16 rows x 11008 past the end of the MLP activation buffer and faulted before the guard in gemm_skinny_kernel /
    vrt_head_kernel).  Bit-identical to one rec_batch call per batch."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    T = 10
    sched = U.rec_schedule(T, vrt_at=range(3, 7))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s_ in range(merge + 2):
        g = [[1, 8, 8] if (s_ + i) % 3 else [1, 6, 10] for i in range(8)]
        grid, pix, ids, am = U.synthetic_batch(cfg, g, n_pre=5 + s_ % 2, n_post=7, seed=900 + s_, ragged=True)
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=merge)
    got = []
    for b in batches:
        got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
    got += runner.flush()
    assert len(got) == len(batches)
    for i in (0, merge - 1, merge, merge + 1):
        b = batches[i]
        d0, c0, l0, v0 = pipeline.rec_batch(model, proc, b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
        d1, c1, l1, v1 = got[i]
        assert c0 == c1 and v0 == v1, f"batch {i}: tokens differ in a decode group of {merge} batches"
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"]) and torch.equal(d0["pred_score"], d1["pred_score"])


@pytest.mark.parametrize("variant", ["untied_head_gqa2_multi_object", "no_prototype_projection_no_mask_head"])
def test_generate_config_variants(variant):
    """The other configurations the reference ships (BASELINE.json configs 3-5): untied lm_head + a GQA group of 2 with an
    OVD-style completion holding TWO VRT runs per image (7B-style head, padt.py:292-297), and
    use_visual_prototype_projection=False (padt.py:191) with the mask head off (padt_decoder.py:235-236)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dataclasses
    import padt_amd
    from padt_amd.modeling import PaDTForConditionalGeneration
    import parity_util as U
    O = U.O
    cfg = padt_amd.small_test_config()
    if variant == "untied_head_gqa2_multi_object":
        cfg = dataclasses.replace(cfg, tie_word_embeddings=False, num_attention_heads=4, num_key_value_heads=2, hidden_size=512)
        cfg = dataclasses.replace(cfg, vision_config=dataclasses.replace(cfg.vision_config, out_hidden_size=512))
    else:
        cfg = dataclasses.replace(cfg, use_visual_prototype_projection=False,
                                  vl_decoder=dict(cfg.vl_decoder, use_mask_loss=False))
    w = U.bf16_weights(cfg, seed=9, std=0.05)
    oc = U.oracle_config(cfg)
    model = PaDTForConditionalGeneration(cfg, w, device="cuda")
    grids = [[1, 8, 8], [1, 10, 12]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=5, n_post=8, ragged=True)
    T = 14
    sched = ["t"] * T
    for i in (2, 3, 4, 8, 9):
        sched[i] = "v"                                               # two VRT runs → two objects per image
    sched[-1] = "e"
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid,
                         max_new_tokens=T, schedule=sched)
    seq = out.sequences.cpu()
    L = ids.shape[1]
    toks = seq[:, L:]
    V = cfg.vocab_size
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    n_tie = 0
    for t in range(T):                                              # margin rule, as in the main parity test
        lg = ores["logits"][t]
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        floor = 2e-2 * lg[torch.isfinite(lg)].abs().max().item()
        for b in range(2):
            second = top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1
            if (top2[b, 0] - second).item() > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: not the oracle argmax"
            else:
                n_tie += 1
                assert (top2[b, 0] - chosen[b]).item() <= floor
    assert n_tie <= T
    hid = out.hidden_states.last_layer_rows().cpu().float()
    for t in range(T):
        mx, rms = rel_err(hid[t], ores["hidden"][t][:, -1].float())
        assert rms < 2e-2 and mx < 8e-2, f"hidden step {t}: rel err max {mx:.3e} rms {rms:.3e}"
    n_m = [g[1] * g[2] // 4 for g in grids]
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, max(n_m)), 2)
    proc.model_embed_token_size = V
    local = proc.assign_to_local_vrt_id(seq.clone(), grid)[:, L:]
    comps, feats, labels, vrts, _ = padt_amd.parseVRTintoCompletion(proc, local, out["hidden_states"], torch.Tensor([False] * 2))
    assert [len(f) for f in feats] == [2, 2] and [f.shape[0] for f in feats[0]] == [3, 2]
    dec = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    odec = O.vl_decode(w, oc, [[f.cpu().float() for f in fs] for fs in feats], out.past_image_embeds.cpu().float(),
                       out.past_high_res_image_embeds.cpu().float(), grid,
                       (out.past_visual_pe[0].cpu(), out.past_visual_pe[1].cpu()))
    assert dec["sample_idx"] == odec["sample_idx"] == [0, 0, 1, 1]
    assert (dec["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item() < 2e-3
    assert (dec["pred_score"].cpu().float() - odec["pred_score"]).abs().max().item() < 5e-2 * (odec["pred_score"].abs().max().item() + 1)
    if variant == "untied_head_gqa2_multi_object":
        mx, rms = rel_err(dec["pred_mask"], odec["pred_mask"])
        assert rms < 2e-2 and mx < 1e-1
        assert torch.equal(dec["pred_mask_valid_hw"][0].cpu(), odec["pred_mask_valid_hw"][0])
    else:
        assert dec["pred_mask"] is None and odec["pred_mask"] is None and dec["pred_mask_valid_hw"] == ()


def test_postprocess_results_end_to_end(setup):
    """vl_decode output → padt_amd.postprocess (HIP mask kernel + host box / RLE logic) vs the oracle's restatement of the
    reference's eval-loop expressions on the SAME decoded tensors: boxes / RLE run lengths bit-exact, masks identical."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline, postprocess
    grids = [[1, 10, 12], [1, 8, 8]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=5, n_post=7, ragged=True)
    T = 10
    sched = U.rec_schedule(T, vrt_at=range(3, 7))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    decoded, comps, labels, vrts = pipeline.rec_batch(model, proc, ids.cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
    sizes = [(173, 140), (112, 112)]                               # (w, h) of the original images
    got = postprocess.postprocess_results(decoded, labels, sizes)
    cpu = {"pred_boxes": decoded["pred_boxes"].float().cpu(), "pred_score": decoded["pred_score"].float().cpu(),
           "pred_mask": decoded["pred_mask"].float().cpu(),
           "pred_mask_valid_hw": tuple(t.cpu() for t in decoded["pred_mask_valid_hw"]), "sample_idx": decoded["sample_idx"]}
    ref = U.O.postprocess_results(cpu, labels, sizes)
    assert len(got) == len(ref) == 2
    # the metric's "box IoU vs ref": HIP boxes vs the fp32 oracle's own end-to-end boxes (its own tokens, features, decoder)
    ores = U.O.generate(w, oc, ids, am, pix, grid, T, schedule=sched)
    st = ores["state"]
    ofeats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in range(3, 7)], 0)] for b in range(2)]
    odec = U.O.vl_decode(w, oc, ofeats, st.proto, st.high_res, grid, st.visual_pe)
    oref = U.O.postprocess_results(odec, labels, sizes)
    for g, r in zip(got, oref):
        iou = postprocess.box_iou_xywh(g["bbox"], r["bbox"])
        inter, union = U.O.mask_ciou_parts(g["mask"], r["mask"])
        assert iou >= 0.97, f"box IoU vs oracle {iou:.4f}"
        assert inter / max(union, 1) >= 0.97, f"mask IoU vs oracle {inter / max(union, 1):.4f}"
    for g, r in zip(got, ref):
        assert g["bbox"] == r["bbox"] and g["category"] == r["category"] and g["sample_idx"] == r["sample_idx"]
        assert abs(g["score"] - r["score"]) < 1e-6
        diff = g["mask"] != r["mask"]
        assert (r["mask_logits_up"].abs().numpy()[diff] < 1e-5).all() and diff.sum() <= 1
        if not diff.any():
            assert g["rle"] == r["rle"]


def test_early_eos_stop_rule_and_chunked_sync(setup):
    """EOS before max_new_tokens: the output is trimmed right after the step in which the last sequence finished
    (padt.py:756-757), identically for one-shot decode, chunked host syncs (sync_every) and a merged decode group."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=7, seed=7, ragged=True)
    T = 12
    sched = ["t", "t", "v", "v", "t", "e"] + ["t"] * (T - 6)          # EOS forced at step 5 → 6 tokens
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
              schedule=sched)
    a = model.generate(sync_every=T, **kw)
    b = model.generate(sync_every=2, **kw)                           # host checks `unfinished` every 2 steps
    c = model.generate(sync_every=2, use_graph=False, **kw)
    L = ids.shape[1]
    assert a.sequences.shape[1] == L + 6 and (a.sequences[:, -1] == cfg.eos_token_id).all()
    assert torch.equal(a.sequences, b.sequences) and torch.equal(a.sequences, c.sequences)
    assert len(a.hidden_states) == 6 and torch.equal(a.hidden_states.last_layer_rows(), b.hidden_states.last_layer_rows())
    ores = U.O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, force_tokens=a.sequences.cpu()[:, L:])
    assert ores["sequences"].shape == a.sequences.shape
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    ref = pipeline.rec_batch(model, proc, ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=2)
    got = []
    for _ in range(3):
        got += runner.submit(ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched, sync_every=3)
    got += runner.flush()
    assert len(got) == 3
    for d, c1, l1, v1 in got:
        assert c1 == ref[1] and v1 == ref[3] and torch.equal(d["pred_boxes"], ref[0]["pred_boxes"])


def test_from_pretrained_directory_round_trip(setup, tmp_path):
    """from_pretrained(config.json + *.safetensors with the checkpoint's key names) builds the same model as the in-memory
    constructor: identical sequences and boxes (the drop-in construction surface, test_demo.py:20-25)."""
    cfg, w, model, U, oc = setup
    import json
    from safetensors.torch import save_file
    from padt_amd.modeling import PaDTForConditionalGeneration
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in w.items()}, str(tmp_path / "model.safetensors"))
    json.dump(cfg.to_dict(), open(tmp_path / "config.json", "w"))
    m2 = PaDTForConditionalGeneration.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16, attn_implementation="flash_attention_2",
                                                      device_map={"": 0}, operands="fp16" if model.dtype == torch.float16 else "bf16")
    assert m2.config.vision_config.spatial_merge_size == 2 and m2.model.embed_tokens.weight.shape[0] == cfg.vocab_size
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=7, seed=3, ragged=True)
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 5))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
              schedule=sched)
    a, b = model.generate(**kw), m2.generate(**kw)
    assert torch.equal(a.sequences, b.sequences)
    assert torch.equal(a.hidden_states.last_layer_rows(), b.hidden_states.last_layer_rows())
    with pytest.raises(FileNotFoundError):
        PaDTForConditionalGeneration.from_pretrained(str(tmp_path / "missing"))


def test_infer_dataset_harness_writes_reference_jsonl(setup, tmp_path):
    """eval/evaluation_scripts/utils.py:176-266 on the throughput runner: rank striding, ragged last batch, both JSONL files
    with the reference's record schema; records equal to the sequential rec_batch + postprocess path."""
    cfg, w, model, U, oc = setup
    import json
    import padt_amd
    from padt_amd import harness, pipeline, postprocess
    T = 9
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    shapes = [[1, 8, 8], [1, 10, 12], [1, 6, 10], [1, 8, 8], [1, 10, 12]]
    dataset = [{"id": 100 + i, "grid": g, "size": (50 + 7 * i, 40 + 5 * i)} for i, g in enumerate(shapes)]

    def prepare(samples):
        grid, pix, ids, am = U.synthetic_batch(cfg, [s["grid"] for s in samples], n_pre=5, n_post=7, seed=samples[0]["id"], ragged=True)
        return {"input_ids": ids.cuda(), "attention_mask": am.cuda(), "pixel_values": pix.cuda(), "image_grid_thw": grid,
                "image_sizes": [s["size"] for s in samples], "ids": [s["id"] for s in samples]}

    seen = []
    for rank in range(2):
        info = harness.infer_dataset(model, proc, dataset, prepare, str(tmp_path), batch_size=2, datasetname="syn", suffix="t",
                                     rank=rank, world=2, max_new_tokens=T, depth=2, merge=2, schedule=sched)
        comp = [json.loads(l) for l in open(info["completions_file"])]
        res = [json.loads(l) for l in open(info["results_file"])]
        assert os.path.basename(info["results_file"]) == f"syn_{rank}_pred_results_t.json"
        assert len(comp) == info["samples"] and all(set(c) == {"image_id", "completion"} for c in comp)
        assert all(set(r) == {"image_id", "score", "category", "bbox", "mask"} for r in res) and len(res) == info["samples"]
        seen += [c["image_id"] for c in comp]
        # the same records through the sequential path
        k = 0
        for idx in pipeline.rank_batches(len(dataset), 2, rank, 2):
            if idx >= len(dataset):
                continue
            b = prepare(dataset[idx: idx + 2])
            decoded, completions, labels, vrts = pipeline.rec_batch(model, proc, b["input_ids"], b["attention_mask"], b["pixel_values"],
                                                                    b["image_grid_thw"], max_new_tokens=T, schedule=sched)
            for r in postprocess.postprocess_results(decoded, labels, b["image_sizes"]):
                got = res[k]
                k += 1
                assert got["image_id"] == b["ids"][r["sample_idx"]] and got["bbox"] == list(r["bbox"]) and got["mask"] == r["rle"]
                assert abs(got["score"] - r["score"]) < 1e-6 and got["mask"]["size"] == [b["image_sizes"][r["sample_idx"]][1], b["image_sizes"][r["sample_idx"]][0]]
        assert k == len(res)
    assert sorted(seen) == [100, 101, 102, 103, 104]


def test_repetition_penalty_and_eos_list(setup):
    """generation_config fidelity: HF's RepetitionPenaltyLogitsProcessor fused into the logit head (seen-id
    bitmap incl. prompt + padding ids) and an EOS id list — against the oracle with the same processor (margin rule)."""
    cfg, w, model, U, oc = setup
    O = U.O
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 10, 12], [1, 8, 8], [1, 6, 10]], n_pre=6, n_post=9, ragged=True)
    T = 12
    sched = U.rec_schedule(T, vrt_at=range(4, 8))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
              schedule=sched)
    L = ids.shape[1]
    base = model.generate(**kw).sequences.cpu()[:, L:]
    pen = 1.6
    out = model.generate(repetition_penalty=pen, **kw)
    toks = out.sequences.cpu()[:, L:]
    assert not torch.equal(toks, base), "penalty changed nothing: the test has no power"
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks, repetition_penalty=pen)
    n_tie = 0
    for t in range(toks.shape[1]):
        lg = ores["logits"][t]
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        floor = 2e-2 * lg[torch.isfinite(lg)].abs().max().item()
        for b in range(3):
            second = top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1
            if (top2[b, 0] - second).item() > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: not the penalised argmax"
            else:
                n_tie += 1
                assert (top2[b, 0] - chosen[b]).item() <= floor
    assert n_tie <= T * 3 // 2
    # graph replay with a different penalty value afterwards: the value lives in device memory, not in the captured graph
    again = model.generate(**kw).sequences.cpu()[:, L:]
    assert torch.equal(again, base)
    # EOS list: the token sample 0 produced at step 1 becomes a second EOS id → that row stops there and is padded
    extra = int(base[0, 1])
    stop = model.generate(eos_token_id=[cfg.eos_token_id, extra], **kw).sequences.cpu()[:, L:]
    assert torch.equal(stop[0, :2], base[0, :2]) and (stop[0, 2:] == cfg.pad_token_id).all()
    for b in (1, 2):
        first = (base[b] == extra).nonzero()
        upto = int(first[0]) + 1 if len(first) else T
        assert torch.equal(stop[b, :upto], base[b, :upto])
    oref = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, force_tokens=stop, eos_token_ids=[cfg.eos_token_id, extra])
    assert torch.equal(oref["sequences"][:, L:], stop[:, : oref["sequences"].shape[1] - L])
    with pytest.raises(NotImplementedError):
        model.generate(eos_token_id=[extra], **kw)


def test_ovd_shaped_completion_through_the_runner(setup):
    """OVD-shaped workload (BASELINE configs[3]: several objects per image, eval/evaluation_scripts/inference_coco.py:101-110) end to
    end on the throughput runner: 7 VRT runs per completion → 7 objects per image through parse + vl_decode, merged decode groups;
    boxes / masks against the oracle's own end-to-end run, object bookkeeping exact."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    from padt_amd.synthetic import multi_object_schedule
    O = U.O
    grids = [[1, 10, 12], [1, 8, 8]]
    T = 48
    sched = multi_object_schedule(T, n_obj=7, n_vrt=3)
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 30), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = [U.synthetic_batch(cfg, grids, n_pre=6, n_post=20, ragged=True, seed=900 + i) for i in range(3)]
    runner = pipeline.PipelinedRunner(model, proc, depth=2, merge=2)
    results = []
    for grid, pix, ids, am in batches:
        results += runner.submit(ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
    results += runner.flush()
    assert len(results) == 3
    segs, cur = [], []
    for t, m in enumerate(sched):
        if m == "v":
            cur.append(t)
        elif cur:
            segs.append(cur)
            cur = []
    for i, ((grid, pix, ids, am), (decoded, completions, labels, vrts)) in enumerate(zip(batches, results)):
        assert decoded["pred_boxes"].shape == (14, 4) and decoded["sample_idx"] == [0] * 7 + [1] * 7
        assert all(len(v) == 7 for v in vrts)
        # the same batch run alone: identical objects, bit for bit (merged decode groups change nothing per sample)
        solo, _, _, svrts = pipeline.rec_batch(model, proc, ids.clone().cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, schedule=sched)
        assert svrts == vrts and torch.equal(solo["pred_boxes"], decoded["pred_boxes"]) and torch.equal(solo["pred_mask"], decoded["pred_mask"])
        if i:
            continue
        # oracle teacher-forced on the HIP tokens → its own features → its own decoder: end-to-end parity of all 14 objects
        gids = proc.assign_to_global_vrt_id(ids.clone(), grid)
        out = model.generate(input_ids=gids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid,
                             max_new_tokens=T, schedule=sched)
        toks = out.sequences.cpu()[:, ids.shape[1]:]
        ores = O.generate(w, oc, gids, am, pix, grid, T, schedule=sched, force_tokens=toks)
        st = ores["state"]
        feats = [[torch.cat([ores["hidden"][t][b:b + 1, -1] for t in sg], 0) for sg in segs] for b in range(2)]
        odec = O.vl_decode(w, oc, feats, st.proto, st.high_res, grid, st.visual_pe)
        db = (decoded["pred_boxes"].cpu().float() - odec["pred_boxes"]).abs().max().item()
        mx, rms = rel_err(decoded["pred_mask"], odec["pred_mask"])
        print(f"\n[OVD-shaped e2e, {model.dtype}] 14 objects: box |d|max {db:.3e} mask rel max {mx:.3e} rms {rms:.3e}")
        # measured: bf16 operands 1.4e-4 / 4.6e-3 rms, fp16 operands 1.9e-5 / 5.0e-4 rms — bounds ≈3x
        assert db < tol(model, 4e-4, 6e-5) and rms < tol(model, 1.4e-2, 1.5e-3), f"OVD e2e: box {db:.3e} mask rms {rms:.3e}"


def test_generate_with_sampling(setup):
    """do_sample=True through generate(): repeatable for a seed (eager == hipGraph), different seeds differ, every drawn token lies in
    the oracle's top-k of that step under teacher forcing (up to the bf16 noise floor at the k-th boundary), top_k=1 degenerates to
    the greedy path, generation_config defaults are picked up."""
    cfg, w, model, U, oc = setup
    O = U.O
    grids = [[1, 8, 8], [1, 10, 12]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=5, n_post=8, ragged=True, seed=31)
    T, K = 10, 5
    sched = ["t"] * T
    sched[-1] = "e"
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    a = model.generate(do_sample=True, top_k=K, temperature=1.5, seed=11, **kw)
    b = model.generate(do_sample=True, top_k=K, temperature=1.5, seed=11, use_graph=False, **kw)
    c = model.generate(do_sample=True, top_k=K, temperature=1.5, seed=12, **kw)
    assert torch.equal(a.sequences, b.sequences) and not torch.equal(a.sequences, c.sequences)
    L = ids.shape[1]
    toks = a.sequences.cpu()[:, L:]
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    for t in range(T - 1):
        lg = ores["logits"][t]
        kth = lg.topk(K, dim=-1).values[:, -1]
        floor = 2e-2 * lg[torch.isfinite(lg)].abs().max().item()
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        assert bool((chosen >= kth - floor).all()), f"step {t}: a drawn token is outside the oracle's top-{K}"
    greedy = model.generate(do_sample=False, **kw)
    assert torch.equal(model.generate(do_sample=True, top_k=1, **kw).sequences, greedy.sequences)
    assert not torch.equal(a.sequences, greedy.sequences)
    model.load_generation_config({"do_sample": True, "top_k": 1, "temperature": 0.1, "top_p": 0.001, "repetition_penalty": 1.0})
    try:
        assert torch.equal(model.generate(**kw).sequences, greedy.sequences)           # Qwen2.5-VL's shipped config: top_k = 1 → greedy
    finally:
        model.load_generation_config({"do_sample": False, "top_k": 50, "temperature": 1.0, "top_p": 1.0})
    with pytest.raises(NotImplementedError):
        model.generate(do_sample=True, top_k=0, top_p=0.9, **kw)


@pytest.mark.parametrize("llm_weights", ["fp8", "fp8+act"])
def test_fp8_llm_weights_against_oracle_on_dequantised_weights(llm_weights):
    """BASELINE configs[4] (PaDT_Pro_7B-style: untied head, GQA group of 2, fp8 weight path): LLM projections quantised to e4m3 with
    power-of-two row scales; "fp8": prefill multiplies the exactly dequantised 16-bit image, decode streams the fp8 image; "fp8+act" (opt-in):
    the prompt pass runs fp8 x fp8 MFMA GEMMs over e4m3 activation rows — the oracle runs the SAME dequantised matrices in fp32
    (parity_util.effective_llm_weights) and, for "fp8+act", quantises the same rows.  Ids by the margin rule, hidden rows, boxes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dataclasses
    import padt_amd
    from padt_amd.modeling import PaDTForConditionalGeneration
    import parity_util as U
    O = U.O
    cfg = padt_amd.small_test_config()
    cfg = dataclasses.replace(cfg, tie_word_embeddings=False, num_attention_heads=4, num_key_value_heads=2, hidden_size=512)
    cfg = dataclasses.replace(cfg, vision_config=dataclasses.replace(cfg.vision_config, out_hidden_size=512))
    w = U.bf16_weights(cfg, seed=19, std=0.05)
    model = PaDTForConditionalGeneration(cfg, w, device="cuda", llm_weights=llm_weights)
    act8 = llm_weights == "fp8+act"
    assert model.W.llm_weights == "fp8" and "llm.0.qkv.wq" in model.W and "llm.0.qkv.wp" not in model.W
    assert model.W.fp8_prefill == act8 and ("llm.0.qkv.w8" in model.W) == act8 and ("llm.0.o.w8" in model.W) == act8   # 512-wide: qkv and o take the fp8 MFMA GEMM
    wo = U.effective_llm_weights(model, w)
    oc = U.oracle_config(cfg)
    grids = [[1, 8, 8], [1, 10, 12]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=5, n_post=8, ragged=True, seed=77)
    T = 12
    sched = U.rec_schedule(T, vrt_at=range(4, 8))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
                         schedule=sched)
    L = ids.shape[1]
    toks = out.sequences.cpu()[:, L:]
    with U.fp8_prefill_hooks(model):                             # prompt pass: e4m3 activation rows into the fp8 x fp8 MFMA GEMMs
        ores = O.generate(wo, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks)
    n_tie = 0
    for t in range(T):
        lg = ores["logits"][t]
        top2 = lg.topk(2, dim=-1).values
        chosen = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        floor = 2e-2 * lg[torch.isfinite(lg)].abs().max().item()
        for b in range(2):
            second = top2[b, 1] if torch.isfinite(top2[b, 1]) else top2[b, 0] - 1
            if (top2[b, 0] - second).item() > floor:
                assert chosen[b] == top2[b, 0], f"step {t} sample {b}: not the oracle argmax"
            else:
                n_tie += 1
                assert (top2[b, 0] - chosen[b]).item() <= floor
    assert n_tie <= T
    hid = out.hidden_states.last_layer_rows().cpu().float()
    worst = 0.0
    for t in range(T):
        mx, rms = rel_err(hid[t], ores["hidden"][t][:, -1].float())
        worst = max(worst, rms)
        # e4m3 activation rows make the prompt pass discontinuous: an upstream difference of one bf16 rounding flips ≈5 % of the e4m3 codes of
        # an activation row, each by a full e4m3 step (12.5 %) — two implementations of the SAME quantised function agree to a few per cent,
        # not to bf16 noise (the GEMM itself is exact against its dequantised operands: test_gemm_fp8_mfma_against_fp32_on_the_dequantised_operands)
        assert (rms < 6e-2 and mx < 2e-1) if act8 else (rms < 5e-3 and mx < 2e-2), f"hidden step {t}: rel err max {mx:.3e} rms {rms:.3e}"
    print(f"\n[{llm_weights} e2e, 512-wide] hidden rel rms worst {worst:.3e}")
    # the quantisation itself is visible against the UN-quantised oracle (sanity: the test is not vacuous)
    ores0 = O.generate(w, oc, ids, am, pix, grid, 1, schedule=sched)
    _, rms0 = rel_err(hid[0], ores0["hidden"][0][:, -1].float())
    assert rms0 > 2e-2


def test_bench_contract_small_config():
    """`python bench.py` prints ONE JSON line with the driver's contract fields, the roofline and cpu_baseline objects and the parity
    read-out — exercised on the small plumbing config so that it runs in seconds (the numbers mean nothing at this size)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "small", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_decode", "from_images", "to_rle", "cpu_baseline", "parity_vs_oracle", "steady_state", "range_guard"):
        assert k in d, k
    assert d["range_guard"]["operands_policy"] == "auto" and d["range_guard"]["batches_rerun_on_bf16"] == 0 and d["steady_state"]["value"] > 0
    assert d["cpu_baseline"]["cores"] == min(32, os.cpu_count() or 1)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["unit"] == "images/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_replay", "how"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    rd = d["roofline_decode"]
    assert rd["bound"] == "hbm" and rd["rows_per_step"] == 64 and rd["us_per_step"] > 0 and rd["us_per_step_alone"] > 0 and rd["bytes_per_step"] > 0
    assert abs(rd["frac"] - rd["achieved"] / rd["peak"]) < 1e-3 and d["from_images"]["value"] > 0
    assert d["to_rle"]["value"] > 0 and d["to_rle"]["records"] == 8 * d["to_rle"]["steps"] and d["dtype"] == "fp16" and "traffic_note" in rf
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "parity"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    same, total = (int(v) for v in cb["parity"]["tokens_equal_oracle_argmax"].split("/"))
    assert total > 0 and same >= total - 2, cb["parity"]
    assert all(iou > 0.9 for iou in cb["parity"]["box_iou_vs_oracle"]), cb["parity"]


def test_pack_results_kernel_matches_the_host_statement():
    """padt_pack_results (one launch, no host round trip) == the indexing statements of pipeline.pack_results on host copies, bit for
    bit: header, sample indices, valid sizes, boxes, scores, mask window, zeros elsewhere; empty batch; no mask head."""
    from padt_amd import pipeline
    g = torch.Generator().manual_seed(5)
    for n, H, Wd, cap, mhw in ((5, 24, 32, 8, 40), (8, 40, 40, 8, 40), (0, 8, 8, 4, 16)):
        dec = {"pred_boxes": torch.rand(n, 4, generator=g), "pred_score": torch.randn(n, 1, generator=g), "pred_mask": torch.randn(n, H, Wd, generator=g),
               "sample_idx": [i % 3 for i in range(n)], "pred_mask_valid_hw": (torch.full((n,), H - 4), torch.full((n,), Wd - 8)) if n else ()}
        ref = pipeline.pack_results(dec, cap, mhw, "cpu")
        dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in dec.items()}
        if n:
            dev["pred_mask_valid_hw"] = tuple(t.cuda() for t in dec["pred_mask_valid_hw"])
        dirty = torch.full((ref.numel(),), 77, dtype=torch.int32, device="cuda")
        got = pipeline.pack_results(dev, cap, mhw, "cuda", out=dirty)
        assert torch.equal(got.cpu(), ref), (n, H, Wd)
        if n:
            nomask = dict(dev, pred_mask=None, pred_mask_valid_hw=())
            ref2 = pipeline.pack_results(dict(dec, pred_mask=None, pred_mask_valid_hw=()), cap, mhw, "cpu")
            assert torch.equal(pipeline.pack_results(nomask, cap, mhw, "cuda").cpu(), ref2)
    with pytest.raises(ValueError, match="exceed the exchange capacity"):
        pipeline.pack_results({"pred_boxes": torch.zeros(5, 4).cuda()}, 4, 16, "cuda")


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_exchange_delivers_every_ranks_results(tmp_path, world):
    """The REAL multi-rank path of bench.py, launched AS TYPED — `python bench.py --gpus N`, which re-launches itself under
    torch.distributed.run — with 2 and with 8 ranks (gloo, all on this GPU — RCCL refuses two ranks per device; the driver's 8-GPU run uses
    nccl = RCCL): device-side pack, one asynchronous all-gather per decode group, and every rank ends up with every rank's boxes / scores /
    mask logits bit for bit; ONE JSON line on stdout.  Small plumbing config."""
    import json
    import subprocess
    import sys
    from padt_amd import pipeline
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PADT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    steps = 5
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--model", "small", "--steps", str(steps), "--warmup", "1",
                        "--merge", "2", "--dump-exchange", str(tmp_path)], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["exchange"]["all_gathers"] >= 3 and d["exchange"]["batches_per_gather"] == 2
    assert d["config"]["global_batch"] == 8 * world and d["config"]["parallelism"] == f"dp{world}" and d["exchange"]["backend"] == "gloo"
    # rank striding (utils.py:181-182): `world` ranks x `steps` batches of 8 cover the items of a dataset of that size exactly once
    covered = sorted(s_ for k in range(world) for s_ in pipeline.rank_batches(8 * steps * world, 8, k, world))
    assert covered == list(range(0, 8 * steps * world, 8))
    dumps = [torch.load(os.path.join(str(tmp_path), f"rank{k}.pt")) for k in range(world)]
    for me in range(world):
        recs = torch.cat(dumps[me]["gathered"], dim=1)                          # GatheredGroup.records: (world, batches rounded up to whole gathers, words)
        assert all(c is None for c in dumps[me]["continuation"])
        assert recs.shape[0] == world and recs.shape[1] >= steps
        for src in range(world):
            local = dumps[src]["local"]
            assert len(local) == steps
            for b in range(recs.shape[1]):
                got = pipeline.unpack_results(recs[src, b][None], batch_per_rank=0)[0]
                if b >= steps:
                    assert got["boxes"].shape[0] == 0               # zero-padded records of the last, partially filled gather
                    continue
                ref = local[b]
                assert torch.equal(got["boxes"], ref["pred_boxes"].float()) and torch.equal(got["scores"], ref["pred_score"].float().reshape(-1))
                H, Wd = ref["pred_mask"].shape[1:]
                assert torch.equal(got["masks"][:, :H, :Wd], ref["pred_mask"].float()) and got["sample_idx"].tolist() == list(ref["sample_idx"])


def test_bench_world_one_exchange_runs_on_rccl():
    """The data-parallel code path at world size 1 with the DEFAULT backend (nccl = RCCL): process group on the device, asynchronous
    all_gather_into_tensor of the packed records on RCCL's stream while the other lane captures / replays its decode graph — the one way to
    execute the exchange on RCCL itself on a one-GPU box (PADT_DIST_FORCE=1)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PADT_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29647")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PADT_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "small", "--steps", "6", "--warmup", "1", "--merge", "2"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["exchange"]["backend"] == "nccl" and d["exchange"]["all_gathers"] >= 3
    assert "roofline" not in d and "cpu_baseline" not in d          # the distributed path prints the contract line only


def test_output_scores_are_the_oracles_processed_logits(setup):
    """generate(output_scores=True) (padt.py:719-720): a T-tuple of (B, V + N) fp32 rows — the step's logits after the logit mask (padt.py:292-301)
    and the logits processors — against the oracle's rows of the same (teacher-forced) run: same -inf support at every step, finite entries
    within the operand type's logit noise; also past_logit_mask (padt.py:196-201), output_logits (== scores when no processor is active) and
    the same rows out of a merged decode group (this batch's own prototype columns of the session-wide rows)."""
    cfg, w, model, U, oc = setup
    import padt_amd
    from padt_amd import pipeline
    grids = [[1, 10, 12], [1, 8, 8]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=6, n_post=9, ragged=True, seed=77)
    T = 9
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    out = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T,
                         schedule=sched, output_scores=True, repetition_penalty=1.3)
    V, N = cfg.vocab_size, sum(g[1] * g[2] // 4 for g in grids)
    assert isinstance(out.scores, tuple) and len(out.scores) == T and out.logits is None
    toks = out.sequences.cpu()[:, ids.shape[1]:]
    ores = U.O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, collect_logits=True, force_tokens=toks, repetition_penalty=1.3)
    worst = 0.0
    for t in range(T):
        s, o = out.scores[t].cpu(), ores["logits"][t]
        assert s.shape == (2, V + N) and s.dtype == torch.float32
        if sched[t] == "e":                                         # forced EOS: the oracle's processor writes 0 at the EOS column, the kernel keeps the logit
            assert torch.isfinite(s).sum(dim=1).tolist() == [1, 1] and torch.isfinite(s[:, cfg.eos_token_id]).all()
            continue
        assert torch.equal(torch.isfinite(s), torch.isfinite(o)), f"step {t}: different -inf support"
        fin = torch.isfinite(o)
        worst = max(worst, ((s[fin] - o[fin]).abs().max() / o[fin].abs().max()).item())
    print(f"\n[output_scores, {model.dtype}] worst |score - oracle| / max|logit| over {T - 1} steps: {worst:.3e}")
    assert worst < tol(model, 2e-2, 3e-3)                           # measured ≈6e-3 (bf16) / 8e-4 (fp16): the logit noise of the operand type
    # past_logit_mask: text columns + the sample's own prototype columns
    m = out.past_logit_mask.cpu()
    assert m.dtype == torch.bool and m.shape == (2, V + N)
    n0 = grids[0][1] * grids[0][2] // 4
    exp = torch.zeros(2, V + N, dtype=torch.bool)
    exp[:, :V] = True
    exp[0, V: V + n0] = True
    exp[1, V + n0:] = True
    assert torch.equal(m, exp)
    # output_logits without a processor == output_scores; with one it is refused
    o2 = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=4,
                        output_scores=True, output_logits=True)
    assert len(o2.logits) == len(o2.scores) and all(torch.equal(a, b) for a, b in zip(o2.logits, o2.scores))
    with pytest.raises(NotImplementedError, match="output_logits"):
        model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=4,
                       schedule=sched, output_logits=True)
    # the second batch of a merged decode group: its scores are the stand-alone run's, bit for bit (rows and columns of the shared session)
    b0 = (ids.clone().cuda(), am.cuda(), pix.cuda(), grid)
    grid1, pix1, ids1, am1 = U.synthetic_batch(cfg, grids, n_pre=6, n_post=9, ragged=True, seed=78)
    alone = model.generate(input_ids=ids1.cuda(), attention_mask=am1.cuda(), pixel_values=pix1.cuda(), image_grid_thw=grid1, max_new_tokens=T,
                           schedule=sched, output_scores=True, lane=5)
    g = model.generate_launch(*b0, max_new_tokens=T, schedule=sched, sync_every=T, lane=6, n_slots=2, keep_scores=True)
    g = model.generate_launch(ids1.cuda(), am1.cuda(), pix1.cuda(), grid1, max_new_tokens=T, schedule=sched, sync_every=T, lane=6, group=g, n_slots=2,
                              keep_scores=True)
    outs = model.generate_collect(g, all_batches=True, output_scores=True)
    assert torch.equal(outs[1].sequences, alone.sequences)
    for t in range(T):
        assert torch.equal(outs[1].scores[t], alone.scores[t]), t
    assert torch.equal(outs[1].past_logit_mask, alone.past_logit_mask)


def test_generate_with_caller_logits_processors_and_stopping_criteria(setup):
    """padt.py:422-423,570-580,717,752 (round 6): caller-supplied `logits_processor` / `stopping_criteria` on the hooked decode loop.  (1) no-op hooks
    reproduce the captured-graph run bit for bit (sequences, scores, per-step hidden rows); (2) a processor's rows are what is selected from and what
    `.scores` returns — the banned token is gone, every token is the arg-max of its returned row; (3) criteria stop single rows (pad afterwards) and the
    loop (HF's MaxLengthCriteria: the sequence ends exactly there); (4) sampling draws from the processed rows; (5) a merged decode group refuses hooks."""
    cfg, w, model, U, oc = setup
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    from transformers.generation.logits_process import LogitsProcessor
    from transformers.generation.stopping_criteria import StoppingCriteria
    grids = [[1, 10, 12], [1, 8, 8]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=6, n_post=9, ragged=True, seed=91)
    L, T = ids.shape[1], 9
    sched = U.rec_schedule(T, vrt_at=range(3, 6))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    base = model.generate(output_scores=True, **kw)
    btok = base.sequences[:, L:].cpu()

    class Same(LogitsProcessor):
        calls = 0

        def __call__(self, input_ids, scores):
            Same.calls += 1
            assert input_ids.shape == (2, L + Same.calls - 1) and scores.shape == (2, cfg.vocab_size + 46) and scores.dtype == torch.float32
            return scores

    class Never(StoppingCriteria):
        def __call__(self, input_ids, scores, **kwargs):
            assert scores is not None and len(scores) == input_ids.shape[1] - L          # output_scores=True: the processed rows so far
            return torch.zeros(input_ids.shape[0], dtype=torch.bool, device=input_ids.device)
    o1 = model.generate(output_scores=True, logits_processor=LogitsProcessorList([Same()]), stopping_criteria=StoppingCriteriaList([Never()]), **kw)
    assert Same.calls == T and torch.equal(o1.sequences, base.sequences)
    assert all(torch.equal(a, b) for a, b in zip(o1.scores, base.scores))
    assert len(o1.hidden_states) == len(base.hidden_states) and torch.equal(o1.hidden_states.buf, base.hidden_states.buf)      # eager steps == graph replays

    # (2) ban what the un-hooked run picked at step 1 (a free text step) in row 0 from step 1 on
    banned = int(btok[0, 1])

    class Ban(LogitsProcessor):
        def __call__(self, input_ids, scores):
            if input_ids.shape[1] >= L + 1:
                scores = scores.clone()
                scores[0, banned] = float("-inf")
            return scores
    o2 = model.generate(output_scores=True, logits_processor=[Ban()], **kw)                 # a plain list of callables works too
    t2 = o2.sequences[:, L:].cpu()
    assert int(t2[0, 0]) == int(btok[0, 0]) and int(t2[0, 1]) != banned and torch.equal(t2[1, :2], btok[1, :2])
    live = torch.ones(2, dtype=torch.bool)
    for t in range(t2.shape[1]):
        am_t = o2.scores[t].argmax(dim=-1).cpu()
        for b in range(2):
            if live[b]:
                assert int(t2[b, t]) == int(am_t[b]), (b, t)
            else:
                assert int(t2[b, t]) == cfg.pad_token_id
            if int(t2[b, t]) == cfg.eos_token_id:
                live[b] = False
        if t >= 1:
            assert o2.scores[t][0, banned].item() == float("-inf")

    # (3) a criterion that stops row 1 once 3 tokens are out: pad from then on; row 0 runs to its EOS
    class StopRow1(StoppingCriteria):
        def __call__(self, input_ids, scores, **kwargs):
            return torch.tensor([False, input_ids.shape[1] >= L + 3], device=input_ids.device)
    o3 = model.generate(stopping_criteria=StoppingCriteriaList([StopRow1()]), **kw)
    t3 = o3.sequences[:, L:].cpu()
    assert torch.equal(t3[0], btok[0]) and torch.equal(t3[1, :3], btok[1, :3]) and (t3[1, 3:] == cfg.pad_token_id).all()
    o4 = model.generate(stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=L + 4)]), **kw)
    assert o4.sequences.shape[1] == L + 4 and torch.equal(o4.sequences, base.sequences[:, : L + 4])
    o5 = model.generate(stopping_criteria=[MaxLengthCriteria(max_length=L + 1)], **kw)       # everything stops after the FIRST token
    assert o5.sequences.shape[1] == L + 1 and torch.equal(o5.sequences, base.sequences[:, : L + 1])
    assert len(o5.hidden_states) == 1

    # (4) sampling from processed rows: only two text columns survive the processor
    keep = torch.tensor([17, 23])

    class Only(LogitsProcessor):
        def __call__(self, input_ids, scores):
            out = torch.full_like(scores, float("-inf"))
            out[:, keep] = scores[:, keep]
            return out
    o6 = model.generate(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=6,
                        do_sample=True, top_k=0, seed=3, logits_processor=LogitsProcessorList([Only()]))
    t6 = o6.sequences[:, L:].cpu()
    assert bool(torch.isin(t6, keep).all())
    # (5) hooks belong to generate()'s own loop
    with pytest.raises(NotImplementedError, match="merged decode group"):
        model.generate_launch(ids.cuda(), am.cuda(), pix.cuda(), grid, max_new_tokens=T, n_slots=2, hooks=dict(processors=[Same()], criteria=None, pass_scores=False))
    with pytest.raises(NotImplementedError, match="output_logits"):
        model.generate(output_logits=True, logits_processor=[Same()], **{k: v for k, v in kw.items() if k != "schedule"})
