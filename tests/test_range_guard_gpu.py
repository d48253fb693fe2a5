"""The fp16-operand range guard (round 5): fp16 ends at 65504 and the un-normalised 16-bit tensors of ViT / LLM — SwiGLU hiddens, q / k / v,
attention outputs, merger hiddens — can exceed it on a real checkpoint ("massive activations" reach down_proj's input).  The product path
must never return a number computed from an overflowed operand:

  * every generate() checks its ViT output rows, prototypes and post-norm hidden rows (prompt pass + every decode step) for inf / NaN on the
    device (padt_check_finite) and csrc/common.h rope_fin turns an overflowing q / k into NaN (an infinite k could otherwise vanish:
    q·k = -inf is a key the soft-max silently drops);
  * operands="fp16": a flagged batch raises PaDTHipError;  operands="auto" (the default): it is re-run on the bf16 instantiation (fp32
    range) — bit for bit what an operands="bf16" model returns — with a RuntimeWarning;  a checkpoint VALUE outside fp16 is caught at load.

Each case drives ONE tensor class past 65504 with weights fp16 itself can hold (so only the activation overflows), checks on the fp32
oracle that the tensor really exceeds the range (the test has power), and asserts flag → fallback → oracle parity."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

FP16_MAX = 65504.0


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def _weights(site):
    """small_test_config weights + the edit that overflows tensor class `site` (all edited values are exact in bf16 and finite in fp16)."""
    import padt_amd
    import parity_util as U
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=21, std=0.05)

    def edit(name, fn):
        w[name] = w[name].clone()
        fn(w[name])
    if site.startswith("llm"):
        # layer 0's o_proj drives residual-stream channel 7 to ~1e5 (the round-4 massive-activation test): the normalised rows of layer 1
        # then carry ~sqrt(D) = 16 in channel 7 and ~0 elsewhere
        edit("model.layers.0.self_attn.o_proj.weight", lambda t: t[7, :].fill_(8192.0))
        if site == "llm_swiglu":          # gate = up = 16 * 32 = 512 → h = silu(512) * 512 = 2.6e5 (tokens whose channel is positive)
            edit("model.layers.1.mlp.gate_proj.weight", lambda t: t[3, 7].fill_(32.0))
            edit("model.layers.1.mlp.up_proj.weight", lambda t: t[3, 7].fill_(32.0))
        elif site == "llm_v":             # v[5] = 16 * 8192 = 1.3e5 → the value rows, the attention output and o_proj's input overflow
            edit("model.layers.1.self_attn.v_proj.weight", lambda t: t[5, 7].fill_(8192.0))
        elif site == "llm_k":             # k[5] = +-1.3e5: +-inf in fp16 — q·k = -inf would be a silently dropped key without rope_fin
            edit("model.layers.1.self_attn.k_proj.weight", lambda t: t[5, 7].fill_(8192.0))
    elif site == "vit_swiglu":
        edit("visual.blocks.0.attn.proj.weight", lambda t: t[7, :].fill_(8192.0))
        edit("visual.blocks.0.mlp.gate_proj.weight", lambda t: t[3, 7].fill_(64.0))
        edit("visual.blocks.0.mlp.up_proj.weight", lambda t: t[3, 7].fill_(64.0))
    else:
        raise ValueError(site)
    return cfg, w, U


def _oracle_peak(U, w, oc, site, ids, am, pix, grid):
    """Largest magnitude the fp32 oracle itself sees in the tensor class the case targets: the INPUT of the edited block's down_proj (the
    SwiGLU hidden) or the OUTPUT of the edited v / k projection — read off the oracle's own `linear` calls during one prompt pass."""
    O = U.O
    probe = {"llm_swiglu": ("model.layers.1.mlp.down_proj.weight", "in"), "vit_swiglu": ("visual.blocks.0.mlp.down_proj.weight", "in"),
             "llm_v": ("model.layers.1.self_attn.v_proj.weight", "out"), "llm_k": ("model.layers.1.self_attn.k_proj.weight", "out")}[site]
    target, side = w[probe[0]], probe[1]
    rec = [0.0]
    orig = O.linear

    def spy(x, wt, *a, **k):
        y = orig(x, wt, *a, **k)
        if wt is target:
            rec[0] = max(rec[0], float((x if side == "in" else y).abs().max()))
        return y
    O.linear = spy
    try:
        O.prefill(w, oc, ids, am, pix, grid)
    finally:
        O.linear = orig
    return rec[0]


@pytest.mark.parametrize("site", ["llm_swiglu", "llm_v", "llm_k", "vit_swiglu"])
def test_fp16_overflow_is_flagged_and_auto_reruns_on_bf16(site):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd._lib import PaDTHipError
    from padt_amd.modeling import PaDTForConditionalGeneration
    cfg, w, U = _weights(site)
    O, oc = U.O, U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=8, ragged=True, seed=41)
    peak = _oracle_peak(U, w, oc, site, ids, am, pix, grid)
    assert peak > FP16_MAX, f"the case has no power: the oracle's {site} tensor peaks at {peak:.3e}"
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 5))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    # operands="fp16": the overflow is REPORTED, never returned
    strict = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="fp16")
    with pytest.raises(PaDTHipError, match="65504"):
        strict.generate(**kw)
    del strict
    # operands="auto": flag → the same batch on the bf16 instantiation, with a warning
    auto = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="auto")
    assert auto.dtype == torch.float16 and auto._fallback is None
    with pytest.warns(RuntimeWarning, match="re-run on the bf16"):
        out = auto.generate(**kw)
    assert auto.overflow_reruns == 1 and auto._fallback is not None
    ref = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="bf16").generate(**kw)
    assert torch.equal(out.sequences, ref.sequences)
    assert torch.equal(out.hidden_states.last_layer_rows(), ref.hidden_states.last_layer_rows())
    assert torch.equal(out.past_image_embeds, ref.past_image_embeds) and torch.equal(out.past_high_res_image_embeds, ref.past_high_res_image_embeds)
    # ... and that answer is the oracle's, at the bf16 operand type's distance
    L = ids.shape[1]
    toks = out.sequences.cpu()[:, L:]
    hid = out.hidden_states.last_layer_rows().cpu().float()
    assert torch.isfinite(hid).all() and torch.isfinite(out.past_high_res_image_embeds).all()
    ores = O.generate(w, oc, ids, am, pix, grid, T, schedule=sched, force_tokens=toks)
    worst = max(rel_err(hid[t], ores["hidden"][t][:, -1].float())[1] for t in range(T))
    print(f"\n[{site}: oracle tensor peaks at {peak:.3e} > 65504] fp16 flagged, bf16 re-run: hidden rel rms worst {worst:.3e}")
    assert worst < 3e-2                                            # bf16 operands, 2 layers (measured ≈ 5e-3 class)
    # the PaDT decoder of the fp16 model takes the twin's bf16 rows as they are
    feats = [[out.hidden_states.last_layer_rows()[2:5, b]] for b in range(2)]
    dec = auto.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
    assert torch.isfinite(dec["pred_boxes"]).all() and dec["pred_boxes"].shape == (2, 4)


def test_range_guard_is_per_batch_inside_a_merged_decode_group():
    """Two batches share one decode session (in-flight batching).  Only batch A's prompt holds the token whose embedding row drives the SwiGLU
    hidden past 65504: A is re-run on the bf16 twin, B keeps its fp16 result bit for bit (rows are independent in every decode kernel)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import parity_util as U
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=22, std=0.05)
    TOK = 77
    for k in ("model.embed_tokens.weight", "model.layers.0.mlp.gate_proj.weight", "model.layers.0.mlp.up_proj.weight"):
        w[k] = w[k].clone()
    w["model.embed_tokens.weight"][TOK, 7] = 59904.0               # finite in fp16; the token's stream row carries it in channel 7
    w["model.layers.0.mlp.gate_proj.weight"][3, 7] = 32.0          # normalised row ≈ 16 in channel 7 → gate = up ≈ 512 → h ≈ 2.6e5
    w["model.layers.0.mlp.up_proj.weight"][3, 7] = 32.0
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 5))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s in range(2):
        grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=8, seed=300 + s, ragged=True)
        ids[ids == TOK] = TOK + 1
        if s == 0:
            ids[0, -3] = TOK                                       # a text position of sample 0's prompt
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    auto = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="auto")
    runner = pipeline.PipelinedRunner(auto, proc, depth=2, merge=2)
    got = []
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for b in batches:
            got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
        got += runner.flush()
    assert len(got) == 2 and auto.overflow_reruns == 1, auto.overflow_reruns
    assert any("re-run on the bf16" in str(r.message) for r in rec)
    bf = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="bf16")
    ref_a = pipeline.rec_batch(bf, proc, batches[0][0].clone(), *batches[0][1:], max_new_tokens=T, schedule=sched)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # batch B alone on the fp16 path: no flag, no warning
        ref_b = pipeline.rec_batch(auto, proc, batches[1][0].clone(), *batches[1][1:], max_new_tokens=T, schedule=sched)
    assert auto.overflow_reruns == 1
    for (d0, c0, l0, v0), (d1, c1, l1, v1) in zip((ref_a, ref_b), got):
        assert c0 == c1 and v0 == v1
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"])
        assert torch.isfinite(d1["pred_boxes"]).all()


def test_range_guard_with_a_vit_stream_flags_a_vit_overflow_in_the_second_batch_of_a_group():
    """with PipelinedRunner(vit_stream=True) the ViT / prototype checks of batch k > 0 run on the ViT stream, which does not wait for
    the prefill stream — the batch's flag must be zeroed on the stream that ORs into it first.  Batch B (k = 1 of a 2-batch decode group) carries
    a pixel column that the edited patch embedding turns into a massive stream channel, which block 0's SwiGLU hidden cannot hold in fp16;
    batch A does not: B is re-run on the bf16 twin, A keeps its fp16 result bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import parity_util as U
    from padt_amd import pipeline
    from padt_amd.modeling import PaDTForConditionalGeneration
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=23, std=0.05)
    COL = 11
    for k in ("visual.patch_embed.proj.weight", "visual.blocks.0.mlp.gate_proj.weight", "visual.blocks.0.mlp.up_proj.weight"):
        w[k] = w[k].clone()
    # batch B's pixel column drives channel 7 of the (fp32) ViT stream to 1.3e5 from the first row on; block 0's norm2 then hands ~sqrt(D) in
    # that channel to gate / up rows 3: silu(g) * u ≈ (64 * 11)^2 = 5e5 > 65504 → the fp16 SwiGLU hidden overflows (batch A: ≈ 64^2, finite)
    w["visual.patch_embed.proj.weight"].view(cfg.vision_config.hidden_size, -1)[7, COL] = 8192.0
    w["visual.blocks.0.mlp.gate_proj.weight"][3, 7] = 64.0
    w["visual.blocks.0.mlp.up_proj.weight"][3, 7] = 64.0
    T = 8
    sched = U.rec_schedule(T, vrt_at=range(2, 5))
    proc = padt_amd.VisonTextProcessingClass(U.FakeProcessor(cfg, 40), 2)
    proc.model_embed_token_size = cfg.vocab_size
    batches = []
    for s in range(2):
        grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=8, seed=310 + s, ragged=True)
        pix = pix.clone()
        pix[:, COL] = 16.0 if s == 1 else 0.0                     # 16 * 8192 = 1.3e5 in channel 7 of every patch row of batch B
        batches.append((ids.cuda(), am.cuda(), pix.cuda(), grid))
    auto = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="auto")
    for rep in range(2):                                           # the race needs the prefill stream to be busy: groups back to back (2 re-runs: a third would switch "auto" to prefers_bf16)
        runner = pipeline.PipelinedRunner(auto, proc, depth=2, merge=2, vit_stream=True)
        before = auto.overflow_reruns
        got = []
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            for b in batches:
                got += runner.submit(b[0].clone(), b[1], b[2], b[3], max_new_tokens=T, schedule=sched)
            got += runner.flush()
        assert len(got) == 2 and auto.overflow_reruns == before + 1, (rep, auto.overflow_reruns)
        assert any("re-run on the bf16" in str(r.message) for r in rec)
    bf = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="bf16")
    ref_b = pipeline.rec_batch(bf, proc, batches[1][0].clone(), *batches[1][1:], max_new_tokens=T, schedule=sched)
    n0 = auto.overflow_reruns
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ref_a = pipeline.rec_batch(auto, proc, batches[0][0].clone(), *batches[0][1:], max_new_tokens=T, schedule=sched)
    assert auto.overflow_reruns == n0
    for (d0, c0, l0, v0), (d1, c1, l1, v1) in zip((ref_a, ref_b), got):
        assert c0 == c1 and v0 == v1
        assert torch.equal(d0["pred_boxes"], d1["pred_boxes"]) and torch.equal(d0["pred_mask"], d1["pred_mask"])
        assert torch.isfinite(d1["pred_boxes"]).all()


def test_auto_stops_paying_twice_when_most_batches_overflow():
    """A checkpoint whose activations exceed fp16 on EVERY batch: after three re-runs (and at least half of the batches seen) an operands="auto"
    model starts new decode groups on its bf16 twin directly — one pass per batch again, results = the bf16 model's, no further warnings."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd.modeling import PaDTForConditionalGeneration
    cfg, w, U = _weights("llm_swiglu")
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 8, 8], [1, 10, 12]], n_pre=5, n_post=8, ragged=True, seed=41)
    T = 6
    sched = U.rec_schedule(T, vrt_at=range(2, 4))
    kw = dict(input_ids=ids.cuda(), attention_mask=am.cuda(), pixel_values=pix.cuda(), image_grid_thw=grid, max_new_tokens=T, schedule=sched)
    auto = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="auto")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        outs = [auto.generate(**kw) for _ in range(3)]
    assert auto.overflow_reruns == 3 and auto.prefers_bf16
    assert sum("now starts every new decode group on the bf16" in str(r.message) for r in rec) == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # fourth batch: straight to the twin, nothing to warn about
        out4 = auto.generate(**kw)
    assert auto.overflow_reruns == 3
    ref = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="bf16").generate(**kw)
    for o in outs + [out4]:
        assert torch.equal(o.sequences, ref.sequences) and torch.equal(o.hidden_states.last_layer_rows(), ref.hidden_states.last_layer_rows())


def test_checkpoint_value_outside_fp16_is_caught_at_load():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import padt_amd
    import parity_util as U
    from padt_amd.modeling import PaDTForConditionalGeneration
    from padt_amd.weights import Fp16RangeError
    cfg = padt_amd.small_test_config()
    w = U.bf16_weights(cfg, seed=23, std=0.05)
    w["model.layers.1.self_attn.o_proj.weight"] = w["model.layers.1.self_attn.o_proj.weight"].clone()
    w["model.layers.1.self_attn.o_proj.weight"][3, 5] = 1.0e5      # a bf16 value fp16 cannot hold
    with pytest.raises(Fp16RangeError, match="o.w"):
        PaDTForConditionalGeneration(cfg, w, device="cuda", operands="fp16")
    with pytest.warns(RuntimeWarning, match="falls back to bf16"):
        m = PaDTForConditionalGeneration(cfg, w, device="cuda", operands="auto")
    assert m.dtype == torch.bfloat16


def test_check_finite_kernel_flags_exactly_the_bad_rows():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x = torch.randn(37, 256, device="cuda", generator=g).to(dt)
        x[:, 5] = torch.finfo(dt).max                              # the largest finite value is not flagged
        flags = torch.zeros(37, dtype=torch.int32, device="cuda")
        ops.check_finite(x, flags, rows_per_flag=1)
        assert int(flags.sum()) == 0
        x[3, 17], x[11, 255], x[36, 0] = float("inf"), float("nan"), float("-inf")
        ops.check_finite(x, flags, rows_per_flag=1)
        assert flags.nonzero().flatten().tolist() == [3, 11, 36]
        one = torch.zeros(4, dtype=torch.int32, device="cuda")
        ops.check_finite(x, one, rows_per_flag=10)                 # rows 0-9, 10-19, 20-29, 30-36
        assert one.tolist() == [1, 1, 0, 1]
        whole = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.check_finite(x[20:30], whole)
        assert int(whole) == 0
        ops.check_finite(x[:, :128], whole)                        # strided view, bad element in column 17
        assert int(whole) == 1
