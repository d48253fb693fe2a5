"""Host logic of the product path against the reference-generated golden vectors and the oracle: processor wrapper,
completion parser (incl. the dropped-sample quirk), 4.50 rope index, ViT index tables, weight layout, rank striding.
No GPU, no HIP calls."""
import os

import numpy as np
import pytest
import torch

import padt_amd
from padt_amd import pipeline
from padt_amd.llm import plan_prompt
from padt_amd.vision import vision_position_ids, window_index
from padt_amd.weights import interleave16, prepare_weights, synthetic_state_dict, weight_shapes

import padt_oracle as O
import parity_util as U


def _t(a):
    return torch.from_numpy(np.asarray(a))


class GoldTok:
    def __init__(self, words, eos):
        self.id2tok, self.eos_token = list(words), eos

    def get_vocab(self):
        return {t: i for i, t in enumerate(self.id2tok)}

    @property
    def vocab(self):
        return self.get_vocab()

    def add_tokens(self, toks):
        for t in toks:
            s = t.content if hasattr(t, "content") else str(t)
            if s not in self.id2tok:
                self.id2tok.append(s)


class GoldProc:
    def __init__(self, tok):
        self.tokenizer = tok

    def batch_decode(self, ids):
        return [self.tokenizer.id2tok[int(i)] for i in ids]

    def __call__(self, *a, **k):
        return {"image_grid_thw": k["image_grid_thw"]} if "image_grid_thw" in k else {}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/tiny_e2e.npz", allow_pickle=False)


def test_processor_wrapper_matches_reference(gold):
    base = [w for w in gold["tokenizer_words"].tolist() if not w.startswith("<|VRT_") and not w.startswith("<|empty_token_")]
    tok = GoldTok(base, "<|im_end|>")
    proc = padt_amd.VisonTextProcessingClass(GoldProc(tok), 2)
    proc.prepare(512)
    grid = _t(gold["grid"])
    proc(image_grid_thw=grid)
    assert len(tok.get_vocab()) == int(gold["vocab_after"])
    assert tok.id2tok == gold["tokenizer_words"].tolist()
    ids = _t(gold["input_ids_local"]).clone()
    g = proc.assign_to_global_vrt_id(ids, grid)
    assert g is ids and torch.equal(ids, _t(gold["input_ids_global"]))           # in place, like the reference
    assert torch.equal(proc.assign_to_local_vrt_id(ids, grid), _t(gold["input_ids_local"]))
    assert proc.pid2vrt([3, 7]) == "<|VRT_3|><|VRT_7|>" and proc.pid2vrt(5) == "<|VRT_5|>"
    assert proc.eos_token if hasattr(proc, "eos_token") else True                # attribute passthrough does not raise for known names
    with pytest.raises(AttributeError):
        proc.not_an_attribute


def test_parser_matches_reference_outputs(gold):
    tok = GoldTok(gold["tokenizer_words"].tolist(), "<|im_end|>")
    proc = padt_amd.VisonTextProcessingClass(GoldProc(tok), 2)
    proc.model_embed_token_size = 512
    comp = _t(gold["comp_local"])
    last = _t(gold["last_hidden"])                                               # (B,T,D)
    T = comp.shape[1]
    hidden = [(last[:, t:t + 1],) for t in range(T)]
    completions, feats, labels, vrts, vf = padt_amd.parseVRTintoCompletion(proc, comp, hidden, torch.Tensor([False, False]))
    assert completions == gold["completions"].tolist()
    assert [";".join(l) for l in labels] == gold["labels"].tolist()              # OVD: both runs labelled "person"
    assert ["|".join(v) for v in vrts] == gold["vrts"].tolist()
    assert torch.equal(feats[0][0], _t(gold["feat_0_0"]))
    assert torch.equal(feats[1][0], _t(gold["feat_1_0"])) and torch.equal(feats[1][1], _t(gold["feat_1_1"]))
    assert vf == [[], []]
    # edge cases: VRT run reaching the end without EOS → sample dropped; <answer> tags with the thinking mask on
    e_ids = _t(gold["edge_ids"])
    e_hidden = [(torch.full((2, 1, 64), float(i)),) for i in range(e_ids.shape[1])]
    e_comp, e_feats, e_labels, _, _ = padt_amd.parseVRTintoCompletion(proc, e_ids, e_hidden, torch.Tensor([False, True]))
    assert [len(f) for f in e_feats] == gold["edge_n_feats"].tolist()
    assert [";".join(l) for l in e_labels] == gold["edge_labels"].tolist()
    assert e_comp == gold["edge_completions"].tolist()
    if len(e_feats[1]):
        assert torch.equal(e_feats[1][0], _t(gold["edge_feat_1_0"]))


def test_vit_index_tables_match_reference(golden_dir):
    z = np.load(f"{golden_dir}/index_tables.npz")
    for name in ("46x46", "46x30", "10x12", "batch"):
        grid = z[f"{name}.grid"].tolist()
        wi, cu = window_index(grid, 2, 112, 14)
        assert torch.equal(wi, _t(z[f"{name}.window_index"])) and cu.tolist() == z[f"{name}.cu_window"].tolist()
        assert torch.equal(vision_position_ids(grid, 2), _t(z[f"{name}.pos_ids"]))


def test_prompt_plan_matches_oracle_rope_index(gold):
    cfg = padt_amd.PaDTConfig(vocab_size=512, image_token_id=500, vision_start_token_id=501, eos_token_id=1, pad_token_id=0,
                              vision_config=padt_amd.VisionConfig(patch_size=2, window_size=16))
    oc = O.OracleConfig(vocab_size=512, image_token_id=500, vision_start_token_id=501, patch_size=2, window_size=16)
    ids, am, grid = _t(gold["input_ids_global"]), _t(gold["attention_mask"]), _t(gold["grid"])
    plan = plan_prompt(cfg, ids, am, grid, "cpu")
    pos, deltas = O.rope_index(oc, ids, grid, am)
    assert torch.equal(plan.rope_deltas, deltas) and torch.equal(deltas, _t(gold["rope_deltas"]))
    packed = torch.cat([pos[:, b, am[b] == 1] for b in range(2)], dim=1)
    assert torch.equal(plan.pos3.long(), packed)
    assert plan.lens == am.sum(1).tolist() and plan.cu.tolist() == [0, plan.lens[0], sum(plan.lens)]
    assert plan.next_pos == [int(pos[:, b, am[b] == 1].max()) + 1 for b in range(2)]
    n_img = [int(g[1] * g[2] // 4) for g in grid]
    assert plan.vrt_off == [0, n_img[0], n_img[0] + n_img[1]]
    assert int((plan.img_index >= 0).sum()) == sum(n_img) and plan.img_index.max() == sum(n_img) - 1
    # 46x46 REC prompt → delta = 23 - 529
    c3 = padt_amd.padt_pro_3b()
    g, pix, i3, a3 = U.synthetic_batch(c3, [[1, 46, 46]], n_pre=15, n_post=33)
    p3 = plan_prompt(c3, i3, a3, g, "cpu")
    assert i3.shape[1] == 577 and int(p3.rope_deltas[0, 0]) == 23 - 529
    with pytest.raises(ValueError, match="Image features and image tokens do not match"):
        bad = i3.clone()
        bad[0, 0] = c3.image_token_id
        plan_prompt(c3, bad, a3, g, "cpu")


@pytest.mark.parametrize("operands", ["fp16", "bf16"])
def test_weight_layout(tmp_path, operands):
    cfg = padt_amd.small_test_config()
    sd = synthetic_state_dict(cfg, seed=1, bias_std=0.02)
    assert set(sd) == set(weight_shapes(cfg)) == set(O.weight_shapes(U.oracle_config(cfg)))
    W = prepare_weights(sd, cfg, device="cpu", operands=operands)
    op = torch.float16 if operands == "fp16" else torch.bfloat16
    assert W.op16 == op and W["vit.0.gu.w"].dtype == op and W["llm.embed"].dtype == op and W["proto.0.w"].dtype == op
    assert all(v.dtype in (torch.bfloat16, torch.float32) for k, v in W.items() if k.startswith("dec."))      # the PaDT decoder's (hi, lo) pairs are bf16
    I, Ip = cfg.vision_config.intermediate_size, W.vit_ipad
    assert Ip % 64 == 0 and W["vit.0.gu.w"].shape == (2 * Ip, 160) and W["vit.0.down.w"].shape == (160, Ip)
    g, u = sd["visual.blocks.0.mlp.gate_proj.weight"].to(op), sd["visual.blocks.0.mlp.up_proj.weight"].to(op)
    gu = W["vit.0.gu.w"].view(Ip // 16, 2, 16, 160)
    assert torch.equal(gu[:, 0].reshape(Ip, 160)[:I], g) and torch.equal(gu[:, 1].reshape(Ip, 160)[:I], u)
    assert (gu[:, 0].reshape(Ip, 160)[I:] == 0).all() and (W["vit.0.down.w"][:, I:] == 0).all()
    hd = cfg.head_dim
    assert W["llm.0.qkv.w"].shape == ((2 + 2) * hd, 256)
    assert torch.equal(W["llm.0.qkv.w"][2 * hd: 3 * hd], sd["model.layers.0.self_attn.k_proj.weight"].to(op))
    with pytest.raises(ValueError, match="operands must be"):
        prepare_weights(sd, cfg, device="cpu", operands="fp8")
    assert W["llm.head"] is W["llm.embed"]                                        # tied (3B); 7B config is untied
    a = torch.arange(32).view(32, 1)
    assert interleave16(a, a + 100).view(-1).tolist()[:34] == list(range(16)) + list(range(100, 116)) + [16, 17]
    # safetensors round trip through from_pretrained's loader, including 5.x-style key nesting
    from safetensors.torch import save_file
    from padt_amd.weights import load_checkpoint_state_dict
    ren = {("model.language_model." + k[6:] if k.startswith("model.") else ("model." + k if k.startswith("visual.") else k)): v.contiguous()
           for k, v in sd.items()}
    save_file(ren, str(tmp_path / "model.safetensors"))
    back = load_checkpoint_state_dict(str(tmp_path))
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)


def test_rank_striding_rule():
    # utils.py:181-182 — every rank walks the same number of batches; starts past the end = "skip the work"
    assert pipeline.rank_batches(100, 16, 0, 8) == [0] and pipeline.rank_batches(100, 16, 7, 8) == [112]
    assert pipeline.rank_batches(300, 16, 1, 8) == [16, 144, 272]
    covered = sorted(s for r in range(8) for s in pipeline.rank_batches(300, 16, r, 8))
    assert covered == list(range(0, 384, 16))


def test_config_from_hf_dict_both_layouts():
    flat = {"vocab_size": 151936, "hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16,
            "num_key_value_heads": 2, "intermediate_size": 11008, "rope_theta": 1e6, "tie_word_embeddings": True,
            "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]},
            "vision_config": {"hidden_size": 1280, "depth": 32, "num_heads": 16, "intermediate_size": 3420, "out_hidden_size": 2048},
            "vl_decoder": {"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "use_mask_loss": False}}
    c = padt_amd.PaDTConfig.from_hf_dict(flat)
    assert c.head_dim == 128 and c.mrope_section == (16, 24, 24) and c.vl_decoder["use_mask_loss"] is False
    nested = {"text_config": {k: v for k, v in flat.items() if k not in ("vision_config", "vl_decoder")},
              "vision_config": flat["vision_config"], "vl_decoder": flat["vl_decoder"], "tie_word_embeddings": True}
    assert padt_amd.PaDTConfig.from_hf_dict(nested).to_dict() == c.to_dict()
    assert padt_amd.padt_pro_7b().head_dim == 128 and not padt_amd.padt_pro_7b().tie_word_embeddings


def test_oracle_logits_warpers_match_installed_transformers():
    """oracle.warp_logits (Temperature → TopK → TopP, the sampling branch's processors, padt.py:570-580) against the installed
    transformers' own warper classes on random scores incl. ties and -inf rows: identical supports and values."""
    import padt_oracle as O
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(4, 500, generator=g) * 3
    scores[:, 17] = scores[:, 3]                      # a tie
    scores[1, 40:60] = float("-inf")                  # masked rows
    ids = torch.zeros(4, 1, dtype=torch.long)
    for T, k, p in [(1.0, 0, 1.0), (0.7, 40, 1.0), (1.3, 50, 0.9), (1.0, 3, 0.5), (2.0, 0, 0.8)]:
        ref = TemperatureLogitsWarper(T)(ids, scores.clone()) if T != 1.0 else scores.clone()
        if k:
            ref = TopKLogitsWarper(k)(ids, ref)
        if p < 1.0:
            ref = TopPLogitsWarper(p)(ids, ref)
        got = O.warp_logits(scores, T, k, p)
        assert torch.equal(torch.isfinite(got), torch.isfinite(ref)), (T, k, p)
        assert torch.allclose(got[torch.isfinite(got)], ref[torch.isfinite(ref)], rtol=0, atol=0), (T, k, p)


def test_rope_index_left_padded_rows_against_installed_transformers():
    """a5 (padt.py:256-277 → HF get_rope_index): the oracle's restatement of transformers 4.50 and the product's packed variant against
    the INSTALLED transformers' Qwen2_5_VLModel.get_rope_index on left-padded, ragged batches with different grids.  The 3-D position
    of every VALID token is identical in both HF versions and must match exactly; what differs between 4.50 (pinned by the reference)
    and 5.15 is only (a) the filler at padded positions (1 vs 0 — never read: those keys are masked) and (b) rope_deltas =
    max + 1 - PADDED length (4.50) vs - valid length (5.15), i.e. exactly the pad count apart — asserted as such."""
    import types
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
    cfg = padt_amd.padt_pro_3b()
    oc = U.oracle_config(cfg)
    grids = [[1, 46, 46], [1, 10, 12], [1, 46, 30], [1, 8, 8]]
    grid, pix, ids, am = U.synthetic_batch(cfg, grids, n_pre=7, n_post=11, ragged=True, seed=5)
    assert int((am == 0).sum()) > 0                                  # really left-padded
    fake = types.SimpleNamespace(config=types.SimpleNamespace(vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=2)))
    fake.get_vision_position_ids = types.MethodType(M.Qwen2_5_VLModel.get_vision_position_ids, fake)
    mm = (ids == cfg.image_token_id).int()
    hf_pos, hf_delta = M.Qwen2_5_VLModel.get_rope_index(fake, ids, mm, image_grid_thw=grid, attention_mask=am)
    pos, deltas = O.rope_index(oc, ids, grid, am)
    valid = am == 1
    assert torch.equal(pos[:, valid], hf_pos[:, valid])
    n_pad = (am == 0).sum(1, keepdim=True)
    assert torch.equal(deltas, hf_delta.to(deltas.dtype) - n_pad)     # 4.50: minus the padded length
    assert bool((pos[:, ~valid] == 1).all())
    plan = plan_prompt(cfg, ids, am, grid, "cpu")
    packed = torch.cat([hf_pos[:, b, am[b] == 1] for b in range(len(grids))], dim=1)
    assert torch.equal(plan.pos3.long(), packed) and torch.equal(plan.rope_deltas, deltas)
    # decode-step position (padt.py:268-277: cache_position + rope_deltas) = the sample's own max + 1 + t, padding-independent
    L = ids.shape[1]
    for b in range(len(grids)):
        assert L + int(deltas[b, 0]) == int(hf_pos[:, b, am[b] == 1].max()) + 1 == plan.next_pos[b]


def test_gumbel_uniform_is_strictly_inside_the_unit_interval():
    """csrc/vrt_head.hip gumbel_noise builds u = ((x >> 9) + 0.5) * 2^-23 in fp32 from a 32-bit hash x: both extremes must stay strictly
    inside (0, 1) so that -log(-log u) is finite (the 24-bit form ((x >> 8) + 0.5) * 2^-24 rounds to exactly 1.0 for x = 0xFFFFFFFF)."""
    import numpy as np
    f = np.float32
    for x in (0, 1, 0x7FFFFFFF, 0xFFFFFFFF):
        u = f(f(f(x >> 9) + f(0.5)) * f(1.0 / 8388608.0))
        assert f(0) < u < f(1), (x, u)
        assert np.isfinite(-np.log(-np.log(u)))
    bad = f(f(f(0xFFFFFFFF >> 8) + f(0.5)) * f(1.0 / 16777216.0))
    assert bad == f(1.0)                                            # what the old expression did


def test_processor_and_parser_on_a_real_bpe_tokenizer(golden_dir):
    """a10 / a11 against the reference's own outputs on a REAL tokenizer (tests/golden/make_golden_tok.py): byte-level BPE inside a
    transformers PreTrainedTokenizerFast, rebuilt from the JSON stored in the fixture — AddedToken semantics of the <|empty_token_i|> /
    <|VRT_k|> additions, per-token strings with leading blanks and punctuation merges through the parser's string tests (REC, OVD with
    multi-token labels, thinking mode, a run cut by max_new_tokens → sample dropped), and the image_prototype branch's
    processor(text=vrts_str) round trip (padt_processor.py:15-28,76,134)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("mk_tok", os.path.join(golden_dir, "make_golden_tok.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(os.path.join(golden_dir, "real_tokenizer.npz"))
    got = mk.run(padt_amd.VisonTextProcessingClass, padt_amd.parseVRTintoCompletion, str(z["tokenizer_json"]))
    assert set(got) == set(z.files) - {"tokenizer_json"}
    for k, v in got.items():
        ref = z[k]
        if isinstance(v, str):
            assert v == str(ref), (k, v, str(ref))
        elif isinstance(v, (int, np.integer)):
            assert int(v) == int(ref), k
        else:
            assert v.shape == ref.shape and (v == ref).all(), k
    # the scenarios bite: VRT ids start right above the embedding rows, labels span tokens, the truncated sample is dropped
    assert got["vocab_after_prepare"] == got["base_vocab"] + mk.EXTRA_ROWS and got["vocab_after_grid"] == got["vocab_after_prepare"] + 16
    assert got["vrt_encode_ids"].tolist()[-5:-1:1].count(got["vocab_after_prepare"] + 15) == 1
    assert json.loads(got["ovd.labels"]) == [["person", "traffic light", "bus"], []] and json.loads(got["truncated.labels"]) == [[], ["bus"]]
    assert json.loads(got["think.labels"]) == [["cat"], ["cat"]] and json.loads(got["rec.n_feats"]) == [[3], [2]]


# ------------------------------------------------------------------------------------------------ generate() argument policy
_REJECTED_WITH_VALUES = [("streamer", object()), ("min_length", 5),
                         ("min_new_tokens", 3), ("num_beams", 4), ("pixel_values_videos", torch.zeros(1)), ("video_grid_thw", torch.zeros(1, 3)),
                         ("inputs_embeds", torch.zeros(1, 2, 4)), ("prefix_allowed_tokens_fn", lambda *a: [0]), ("assistant_model", object()),
                         ("negative_prompt_ids", torch.zeros(1, 2)), ("output_attentions", True),
                         ("stop_strings", ["x"]), ("num_return_sequences", 2), ("generation_config", object())]


@pytest.mark.parametrize("name,value", _REJECTED_WITH_VALUES, ids=[n for n, _ in _REJECTED_WITH_VALUES])
def test_generate_rejects_arguments_it_does_not_implement(name, value):
    """padt.py:414-434,445,511-533,570-580: the reference honours these; the MI355X path must say so instead of ignoring them.  The check
    runs before anything touches the model, so an uninitialised instance is enough (no GPU, no library)."""
    from padt_amd.modeling import PaDTForConditionalGeneration
    m = PaDTForConditionalGeneration.__new__(PaDTForConditionalGeneration)
    with pytest.raises(NotImplementedError, match=name):
        m.generate(input_ids=torch.zeros((1, 4), dtype=torch.long), max_new_tokens=4, **{name: value})


def test_hook_lists_chain_processors_and_or_criteria():
    """padt.py:717,752: `logits_processor(input_ids, scores)` chains, `stopping_criteria(input_ids, scores)` ORs — for HF's list objects (callables
    themselves) and for plain lists of callables alike (llm._call_hooks, the host side of the hooked decode loop)."""
    from transformers import LogitsProcessorList, StoppingCriteriaList, MaxLengthCriteria
    from transformers.generation.logits_process import LogitsProcessor
    from padt_amd.llm import _call_hooks

    class Add(LogitsProcessor):
        def __init__(self, v):
            self.v = v

        def __call__(self, input_ids, scores):
            return scores + self.v * input_ids.shape[1]
    ids = torch.zeros((2, 3), dtype=torch.long)
    sc = torch.zeros((2, 5))
    want = sc + 3 * 3
    assert torch.equal(_call_hooks([Add(1), Add(2)], ids, sc, chain=True), want)
    assert torch.equal(_call_hooks(LogitsProcessorList([Add(1), Add(2)]), ids, sc, chain=True), want)
    crit = [lambda i, s: torch.tensor([True, False]), lambda i, s: torch.tensor([False, False])]
    assert _call_hooks(crit, ids, None, chain=False).tolist() == [True, False]
    assert _call_hooks(StoppingCriteriaList([MaxLengthCriteria(max_length=3)]), ids, None, chain=False).tolist() == [True, True]
    assert _call_hooks(StoppingCriteriaList([MaxLengthCriteria(max_length=4)]), ids, None, chain=False).tolist() == [False, False]


def test_generate_argument_policy_defaults_unknowns_and_max_length():
    from padt_amd.modeling import PaDTForConditionalGeneration, check_generate_kwargs
    # the callers' own call (eval/test_demo.py:96-103, utils.py:224-232) and the reference's default-valued arguments pass
    ok = dict(attn_implementation="flash_attention_2", tokenizer=None, streamer=None, num_beams=1,
              min_length=0, generation_config=None, pixel_values_videos=None,
              bos_token_id=151643, decoder_start_token_id=None, return_legacy_cache=True)       # benign GenerationConfig fields
    assert check_generate_kwargs(dict(ok), 16, None, 9) == 16
    # a tensor / list value of a policy-checked argument is compared by identity, never by `in` (no "Boolean value of Tensor" surprise)
    with pytest.raises(NotImplementedError, match="num_beams"):
        check_generate_kwargs({"num_beams": torch.tensor(2)}, 4, None, 9)
    with pytest.raises(NotImplementedError, match="bad_words_ids"):
        check_generate_kwargs({"bad_words_ids": [[1, 2]]}, 4, None, 9)
    assert check_generate_kwargs({}, None, None, 9) == 1024
    # max_length counts the (padded) prompt (padt.py:511-520); max_new_tokens wins when both are given
    assert check_generate_kwargs({}, None, 40, 9) == 31
    assert check_generate_kwargs({}, 7, 40, 9) == 7
    with pytest.raises(ValueError, match="max_length"):
        check_generate_kwargs({}, None, 9, 9)
    # an unknown keyword: HF's _validate_model_kwargs error (padt.py:440), not silence
    with pytest.raises(ValueError, match="not used by the model.*max_new_token"):
        check_generate_kwargs({"max_new_token": 3}, None, None, 9)
    m = PaDTForConditionalGeneration.__new__(PaDTForConditionalGeneration)
    with pytest.raises(NotImplementedError, match="synced_gpus"):
        m.generate(input_ids=torch.zeros((1, 4), dtype=torch.long), synced_gpus=True)
    # pad_token_id: the checkpoint's own value passes the policy, another one is refused (finished rows are padded on the device, padt.py:749);
    # output_logits with an active logits processor is refused (only the processed rows are kept), output_scores never is
    from types import SimpleNamespace
    m.generation_config = SimpleNamespace(pad_token_id=151643, repetition_penalty=1.1)
    with pytest.raises(NotImplementedError, match="pad_token_id"):
        m.generate(input_ids=torch.zeros((1, 4), dtype=torch.long), max_new_tokens=2, pad_token_id=0)
    with pytest.raises(NotImplementedError, match="output_logits"):
        m.generate(input_ids=torch.zeros((1, 4), dtype=torch.long), max_new_tokens=2, pad_token_id=151643, output_logits=True)


def test_import_sets_hardware_queue_default_but_never_overrides_the_user():
    """padt_amd/__init__.py: GPU_MAX_HW_QUEUES defaults to 8 (a fifth HIP stream must not share an HSA queue with the prefill stream:
    profiles/r05_to_rle_hw_queue_stall.log); an explicit setting wins."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, sys; sys.path.insert(0, %r); import padt_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "2"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip() == "2"
