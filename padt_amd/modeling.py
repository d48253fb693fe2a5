"""``PaDTForConditionalGeneration`` — drop-in surface of ``src/PaDT/models/padt.py`` on the MI355X HIP path.

Kept callable exactly as the reference's callers use them (eval/test_demo.py:20-113, eval/evaluation_scripts/utils.py:
194-246): ``from_pretrained``, ``generate(**processor_outputs, use_cache=True, max_new_tokens=…, do_sample=False,
output_hidden_states=True, return_dict_in_generate=True[, synced_gpus=False])`` → object with ``.sequences /
.hidden_states / .past_image_embeds / .past_logit_mask / .past_high_res_image_embeds / .past_visual_pe`` (attribute and
``['key']`` access, padt.py:786-798), ``vl_decode(feats, low, high, image_grid_thw, visual_pe)`` → dict (padt.py:397-412),
``.config.vision_config.spatial_merge_size`` and ``.model.embed_tokens.weight.shape[0]``.

There is no HuggingFace modeling code and no PyTorch compute on this path: the methods sequence kernels from
libpadt_hip.so (see vision.py / llm.py / decoder.py) and fail loudly if that library is missing.
"""
import json
import os
from types import SimpleNamespace
from typing import Optional, Sequence

import torch

from . import _lib, ops
from .config import PaDTConfig
from .decoder import PaDTDecoder
from .llm import MODE, LanguageModel, plan_prompt
from .vision import VisionEncoder
from .weights import Fp16RangeError, load_checkpoint_state_dict, prepare_weights, synthetic_state_dict


class StepHiddenStates:
    """Lazy stand-in for HF's ``hidden_states`` (T-tuple of per-layer tuples): ``hs[step][-1]`` is the last-layer,
    post-final-norm state that predicted completion token ``step`` — (B, L, D) for step 0, (B, 1, D) afterwards
    (padt.py:732-737; consumed at padt_processor.py:125).  Only the last layer is retained."""

    class _Layers:
        def __init__(self, last):
            self._last = last

        def __getitem__(self, i):
            if i in (-1,):
                return self._last()
            raise IndexError("only the last layer ([-1]) is retained on the MI355X path")

        def __len__(self):
            return 1

    def __init__(self, hidden_buf, n_steps, prefill_packed, lens, L_pad):
        self.buf, self.n, self.prefill, self.lens, self.L = hidden_buf, n_steps, prefill_packed, lens, L_pad
        self._prefill_padded = None

    def __len__(self):
        return self.n

    def _step0(self):
        if self._prefill_padded is None:
            B, D = len(self.lens), self.prefill.shape[1]
            out = torch.zeros((B, self.L, D), device=self.prefill.device, dtype=self.prefill.dtype)
            o = 0
            for b, l in enumerate(self.lens):
                out[b, self.L - l:] = self.prefill[o:o + l]          # left padding, as the reference's callers produce
                o += l
            self._prefill_padded = out
        return self._prefill_padded

    def __getitem__(self, step):
        if isinstance(step, slice):
            return [self[i] for i in range(*step.indices(self.n))]
        if step < 0:
            step += self.n
        if not 0 <= step < self.n:
            raise IndexError(step)
        if step == 0:
            return self._Layers(self._step0)
        return self._Layers(lambda s=step: self.buf[s].unsqueeze(1))

    def last_layer_rows(self):
        """(T, B, D) tensor of the per-step last-position states (step 0 = last prompt position)."""
        return self.buf[: self.n]


class CustomGenerateDecoderOnlyOutput(dict):
    """Attribute + item access like HF's ModelOutput (padt.py:40-45)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# ------------------------------------------------------------------------------------------------ generate() argument policy
# HF / reference generate() arguments that cannot change what this path returns: accepted when they hold one of the listed values
# (None = any value), otherwise NotImplementedError.
_GENERATE_IGNORED = {
    "attn_implementation": None, "tokenizer": None, "assistant_tokenizer": None, "cache_implementation": None, "logits_to_keep": None,
    "generation_config": (None,), "output_attentions": (False, None),
    # GenerationConfig fields callers pass routinely and the reference's generate() accepts without effect on this path:
    "bos_token_id": None, "decoder_start_token_id": None, "return_legacy_cache": None,
    "num_beams": (1, None), "num_beam_groups": (1, None), "num_return_sequences": (1, None), "penalty_alpha": (None,), "min_new_tokens": (0, None),
    "min_length": (0, None), "no_repeat_ngram_size": (0, None), "bad_words_ids": (None,), "force_words_ids": (None,), "token_healing": (False, None),
    "length_penalty": (1.0, None), "early_stopping": (False, None), "typical_p": (1.0, None), "min_p": (None,), "epsilon_cutoff": (0.0, None),
    "eta_cutoff": (0.0, None), "encoder_repetition_penalty": (1.0, None), "guidance_scale": (None, 1.0), "renormalize_logits": (False, None),
}
# arguments the reference honours (padt.py:418-424,445,511-533,570-580,719-737) and this path does not implement
_GENERATE_REJECTED = ("inputs", "prefix_allowed_tokens_fn", "assistant_model", "streamer",
                      "negative_prompt_ids", "negative_prompt_attention_mask", "pixel_values_videos", "video_grid_thw", "second_per_grid_ts",
                      "inputs_embeds", "past_key_values", "position_ids", "cache_position", "rope_deltas", "stop_strings", "max_time",
                      "sequence_bias", "suppress_tokens", "begin_suppress_tokens", "forced_bos_token_id", "forced_eos_token_id",
                      "exponential_decay_length_penalty", "prompt_lookup_num_tokens", "dola_layers", "labels")


def check_generate_kwargs(kwargs: dict, max_new_tokens, max_length, prompt_len) -> int:
    """The drop-in surface's argument policy (see generate()): → the effective max_new_tokens.  Raises NotImplementedError naming an
    argument the reference honours and this path does not, ValueError (HF's _validate_model_kwargs wording) for an unknown one."""
    for k in list(kwargs):
        v = kwargs[k]
        if k in _GENERATE_REJECTED:
            if v is None:
                continue                                              # the reference's own default
            raise NotImplementedError(f"generate({k}=...) is not implemented on the MI355X path (the reference honours it, padt.py:414-580): "
                                      "remove the argument or run the reference for this call")
        if k in _GENERATE_IGNORED:
            ok = _GENERATE_IGNORED[k]
            if ok is None or any(v is o or (not isinstance(v, (torch.Tensor, list, tuple, dict)) and v == o) for o in ok):
                continue
            raise NotImplementedError(f"generate({k}={v!r}) is not implemented on the MI355X path (supported: {ok})")
        raise ValueError(f"The following `model_kwargs` are not used by the model: ['{k}'] (note: typos in the generate arguments will also "
                         "show up in this list)")
    if max_new_tokens is None:
        if max_length is not None:
            if prompt_len is None or int(max_length) <= int(prompt_len):
                raise ValueError(f"Input length of input_ids is {prompt_len}, but `max_length` is set to {max_length}. This can lead to unexpected "
                                 "behavior. You should consider increasing `max_length` or, better yet, setting `max_new_tokens`.")
            return int(max_length) - int(prompt_len)
        return 1024
    return int(max_new_tokens)


class PaDTForConditionalGeneration:
    def __init__(self, config: PaDTConfig, state_dict, device="cuda", dtype=torch.bfloat16, llm_weights: str = "bf16", operands=None,
                 state_dict_factory=None, precision: str = "default"):
        """dtype: the checkpoint's (the reference's torch_dtype argument; PaDT checkpoints are bf16).  operands: the 16-bit MFMA operand type
        ViT / LLM compute in, fp32 accumulation either way (weights.prepare_weights):
          "auto" (default; env PADT_OPERANDS) — fp16 operands (8x closer to the fp32 reference than bf16 at the same MFMA rate) UNDER A RANGE
                 GUARD: every generate() checks its ViT output rows, prototypes and the post-norm hidden rows of the prompt pass and of every
                 decode step for inf / NaN on the device (fp16 ends at 65504; an overflow of any un-normalised fp16 tensor — SwiGLU hidden,
                 q / k / v, attention output, merger hidden — arrives there, csrc/common.h rope_fin); a flagged batch is RE-RUN on the bf16
                 instantiation of the same kernels (fp32 range; built lazily from the same checkpoint, +13 GB at 3B) and a warning is logged;
          "fp16" — the same guard, but a flagged batch raises PaDTHipError (no second weight set is ever built);
          "bf16" — bf16 operands (the reference's own dtype): nothing to guard but NaN weights; a flagged batch raises.
        state_dict_factory: callable returning the checkpoint's state dict again (from_pretrained / from_synthetic pass one so that the
        fallback does not have to keep a second copy of the checkpoint alive); default: the dict given here is retained.
        precision: "default" — 16-bit MFMA operands as above (boxes within the north star's 1e-3 of the fp32 reference, mask logits at the
        3.7e-3 floor of the operand type); "reference" — ViT / LLM on the split-precision machinery of the PaDT decoder (fp32 streams,
        (hi, lo) bf16 GEMM operands at twice the MFMA work, fp32 attention and an fp32 KV cache: padt_amd/reference.py): every float output within
        1e-3 (measured 3e-6 on boxes, 5e-5 on mask logits), at a fifth of the default throughput (bench.py `reference_precision`); needs 16-bit LLM weights."""
        if dtype != torch.bfloat16:
            raise ValueError("PaDT checkpoints are bf16: pass torch_dtype=torch.bfloat16 (the MFMA operand type is chosen with operands=)")
        _lib.load()                                            # fail loudly before touching any weight
        self.config = config
        self.device = torch.device(device)
        operands = operands or os.environ.get("PADT_OPERANDS", "auto")
        if operands not in ("auto", "fp16", "bf16"):
            raise ValueError("operands must be 'auto', 'fp16' or 'bf16'")
        self.operands = operands
        self._llm_weights = llm_weights
        self._sd_factory = state_dict_factory if state_dict_factory is not None else (lambda sd=state_dict: sd) if operands == "auto" else None
        self._fallback = None                                  # the bf16-operand twin, built on the first flagged batch (operands="auto")
        self.overflow_reruns = 0                               # batches answered by the twin so far
        self._batches_seen = 0
        self.prefers_bf16 = False                              # "auto": set once most batches overflow — new decode groups then start on the twin directly
        try:
            self.W = prepare_weights(state_dict, config, self.device, llm_weights=llm_weights, operands="bf16" if operands == "bf16" else "fp16")
        except Fp16RangeError as e:                            # a weight fp16 cannot hold: "auto" multiplies bf16 operands from the start
            if operands != "auto":
                raise
            import warnings
            warnings.warn("padt_amd: %s — operands='auto' falls back to bf16 MFMA operands for this checkpoint" % e, RuntimeWarning, stacklevel=2)
            self.W = prepare_weights(state_dict, config, self.device, llm_weights=llm_weights, operands="bf16")
        self.dtype = self.W.op16                               # what pixel_values / hidden_states / past_image_embeds are held in
        self.visual = VisionEncoder(config, self.W, self.device)
        self.lm = LanguageModel(config, self.W, self.device)
        if precision not in ("default", "reference"):
            raise ValueError("precision must be 'default' or 'reference'")
        self.precision = precision
        self.ref = None
        if precision == "reference":
            if llm_weights != "bf16" or self.dtype != torch.float16:
                raise ValueError("precision='reference' needs 16-bit LLM weights and fp16 attention operands (operands='auto' or 'fp16')")
            from .reference import ReferencePath
            self.ref = ReferencePath(config, state_dict, self)
        self.vl_decoder = PaDTDecoder(config, self.W, self.device, torch.bfloat16)
        self.model = SimpleNamespace(embed_tokens=SimpleNamespace(weight=self.W["llm.embed"]))
        self.use_visual_prototype_projection = config.use_visual_prototype_projection
        self.rope_deltas = None
        # defaults HF's generate() takes from the checkpoint's generation_config.json (padt.py:436 _prepare_generation_config,
        # :570-580 logits processors / stopping criteria); from_pretrained fills it, explicit generate() kwargs override it
        self.generation_config = SimpleNamespace(repetition_penalty=1.0, eos_token_id=[config.eos_token_id],
                                                 pad_token_id=config.pad_token_id, do_sample=False, temperature=1.0,
                                                 top_k=50, top_p=1.0)          # HF GenerationConfig defaults

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.bfloat16, attn_implementation=None,
                        device_map=None, config=None, llm_weights: str = "bf16", operands=None, precision: str = "default", **_):
        """Loads ``config.json`` + ``*.safetensors`` (checkpoint key layout of PaDT-MLLM/PaDT_*).  ``attn_implementation``
        is accepted and ignored: attention is always the HIP flash kernel."""
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: checkpoints must be a local directory (no hub access on this path)")
        if config is None:
            config = PaDTConfig.from_hf_dict(json.load(open(os.path.join(path, "config.json"))))
        elif isinstance(config, dict):
            config = PaDTConfig.from_hf_dict(config)
        device = "cuda"
        if isinstance(device_map, dict) and "" in device_map:
            d = device_map[""]
            device = f"cuda:{d}" if isinstance(d, int) else str(d)
        elif isinstance(device_map, (str, torch.device)):
            device = str(device_map)
        model = cls(config, load_checkpoint_state_dict(path), device=device, dtype=torch_dtype, llm_weights=llm_weights, operands=operands,
                    state_dict_factory=lambda: load_checkpoint_state_dict(path), precision=precision)
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            model.load_generation_config(json.load(open(gpath)))
        return model

    def load_generation_config(self, d: dict):
        """generation_config.json → defaults of generate(), as HF's GenerationMixin applies them (padt.py:436): repetition_penalty,
        eos_token_id (int or list), pad_token_id, do_sample / temperature / top_k / top_p."""
        g = self.generation_config
        if d.get("repetition_penalty") is not None:
            g.repetition_penalty = float(d["repetition_penalty"])
        if d.get("eos_token_id") is not None:
            e = d["eos_token_id"]
            g.eos_token_id = [int(e)] if isinstance(e, int) else [int(x) for x in e]
        if d.get("pad_token_id") is not None:
            g.pad_token_id = int(d["pad_token_id"])
        for k, cast in (("do_sample", bool), ("temperature", float), ("top_k", int), ("top_p", float)):
            if d.get(k) is not None:
                setattr(g, k, cast(d[k]))
        return g

    @classmethod
    def from_synthetic(cls, config: PaDTConfig, seed=0, device="cuda", state_dict=None, **kw):
        """Random-init weights of the given architecture (no checkpoints offline)."""
        kw_llm, kw_op, kw_prec = kw.pop("llm_weights", "bf16"), kw.pop("operands", None), kw.pop("precision", "default")
        make = (lambda: state_dict) if state_dict is not None else (lambda: synthetic_state_dict(config, seed=seed, device=device, dtype=torch.bfloat16, **kw))
        return cls(config, make(), device=device, llm_weights=kw_llm, operands=kw_op, state_dict_factory=make, precision=kw_prec)

    def fallback_model(self):
        """The bf16-operand twin of an operands="auto" model (same checkpoint, same kernels in their bf16 instantiation: fp32 range), built on
        first use; its generation defaults follow this model's."""
        if self._fallback is None:
            if self.operands != "auto" or self._sd_factory is None:
                raise _lib.PaDTHipError("no bf16 fallback: the model was built with operands=%r" % self.operands)
            if self.precision != "default":
                raise _lib.PaDTHipError("precision='reference' has no bf16 twin: a non-finite batch is an error there")
            fb = PaDTForConditionalGeneration(self.config, self._sd_factory(), device=self.device, llm_weights=self._llm_weights, operands="bf16")
            fb.generation_config = self.generation_config
            self._fallback = fb
        return self._fallback

    def eval(self):
        return self

    # ------------------------------------------------------------------ generate (padt.py:414-616 → 618-800)
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None, use_cache=True,
                 max_new_tokens=None, do_sample=None, output_hidden_states=True, return_dict_in_generate=True,
                 synced_gpus=False, schedule: Optional[Sequence[Optional[str]]] = None, sync_every: int = 16,
                 use_graph: bool = True, lane: int = 0, repetition_penalty: Optional[float] = None, eos_token_id=None,
                 temperature: Optional[float] = None, top_k: Optional[int] = None, top_p: Optional[float] = None,
                 seed: Optional[int] = None, max_length: Optional[int] = None, output_scores: bool = False, output_logits: bool = False,
                 pad_token_id: Optional[int] = None, logits_processor=None, stopping_criteria=None, **kwargs):
        """Greedy generation over the unified text‖VRT vocabulary.

        ``logits_processor`` / ``stopping_criteria`` (padt.py:422-423,570-580,717,752; round 6): HF ``LogitsProcessorList`` / ``StoppingCriteriaList``
        objects or plain lists of callables ``f(input_ids, scores)``.  They run where the reference runs them — the processors on the step's fp32
        score rows AFTER the logit mask and the built-in processors (repetition penalty), before the arg-max / the sampler; the criteria on the
        sequences including the step's token, OR-ed into the stop state next to EOS and the length limit — on the HOOKED loop: decode steps launched
        kernel by kernel (no captured graph: the callables are host code), one host sync per step, ``.scores`` = the processed rows.  Not available
        inside ``pipeline.PipelinedRunner``'s merged decode groups.

        Arguments of the reference's ``generate`` (padt.py:414-434 + the HF generation kwargs it forwards) that this path does not
        implement are REJECTED by name (``NotImplementedError``: ``streamer``, ``prefix_allowed_tokens_fn``, ``min_length``,
        ``num_beams`` > 1, video inputs, ``inputs_embeds`` …), arguments that cannot change the
        result here are accepted and ignored (``use_cache``, ``attn_implementation``, ``synced_gpus=False`` …), anything else raises the
        ``ValueError`` HF's ``_validate_model_kwargs`` raises (padt.py:440) — a caller never gets silently different behaviour.
        ``max_length`` (padt.py:511-520): total length incl. the (padded) prompt; ``max_new_tokens`` wins when both are given, as in HF.

        ``schedule`` (synthetic weights only): per-step logits-processor code — 't' text rows only, 'v' the sample's own
        VRT rows only, 'e' force EOS, None free — applied where HF's ``logits_processor`` sits (padt.py:717).
        ``lane`` selects an independent decode session (KV caches, token ring, graph) so several batches can be in
        flight on different HIP streams (pipeline.PipelinedRunner).
        ``repetition_penalty`` / ``eos_token_id`` (int or list) / ``do_sample``: default to the checkpoint's
        generation_config.json (self.generation_config), exactly the entries HF's generate turns into a logits processor /
        stopping criterion / sampling switch (padt.py:436,570-580,740-743); explicit arguments override.
        ``output_scores=True`` (padt.py:719-720): ``.scores`` = T-tuple of (B, table rows) fp32 tensors — the step's logits after the logit mask
        (padt.py:292-301) and the logits processors (repetition penalty, the synthetic ``schedule``), i.e. what the arg-max / the sampler saw.
        ``output_logits=True`` (padt.py:721-724, the rows BEFORE the processors): served when no processor is active (repetition_penalty == 1,
        no schedule — then they ARE the scores), rejected otherwise.  ``pad_token_id``: must be the config's (the greedy kernel pads finished
        rows with it, padt.py:749).
        ``do_sample=True``: multinomial sampling after HF's Temperature → TopK → TopP warpers (``temperature`` / ``top_k`` /
        ``top_p``, defaults from generation_config, HF's own defaults 1.0 / 50 / 1.0) on a device counter-based generator keyed
        by ``seed`` (default: drawn from torch's global generator, so torch.manual_seed makes runs repeatable).  The draws are
        not torch.multinomial's; the distribution is.
        """
        if synced_gpus:
            raise NotImplementedError("generate(synced_gpus=True) is the ZeRO-3 / FSDP lock-step loop (padt.py:445,670): every rank holds a full "
                                      "replica on this path — pass synced_gpus=False")
        max_new_tokens = check_generate_kwargs(kwargs, max_new_tokens, max_length, None if input_ids is None else input_ids.shape[1])
        if pad_token_id is not None and int(pad_token_id) != int(self.generation_config.pad_token_id):
            raise NotImplementedError(f"generate(pad_token_id={pad_token_id}): finished rows are padded with the checkpoint's pad token "
                                      f"({self.generation_config.pad_token_id}) on this path")
        processors = logits_processor if logits_processor is not None and len(logits_processor) > 0 else None
        criteria = stopping_criteria if stopping_criteria is not None and len(stopping_criteria) > 0 else None
        hooks = None
        if processors is not None or criteria is not None:
            hooks = dict(processors=processors, criteria=criteria, pass_scores=bool(output_scores))
            use_graph, sync_every = False, 1                          # host callables between the kernels of every step
        if output_logits:
            pen = self.generation_config.repetition_penalty if repetition_penalty is None else repetition_penalty
            if float(pen) != 1.0 or schedule is not None or processors is not None:
                raise NotImplementedError("generate(output_logits=True) with a logits processor active (repetition_penalty != 1, a schedule or a "
                                          "logits_processor): only the processed rows are kept on this path — ask for output_scores=True")
        ctx = self.generate_launch(input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens, do_sample,
                                   schedule, sync_every, use_graph, lane, repetition_penalty=repetition_penalty,
                                   eos_token_id=eos_token_id, temperature=temperature, top_k=top_k, top_p=top_p, seed=seed,
                                   keep_scores=bool(output_scores or output_logits or hooks is not None), hooks=hooks)
        return self.generate_collect(ctx, output_hidden_states, return_dict_in_generate, output_scores=bool(output_scores),
                                     output_logits=bool(output_logits))

    @torch.no_grad()
    def generate_launch(self, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens=1024, do_sample=None,
                        schedule=None, sync_every=16, use_graph=True, lane=0, decode_stream=None, group=None, n_slots=1,
                        repetition_penalty=None, eos_token_id=None, temperature=None, top_k=None, top_p=None, seed=None,
                        vit_stream=None, inputs_ready=None, keep_scores=False, hooks=None):
        """Asynchronous half of generate(): host integer prep + every kernel up to the first host sync point, enqueued on
        the current stream (the decode steps on ``decode_stream`` if given, ordered after the prefill by an event).
        Returns a group context for generate_collect().

        Merged decode (pipeline.PipelinedRunner(merge=n)): ``n_slots`` batches share ONE decode session of n_slots*B
        rows — each call adds a batch (its ViT + prefill run now, its KV rows / prototypes land in the shared session),
        pass the returned context back as ``group`` for the next batch; the decode steps of all of them run together
        once the group is full (or on launch_decode()).  Every sample's math is unchanged (rows are independent in every
        decode kernel); the weights are streamed once per step for all rows.  Returns None instead of adding when the
        batch does not fit the group's session (caller closes the group and starts a new one).
        """
        # operands="auto" whose checkpoint keeps overflowing fp16: new decode groups start on the bf16 twin (a group stays with its owner)
        owner = group["owner"] if group is not None else (self.fallback_model() if self.prefers_bf16 else self)
        if owner is not self:
            return owner.generate_launch(input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens, do_sample, schedule, sync_every,
                                         use_graph, lane, decode_stream, group, n_slots, repetition_penalty, eos_token_id, temperature, top_k, top_p,
                                         seed, vit_stream, inputs_ready, keep_scores, hooks)
        self._batches_seen += 1
        gc = self.generation_config
        do_sample = gc.do_sample if do_sample is None else do_sample
        repetition_penalty = gc.repetition_penalty if repetition_penalty is None else repetition_penalty
        temperature = gc.temperature if temperature is None else float(temperature)
        top_k = gc.top_k if top_k is None else int(top_k)
        top_p = gc.top_p if top_p is None else float(top_p)
        if do_sample and top_k == 1:
            do_sample = False                                     # sampling among the single best token IS the arg-max
        if do_sample:
            if temperature <= 0:
                raise ValueError("temperature must be strictly positive (HF TemperatureLogitsWarper)")
            if top_p < 1.0 and not (0 < top_k <= 1024):
                raise NotImplementedError("top_p < 1 is supported together with 0 < top_k <= 1024 (the nucleus is taken over the top-k survivors)")
            if seed is None:
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        if pixel_values is None or image_grid_thw is None:
            raise ValueError("pixel_values and image_grid_thw are required (text-only input crashes in the reference too, "
                             "padt.py:292 with image_prototypes unbound)")
        cfg, dev = self.config, self.device
        eos_list = list(gc.eos_token_id) if eos_token_id is None else ([int(eos_token_id)] if isinstance(eos_token_id, int) else [int(e) for e in eos_token_id])
        if cfg.eos_token_id not in eos_list or len(eos_list) > 4:
            raise NotImplementedError("eos_token_id must contain config.eos_token_id and hold at most 4 ids")
        samp = (float(temperature), int(top_k), float(top_p), int(seed)) if do_sample else None
        gen_key = (float(repetition_penalty), tuple(eos_list), samp, bool(keep_scores))
        grid = image_grid_thw.detach().cpu().long()
        B = input_ids.shape[0]
        T_max = int(max_new_tokens)
        k = 0 if group is None else len(group["subs"])
        row0 = k * B
        proto_row0 = 0 if group is None else group["proto_rows"]
        plan = plan_prompt(cfg, input_ids, attention_mask, grid, dev, row0=row0, proto_row0=proto_row0)
        n_proto = plan.vrt_off[-1]
        need_s = max(plan.lens) + T_max
        if hooks is not None and (group is not None or n_slots != 1):
            raise NotImplementedError("logits_processor / stopping_criteria run on generate()'s own (un-merged) decode loop, not inside a merged decode group")
        if group is None:
            sess = self.lm.session(B * n_slots, need_s, n_proto * n_slots, T_max, lane=lane)
            group = dict(sess=sess, subs=[], proto_rows=0, B=B, n_slots=n_slots, T_max=T_max, sync_every=sync_every,
                         use_graph=use_graph, decode_stream=decode_stream, done=0, launched=False, schedule=schedule,
                         gen_key=gen_key, eos_list=eos_list, lane=lane, owner=self)
            sess.gen_cfg.copy_(ops.gen_cfg_tensor(gen_key[0], gen_key[1], "cpu", do_sample=samp is not None, seed=samp[3] if samp else 0,
                                                  temperature=samp[0] if samp else 1.0, top_k=samp[1] if samp else 0,
                                                  top_p=samp[2] if samp else 1.0).to(dev, non_blocking=True))
            sess.do_sample = samp is not None
            sess.keep_scores = bool(keep_scores)
            sess.hooks = None
            if hooks is not None:
                ids_dev = input_ids.detach().to(dev).long().contiguous()
                hk = dict(hooks, B=B, table_rows=cfg.vocab_size + n_proto, t=0)
                # `input_ids` as the reference's loop holds them (padt.py:751): the prompt and the tokens selected so far
                hk["sequences"] = lambda hk=hk, s=sess: ops.assemble_sequences(ids_dev, s.tokens[:B], hk["t"], cfg.vocab_size, 0)
                sess.hooks = hk
                group["hooks"] = hooks
            if gen_key[0] != 1.0:
                sess.seen.zero_()
            # neutral state for every row; the batches overwrite their own rows (unused rows stay finished / empty)
            st = torch.zeros(T_max + 1, dtype=torch.int32)
            if schedule is not None:
                for i, mm in enumerate(schedule[: T_max]):
                    st[i] = MODE[mm]
            sess.mode_table[: T_max + 1].copy_(st.to(dev, non_blocking=True))
            sess.step.zero_()
            sess.err.zero_()
            sess.unfinished.zero_()
            sess.slot.zero_()
            sess.lens.fill_(1)
            sess.pos3.zero_()
            sess.cur_tok.fill_(cfg.pad_token_id)
            sess.vrt_off.zero_()
            if self.ref is not None:                              # reference precision: eager split-precision steps + fp32 hidden rows
                sess.step_fn = self.ref.step
                if sess.hid32 is None or sess.hid32.shape[0] < T_max:
                    sess.hid32 = torch.zeros((sess.t_max, sess.B, cfg.hidden_size), device=dev, dtype=torch.float32)
        else:
            sess = group["sess"]
            if (group["launched"] or k >= group["n_slots"] or B != group["B"] or T_max != group["T_max"]
                    or schedule != group["schedule"] or gen_key != group["gen_key"] or sess.s_max < need_s
                    or sess.np_max < proto_row0 + n_proto):
                return None
        rows = slice(row0, row0 + B)
        sess.nf[rows].zero_()
        # this batch's range guard (ViT rows, prototypes, prompt-pass hidden rows).  Its zero fill must be ordered before EVERY check that ORs
        # into it: with a ViT stream the flag is zeroed ON that stream (the fill on the current stream would sit behind the
        # previous batch's prefill while the ViT checks of this batch already run — ADVICE r05); the prefill's check comes after
        # cur.wait_stream(vit_stream) below
        # The flag is slot k of the session's per-batch vector (read back with every other flag by ONE ops.collect_summary launch).
        nf = sess.nf_batch[k: k + 1]
        if vit_stream is None or self.ref is not None:
            nf.zero_()

        # ---- ViT → prototypes → session table
        if self.ref is not None:
            low, high, pe = self.ref.visual(pixel_values.to(dev), grid, nf=nf)
            proto = self.ref.prototypes(low, sess, proto_row0, nf=nf)     # fp32 rows (→ past_image_embeds); sess.proto gets their fp16 image
        elif vit_stream is None:
            low, high, pe = self.visual(pixel_values.to(dev), grid, nf=nf)
            proto = self.lm.prototypes(low, out=sess.proto[proto_row0: proto_row0 + n_proto], nf=nf)
        else:
            # the ViT of this batch on its own stream: it needs the inputs (event `inputs_ready` of the caller's stream) and nothing of the
            # current stream — so it runs while the PREVIOUS batch's prefill is still on the current stream — except for the first batch of a
            # group, whose session may just have been (re)allocated or reset on the current stream
            cur = torch.cuda.current_stream()
            if inputs_ready is not None:
                vit_stream.wait_event(inputs_ready)
            if k == 0 or inputs_ready is None:
                vit_stream.wait_stream(cur)
            with torch.cuda.stream(vit_stream):
                nf.zero_()
                low, high, pe = self.visual(pixel_values.to(dev), grid, nf=nf)
                proto = self.lm.prototypes(low, out=sess.proto[proto_row0: proto_row0 + n_proto], nf=nf)
            cur.wait_stream(vit_stream)                           # the prefill below reads low / the prototype rows
            for t_ in (low, high, pe[0], pe[1]):                   # allocated on vit_stream, read on the prefill / decode streams
                t_.record_stream(cur)
                if decode_stream is not None:
                    t_.record_stream(decode_stream)
        # ---- per-generate device state of this batch's rows
        off = torch.tensor([proto_row0 + o for o in plan.vrt_off], dtype=torch.int32)
        sess.vrt_off[row0: row0 + B + 1].copy_(off.to(dev, non_blocking=True))
        if row0 + B + 1 < sess.vrt_off.numel():
            sess.vrt_off[row0 + B + 1:].fill_(proto_row0 + n_proto)     # rows not (yet) in use: empty VRT range
        sess.unfinished[rows].fill_(1)
        lens_t = torch.tensor(plan.lens, dtype=torch.int32)
        sess.slot[rows].copy_(lens_t.to(dev, non_blocking=True))        # next append index
        sess.lens[rows].copy_((lens_t + 1).to(dev, non_blocking=True))  # keys visible to the next token
        sess.pos3[:, rows].copy_(torch.tensor([plan.next_pos] * 3, dtype=torch.int32).to(dev, non_blocking=True))
        self.rope_deltas = plan.rope_deltas
        if gen_key[0] != 1.0:
            # RepetitionPenaltyLogitsProcessor sees the caller's full (B, L) input_ids, padding included
            ids_full = input_ids.detach().to(dev).long()
            if proto_row0:
                ids_full = torch.where(ids_full >= cfg.vocab_size, ids_full + proto_row0, ids_full)
            rws = (torch.arange(B, device=dev, dtype=torch.int32) + row0)[:, None].expand(B, ids_full.shape[1]).contiguous()
            ops.seen_init(ids_full.reshape(-1).contiguous(), rws.reshape(-1), sess.seen)

        # ---- prefill; the first token is selected together with the other batches of the group (launch_decode)
        if self.ref is not None:
            hn_all = self.ref.prefill(plan, low, sess, nf=nf)             # fp32 rows
            ops.gather_rows(hn_all, plan.last_idx, out=sess.hid32[0, rows])
            ops.cast_f32_x16(sess.hid32[0, rows], out=sess.hn_first[rows])
        else:
            hn_all = self.lm.prefill(plan, low, sess, nf=nf)
            ops.gather_rows(hn_all, plan.last_idx, out=sess.hn_first[rows])
        group["subs"].append(dict(plan=plan, low=low, high=high, pe=pe, proto=proto, hn_all=hn_all, input_ids=input_ids,
                                  n_proto=n_proto, row0=row0, proto_row0=proto_row0, nf=nf,
                                  inputs=(attention_mask, pixel_values, image_grid_thw)))       # what a re-run on the bf16 twin needs
        group["proto_rows"] = proto_row0 + n_proto
        if len(group["subs"]) == group["n_slots"]:
            self.launch_decode(group)
        return group

    @torch.no_grad()
    def launch_decode(self, group):
        """First-token selection + the first chunk of decode steps (hipGraph replays, no host sync) for every batch of the
        group; ordered after their prefills (current stream) by an event when a decode stream is used."""
        if group["owner"] is not self:
            return group["owner"].launch_decode(group)
        if group["launched"]:
            return
        group["launched"] = True
        sess, T_max = group["sess"], group["T_max"]
        n = min(group["sync_every"], T_max - 1)
        if sess.hooks is not None:
            n = 0                                                    # hooked loop: the stop state is read after EVERY token, the first included (padt.py:752-757)

        def go():
            sess.head_and_select(sess.hn_first, advance=False)
            sess.run_steps(n, use_graph=group["use_graph"])
        ds = group["decode_stream"]
        if ds is not None:
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(ds):
                ds.wait_event(ev)
                go()
        else:
            go()
        group["done"] = 1 + n

    @torch.no_grad()
    def generate_collect(self, group, output_hidden_states=True, return_dict_in_generate=True, all_batches=False, output_scores=False,
                         output_logits=False):
        """Synchronising half of generate(): remaining decode chunks, trimming to the reference's stop rule, output object — of the group's only
        batch, or a list over its batches.  Everything the host needs from the device per chunk — the table-range assert (padt.py:203), `any
        row unfinished`, the range guard's per-row / per-batch flags, each row's first EOS step — comes from ONE kernel + ONE pinned D2H copy
        (ops.collect_summary; round 6: no ATen reductions, no `.item()` round trips)."""
        if group["owner"] is not self:
            return group["owner"].generate_collect(group, output_hidden_states, return_dict_in_generate, all_batches, output_scores, output_logits)
        cfg, dev = self.config, self.device
        self.launch_decode(group)
        sess, T_max = group["sess"], group["T_max"]
        n_rows, n_sub = sess.B, len(group["subs"])

        def summary(done):
            ops.collect_summary(sess.err, sess.unfinished, sess.nf, sess.nf_batch, n_sub, sess.tokens, done, cfg.eos_token_id, sess.gen_cfg, sess.summary)
            sess.summary_host.copy_(sess.summary, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return sess.summary_host.tolist()
        done_steps = group["done"]
        flags = summary(done_steps)
        while done_steps < T_max and flags[1]:
            n = min(group["sync_every"], T_max - done_steps)
            sess.run_steps(n, use_graph=group["use_graph"])
            done_steps += n
            flags = summary(done_steps)
        group["done"] = done_steps
        if flags[0] != 0:
            raise AssertionError("input_ids.max() >= extended table rows (padt.py:203)")
        nf_rows, nf_batch = flags[2: 2 + n_rows], flags[2 + n_rows: 2 + n_rows + n_sub]
        first_eos = flags[2 + n_rows + n_sub: 2 + 2 * n_rows + n_sub]
        outs = []
        for k_sub, sub in enumerate(group["subs"]):
            plan, row0, B = sub["plan"], sub["row0"], group["B"]
            if nf_batch[k_sub] != 0 or any(nf_rows[row0: row0 + B]):
                outs.append(self._non_finite_batch(group, sub, output_hidden_states, return_dict_in_generate))
                continue
            # the reference stops right after the step in which the last sequence finished (padt.py:756-757); any id of the EOS list finishes a row
            fe = first_eos[row0: row0 + B]
            n_steps = max(fe) + 1 if all(f >= 0 for f in fe) else done_steps
            # [input_ids | tokens] with the group's session-global VRT ids shifted back to this batch's own (one launch, padt.py:751)
            sequences = ops.assemble_sequences(sub["input_ids"].to(dev), sess.tokens[row0: row0 + B], n_steps, cfg.vocab_size, sub["proto_row0"])
            # clone: the session's hidden_buf is reused by the lane's next generate(); the caller owns what it gets back
            hbuf = sess.hid32 if self.ref is not None else sess.hidden_buf      # reference precision keeps the per-step rows in fp32
            hidden = StepHiddenStates(hbuf[:n_steps, row0: row0 + B].clone(), n_steps, sub["hn_all"],
                                      plan.lens, plan.L_pad)
            table_rows = cfg.vocab_size + sub["n_proto"]
            scores = None
            if output_scores or output_logits:
                # per step (B, table rows): text columns + this batch's own prototype columns of the session-wide rows (padt.py:719-724)
                V, p0 = cfg.vocab_size, sub["proto_row0"]
                rows_t = sess.scores[:n_steps, row0: row0 + B]
                if p0 == 0:
                    scores = tuple(rows_t[t, :, :table_rows].clone() for t in range(n_steps))
                else:
                    scores = tuple(torch.cat([rows_t[t, :, :V], rows_t[t, :, V + p0: V + p0 + sub["n_proto"]]], dim=1) for t in range(n_steps))
            out = CustomGenerateDecoderOnlyOutput(
                sequences=sequences, scores=scores if output_scores else None, logits=scores if output_logits else None, attentions=None,
                hidden_states=hidden if output_hidden_states else None, past_key_values=sess,
                past_image_embeds=sub["proto"].clone(),
                past_logit_mask=ops.logit_mask(sess.vrt_off[row0: row0 + B + 1], cfg.vocab_size, table_rows, sub["proto_row0"], B),
                past_high_res_image_embeds=sub["high"], past_visual_pe=sub["pe"])
            outs.append(out if return_dict_in_generate else sequences)
        return outs if all_batches else outs[0]

    def _non_finite_batch(self, group, sub, output_hidden_states, return_dict_in_generate):
        """A batch whose range guard fired: inf / NaN reached its ViT output, prototypes or hidden rows.  With fp16 operands that is an
        overflow of an un-normalised 16-bit tensor (65504); operands="auto" answers it with the SAME batch on the bf16 instantiation."""
        what = ("a non-finite value reached the ViT output / prototypes / post-norm hidden rows of a batch (%s MFMA operands)"
                % ("fp16" if self.dtype == torch.float16 else "bf16"))
        if self.operands != "auto" or self.dtype != torch.float16:
            raise _lib.PaDTHipError(what + (": an un-normalised activation exceeded fp16's 65504 — build the model with operands='auto' (re-runs such "
                                            "batches on bf16 operands) or operands='bf16'" if self.dtype == torch.float16 else
                                            ": NaN / inf weights or inputs?"))
        import warnings
        fb = self.fallback_model()
        self.overflow_reruns += 1
        warnings.warn("padt_amd: " + what + " — fp16 range exceeded; the batch is re-run on the bf16-operand instantiation "
                      "(%d so far; operands='bf16' avoids the second pass)" % self.overflow_reruns, RuntimeWarning, stacklevel=3)
        if not self.prefers_bf16 and self.overflow_reruns >= 3 and 2 * self.overflow_reruns >= self._batches_seen:
            # this checkpoint's activations do not fit fp16 as a rule, not as an exception: stop paying for two passes
            self.prefers_bf16 = True
            warnings.warn("padt_amd: %d of %d batches exceeded fp16's range — operands='auto' now starts every new decode group on the bf16 "
                          "instantiation" % (self.overflow_reruns, self._batches_seen), RuntimeWarning, stacklevel=3)
        am, pix, grid = sub["inputs"]
        pen, eos, samp = group["gen_key"][:3]
        kw = dict(do_sample=False)
        if samp is not None:
            kw = dict(do_sample=True, temperature=samp[0], top_k=samp[1], top_p=samp[2], seed=samp[3])
        if group.get("hooks") is not None:
            kw.update(logits_processor=group["hooks"]["processors"], stopping_criteria=group["hooks"]["criteria"])
        return fb.generate(input_ids=sub["input_ids"], attention_mask=am, pixel_values=pix, image_grid_thw=grid, max_new_tokens=group["T_max"],
                           schedule=group["schedule"], sync_every=group["sync_every"], use_graph=group["use_graph"], lane=("fb", group["lane"]),
                           repetition_penalty=pen, eos_token_id=list(eos), output_hidden_states=output_hidden_states,
                           return_dict_in_generate=return_dict_in_generate,
                           output_scores=bool(group["gen_key"][3]) and (group.get("hooks") is None or group["hooks"]["pass_scores"]), **kw)

    # ------------------------------------------------------------------ vl_decode (padt.py:342-412)
    @torch.no_grad()
    def vl_decode(self, object_vp_feats, low_res_image_embeds, high_res_image_embeds, image_grid_thws, visual_pes):
        cfg, dev = self.config, self.device
        flat = sum(object_vp_feats, [])
        if len(flat) == 0:                                        # padt.py:406-412 (the dummy pass of 383-393 is skipped)
            f32 = torch.float32                                   # same dtypes as the non-empty result (boxes / scores / logits are fp32)
            return {"pred_boxes": torch.zeros((0, 4), device=dev, dtype=f32),
                    "pred_score": torch.zeros((0, 1), device=dev, dtype=f32),
                    "pred_mask": torch.zeros((0, 8, 8), device=dev, dtype=f32),
                    "pred_mask_valid_hw": (), "sample_idx": []}
        grids = [[int(x) for x in g] for g in torch.as_tensor(image_grid_thws).tolist()]
        patch_num = [g[0] * g[1] * g[2] for g in grids]
        patch_off = [0]
        for n in patch_num:
            patch_off.append(patch_off[-1] + n)
        obj_sample, n_vp = [], []
        for si, feats in enumerate(object_vp_feats):
            for f in feats:
                obj_sample.append(si)
                n_vp.append(int(f.shape[0]))
        # the PaDT decoder reads fp32 or bf16 rows (its own operand pairs are bf16 (hi, lo)): fp16 rows of an fp16-operand model enter it
        # as fp32 (exact), never through a bf16 rounding
        def dec_in(t):
            t = t.to(dev).contiguous()
            if t.dtype == torch.float16:
                t = ops.cast_x16_f32(t)
            if not self.W.dec_hp and t.dtype == torch.float32:                      # the bf16-storage decoder (PADT_DECODER_HP=0) reads bf16 rows
                t = ops.cast_f32_bf16(t)
            return t
        feats_cat = torch.cat([f.to(dev) for f in flat], dim=0)
        feats_cat = dec_in(feats_cat if feats_cat.dtype in (torch.float16, torch.float32) else feats_cat.to(torch.bfloat16))
        low_res_image_embeds = dec_in(low_res_image_embeds)
        high_res_image_embeds = dec_in(high_res_image_embeds)
        bbox, score, masks, hw = self.vl_decoder.forward_objects(
            feats_cat, n_vp, low_res_image_embeds, high_res_image_embeds, visual_pes, obj_sample, patch_off, patch_num, grids)
        # "sample_idx_t": the same list as a device tensor (cached with the decoder's plan) for the device-side result pack
        # "sample_idx_t" / "pred_mask_src_hw": device copies for the device-side consumers (result pack, mask post-processing) — shared with the
        # decoder's cached plan: read-only
        return {"pred_boxes": bbox, "pred_score": score, "pred_mask": masks, "pred_mask_valid_hw": hw,
                "sample_idx": obj_sample, "sample_idx_t": self.vl_decoder.last_sample_t, "pred_mask_src_hw": self.vl_decoder.last_src_hw}

    def forward(self, *args, is_main=True, **kwargs):
        if is_main:
            raise NotImplementedError("teacher-forced forward_main is a training surface (out of scope); use generate()")
        return self.vl_decode(*args, **kwargs)

    __call__ = forward
