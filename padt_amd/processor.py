"""Processor wrapper and completion parser — drop-in for ``padt_processor.py`` (same names, arguments, return values and
error behaviour, including the quirks listed in SURVEY.md Appendix C.5/6/10).  Pure host logic: token ids and strings.
"""
from typing import List, Tuple

import torch

try:  # the real tokenizer path uses HF's AddedToken; fake tokenizers in tests accept plain objects with .content
    from transformers.tokenization_utils import AddedToken
except Exception:  # pragma: no cover
    class AddedToken(str):
        def __new__(cls, content, **kw):
            o = str.__new__(cls, content)
            o.content = content
            return o


class VisonTextProcessingClass:
    """Drop-in for the reference's processor wrapper (padt_processor.py:4-57): everything not defined here is forwarded to the
    wrapped HF processor; on top of it the wrapper owns the VRT vocabulary — ``<|VRT_k|>`` must tokenise to
    ``embedding_rows + k`` — and the local ↔ global VRT id shift of a batch."""

    EMPTY_TOKEN = "<|empty_token_%d|>"
    VRT_TOKEN = "<|VRT_%d|>"

    def __init__(self, processing_class, spatial_merge_size=2):
        self.processing_class = processing_class
        self.spatial_merge_size = spatial_merge_size
        self.model_embed_token_size = self._vocab_size()

    def _vocab_size(self) -> int:
        return len(self.processing_class.tokenizer.get_vocab())

    def __getattr__(self, name: str):
        # reached only when normal lookup fails: forward to the wrapped processor (tokenizer, batch_decode, apply_chat_template …)
        inner = self.__dict__.get("processing_class")
        if inner is not None and hasattr(inner, name):
            return getattr(inner, name)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    # ---- vocabulary management
    def prepare(self, model_embed_token_size):
        """Fill the gap between the tokenizer's vocabulary and the embedding table with ``<|empty_token_i|>`` specials, so the
        VRT tokens added later start exactly at row ``model_embed_token_size`` (padt_processor.py:15-21)."""
        self.model_embed_token_size = model_embed_token_size
        missing = model_embed_token_size - self._vocab_size()
        assert missing > 0 or self.EMPTY_TOKEN % 0 in self.tokenizer.vocab
        if missing > 0:
            self.tokenizer.add_tokens([AddedToken(self.EMPTY_TOKEN % i, lstrip=False, rstrip=False, special=True, normalized=False)
                                       for i in range(missing)])
        return True

    def set_image_grid_thw(self, image_grid_thw):
        """Make sure ``<|VRT_0|> … <|VRT_{n-1}|>`` exist for n = the largest merged-patch count in the batch (:23-28)."""
        wanted = int(image_grid_thw.prod(dim=-1).max()) // self.spatial_merge_size ** 2
        present = self._vocab_size() - self.model_embed_token_size
        if present < wanted:
            self.processing_class.tokenizer.add_tokens(
                [AddedToken(self.VRT_TOKEN % i, lstrip=False, rstrip=False, special=False, normalized=False)
                 for i in range(present, wanted)])
        return True

    def __call__(self, *args, **kwargs):
        batch = self.processing_class(*args, **kwargs)
        if 'image_grid_thw' in batch:
            self.set_image_grid_thw(batch['image_grid_thw'])
        return batch

    # ---- VRT ids: local (per image, 0-based) ↔ global (offset by the merged patches of the samples before it)
    def _shift_vrt_ids(self, input_ids, image_grid_thw, sign: int):
        is_vrt = input_ids >= self.model_embed_token_size
        if bool(is_vrt.any()):
            merged = image_grid_thw.prod(dim=-1) // self.spatial_merge_size ** 2
            start = (merged.cumsum(0) - merged).to(input_ids.device)           # exclusive prefix sum: first patch of each sample
            input_ids[is_vrt] += sign * start[:, None].expand_as(input_ids)[is_vrt]
        return input_ids                                                       # modified in place, like the reference

    def assign_to_global_vrt_id(self, input_ids, image_grid_thw):
        return self._shift_vrt_ids(input_ids, image_grid_thw, +1)

    def assign_to_local_vrt_id(self, input_ids, image_grid_thw):
        return self._shift_vrt_ids(input_ids, image_grid_thw, -1)

    def pid2vrt(self, patch_ids):
        ids = [patch_ids] if type(patch_ids) == int else [int(i) for i in patch_ids]
        return ''.join(self.VRT_TOKEN % i for i in ids)


def _scan_completion(toks: List[str], eos_token: str, free_form: bool) -> List[Tuple[str, List[int], str]]:
    """String state machine of padt_processor.py:92-144 on one sample's per-token strings → [(label, VRT step indices, VRT text)].

    Two flags: inside an ``<answer>`` span (or ``free_form`` = no thinking section expected, so objects may appear anywhere) and
    inside a quoted object name.  A run of ``<|VRT_k|>`` tokens closes the object that the last quoted name labelled.  List
    indexing is deliberately unguarded: the reference looks one and two tokens ahead and lets the IndexError escape when a
    tag or a VRT run touches the end of the completion — the caller drops that sample (padt_processor.py:146-150)."""
    objects: List[Tuple[str, List[int], str]] = []
    answering = quoting = False
    name = ""
    i = 0
    while i < len(toks):
        tok = toks[i]
        if eos_token in tok:
            break
        opens = (not answering) and '<' in tok and '</' not in tok and 'answer' in toks[i + 1] and '>' in toks[i + 2]
        if opens:
            answering = True
            i += 3
            continue
        if answering or free_form:
            if '</' in tok and 'answer' in toks[i + 1] and '>' in toks[i + 2]:
                break
            if '"' in tok:
                if quoting:                                   # closing quote: text before it still belongs to the name
                    name = (name + tok.split('"')[0]).strip()
                else:                                         # opening quote: the name starts after it
                    name = tok.split('"')[1]
                quoting = not quoting
                i += 1
                continue
            if '<|VRT_' in tok:
                quoting = False
                steps, text = [], ""
                while '<|VRT_' in toks[i]:                    # IndexError when the run reaches the end of the completion
                    steps.append(i)
                    text += toks[i]
                    i += 1
                objects.append((name, steps, text))
                continue
            if quoting:
                name += tok
        i += 1
    return objects


def parseVRTintoCompletion(processor, completion_ids, hidden_states, need_thinking_mask=None, image_prototype=None,
                           image_grid_thw=None):
    """padt_processor.py:60-151.  ``hidden_states[step][-1][batch_idx]`` must be the (Lq, D) last-layer state that
    PREDICTED completion token ``step`` (our generate() returns a lazy sequence with exactly that indexing).
    → (completions, per-sample lists of object features, labels, VRT strings, prototype features)."""
    if image_grid_thw is not None:
        patch_starts = torch.nn.functional.pad((image_grid_thw.cumprod(-1)[:, -1] // 4).cumsum(-1), (1, 0), 'constant', 0)
    if need_thinking_mask is None:
        need_thinking_mask = torch.ones(len(completion_ids)).to(torch.bool)
    completions, all_feats, all_labels, all_vrts, all_proto = [], [], [], [], []
    for b, completion in enumerate(completion_ids):
        toks = processor.batch_decode(completion)
        completions.append(''.join(toks))
        feats, labels, vrts, protos = [], [], [], []
        try:
            for name, steps, text in _scan_completion(toks, processor.tokenizer.eos_token, not need_thinking_mask[b].item()):
                feats.append(torch.cat([hidden_states[s][-1][b] for s in steps], dim=0))
                labels.append(name)
                vrts.append(text)
                if image_prototype is not None and image_grid_thw is not None:
                    rows = processor(text=text, return_tensors='pt')['input_ids'].to(image_grid_thw.device)[0, ...] \
                        + patch_starts[b] - processor.model_embed_token_size
                    protos.append(image_prototype[rows])
        except Exception:  # the reference swallows everything per sample and returns empty lists for it (padt_processor.py:146-150)
            feats, labels, vrts, protos = [], [], [], []
        all_feats.append(feats)
        all_labels.append(labels)
        all_vrts.append(vrts)
        all_proto.append(protos)
    return completions, all_feats, all_labels, all_vrts, all_proto
