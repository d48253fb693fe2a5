#!/usr/bin/env python
"""Golden fixture for the processor wrapper and the completion parser on a REAL tokenizer.

Every other a10 / a11 fixture drives a hand-made word-level FakeTokenizer.  This one builds, in memory and without any file, what the
reference's callers actually wrap: a byte-level BPE (`tokenizers`: ByteLevel pre-tokenizer with the GPT-2 regex, ByteLevel decoder, merges
trained on a few sentences of the PaDT answer templates) inside a `transformers.PreTrainedTokenizerFast` with `<|im_end|>` / `<|endoftext|>`
specials, and runs the reference's OWN `VisonTextProcessingClass` (prepare / set_image_grid_thw / __call__ / assign_to_*_vrt_id) and
`parseVRTintoCompletion` (padt_processor.py:15-28,31-57,60-151) on it, unmodified:
  * `AddedToken` semantics: `<|empty_token_i|>` specials filling the gap to the embedding table, `<|VRT_k|>` non-special added tokens that
    must tokenise to `model_embed_token_size + k` and survive in the middle of text;
  * per-token strings of a byte-level BPE (`Ġ`-spaces decoded to leading blanks, `"` glued to neighbours or not, `<`, `answer`, `>` splits,
    ` </`, `><` merges) through the parser's string tests, with and without thinking mode;
  * REC, OVD (several objects, labels spanning tokens), a VRT run truncated by max_new_tokens (the unguarded look-ahead → IndexError →
    sample dropped), and the `image_prototype` branch (`processor(text=vrts_str)` round trip at :134).
The processor's `batch_decode` follows transformers 4.50 (`[decode(seq) for seq in sequences]`: a 1-D id tensor gives one string PER TOKEN —
what padt_processor.py:76 relies on); transformers 5.15, installed here, joins them, so the stand-in processor states the 4.50 rule.

Writes tests/golden/real_tokenizer.npz (inputs, the serialised tokenizer and the reference's outputs); tests/test_host_logic_cpu.py rebuilds
the tokenizer from the stored JSON and runs padt_amd.processor on the same inputs.

    python tests/golden/make_golden_tok.py          (needs /root/reference: build container only)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

CORPUS = [
    'In this image, the "white dog" is located at .',
    'The "person" , the "traffic light" and the "fire hydrant" are here .',
    '<think> the cat sits on the mat , so the answer is the cat . </think><answer> the "cat" </answer>',
    '"stop sign" , "bus" , "dining table" .',
    'There is no such object in the image .',
]
COMPLETIONS = {
    # name → (texts per sample, need_thinking_mask per sample)
    "rec": (['In this image, the "white dog" is located at <|VRT_3|><|VRT_4|><|VRT_7|>.<|im_end|>',
             'The "traffic light" <|VRT_0|><|VRT_9|> .<|im_end|><|endoftext|><|endoftext|>'], [False, False]),
    "ovd": (['"person" <|VRT_1|><|VRT_2|> , "traffic light" <|VRT_5|><|VRT_9|><|VRT_10|> , "bus"<|VRT_11|>.<|im_end|>',
             'There is no such object in the image .<|im_end|>'], [False, False]),
    "think": (['<think> the cat sits on the mat , so the answer is the cat . </think><answer> the "cat" <|VRT_2|><|VRT_6|> </answer><|im_end|>',
               'the "dog" <|VRT_1|> <answer> the "cat" <|VRT_3|></answer> the "bus" <|VRT_4|><|im_end|>'], [True, True]),
    "truncated": (['The "person" <|VRT_1|><|VRT_2|>', 'The "bus" <|VRT_4|> .<|im_end|>'], [False, False]),
}
GRID = [[1, 8, 8], [1, 6, 10]]                                       # 16 and 15 merged patches
EXTRA_ROWS = 13                                                      # embedding rows beyond the tokenizer's vocabulary (prepare() fills them)


def train_tokenizer_json() -> str:
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=400, min_frequency=1, special_tokens=[], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                             show_progress=False)
    tok.train_from_iterator(CORPUS * 40, tr)
    return tok.to_str()


class TinyProcessor:
    """The surface of an HF processor that padt_processor.py touches: .tokenizer, batch_decode (transformers 4.50 rule: one string per
    element of the outer sequence), __call__(text=..., return_tensors='pt')."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def batch_decode(self, sequences, **kw):
        return [self.tokenizer.decode(seq, **kw) for seq in sequences]

    def __call__(self, text=None, return_tensors=None, **kw):
        return self.tokenizer(text if isinstance(text, list) else [text], return_tensors=return_tensors, **kw)


def processor_from_json(js: str) -> TinyProcessor:
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast
    t = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(js), eos_token="<|im_end|>", pad_token="<|endoftext|>")
    return TinyProcessor(t)


def scenario_inputs(wrapper, seed=0):
    """Token ids of every completion (padded with the pad id), per-step hidden rows and an image prototype table — built through the
    WRAPPED processor, after prepare() and set_image_grid_thw()."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    pad = wrapper.tokenizer.pad_token_id
    for name, (texts, think) in COMPLETIONS.items():
        rows = [wrapper.tokenizer(t)["input_ids"] for t in texts]
        T = max(len(r) for r in rows)
        ids = torch.tensor([r + [pad] * (T - len(r)) for r in rows], dtype=torch.int64)
        hidden = [(torch.randn(len(texts), 1, 8, generator=g),) for _ in range(T)]          # hidden_states[step][-1][batch] → (1, D)
        out[name] = (ids, hidden, torch.tensor(think))
    return out


def run(wrapper_cls, parse_fn, js):
    """The whole scripted use of the wrapper + parser → dict of plain arrays / strings (what the fixture stores and the test recomputes)."""
    proc = processor_from_json(js)
    w = wrapper_cls(proc, 2)
    base_vocab = len(proc.tokenizer.get_vocab())
    assert w.model_embed_token_size == base_vocab
    w.prepare(base_vocab + EXTRA_ROWS)
    res = {"base_vocab": base_vocab, "vocab_after_prepare": len(proc.tokenizer.get_vocab())}
    grid = torch.tensor(GRID)
    w.set_image_grid_thw(grid)
    res["vocab_after_grid"] = len(proc.tokenizer.get_vocab())
    w.set_image_grid_thw(torch.tensor([[1, 4, 4]]))                  # fewer VRTs than present: nothing added
    res["vocab_after_small_grid"] = len(proc.tokenizer.get_vocab())
    enc = w(text='a "dog" <|VRT_3|><|VRT_11|> and<|VRT_15|>.', return_tensors="pt")["input_ids"][0]
    res["vrt_encode_ids"] = enc.numpy()
    res["vrt_encode_tokens"] = json.dumps(proc.batch_decode(enc))
    res["empty_token_id"] = proc.tokenizer.convert_tokens_to_ids("<|empty_token_5|>")
    res["pid2vrt"] = w.pid2vrt(torch.tensor([3, 0, 15]))
    g = torch.Generator().manual_seed(5)
    proto = torch.randn(31, 8, generator=g)                          # 16 + 15 prototype rows of the batch
    for name, (ids, hidden, think) in scenario_inputs(w).items():
        res[name + ".ids"] = ids.numpy()
        res[name + ".tokens"] = json.dumps([proc.batch_decode(r) for r in ids])
        glob = w.assign_to_global_vrt_id(ids.clone(), grid)
        res[name + ".ids_global"] = glob.numpy()
        back = w.assign_to_local_vrt_id(glob.clone(), grid)
        assert torch.equal(back, ids)
        comps, feats, labels, vrts, pfeats = parse_fn(w, ids, hidden, think, image_prototype=proto, image_grid_thw=grid)
        res[name + ".completions"] = json.dumps(comps)
        res[name + ".labels"] = json.dumps(labels)
        res[name + ".vrts"] = json.dumps(vrts)
        res[name + ".n_feats"] = json.dumps([[int(f.shape[0]) for f in fs] for fs in feats])
        res[name + ".feats"] = (torch.cat([f for fs in feats for f in fs], 0) if any(len(fs) for fs in feats) else torch.zeros(0, 8)).numpy()
        res[name + ".proto_feats"] = (torch.cat([f for fs in pfeats for f in fs], 0) if any(len(fs) for fs in pfeats) else torch.zeros(0, 8)).numpy()
    return res


def main():
    sys.path.insert(0, HERE)
    import make_golden as MG
    MG.install_shims()
    from PaDT.models.padt_processor import VisonTextProcessingClass, parseVRTintoCompletion
    js = train_tokenizer_json()
    res = run(VisonTextProcessingClass, parseVRTintoCompletion, js)
    res["tokenizer_json"] = js
    path = os.path.join(HERE, "real_tokenizer.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")
    for name in COMPLETIONS:
        print(name, json.loads(res[name + ".tokens"])[0][:14], "…", res[name + ".labels"], res[name + ".n_feats"])


if __name__ == "__main__":
    main()
