"""Processor wrapper and completion parser — drop-in for ``padt_processor.py`` (same names, arguments, return values and
error behaviour, including the quirks listed in SURVEY.md Appendix C.5/6/10).  Pure host logic: token ids and strings.
"""
import torch

try:  # the real tokenizer path uses HF's AddedToken; fake tokenizers in tests accept plain objects with .content
    from transformers.tokenization_utils import AddedToken
except Exception:  # pragma: no cover
    class AddedToken(str):
        def __new__(cls, content, **kw):
            o = str.__new__(cls, content)
            o.content = content
            return o


class VisonTextProcessingClass(object):
    """padt_processor.py:4-57."""

    def __init__(self, processing_class, spatial_merge_size=2):
        self.processing_class = processing_class
        self.spatial_merge_size = spatial_merge_size
        self.model_embed_token_size = len(processing_class.tokenizer.get_vocab())

    def __getattr__(self, name: str):
        if hasattr(self.processing_class, name):
            return getattr(self.processing_class, name)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def prepare(self, model_embed_token_size):
        """Pad the tokenizer with <|empty_token_i|> up to the embedding rows so <|VRT_k|> gets id rows+k (:15-21)."""
        self.model_embed_token_size = model_embed_token_size
        need_pad_size = model_embed_token_size - len(self.tokenizer.get_vocab())
        assert '<|empty_token_0|>' in self.tokenizer.vocab or need_pad_size > 0
        if need_pad_size > 0:
            self.tokenizer.add_tokens([AddedToken("<|empty_token_%d|>" % i, lstrip=False, rstrip=False, special=True,
                                                  normalized=False) for i in range(need_pad_size)])
        return True

    def set_image_grid_thw(self, image_grid_thw):
        """Grow the vocabulary to the largest merged-patch count in the batch (:23-28)."""
        max_visual_patch_num = image_grid_thw.cumprod(-1).max(dim=0)[0][-1] // (self.spatial_merge_size) ** 2
        have = len(self.processing_class.tokenizer.get_vocab()) - self.model_embed_token_size
        if have < max_visual_patch_num:
            self.processing_class.tokenizer.add_tokens(
                [AddedToken("<|VRT_%d|>" % i, lstrip=False, rstrip=False, special=False, normalized=False)
                 for i in range(have, int(max_visual_patch_num))])
        return True

    def __call__(self, *args, **kwargs):
        parent_ret = self.processing_class(*args, **kwargs)
        if 'image_grid_thw' in parent_ret:
            self.set_image_grid_thw(parent_ret['image_grid_thw'])
        return parent_ret

    def _offsets(self, input_ids, image_grid_thw):
        per = torch.nn.functional.pad((image_grid_thw.cumprod(-1)[:, -1] // (self.spatial_merge_size) ** 2).cumsum(dim=-1),
                                      (1, 0), 'constant', 0)
        return per[:-1, None].expand(-1, input_ids.shape[1]).to(input_ids.device)

    def assign_to_global_vrt_id(self, input_ids, image_grid_thw):
        """In place, like the reference (:36-42): ids >= embed rows get the sample's cumulative patch offset added."""
        visual_patch_mask = input_ids >= self.model_embed_token_size
        if visual_patch_mask.sum() > 0:
            input_ids[visual_patch_mask] += self._offsets(input_ids, image_grid_thw)[visual_patch_mask]
        return input_ids

    def assign_to_local_vrt_id(self, input_ids, image_grid_thw):
        visual_patch_mask = input_ids >= self.model_embed_token_size
        if visual_patch_mask.sum() > 0:
            input_ids[visual_patch_mask] -= self._offsets(input_ids, image_grid_thw)[visual_patch_mask]
        return input_ids

    def pid2vrt(self, patch_ids):
        if type(patch_ids) == int:
            patch_ids = [patch_ids]
        else:
            patch_ids = [int(i) for i in patch_ids]
        return ''.join(['<|VRT_%d|>' % i for i in patch_ids])


def parseVRTintoCompletion(processor, completion_ids, hidden_states, need_thinking_mask=None, image_prototype=None,
                           image_grid_thw=None):
    """padt_processor.py:60-151.  ``hidden_states[step][-1][batch_idx]`` must be the (Lq, D) last-layer state that
    PREDICTED completion token ``step`` (our generate() returns a lazy sequence with exactly that indexing)."""
    ret_list, ret_completions, ret_labels, ret_vrts, ret_vrts_feats = [], [], [], [], []
    if image_grid_thw is not None:
        vision_patch_nums = torch.nn.functional.pad((image_grid_thw.cumprod(-1)[:, -1] // 4).cumsum(-1), (1, 0), 'constant', 0)
    if need_thinking_mask is None:
        need_thinking_mask = torch.ones(len(completion_ids)).to(torch.bool)

    for batch_idx, completion in enumerate(completion_ids):
        toks = processor.batch_decode(completion)
        ret_completions.append(''.join(toks))
        s_list, s_labels, s_vrts, s_vfeats = [], [], [], []
        i = 0
        without_thinking = not need_thinking_mask[batch_idx].item()
        in_answer = False
        in_name = False
        label = ""
        try:
            while i < len(toks):
                if processor.tokenizer.eos_token in toks[i]:
                    break
                if in_answer is False and '<' in toks[i] and '</' not in toks[i] and 'answer' in toks[i + 1] and '>' in toks[i + 2]:
                    in_answer = True
                    i += 3
                    continue
                if in_answer is True or without_thinking:
                    if '</' in toks[i] and 'answer' in toks[i + 1] and '>' in toks[i + 2]:
                        in_answer = False
                        break
                    else:
                        if '"' in toks[i] and in_name is False:
                            in_name = True
                            label = toks[i].split('"')[1]
                            i += 1
                            continue
                        if '"' in toks[i] and in_name is True:
                            in_name = False
                            label += toks[i].split('"')[0]
                            label = label.strip()
                            i += 1
                            continue
                        if '<|VRT_' in toks[i]:
                            in_name = False
                            run_states, run_str = [], ""
                            while '<|VRT_' in toks[i]:                 # IndexError at end-of-completion → sample dropped
                                run_states.append(hidden_states[i][-1][batch_idx])
                                run_str += toks[i]
                                i += 1
                            s_list.append(torch.cat(run_states, dim=0))
                            s_labels.append(label)
                            s_vrts.append(run_str)
                            if image_prototype is not None and image_grid_thw is not None:
                                ids = processor(text=run_str, return_tensors='pt')['input_ids'].to(image_grid_thw.device)[0, ...] \
                                    + vision_patch_nums[batch_idx] - processor.model_embed_token_size
                                s_vfeats.append(image_prototype[ids])
                            continue
                        if in_name:
                            label += toks[i]
                i += 1
            ret_list.append(s_list)
            ret_labels.append(s_labels)
            ret_vrts.append(s_vrts)
            ret_vrts_feats.append(s_vfeats)
        except:  # noqa: E722 — the reference swallows everything per sample (padt_processor.py:146-150)
            ret_list.append([])
            ret_labels.append([])
            ret_vrts.append([])
            ret_vrts_feats.append([])
    return ret_completions, ret_list, ret_labels, ret_vrts, ret_vrts_feats
