"""Benchmark / self-test support shipped WITH the package (round 6: bench.py and smoke() no longer import from tests/, so an installed
package can be benchmarked).  Nothing on the product path imports this module.  Synthetic inputs for a model with random-init weights (no datasets / checkpoints offline): prompts of
the reference's shape, N(0,1) pixel rows, a scripted completion schedule (random weights never emit EOS), and a minimal
tokenizer stand-in so ``parseVRTintoCompletion`` can run on generated ids."""
import torch

from padt_amd.config import PaDTConfig


def synthetic_batch(cfg: PaDTConfig, grids, n_pre=15, n_post=33, seed=1234, ragged=False):
    """n_pre text ids (the last one is <|vision_start|>) + N x <|image_pad|> + n_post text ids per sample; pixel_values
    ~ N(0,1) rounded to bf16.  ``ragged`` varies the text lengths per sample; rows are left-padded (padding_side="left",
    test_demo.py:79).  Returns (image_grid_thw, pixel_values fp32, input_ids, attention_mask) on CPU."""
    g = torch.Generator().manual_seed(seed)
    grid = torch.tensor(grids, dtype=torch.long)
    P = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    pix = torch.randn(P, cfg.patch_dim, generator=g).to(torch.bfloat16).float()
    rows = []
    hi = min(cfg.vocab_size, cfg.image_token_id) - 1
    for b, (t, h, w) in enumerate(grids):
        n = t * h * w // cfg.merge_unit
        pre = torch.randint(0, hi, (n_pre - 1 + (b % 3 if ragged else 0),), generator=g).tolist()
        post = torch.randint(0, hi, (n_post + (2 * b % 5 if ragged else 0),), generator=g).tolist()
        rows.append(pre + [cfg.vision_start_token_id] + [cfg.image_token_id] * n + post)
    L = max(len(r) for r in rows)
    ids = torch.full((len(rows), L), cfg.pad_token_id, dtype=torch.long)
    am = torch.zeros((len(rows), L), dtype=torch.long)
    for b, r in enumerate(rows):
        ids[b, L - len(r):] = torch.tensor(r)
        am[b, L - len(r):] = 1
    return grid, pix, ids, am


def rec_schedule(t_new=28, vrt_at=range(11, 16)):
    """REC-shaped completion: text … one run of VRTs … text, EOS forced at the last step (1 object x 5 VRTs by default)."""
    s = ["t"] * t_new
    for i in vrt_at:
        s[i] = "v"
    s[-1] = "e"
    return s


def multi_object_schedule(t_new=120, n_obj=7, n_vrt=5):
    """OVD-shaped completion (src/preprocess/process_coco.py:135-164 answer template): n_obj runs of n_vrt VRTs separated by text
    (label, quotes, separators), evenly spread, EOS forced at the last step."""
    gap = (t_new - 2 - n_obj * n_vrt) // (n_obj + 1)
    if gap < 1:
        raise ValueError("t_new too short for the requested objects")
    s = ["t"] * t_new
    pos = gap
    for _ in range(n_obj):
        for i in range(pos, pos + n_vrt):
            s[i] = "v"
        pos += n_vrt + gap
    s[-1] = "e"
    return s


class FakeTokenizer:
    def __init__(self, cfg: PaDTConfig, n_vrt: int):
        self.cfg, self.n_vrt = cfg, n_vrt
        self.eos_token = "<|im_end|>"

    def tok(self, i: int) -> str:
        c = self.cfg
        if i == c.eos_token_id:
            return self.eos_token
        if i == c.pad_token_id:
            return "<|endoftext|>"
        if i >= c.vocab_size:
            return "<|VRT_%d|>" % (i - c.vocab_size)
        return " w%d" % i

    def get_vocab(self):
        return {self.tok(i): i for i in range(self.cfg.vocab_size + self.n_vrt)}

    @property
    def vocab(self):
        return self.get_vocab()


class FakeProcessor:
    """Just enough of an HF processor for VisonTextProcessingClass / parseVRTintoCompletion on token ids."""
    def __init__(self, cfg, n_vrt):
        self.tokenizer = FakeTokenizer(cfg, n_vrt)

    def batch_decode(self, ids):
        return [self.tokenizer.tok(int(i)) for i in ids]
