"""The reference callers' loop body (eval/test_demo.py:84-113, eval/evaluation_scripts/utils.py:220-246) as one function,
plus the data-parallel result exchange.

processor outputs → assign_to_global_vrt_id → generate → assign_to_local_vrt_id → parseVRTintoCompletion → vl_decode.
Images are independent, so multi-GPU = one full replica per GPU, rank-strided batches (utils.py:181-182) and ONE
all-gather of packed results per batch over RCCL/xGMI; there is no collective on the forward path.
"""
from typing import List, Optional, Sequence

import torch

from .processor import parseVRTintoCompletion


def rec_batch(model, processor, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens=1024,
              schedule: Optional[Sequence] = None, need_thinking_mask=None, **gen_kw):
    """→ (decoded dict of vl_decode, completions, labels, vrts).  ``input_ids`` carries LOCAL VRT ids (as a processor
    emits them) and is updated in place exactly like the reference's callers do."""
    B = input_ids.shape[0]
    ids = processor.assign_to_global_vrt_id(input_ids, image_grid_thw)
    out = model.generate(input_ids=ids, attention_mask=attention_mask, pixel_values=pixel_values,
                         image_grid_thw=image_grid_thw, use_cache=True, max_new_tokens=max_new_tokens, do_sample=False,
                         output_hidden_states=True, return_dict_in_generate=True, schedule=schedule, **gen_kw)
    L = input_ids.shape[1]
    seq_local = processor.assign_to_local_vrt_id(out["sequences"].cpu(), image_grid_thw.cpu())
    completion_ids = seq_local[:, L:]
    mask = need_thinking_mask if need_thinking_mask is not None else torch.Tensor([False] * B)
    completions, feats, labels, vrts, _ = parseVRTintoCompletion(processor, completion_ids, out["hidden_states"], mask)
    decoded = model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, image_grid_thw, out.past_visual_pe)
    return decoded, completions, labels, vrts


class PipelinedRunner:
    """Keeps `depth` batches in flight: ViT + prefill of batch i+1 (MFMA-bound) run on a normal-priority HIP stream while
    the decode steps, parse and PaDT decoder of batch i (HBM- / launch-latency-bound) run on a HIGH-priority stream, each
    batch with its own decode session ("lane": KV caches, token ring, captured graph).  Results are identical to
    rec_batch (same kernels, same order within a batch); only the interleaving across batches changes.

        r = PipelinedRunner(model, processor); for b in batches: done = r.submit(**b); ...; rest = r.flush()
    """

    def __init__(self, model, processor, depth: int = 2):
        self.model, self.processor, self.depth = model, processor, depth
        self.prefill_stream = torch.cuda.Stream(device=model.device)
        # one decode stream per lane: a batch's host-synchronising collect must not queue behind the next batch's decode
        self.decode_streams = [torch.cuda.Stream(device=model.device, priority=-1) for _ in range(depth)]
        self.pending = []
        self.count = 0
        self.trace = None          # set to [] to collect (batch, tag, event) marks for a stream timeline (bench.py --timeline)

    def _mark(self, batch, tag, stream):
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.trace.append((batch, tag, ev))

    def submit(self, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens=1024, schedule=None,
               need_thinking_mask=None, sync_every=None):
        lane = self.count % self.depth
        bid = self.count
        self.count += 1
        self.prefill_stream.wait_stream(torch.cuda.current_stream())   # inputs were produced on the caller's stream
        ids = self.processor.assign_to_global_vrt_id(input_ids, image_grid_thw)
        self._mark(bid, "prefill_begin", self.prefill_stream)
        with torch.cuda.stream(self.prefill_stream):
            ctx = self.model.generate_launch(ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens, False,
                                             schedule, sync_every or max_new_tokens, True, lane, self.decode_streams[lane])
        self._mark(bid, "prefill_end", self.prefill_stream)
        self._mark(bid, "decode_end", self.decode_streams[lane])
        self.pending.append((lane, ctx, input_ids.shape, image_grid_thw, need_thinking_mask, bid))
        return self._finish_oldest() if len(self.pending) >= self.depth else None

    def _finish_oldest(self):
        lane, ctx, shape, grid, mask, bid = self.pending.pop(0)
        B, L = shape
        with torch.cuda.stream(self.decode_streams[lane]):
            out = self.model.generate_collect(ctx)
            seq_local = self.processor.assign_to_local_vrt_id(out["sequences"].cpu(), grid.cpu())
            m = mask if mask is not None else torch.Tensor([False] * B)
            completions, feats, labels, vrts, _ = parseVRTintoCompletion(self.processor, seq_local[:, L:], out["hidden_states"], m)
            self._mark(bid, "vl_begin", self.decode_streams[lane])
            decoded = self.model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
            self._mark(bid, "vl_end", self.decode_streams[lane])
        self.decode_streams[lane].synchronize()
        return decoded, completions, labels, vrts

    def flush(self):
        res = []
        while self.pending:
            res.append(self._finish_oldest())
        return res


# --------------------------------------------------------------------------------------------- result exchange (RCCL)
def pack_results(decoded: dict, cap: int, mask_hw: int, device) -> dict:
    """Fixed-capacity buffers so every rank contributes the same shapes to one all_gather."""
    n = decoded["pred_boxes"].shape[0]
    if n > cap:
        raise ValueError(f"{n} objects exceed the exchange capacity {cap}")
    count = torch.tensor([n], dtype=torch.int32, device=device)
    sample = torch.full((cap,), -1, dtype=torch.int32, device=device)
    boxes = torch.zeros((cap, 4), dtype=torch.float32, device=device)
    scores = torch.zeros((cap,), dtype=torch.float32, device=device)
    hw = torch.zeros((cap, 2), dtype=torch.int32, device=device)
    masks = torch.zeros((cap, mask_hw, mask_hw), dtype=torch.bfloat16, device=device)
    if n:
        sample[:n] = torch.tensor(decoded["sample_idx"], dtype=torch.int32, device=device)
        boxes[:n] = decoded["pred_boxes"].float()
        scores[:n] = decoded["pred_score"].float().reshape(-1)
        if decoded["pred_mask"] is not None:
            H, Wd = decoded["pred_mask"].shape[1:]
            if H > mask_hw or Wd > mask_hw:
                raise ValueError(f"mask {H}x{Wd} exceeds exchange capacity {mask_hw}")
            masks[:n, :H, :Wd] = decoded["pred_mask"].to(torch.bfloat16)
            hw[:n, 0] = decoded["pred_mask_valid_hw"][0].to(torch.int32)
            hw[:n, 1] = decoded["pred_mask_valid_hw"][1].to(torch.int32)
    return {"count": count, "sample_idx": sample, "boxes": boxes, "scores": scores, "valid_hw": hw, "masks": masks}


def all_gather_results(packed: dict, group=None) -> dict:
    """One all_gather per field (5 small + 1 mask buffer; ≈0.55 MB/rank for REC bs 8 → latency-bound on xGMI)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = {}
    for k, t in packed.items():
        buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf.view(-1), t.contiguous().view(-1), group=group)
        out[k] = buf
    return out


def unpack_results(gathered: dict, batch_per_rank: int) -> List[dict]:
    """→ list over ranks of {global_sample_idx, boxes, scores, valid_hw, masks} trimmed to each rank's count."""
    world = gathered["count"].shape[0]
    res = []
    for r in range(world):
        n = int(gathered["count"][r, 0])
        res.append({
            "sample_idx": (gathered["sample_idx"][r, :n].long() + r * batch_per_rank),
            "boxes": gathered["boxes"][r, :n], "scores": gathered["scores"][r, :n],
            "valid_hw": gathered["valid_hw"][r, :n], "masks": gathered["masks"][r, :n]})
    return res


def rank_batches(n_items: int, batch_size: int, rank: int, world: int):
    """Start indices rank ``rank`` processes — the reference's rule (utils.py:181-182): every rank walks the same number
    of batches; starts past the end mean "enter the loop, skip the work"."""
    import math
    all_number = math.ceil(n_items / (world * batch_size)) * world * batch_size
    return list(range(rank * batch_size, all_number, world * batch_size))
