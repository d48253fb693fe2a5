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
    """Throughput runner: keeps `depth` decode groups in flight, each group = `merge` consecutive batches.

    * ViT + prefill of a batch (MFMA-bound) run on a normal-priority HIP stream while the decode steps, parse and PaDT
      decoder of the previous group (HBM- / launch-latency-bound) run on that group's HIGH-priority stream; every group
      has its own decode session ("lane": KV caches, token ring, captured graph).
    * merge > 1 (in-flight batching): the decode steps of `merge` consecutive batches share one session of merge*B rows,
      so every step streams the 5.5 GB of LLM weights ONCE for all of them.  ViT, prefill, parse and the PaDT decoder
      still run per batch; per-sample results are identical to rec_batch (decode kernels treat rows independently, the
      GPU tests check bit-equality) — only the interleaving across batches changes.

        r = PipelinedRunner(model, processor); for b in batches: done += r.submit(**b); ...; done += r.flush()
    submit()/flush() return the (decoded, completions, labels, vrts) tuples of the batches that completed, in order.
    """

    def __init__(self, model, processor, depth: int = 2, merge: int = 1, shared_prefill_stream: bool = True, use_graph: bool = True,
                 vit_stream: bool = False):
        self.model, self.processor, self.depth, self.merge = model, processor, depth, max(1, merge)
        import padt_amd
        if padt_amd.hw_queue_note:                             # the runner is the component that needs more than 4 hardware queues
            import warnings
            warnings.warn(padt_amd.hw_queue_note, RuntimeWarning, stacklevel=2)
            padt_amd.hw_queue_note = None
        # vit_stream: the ViT of batch b + 1 on its own stream, concurrent with the LLM prefill of batch b (the two phases leave different
        # tails on the 256 CUs); per-sample results are unchanged (same kernels, same order per batch)
        self.vit_stream = torch.cuda.Stream(device=model.device) if vit_stream else None
        self.use_graph = use_graph     # False: decode steps launched kernel by kernel (counter passes under rocprofv3; same results)
        # shared_prefill_stream=False gives every lane its own prefill stream: GEMMs of two batches may then co-run and
        # fill each other's partial waves (152-tile o_proj on 256 CUs), at the price of L2 / HBM contention
        n_pre = 1 if shared_prefill_stream else depth
        self.prefill_streams = [torch.cuda.Stream(device=model.device) for _ in range(n_pre)]
        # one decode stream per lane: a group's host-synchronising collect must not queue behind the next group's decode
        self.decode_streams = [torch.cuda.Stream(device=model.device, priority=-1) for _ in range(depth)]      # high priority
        self.pending = []          # launched groups, oldest first
        self.cur = None            # group still accepting batches
        self.n_groups = 0
        self.n_batches = 0
        self.trace = None          # set to [] to collect (batch, tag, event) marks for a stream timeline (bench.py --timeline)

    def _mark(self, batch, tag, stream):
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.trace.append((batch, tag, ev))

    def _close_cur(self):
        g = self.cur
        self.cur = None
        if g is None:
            return
        with torch.cuda.stream(g["pre"]):
            self.model.launch_decode(g["ctx"])                    # no-op when the group filled up and launched itself
        self._mark(g["bids"][-1], "decode_end", self.decode_streams[g["lane"]])
        self.pending.append(g)

    def submit(self, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens=1024, schedule=None,
               need_thinking_mask=None, sync_every=None, repetition_penalty=None, eos_token_id=None, **sampling):
        done = []
        bid = self.n_batches
        self.n_batches += 1
        ids = self.processor.assign_to_global_vrt_id(input_ids, image_grid_thw)
        sched = tuple(schedule) if schedule is not None else None
        for attempt in range(2):
            if self.cur is None:
                while len(self.pending) > self.depth - 1:         # the lane about to be reused must have been collected
                    done += self._finish_oldest()
                lane = self.n_groups % self.depth
                self.n_groups += 1
                self.cur = dict(lane=lane, ctx=None, meta=[], bids=[], pre=self.prefill_streams[lane % len(self.prefill_streams)])
            g = self.cur
            pre = g["pre"]
            pre.wait_stream(torch.cuda.current_stream())          # inputs were produced on the caller's stream
            ev_in = torch.cuda.current_stream().record_event() if self.vit_stream is not None else None
            # ... and are consumed on the prefill / decode streams, possibly long after the caller dropped them: tell the
            # caching allocator (record_stream) so their blocks are not handed to the caller's next batch while a lagging
            # side stream still reads them; the group also holds references until its results were collected
            for t in (ids, attention_mask, pixel_values, image_grid_thw):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(pre)
                    t.record_stream(self.decode_streams[g["lane"]])
                    if self.vit_stream is not None:
                        t.record_stream(self.vit_stream)
            self._mark(bid, "prefill_begin", pre)
            with torch.cuda.stream(pre):
                ctx = self.model.generate_launch(ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens, False,
                                                 sched, sync_every or max_new_tokens, self.use_graph, g["lane"],
                                                 self.decode_streams[g["lane"]], group=g["ctx"], n_slots=self.merge,
                                                 repetition_penalty=repetition_penalty, eos_token_id=eos_token_id,
                                                 vit_stream=self.vit_stream, inputs_ready=ev_in, **sampling)
            if ctx is not None:
                break
            self._close_cur()                                     # batch does not fit this group's session: start a new one
        if ctx is None:
            raise RuntimeError("batch rejected by an empty decode group")
        self._mark(bid, "prefill_end", pre)
        g["ctx"] = ctx
        g["meta"].append((input_ids.shape, image_grid_thw, need_thinking_mask))
        g.setdefault("keep", []).append((ids, attention_mask, pixel_values))
        g["bids"].append(bid)
        if len(g["meta"]) == self.merge:
            self._close_cur()
        while len(self.pending) >= self.depth:
            done += self._finish_oldest()
        return done

    def _finish_oldest(self):
        g = self.pending.pop(0)
        lane = g["lane"]
        res = []
        with torch.cuda.stream(self.decode_streams[lane]):
            outs = self.model.generate_collect(g["ctx"], all_batches=True)
            for out, (shape, grid, mask), bid in zip(outs, g["meta"], g["bids"]):
                B, L = shape
                seq_local = self.processor.assign_to_local_vrt_id(out["sequences"].cpu(), grid.cpu())
                m = mask if mask is not None else torch.Tensor([False] * B)
                completions, feats, labels, vrts, _ = parseVRTintoCompletion(self.processor, seq_local[:, L:], out["hidden_states"], m)
                self._mark(bid, "vl_begin", self.decode_streams[lane])
                decoded = self.model.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)
                self._mark(bid, "vl_end", self.decode_streams[lane])
                res.append((decoded, completions, labels, vrts))
        self.decode_streams[lane].synchronize()
        return res

    def flush(self):
        res = []
        self._close_cur()
        while self.pending:
            res += self._finish_oldest()
        return res


# --------------------------------------------------------------------------------------------- result exchange (RCCL)
# ONE contiguous fixed-capacity record per rank and batch → ONE all_gather_into_tensor (one latency hop on xGMI instead of one per
# field).  32-bit words:  [n, cap, mask_hw, has_mask | sample_idx (cap) | valid_hw (2 cap) | boxes f32 (4 cap) | scores f32 (cap) |
# mask logits f32 (cap * mask_hw^2)] — fp32 throughout: what travels is bit for bit what vl_decode returned.
_HDR = 4


def _record_words(cap: int, mask_hw: int) -> int:
    return _HDR + cap * (1 + 2 + 4 + 1) + cap * mask_hw * mask_hw


def pack_results(decoded: dict, cap: int, mask_hw: int, device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vl_decode output → one int32 record (floats bit-cast) of fixed size, so every rank contributes the same shape.
    On the GPU this is ONE kernel launch (padt_pack_results: no host round trip, no H2D copy); host tensors (the protocol tests) are
    packed with the equivalent indexing statements below."""
    n = decoded["pred_boxes"].shape[0]
    if n > cap:
        raise ValueError(f"{n} objects exceed the exchange capacity {cap}")
    words = _record_words(cap, mask_hw)
    has_mask = decoded.get("pred_mask") is not None and n > 0
    if has_mask and (decoded["pred_mask"].shape[1] > mask_hw or decoded["pred_mask"].shape[2] > mask_hw):
        raise ValueError(f"mask {tuple(decoded['pred_mask'].shape[1:])} exceeds exchange capacity {mask_hw}")
    if torch.device(device).type == "cuda":
        from . import ops
        buf = out if out is not None else torch.empty(words, dtype=torch.int32, device=device)
        sidx = decoded.get("sample_idx_t")
        if n and (sidx is None or sidx.numel() != n):                 # a dict that did not come from vl_decode
            sidx = torch.tensor(decoded["sample_idx"], dtype=torch.int32).to(device, non_blocking=True)
        hw = decoded["pred_mask_valid_hw"] if has_mask else (None, None)
        # the decoded tensors were allocated on the runner lane's stream; the pack kernel reads them on the caller's: tell the caching
        # allocator, or their blocks could be handed to the lane's next batch while the pack is still queued here
        cur = torch.cuda.current_stream()
        for t in (sidx, hw[0], hw[1], decoded["pred_boxes"], decoded["pred_score"], decoded["pred_mask"] if has_mask else None):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        return ops.pack_results(buf, n, cap, mask_hw, sidx, hw[0], hw[1], decoded["pred_boxes"] if n else None,
                                decoded["pred_score"] if n else None, decoded["pred_mask"] if has_mask else None)
    buf = out if out is not None else torch.empty(words, dtype=torch.int32, device=device)
    buf.zero_()
    fbuf = buf.view(torch.float32)
    buf[:_HDR] = torch.tensor([n, cap, mask_hw, 1 if has_mask else 0], dtype=torch.int32, device=device)
    o = _HDR
    if n:
        buf[o: o + n] = torch.tensor(decoded["sample_idx"], dtype=torch.int32, device=device)
    o += cap
    hw = buf[o: o + 2 * cap].view(cap, 2)
    o += 2 * cap
    if n:
        fbuf[o: o + 4 * cap].view(cap, 4)[:n] = decoded["pred_boxes"].to(device=device, dtype=torch.float32)
    o += 4 * cap
    if n:
        fbuf[o: o + cap][:n] = decoded["pred_score"].to(device=device, dtype=torch.float32).reshape(-1)
    o += cap
    if has_mask:
        H, Wd = decoded["pred_mask"].shape[1:]
        fbuf[o:].view(cap, mask_hw, mask_hw)[:n, :H, :Wd] = decoded["pred_mask"].to(device=device, dtype=torch.float32)
        hw[:n, 0] = decoded["pred_mask_valid_hw"][0].to(device=device, dtype=torch.int32)
        hw[:n, 1] = decoded["pred_mask_valid_hw"][1].to(device=device, dtype=torch.int32)
    return buf


def all_gather_results(packed: torch.Tensor, group=None) -> torch.Tensor:
    """The path's only collective: one all_gather_into_tensor of the fixed-size records → (world, words) int32."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world, packed.numel()), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1), packed.contiguous(), group=group)
    return out


def unpack_results(gathered: torch.Tensor, batch_per_rank: int) -> List[dict]:
    """(world, words) records → list over ranks of {sample_idx (global), boxes, scores, valid_hw, masks} trimmed to each rank's count."""
    res = []
    for r in range(gathered.shape[0]):
        rec = gathered[r]
        frec = rec.view(torch.float32)
        n, cap, mask_hw, has_mask = (int(v) for v in rec[:_HDR].tolist())
        has_mask &= 1                                                  # bits 8 and up: continuation marker (ResultExchange)
        o = _HDR
        sidx = rec[o: o + cap][:n].long() + r * batch_per_rank
        o += cap
        hw = rec[o: o + 2 * cap].view(cap, 2)[:n]
        o += 2 * cap
        boxes = frec[o: o + 4 * cap].view(cap, 4)[:n]
        o += 4 * cap
        scores = frec[o: o + cap][:n]
        o += cap
        masks = frec[o:].view(cap, mask_hw, mask_hw)[:n] if has_mask else None
        res.append({"sample_idx": sidx, "boxes": boxes, "scores": scores, "valid_hw": hw, "masks": masks})
    return res


def rank_batches(n_items: int, batch_size: int, rank: int, world: int):
    """Start indices rank ``rank`` processes — the reference's rule (utils.py:181-182): every rank walks the same number
    of batches; starts past the end mean "enter the loop, skip the work"."""
    import math
    all_number = math.ceil(n_items / (world * batch_size)) * world * batch_size
    return list(range(rank * batch_size, all_number, world * batch_size))


def _cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' → [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def plan_rank_affinity(local_rank: int, local_world: int, gpu_numa: List[int], node_cpus: dict, allowed: List[int]) -> List[int]:
    """Host cores for one rank of a one-process-per-GPU job (the reference launches its eval the same way, eval_coco.sh:5 / utils.py:181-182):
    the cores of the NUMA node its GPU hangs off, divided evenly among the ranks whose GPUs share that node; without topology information an
    even slice of the allowed cores.  Pure function of its arguments (tested on CPU); every rank gets a disjoint, non-empty set when
    len(allowed) >= local_world."""
    allowed = sorted(allowed)
    node = gpu_numa[local_rank] if local_rank < len(gpu_numa) else -1
    cpus = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node >= 0 else []
    if cpus:
        peers = [r for r in range(local_world) if r < len(gpu_numa) and gpu_numa[r] == node]
        i, n = peers.index(local_rank), len(peers)
    else:
        cpus, i, n = allowed, local_rank, local_world
    per = max(1, len(cpus) // n)
    mine = cpus[i * per: (i + 1) * per] if i < n - 1 else cpus[i * per:]
    return mine or cpus[i % len(cpus): i % len(cpus) + 1]


def pin_rank_to_local_cores(local_rank: int, local_world: int) -> dict:
    """sched_setaffinity of this process to plan_rank_affinity()'s cores, read from sysfs (GPU PCI address → numa_node → cpulist); best effort:
    returns what was done ({'cpus': n, 'numa_node': k, 'first': c} or {'skipped': reason}).  Each rank runs a host enqueue thread of ≈7 ms per
    batch plus the parser; 8 ranks bouncing between sockets cost the decode groups their launch cadence."""
    import os
    try:
        allowed = sorted(os.sched_getaffinity(0))
        numa = []
        for r in range(local_world):
            node = -1
            try:
                p = torch.cuda.get_device_properties(r)
                bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
                node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
            except Exception:                                          # noqa: BLE001 — no sysfs entry / fewer devices than ranks
                pass
            numa.append(node)
        nodes = {}
        base = "/sys/devices/system/node"
        if os.path.isdir(base):
            for d in os.listdir(base):
                if d.startswith("node") and d[4:].isdigit():
                    try:
                        nodes[int(d[4:])] = _cpulist(open(os.path.join(base, d, "cpulist")).read())
                    except OSError:
                        pass
        mine = plan_rank_affinity(local_rank, local_world, numa, nodes, allowed)
        os.sched_setaffinity(0, mine)
        return {"cpus": len(mine), "numa_node": numa[local_rank] if local_rank < len(numa) else -1, "first": mine[0]}
    except Exception as e:                                             # noqa: BLE001
        return {"skipped": f"{type(e).__name__}: {e}"[:120]}


def _split_decoded(decoded: dict, cap: int) -> List[dict]:
    """A vl_decode output with more than `cap` objects → consecutive chunks of at most `cap` objects each (views, no copies)."""
    n = decoded["pred_boxes"].shape[0]
    has_mask = decoded.get("pred_mask") is not None
    chunks = []
    for a in range(0, max(n, 1), cap):
        b = min(n, a + cap)
        c = {"pred_boxes": decoded["pred_boxes"][a:b], "pred_score": decoded["pred_score"][a:b], "sample_idx": list(decoded["sample_idx"][a:b]),
             "pred_mask": decoded["pred_mask"][a:b] if has_mask else None,
             "pred_mask_valid_hw": tuple(t[a:b] for t in decoded["pred_mask_valid_hw"]) if has_mask else ()}
        if decoded.get("sample_idx_t") is not None:
            c["sample_idx_t"] = decoded["sample_idx_t"][a:b]
        chunks.append(c)
    return chunks


class GatheredGroup:
    """What ONE completed gather delivered to this rank: `records` (world, per_gather, words) int32 — slot b of rank r is the record of that
    rank's b-th batch of the group (at most `cap` objects) — and, only when some rank had a batch over capacity, `continuation`
    (world, k, words): the overflow records (word 3 of their header carries 1 + the slot they continue in bits 8 and up).  Always read a
    batch through batch_record(): it re-assembles slot + continuation records; indexing `records` alone drops the objects beyond `cap`.
    `records` is a view of the exchange's double buffer: consume (unpack / clone()) it before the second-next gather is issued."""

    def __init__(self, records: torch.Tensor, continuation: Optional[torch.Tensor] = None):
        self.records, self.continuation = records, continuation

    @property
    def world(self) -> int:
        return self.records.shape[0]

    @property
    def per(self) -> int:
        return self.records.shape[1]

    def clone(self) -> "GatheredGroup":
        return GatheredGroup(self.records.clone(), None if self.continuation is None else self.continuation.clone())

    def cpu(self) -> "GatheredGroup":
        return GatheredGroup(self.records.cpu(), None if self.continuation is None else self.continuation.cpu())

    def batch_record(self, src_rank: int, slot: int) -> dict:
        """One batch's results as rank `src_rank` computed them: the slot's record with its continuation records (if any) appended in order."""
        parts = [unpack_results(self.records[src_rank, slot][None], batch_per_rank=0)[0]]
        if self.continuation is not None:
            for j in range(self.continuation.shape[1]):
                if (int(self.continuation[src_rank, j, 3]) >> 8) == slot + 1:
                    parts.append(unpack_results(self.continuation[src_rank, j][None], batch_per_rank=0)[0])
        if len(parts) == 1:
            return parts[0]
        masks = [p["masks"] for p in parts]
        return {"sample_idx": torch.cat([p["sample_idx"] for p in parts]), "boxes": torch.cat([p["boxes"] for p in parts]),
                "scores": torch.cat([p["scores"] for p in parts]), "valid_hw": torch.cat([p["valid_hw"] for p in parts]),
                "masks": torch.cat(masks) if all(m is not None for m in masks) else None}


class ResultExchange:
    """The data-parallel exchange without lock-step: every rank packs `per_gather` consecutive batch records (normally one decode group)
    into one buffer and issues ONE asynchronous all_gather_into_tensor for them (RCCL runs it on its own stream); the handle is only waited
    on when the NEXT gather is issued or at flush(), so ranks meet once per decode group at most and a straggler delays nobody's compute.
    Every rank must add() the same number of batches.

        ex = ResultExchange(cap, mask_hw, per_gather=8, device=dev)
        for decoded in ...: done += ex.add(decoded)        # → list of GatheredGroup whose gather completed
        done += ex.flush()
        for g in done: for b in range(g.per): rec = g.batch_record(src_rank, b)

    A batch with more objects than `cap` (an OVD image can carry tens of objects, padt.py:370) does NOT raise — on one rank that would leave
    the other ranks waiting in the gather for ever: its first `cap` objects travel in the batch's slot, the rest as CONTINUATION records.  The
    gather buffer's last word announces how many continuation records the rank holds for this gather; when the gather completes every rank
    sees every rank's count, so all ranks agree — without a further message — on k = the largest count and run ONE more (synchronous)
    gather of k records per rank (GatheredGroup.continuation).  With no rank over capacity (the normal case) nothing changes: one collective
    per decode group.  The announced counts reach the host through a pinned buffer filled on a side stream behind the gather (round 5): the
    common k = 0 case costs no synchronisation of the caller's stream."""

    def __init__(self, cap: int, mask_hw: int, per_gather: int, device, group=None):
        import torch.distributed as dist
        self.cap, self.mask_hw, self.per, self.device, self.group = cap, mask_hw, max(1, per_gather), device, group
        self.world = dist.get_world_size(group)
        self.words = _record_words(cap, mask_hw)
        n = self.per * self.words + 1                                  # + the continuation count
        self._bufs = [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(2)]                         # double buffer
        self._outs = [torch.empty((self.world, n), dtype=torch.int32, device=device) for _ in range(2)]
        self._cur, self._fill, self._inflight = 0, 0, None
        self._cont = []                                                # continuation records of the gather being filled
        self.n_gathers = 0
        self.n_continuation_gathers = 0
        self.wait_ms_total, self.wait_ms_max = 0.0, 0.0                # host time spent waiting for a gather to complete (_wait): what a straggler costs
        self._cuda = torch.device(device).type == "cuda"
        if self._cuda:
            self._side = torch.cuda.Stream(device=device)
            self._counts = [torch.zeros(self.world, dtype=torch.int32).pin_memory() for _ in range(2)]

    def _slots(self, flat: torch.Tensor) -> torch.Tensor:
        return flat[..., : self.per * self.words].unflatten(-1, (self.per, self.words))

    def add(self, decoded: dict) -> List[GatheredGroup]:
        chunks = _split_decoded(decoded, self.cap) if decoded["pred_boxes"].shape[0] > self.cap else [decoded]
        pack_results(chunks[0], self.cap, self.mask_hw, self.device, out=self._slots(self._bufs[self._cur])[self._fill])
        for c in chunks[1:]:
            rec = pack_results(c, self.cap, self.mask_hw, self.device)
            rec[3] += (self._fill + 1) << 8
            self._cont.append(rec)
        self._fill += 1
        return self._launch() if self._fill == self.per else []

    def _wait(self) -> List[GatheredGroup]:
        if self._inflight is None:
            return []
        import torch.distributed as dist
        import time
        work, out, cont, ev, counts = self._inflight
        self._inflight = None
        t0 = time.perf_counter()
        if ev is not None:
            ev.synchronize()                                           # the gather and the copy of the announced counts behind it are done
            torch.cuda.current_stream().wait_event(ev)                 # ... and the caller's stream may read the gathered records
            k = int(counts.max())
        else:
            work.wait()
            k = int(out[:, -1].max().item())                           # every rank computes the same k from the same gathered words
        dt = (time.perf_counter() - t0) * 1e3
        self.wait_ms_total += dt
        self.wait_ms_max = max(self.wait_ms_max, dt)
        more = None
        if k:
            mine = torch.zeros((k, self.words), dtype=torch.int32, device=self.device)
            for i, rec in enumerate(cont):
                mine[i] = rec
            more = torch.empty((self.world, k, self.words), dtype=torch.int32, device=self.device)
            dist.all_gather_into_tensor(more.view(-1), mine.view(-1), group=self.group)
            self.n_continuation_gathers += 1
        return [GatheredGroup(self._slots(out), more)]

    def _launch(self) -> List[GatheredGroup]:
        import torch.distributed as dist
        done = self._wait()                                            # at most one gather in flight: its buffers are free again
        buf, out = self._bufs[self._cur], self._outs[self._cur]
        if self._fill < self.per:
            self._slots(buf)[self._fill:].zero_()                      # partially filled last group: n = 0 records
        buf[-1:].fill_(len(self._cont))
        work = dist.all_gather_into_tensor(out.view(-1), buf, group=self.group, async_op=True)
        ev, counts = None, None
        if self._cuda:
            counts = self._counts[self._cur]
            with torch.cuda.stream(self._side):
                work.wait()                                            # orders the SIDE stream behind the collective; the caller's stream goes on
                counts.copy_(out[:, -1], non_blocking=True)
                ev = self._side.record_event()
        self._inflight = (work, out, self._cont, ev, counts)
        self._cont = []
        self.n_gathers += 1
        self._cur ^= 1
        self._fill = 0
        return done

    def flush(self) -> List[GatheredGroup]:
        done = self._launch() if self._fill else []
        return done + self._wait()

    def batch_record(self, gathered: GatheredGroup, src_rank: int, slot: int) -> dict:
        return gathered.batch_record(src_rank, slot)
