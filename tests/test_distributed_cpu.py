"""world_size-2 gloo run of the data-parallel result exchange (the only collective on the path):
each rank packs its own vl_decode outputs into ONE fixed-capacity record, ONE all_gather_into_tensor, every rank unpacks all
ranks' results with globally re-based sample indices — fp32 payload, bit for bit what vl_decode returned."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from padt_amd import pipeline


def _ship(o):
    """Tensors → numpy before they go through a multiprocessing queue: a tensor travels as a shared-memory handle that the parent may
    only open after the worker is gone (FileNotFoundError, seen as a flaky failure); arrays are pickled by value."""
    if isinstance(o, torch.Tensor):
        return ("__tensor__", o.detach().cpu().numpy().copy())
    if isinstance(o, (list, tuple)):
        return type(o)(_ship(x) for x in o)
    return o


def _unship(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == "__tensor__":
        return torch.from_numpy(o[1])
    if isinstance(o, (list, tuple)):
        return type(o)(_unship(x) for x in o)
    return o


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_decoded(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return {"pred_boxes": torch.rand(n, 4, generator=g), "pred_score": torch.randn(n, 1, generator=g),
            "pred_mask": torch.randn(n, 24, 32, generator=g), "sample_idx": [i % 4 for i in range(n)],
            "pred_mask_valid_hw": (torch.full((n,), 6), torch.full((n,), 8))}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 3 if rank == 0 else 0                                   # rank 1 has no objects (empty vl_decode) — still participates
    dec = _fake_decoded(rank, n) if n else {"pred_boxes": torch.zeros(0, 4), "pred_score": torch.zeros(0, 1),
                                             "pred_mask": torch.zeros(0, 8, 8), "sample_idx": [], "pred_mask_valid_hw": ()}
    packed = pipeline.pack_results(dec, cap=8, mask_hw=32, device="cpu")
    gathered = pipeline.all_gather_results(packed)
    res = pipeline.unpack_results(gathered, batch_per_rank=4)
    assert gathered.shape == (world, packed.numel()) and gathered.dtype == torch.int32
    q.put(_ship((rank, [r["boxes"].clone() for r in res], [r["sample_idx"].clone() for r in res],
                 [None if r["masks"] is None else r["masks"].clone() for r in res], [r["scores"].clone() for r in res], [r["valid_hw"].clone() for r in res])))
    dist.barrier()
    dist.destroy_process_group()


def test_result_all_gather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [_unship(q.get(timeout=120)) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_decoded(0, 3)
    for rank, boxes, sidx, masks, scores, hw in got:
        assert len(boxes) == 2 and boxes[0].shape == (3, 4) and boxes[1].shape == (0, 4)
        assert torch.equal(boxes[0], ref["pred_boxes"]) and torch.equal(scores[0], ref["pred_score"].reshape(-1))
        assert sidx[0].tolist() == [0, 1, 2] and sidx[1].tolist() == []
        assert torch.equal(masks[0][:, :24, :32], ref["pred_mask"])              # fp32 on the wire: bit-exact
        assert (masks[0][:, 24:] == 0).all() and masks[1] is None
        assert hw[0].tolist() == [[6, 8]] * 3


def test_pack_capacity_errors():
    with pytest.raises(ValueError, match="exceed the exchange capacity"):
        pipeline.pack_results(_fake_decoded(0, 5), cap=4, mask_hw=32, device="cpu")
    with pytest.raises(ValueError, match="exceeds exchange capacity"):
        pipeline.pack_results(_fake_decoded(0, 2), cap=4, mask_hw=16, device="cpu")


def _worker_exchange(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = pipeline.ResultExchange(cap=8, mask_hw=32, per_gather=3, device="cpu")
    done = []
    for b in range(7):                                          # 7 batches, 3 per gather → gathers of 3, 3 and 1 (+ 2 empty records)
        n = (b + rank) % 4                                      # ragged object counts, some batches empty
        dec = _fake_decoded(10 * rank + b, n) if n else {"pred_boxes": torch.zeros(0, 4), "pred_score": torch.zeros(0, 1),
                                                         "pred_mask": torch.zeros(0, 8, 8), "sample_idx": [], "pred_mask_valid_hw": ()}
        done += [t.clone() for t in ex.add(dec)]
    done += [t.clone() for t in ex.flush()]
    q.put(_ship((rank, ex.n_gathers, [d.records.clone() for d in done])))
    dist.barrier()
    dist.destroy_process_group()


def test_result_exchange_groups_gather_asynchronously_and_in_order():
    """ResultExchange: one asynchronous all-gather per group of batches, at most one in flight, results delivered in submission order;
    every rank sees every rank's records bit for bit (incl. empty batches and the zero-padded tail group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [_unship(q.get(timeout=120)) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_gathers, done in got:
        assert n_gathers == 3 and len(done) == 3
        recs = torch.cat(done, dim=1)                           # (world, 9, words): 7 real batches + 2 zero records (GatheredGroup.records)
        assert recs.shape[:2] == (2, 9)
        for r in range(2):
            for b in range(9):
                res = pipeline.unpack_results(recs[r, b][None], batch_per_rank=0)[0]
                n = (b + r) % 4 if b < 7 else 0
                assert res["boxes"].shape == (n, 4)
                if n:
                    ref = _fake_decoded(10 * r + b, n)
                    assert torch.equal(res["boxes"], ref["pred_boxes"]) and torch.equal(res["masks"][:, :24, :32], ref["pred_mask"])
                    assert res["sample_idx"].tolist() == ref["sample_idx"]


def _worker_overflow(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = pipeline.ResultExchange(cap=8, mask_hw=32, per_gather=2, device="cpu")
    done = []
    for b, n in enumerate(_OVERFLOW_COUNTS[rank]):
        done += [t.clone() for t in ex.add(_fake_decoded(10 * rank + b, n))]
    done += [t.clone() for t in ex.flush()]
    out = []
    for g, t in enumerate(done):
        for src in range(world):
            for slot in range(ex.per):
                rec = ex.batch_record(t, src, slot)
                # numpy: pickled by value (tensors travel as shared-memory handles the parent may open after this process is gone)
                out.append((g, src, slot, rec["boxes"].numpy().copy(), rec["sample_idx"].tolist(), rec["masks"].numpy().copy() if rec["masks"] is not None else None))
    q.put((rank, ex.n_gathers, ex.n_continuation_gathers, [t.per + (t.continuation.shape[1] if t.continuation is not None else 0) for t in done], out))
    dist.barrier()
    dist.destroy_process_group()


_OVERFLOW_COUNTS = {0: [3, 8, 2, 1], 1: [5, 19, 9, 0]}          # cap 8: rank 1 overflows in gather 0 (19 → 8 + 8 + 3) and in gather 1 (9 → 8 + 1)


def test_result_exchange_over_capacity_batch_is_split_not_raised():
    """A batch with more objects than the record capacity travels as continuation records: no rank raises (which would leave the others
    hanging in the gather), every rank — also the one that had nothing to continue — joins the extra gather the announced counts call for,
    and batch_record() re-assembles the batch bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overflow, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [_unship(q.get(timeout=120)) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_gathers, n_cont, widths, out in got:
        assert n_gathers == 2 and n_cont == 2 and widths == [2 + 2, 2 + 1]
        for g, src, slot, boxes, sidx, masks in out:
            b = 2 * g + slot
            n = _OVERFLOW_COUNTS[src][b]
            ref = _fake_decoded(10 * src + b, n)
            assert boxes.shape == (n, 4) and (boxes == ref["pred_boxes"].numpy()).all() and sidx == ref["sample_idx"]
            if n:
                assert (masks[:, :24, :32] == ref["pred_mask"].numpy()).all()


# ---------------------------------------------------------------------------------------------- configs[3]: 64 images per step over 8 ranks
_OVD_ITEMS, _OVD_BS, _OVD_CAP = 150, 8, 16                       # 150 images: 3 steps of 8 ranks x 8 images, the last one ragged (items 128..149 + 42 skips)


def _ovd_objects(item):
    """Deterministic per-image object list of the OVD stand-in: 3..11 objects (≈7 on average) → 24..88 per batch of 8 > cap 16."""
    g = torch.Generator().manual_seed(9000 + item)
    n = 3 + item % 9
    return torch.rand(n, 4, generator=g), torch.randn(n, 1, generator=g), torch.randn(n, 6, 6, generator=g)


def _ovd_decoded(start):
    """vl_decode stand-in for the batch of items start .. start+7 (items past the end of the dataset are skipped: utils.py:183-186)."""
    boxes, scores, masks, sidx = [], [], [], []
    for j in range(_OVD_BS):
        if start + j >= _OVD_ITEMS:
            continue
        b, s, m = _ovd_objects(start + j)
        boxes.append(b), scores.append(s), masks.append(m), sidx.extend([j] * b.shape[0])
    if not boxes:
        return {"pred_boxes": torch.zeros(0, 4), "pred_score": torch.zeros(0, 1), "pred_mask": torch.zeros(0, 8, 8), "sample_idx": [], "pred_mask_valid_hw": ()}
    n = len(sidx)
    return {"pred_boxes": torch.cat(boxes), "pred_score": torch.cat(scores), "pred_mask": torch.cat(masks), "sample_idx": sidx,
            "pred_mask_valid_hw": (torch.full((n,), 6), torch.full((n,), 6))}


def _worker_ovd(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    starts = pipeline.rank_batches(_OVD_ITEMS, _OVD_BS, rank, world)
    ex = pipeline.ResultExchange(cap=_OVD_CAP, mask_hw=8, per_gather=2, device="cpu")
    done = []
    for st in starts:
        done += [t.clone() for t in ex.add(_ovd_decoded(st))]
    done += [t.clone() for t in ex.flush()]
    # every rank re-assembles EVERY rank's batches from what the gathers delivered: item id → (boxes, scores, masks)
    seen = {}
    for g, t in enumerate(done):
        for src in range(world):
            src_starts = pipeline.rank_batches(_OVD_ITEMS, _OVD_BS, src, world)
            for slot in range(t.per):
                b = g * ex.per + slot
                if b >= len(src_starts):
                    continue
                rec = t.batch_record(src, slot)
                for j in sorted(set(rec["sample_idx"].tolist())):
                    sel = rec["sample_idx"] == j
                    item = src_starts[b] + j
                    assert item not in seen, ("delivered twice", item)
                    seen[item] = (rec["boxes"][sel].numpy().copy(), rec["scores"][sel].numpy().copy(), rec["masks"][sel][:, :6, :6].numpy().copy())
    q.put((rank, len(starts), ex.n_gathers, ex.n_continuation_gathers, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_ovd_step_of_64_images_is_delivered_exactly_once_across_8_ranks():
    """BASELINE configs[3] (OVD, batch 64 = 8 ranks x 8 images) through the reference's rank-strided walk (utils.py:181-182) and the exchange:
    every rank ends up with every image of a 150-image dataset exactly once, objects bit for bit, although every batch exceeds the record
    capacity (24..88 objects at cap 16 → continuation records) and the last step is ragged (some ranks only skip)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ovd, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = sorted(s + j for r in range(world) for s in pipeline.rank_batches(_OVD_ITEMS, _OVD_BS, r, world) for j in range(_OVD_BS))
    assert covered == list(range(192))                           # 3 steps x 64: every item index once, 150..191 are the "skip" slots
    for rank, n_b, n_g, n_cont, seen in got:
        assert n_b == 3 and n_g == 2 and n_cont == 2
        assert sorted(seen) == list(range(_OVD_ITEMS)), (rank, len(seen))
        for item, (bx, sc, mk) in seen.items():
            b, s, m = _ovd_objects(item)
            assert (bx == b.numpy()).all() and (sc == s.reshape(-1).numpy()).all() and (mk == m.numpy()).all()


def test_rank_affinity_plan_gives_disjoint_numa_local_core_sets():
    """pipeline.plan_rank_affinity: 8 ranks, GPUs 0-3 on NUMA node 0 and 4-7 on node 1 (the MI355X box's layout class) → every rank gets a
    disjoint quarter of its GPU's node; without topology an even slice; never an empty set."""
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    allowed = list(range(256))
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    sets = [pipeline.plan_rank_affinity(r, 8, numa, nodes, allowed) for r in range(8)]
    assert all(len(s) == 32 for s in sets) and len(set().union(*map(set, sets))) == 256
    for r, s in enumerate(sets):
        assert set(s) <= set(nodes[numa[r]])
    flat = [pipeline.plan_rank_affinity(r, 8, [-1] * 8, {}, list(range(20))) for r in range(8)]
    assert sorted(c for s in flat for c in s) == list(range(20)) and all(flat)
    tiny = [pipeline.plan_rank_affinity(r, 8, [-1] * 8, {}, [3, 5]) for r in range(8)]
    assert all(len(s) >= 1 and set(s) <= {3, 5} for s in tiny)
    assert pipeline._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
