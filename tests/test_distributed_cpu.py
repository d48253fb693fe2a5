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
