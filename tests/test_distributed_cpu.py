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
    q.put((rank, [r["boxes"].clone() for r in res], [r["sample_idx"].clone() for r in res],
           [None if r["masks"] is None else r["masks"].clone() for r in res], [r["scores"].clone() for r in res], [r["valid_hw"].clone() for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_result_all_gather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
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
