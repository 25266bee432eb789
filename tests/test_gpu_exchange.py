"""The exchange step of the multi-GPU path through the C ABI: an RCCL communicator made by the library (hgx_comm_create, here
of one rank — the test box has one GPU), hgx_liftover_exchange = one in-place all-gather of self-describing slots, and the
decoded slots against the records of the direct run.  Also hal_amd.shard.SlotExchange over torch.distributed (the same slots,
the launcher's communicator)."""
import numpy as np
import pytest

from test_gpu_liftover import _rand_alignment

pytestmark = pytest.mark.gpu


def _batch(hal, al, n, seed):
    import torch
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, ss, length = al.sequences(src)[0]
    g = torch.Generator().manual_seed(seed)
    starts = torch.randint(0, length - 400, (n,), generator=g)
    lens = torch.randint(1, 400, (n,), generator=g)
    strand = torch.tensor([ord("+-."[i % 3]) for i in range(n)], dtype=torch.uint8)
    return src, tgt, (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()


def test_exchange_on_a_one_rank_rccl_group(hal, tmp_path):
    import torch
    from hal_amd import shard
    al, _ = _rand_alignment(hal, tmp_path, 2)
    src, tgt, gs, ge, st = _batch(hal, al, 3000, 5)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=3000)
    comm = hal.Comm(hal.Comm.unique_id(), 0, 1, 0)
    ptr, nrec = plan.run(gs, ge, st)
    direct = plan.records_to_tensor(ptr, nrec).clone()
    pristine = direct.clone()
    slot = (plan.wire_capacity() + 7) // 8 * 8
    ex = shard.SlotExchange(1, 0, slot, "cuda", backend="c_abi", comm=comm)
    for first_query in (0, 1 << 33):  # a shard's global offset travels in the header
        ex.submit(plan, first_query=first_query)
        buf = ex.wait()
        torch.cuda.synchronize()
        (blob,) = ex.slots(buf)
        assert blob.numel() == ex.last_bytes
        recs, fq, nq = shard.decode_blob(blob)
        assert (fq, nq) == (first_query, 3000)
        want = shard.offset_query_index(direct, first_query)
        assert torch.equal(recs.cpu(), want.cpu())
    # the writer's collation (hgx_liftover_gather: send / recv to the root; here the root is the only rank) in the 8-byte form:
    # the device writes the bytes the torch encoder writes, and they decode to the records without their source start
    gx = shard.SlotExchange(1, 0, slot, "cuda", backend="c_abi", comm=comm, root=0, bed_only=True)
    gx.submit(plan, first_query=12345)
    buf = gx.wait()
    torch.cuda.synchronize()
    (blob8,) = gx.slots(buf)
    assert blob8.numel() == gx.last_bytes == 32 + (2 * 3000 + 7) // 8 * 8 + 8 * nrec
    assert torch.equal(blob8.cpu(), shard.encode_blob(pristine.cpu(), 3000, first_query=12345, fmt=8))
    recs8, fq, nq = shard.decode_blob(blob8)
    want8 = shard.offset_query_index(pristine.clone(), 12345).cpu().contiguous().view(torch.int64).view(-1, 5)
    want8[:, 3] = -1
    assert (fq, nq) == (12345, 3000) and torch.equal(recs8.cpu().contiguous().view(torch.int64).view(-1, 5), want8)
    b8, fmt8 = plan.wire_blob(first_query=0, bed_only=True)
    assert fmt8 == 8 and b8.numel() == blob8.numel()
    # a slot that is too small: the collective is still carried out, the call reports it, the slot says so to the others
    small = shard.SlotExchange(1, 0, 64, "cuda", backend="c_abi", comm=comm)
    with pytest.raises(hal.HgxError, match="need"):
        small.submit(plan, first_query=0)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="did not fit"):
        small.slots(small._bufs[0])
    comm.close()


def test_slot_exchange_over_torch_distributed(hal, tmp_path):
    import os
    import torch
    import torch.distributed as dist
    from hal_amd import shard
    al, _ = _rand_alignment(hal, tmp_path, 2)
    src, tgt, gs, ge, st = _batch(hal, al, 2000, 6)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=2000)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ptr, nrec = plan.run(gs, ge, st)
        direct = plan.records_to_tensor(ptr, nrec).clone()
        ex = shard.SlotExchange(1, 0, plan.wire_capacity(), "cuda", backend="torch")
        ex.submit(plan, first_query=77)
        buf = ex.wait()
        torch.cuda.synchronize()
        recs, fq, nq = shard.decode_blob(ex.slots(buf)[0])
        assert (fq, nq) == (77, 2000) and torch.equal(recs.cpu(), shard.offset_query_index(direct, 77).cpu())
        # gather to the writer rank over torch.distributed (dist.gather)
        gx = shard.SlotExchange(1, 0, plan.wire_capacity(), "cuda", backend="torch", root=0, bed_only=True)
        gx.submit(plan, first_query=5)
        buf = gx.wait()
        torch.cuda.synchronize()
        recs8, fq, nq = shard.decode_blob(gx.slots(buf)[0])
        assert (fq, nq, gx.last_format) == (5, 2000, 8) and recs8.shape == recs.shape
    finally:
        dist.destroy_process_group()
