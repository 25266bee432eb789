"""Multi-GPU host logic of the hot path: one process per GPU, contiguous shards of independent units (query
intervals: liftover/impl/halLiftover.cpp:46-92 clears all per-line state; reference columns:
api/impl/halColumnIterator.cpp:785-787), and the single exchange step that collates fixed-width output records
(SURVEY 8(e)): an all-gather of per-rank counts, then an all-gather of payloads padded to the largest shard.
torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist

RECORD_BYTES = 40  # sizeof(hgx_record), include/hgx.h


def shard_bounds(n, world, rank):
    """Rank r owns [r*n/world, (r+1)*n/world): rank-major concatenation of the shards is input order."""
    return (rank * n) // world, ((rank + 1) * n) // world


def all_gather_counts(n, device):
    """Every rank's record count, as a python list, on every rank."""
    world = dist.get_world_size()
    cnt = torch.tensor([n], dtype=torch.int64, device=device)
    counts = torch.empty(world, dtype=torch.int64, device=device)
    if device.type == "cuda":
        dist.all_gather_into_tensor(counts, cnt)
    else:  # gloo has no all_gather_into_tensor on every build
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, cnt)
        counts = torch.cat(parts)
    return [int(c) for c in counts.tolist()]


def all_gather_records(recs, trim=True):
    """recs: uint8 [n, 40] tensor of hgx_record (any device).  Returns (records, counts): with trim the rank-major
    concatenation [sum(counts), 40] on every rank; without it the padded [world * max(counts), 40] buffer, which
    avoids the device-side compaction when the caller only forwards the bytes."""
    world = dist.get_world_size()
    dev = recs.device
    counts = all_gather_counts(recs.shape[0], dev)
    mx = max(counts) if counts else 0
    mine = recs
    if recs.shape[0] != mx:
        mine = torch.zeros((mx, RECORD_BYTES), dtype=torch.uint8, device=dev)
        mine[:recs.shape[0]] = recs
    if dev.type == "cuda":
        allrec = torch.empty((world * mx, RECORD_BYTES), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(allrec, mine.contiguous())
        bufs = list(allrec.view(world, mx, RECORD_BYTES).unbind(0))
    else:
        bufs = [torch.empty((mx, RECORD_BYTES), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(bufs, mine.contiguous())
        allrec = None
    if not trim:
        return (allrec if allrec is not None else torch.cat(bufs, dim=0)), counts
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0), counts


def offset_query_index(recs, first_query):
    """Records carry the query index within the shard (hgx_record.query, first 8 bytes, little endian); add the shard's
    first global index so that the gathered stream reads as one batch."""
    if recs.shape[0]:
        q = recs[:, :8].contiguous().view(torch.int64)
        q += first_query
        recs[:, :8] = q.view(torch.uint8).view(-1, 8)
    return recs


PACKED_BYTES = 20


def can_pack(max_position, num_queries, num_target_sequences):
    """The 20-byte wire form holds 31-bit coordinates and query indices and a 16-bit sequence index."""
    return max_position < (1 << 31) and num_queries < (1 << 31) and num_target_sequences < (1 << 16)


def pack_records(recs):
    """hgx_record (40 bytes: four int64, an int32 and two bytes) -> 20 bytes (five int32): query, tgt_start, tgt_end,
    src_start, tgt_seq << 16 | strand << 8 | tgt_reversed.  Halves what the all-gatherv moves over xGMI; valid under
    can_pack().  uint8 [n, 40] in, uint8 [n, 20] out, same device."""
    n = recs.shape[0]
    r64 = recs.contiguous().view(torch.int64).view(n, 5)
    out = torch.empty((n, 5), dtype=torch.int32, device=recs.device)
    out[:, :4] = r64[:, :4]
    tail = r64[:, 4]
    out[:, 4] = ((tail & 0xFFFF) << 16) | (((tail >> 32) & 0xFF) << 8) | ((tail >> 40) & 0xFF)
    return out.view(torch.uint8).view(n, PACKED_BYTES)


def unpack_records(packed):
    """inverse of pack_records"""
    n = packed.shape[0]
    p = packed.contiguous().view(torch.int32).view(n, 5).to(torch.int64)
    out = torch.zeros((n, 5), dtype=torch.int64, device=packed.device)
    out[:, :4] = p[:, :4]
    w = p[:, 4] & 0xFFFFFFFF
    out[:, 4] = ((w >> 16) & 0xFFFF) | (((w >> 8) & 0xFF) << 32) | ((w & 0xFF) << 40)
    return out.view(torch.uint8).view(n, RECORD_BYTES)


class RecordCollator:
    """The all-gatherv of one batch overlapped with the mapping of the next: submit() exchanges the counts (a few bytes,
    synchronous) and starts the payload all-gather asynchronously on the communicator's own stream; wait() returns the
    previous batch.  At most one exchange is in flight, its buffers are kept alive here until it has completed."""

    def __init__(self):
        self._pending = None

    def submit(self, recs):
        world = dist.get_world_size()
        dev = recs.device
        width = recs.shape[1]  # 40 (hgx_record) or 20 (pack_records)
        counts = all_gather_counts(recs.shape[0], dev)
        mx = max(counts) if counts else 0
        mine = recs
        if recs.shape[0] != mx:
            mine = torch.zeros((mx, width), dtype=torch.uint8, device=dev)
            mine[:recs.shape[0]] = recs
        mine = mine.contiguous()
        if dev.type == "cuda":
            out = torch.empty((world * mx, width), dtype=torch.uint8, device=dev)
            work = dist.all_gather_into_tensor(out, mine, async_op=True)
            bufs = None
        else:
            bufs = [torch.empty((mx, width), dtype=torch.uint8) for _ in range(world)]
            work = dist.all_gather(bufs, mine, async_op=True)
            out = None
        self._pending = (work, out, bufs, mine, counts, mx, width)

    def wait(self, trim=True):
        """(records, counts) of the submitted batch, or None when nothing is in flight."""
        if self._pending is None:
            return None
        work, out, bufs, _mine, counts, mx, width = self._pending
        self._pending = None
        work.wait()
        if bufs is None:
            if not trim:
                return out, counts
            bufs = list(out.view(len(counts), mx, width).unbind(0))
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0), counts
