"""Multi-GPU host logic of the hot path: one process per GPU, contiguous shards of independent units (query
intervals: liftover/impl/halLiftover.cpp:46-92 clears all per-line state; reference columns:
api/impl/halColumnIterator.cpp:785-787), and the single exchange step that collates fixed-width output records
(SURVEY 8(e)): an all-gather of per-rank counts, then an all-gather of payloads padded to the largest shard; for writers of text
one gather to a root, or to several writers whose texts are placed by their sizes (SlotExchange(group=), text_placement).
torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import re
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


# ---- self-describing wire blobs (include/hgx.h: hgx_liftover_wire_blob) ------------------------------------------------
BLOB_HEADER = 32
_STRAND_CODES = (ord("+"), ord("-"), ord("."))


def _header(fmt, first_query, nq, nrec):
    import struct
    return torch.frombuffer(bytearray(struct.pack("<4sIqQQ", b"HGXW", fmt, first_query, nq, nrec)), dtype=torch.uint8)


def encode_blob(recs, num_queries, first_query=0, fmt=None):
    """Reference encoder (torch ops, any device) of what hgx_liftover_wire_blob writes: recs uint8 [n, 40] hgx_record
    rows grouped by query (query index relative to the shard), num_queries intervals in the batch.  fmt None picks 12
    when every field fits, else 20 (the caller knows whether 31 bits hold its coordinates; 40 = raw rows)."""
    n = recs.shape[0]
    dev = recs.device
    r64 = recs.contiguous().view(torch.int64).view(n, 5)
    q, ts, te, ss, tail = r64[:, 0], r64[:, 1], r64[:, 2], r64[:, 3], r64[:, 4]
    seq, strand, rev = tail & 0xFFFFFFFF, (tail >> 32) & 0xFF, (tail >> 40) & 0xFF
    counts = torch.bincount(q, minlength=num_queries) if n else torch.zeros(num_queries, dtype=torch.int64, device=dev)
    code = torch.full_like(strand, 3)
    for k, ch in enumerate(_STRAND_CODES):
        code = torch.where(strand == ch, torch.full_like(code, k), code)
    fits = bool(n == 0 or (((te - ts) >= 0) & ((te - ts) < (1 << 22)) & (ts >= 0) & (ts < (1 << 32)) & (ss >= 0) & (ss < (1 << 32)) &
                           (seq < 128) & (code < 3) & (rev <= 1)).all()) and bool(num_queries == 0 or (counts < 65536).all())
    if fmt is None:
        fmt = 12 if fits else 20
    if fmt == 8:  # (the 12-byte form without the source start: all a writer of BED lines needs)
        assert fits, "fields do not fit the 8-byte form"
        c16 = torch.zeros((2 * num_queries + 7) // 8 * 4, dtype=torch.int16, device=dev)
        c16[:num_queries] = counts.to(torch.int32).to(torch.int16)
        w = torch.empty((n, 2), dtype=torch.int64, device=dev)
        w[:, 0] = ts
        w[:, 1] = (te - ts) | (seq << 22) | (code << 29) | (rev << 31)
        body = torch.cat([c16.view(torch.uint8), (w & 0xFFFFFFFF).to(torch.int32).view(torch.uint8).view(-1) if n else
                          torch.empty(0, dtype=torch.uint8, device=dev)])
    elif fmt == 12:
        assert fits, "fields do not fit the 12-byte form"
        c16 = torch.zeros((2 * num_queries + 7) // 8 * 4, dtype=torch.int16, device=dev)
        c16[:num_queries] = counts.to(torch.int32).to(torch.int16)  # (uint16 bit pattern)
        w = torch.empty((n, 3), dtype=torch.int64, device=dev)
        w[:, 0], w[:, 1] = ts, ss
        w[:, 2] = (te - ts) | (seq << 22) | (code << 29) | (rev << 31)
        body = torch.cat([c16.view(torch.uint8), (w & 0xFFFFFFFF).to(torch.int32).view(torch.uint8).view(-1) if n else
                          torch.empty(0, dtype=torch.uint8, device=dev)])
    elif fmt == 20:
        body = pack_records(recs).view(-1)
    else:
        body = recs.contiguous().view(-1)
    return torch.cat([_header(fmt, first_query, num_queries, n).to(dev), body])


def decode_blob(blob):
    """blob (uint8, any device) -> (records uint8 [n, 40] with GLOBAL query indices, first_query, num_queries)."""
    import struct
    magic, fmt, first_query, nq, nrec = struct.unpack("<4sIqQQ", bytes(blob[:BLOB_HEADER].cpu().numpy()))
    if magic != b"HGXW":
        raise ValueError("not a wire blob")
    dev = blob.device
    body = blob[BLOB_HEADER:]
    if fmt == 20:
        recs = unpack_records(body[:PACKED_BYTES * nrec].view(nrec, PACKED_BYTES))
        return offset_query_index(recs, first_query), first_query, nq
    if fmt == 40:
        recs = body[:RECORD_BYTES * nrec].view(nrec, RECORD_BYTES).clone()
        return offset_query_index(recs, first_query), first_query, nq
    if fmt not in (8, 12):
        raise ValueError("unknown wire format %d" % fmt)
    cbytes = (2 * nq + 7) // 8 * 8
    counts = body[:2 * nq].contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    words = fmt // 4
    w = (body[cbytes:cbytes + fmt * nrec].contiguous().view(torch.int32).view(nrec, words).to(torch.int64)) & 0xFFFFFFFF
    if fmt == 8:  # (no source start on the wire: -1 in the record)
        w = torch.stack([w[:, 0], torch.full_like(w[:, 0], -1), w[:, 1]], dim=1)
    out = torch.zeros((nrec, 5), dtype=torch.int64, device=dev)
    out[:, 0] = torch.repeat_interleave(torch.arange(nq, dtype=torch.int64, device=dev), counts) + first_query
    out[:, 1] = w[:, 0]
    out[:, 2] = w[:, 0] + (w[:, 2] & ((1 << 22) - 1))
    out[:, 3] = w[:, 1]
    code = (w[:, 2] >> 29) & 3
    strand = torch.tensor(_STRAND_CODES, dtype=torch.int64, device=dev)[code.clamp(max=2)]
    out[:, 4] = ((w[:, 2] >> 22) & 127) | (strand << 32) | (((w[:, 2] >> 31) & 1) << 40)
    return out.view(torch.uint8).view(nrec, RECORD_BYTES), first_query, nq


def split_blobs(gathered, counts):
    """The padded all-gather buffer of RecordCollator.wait(trim=False) -> the ranks' blobs."""
    world = len(counts)
    mx = gathered.numel() // world if world else 0
    return [gathered.reshape(world, mx)[r, :counts[r]] for r in range(world)]


class RecordCollator:
    """The all-gatherv of a stream of batches, overlapped with the mapping of the following ones and free of host waits in
    the steady state.  submit(batch k) starts the exchange of the ranks' sizes for batch k and the payload all-gather of
    batch k-1, whose sizes have arrived in the meantime; wait() hands out the oldest batch whose payload is in flight
    (batch k-2 in a wait(); submit() loop) after making the current stream wait for it.  Buffers are kept alive here until
    their exchange has completed.  drain() completes everything that was submitted."""

    def __init__(self):
        self._staged = None   # sizes under way, payload not yet started
        self._inflight = []   # payload all-gathers, oldest first
        self._slots = None    # reusable (size, sizes, pinned host copy, event) sets: allocating them per batch costs more
        self._turn = 0        # host time than the collective itself

    # -- stage A: the sizes --------------------------------------------------------------------------------------------
    def submit(self, recs, backing=None):
        """recs: uint8 [n, width] rows or a 1-D blob.  backing: the allocation recs is a prefix of (wire_blob's), sent
        without a padding copy when it is long enough for the largest rank."""
        world = dist.get_world_size()
        dev = recs.device
        if recs.dim() == 1:  # a wire blob: rows of one byte
            if backing is None and recs._base is not None and recs._base.dim() == 1 and recs._base.dtype == torch.uint8:
                backing = recs._base  # LiftoverPlan.wire_blob returns a prefix of a larger allocation
            recs = recs.view(-1, 1)
        width = recs.shape[1]  # 40 (hgx_record), 20 (pack_records) or 1 (blob bytes)
        stage = {"recs": recs, "backing": backing, "width": width, "world": world}
        if dev.type == "cuda":
            if self._slots is None:  # three sets: one staged, one being read by _launch, one free
                self._slots = [(torch.zeros(1, dtype=torch.int64, device=dev), torch.empty(world, dtype=torch.int64, device=dev),
                                torch.empty(world, dtype=torch.int64, pin_memory=True), torch.cuda.Event()) for _ in range(3)]
            size, sizes, host, ev = self._slots[self._turn % 3]
            self._turn += 1
            size.fill_(recs.shape[0])
            dist.all_gather_into_tensor(sizes, size)  # stream ordered; the host does not wait
            host.copy_(sizes, non_blocking=True)
            ev.record()
            stage.update(host=host, event=ev)
        else:
            stage["counts"] = all_gather_counts(recs.shape[0], dev)
        previous, self._staged = self._staged, stage
        if previous is not None:
            self._launch(previous)

    # -- stage B: the payload ------------------------------------------------------------------------------------------
    def _launch(self, stage):
        recs, width, world = stage["recs"], stage["width"], stage["world"]
        dev = recs.device
        if "counts" not in stage:
            stage["event"].synchronize()  # recorded one batch ago
            stage["counts"] = [int(c) for c in stage["host"].tolist()]
        counts = stage["counts"]
        mx = max(counts) if counts else 0
        backing = stage["backing"]
        if recs.shape[0] == mx:
            mine = recs
        elif backing is not None and width == 1 and backing.numel() >= mx and backing.data_ptr() == recs.data_ptr():
            mine = backing[:mx].view(-1, 1)  # bytes after the blob are padding nobody reads
        else:
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
        self._inflight.append((work, out, bufs, (mine, recs, backing), counts, mx, width))

    def wait(self, trim=True, flush=False):
        """(records, counts) of the oldest batch whose payload exchange is under way, or None when there is none.
        flush: also start (and complete) the exchange of a batch whose sizes only have been exchanged so far — the last
        batch of a stream, or the only one."""
        if not self._inflight:
            if self._staged is None or not flush:
                return None
            staged, self._staged = self._staged, None
            self._launch(staged)
        work, out, bufs, _keep, counts, mx, width = self._inflight.pop(0)
        work.wait()
        if not trim:
            return (out if bufs is None else torch.cat(bufs, dim=0)), counts
        if bufs is None:
            bufs = list(out.view(len(counts), mx, width).unbind(0))
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0), counts

    def drain(self, trim=True):
        """complete every submitted exchange; returns the batches in submission order"""
        done = []
        while True:
            r = self.wait(trim=trim, flush=True)
            if r is None:
                return done
            done.append(r)


class SlotExchange:
    """ONE collective per batch.  Every rank owns a slot of `slot_bytes` in a buffer of world_size equal slots; its records
    travel as a self-describing wire blob (hgx_liftover_wire_blob: the header carries the sizes, so no exchange of sizes comes
    first) and one all-gather fills in the other slots.  backend "c_abi": hgx_liftover_exchange — RCCL called by the library
    itself on the current stream; "torch": torch.distributed's all-gather of the same slots (the launcher's own communicator;
    gloo on CPU for the tests).  submit() starts the exchange of the plan's last run, wait() hands out the oldest buffer under
    way (three rotate, so the mapping of the next batches overlaps the exchange); slots(buf) views the ranks' blobs."""

    def __init__(self, world, rank, slot_bytes, device, backend="torch", comm=None, root=None, bed_only=False, group=None):
        """root: None — every rank gets every rank's blob (one all-gather); a rank — only that rank does (the writer's collation:
        a rank sends its slot once instead of receiving all the others'; hgx_liftover_gather / torch.distributed.gather), the
        other ranks' buffers keep their own slot only.  group: several writers instead of one root — the ranks in groups of
        `group` consecutive ranks, the first of a group receives the group's blobs (hgx_liftover_gather_writers / grouped
        isend / irecv; buffers of `group` slots, rank r's at r - writer); what the writers do with them: text_placement below.
        bed_only: blobs in the 8-byte form (no source coordinates)."""
        self.world, self.rank, self.slot = world, rank, (int(slot_bytes) + 7) // 8 * 8
        self.backend, self.comm = backend, comm
        self.root, self.bed_only, self.group = root, bed_only, group
        if group is not None:
            if root is not None or group < 1:
                raise ValueError("either one root or groups of at least one rank")
            self.writer = writer_of(rank, group)
            self.members = list(range(self.writer, min(self.writer + group, world)))
        self._bufs = [torch.zeros((world if group is None else group) * self.slot, dtype=torch.uint8, device=device) for _ in range(3)]
        self._turn, self._inflight = 0, []
        self.last_bytes, self.last_format = 0, None

    def _mine(self, buf):
        at = self.rank if self.group is None else 0  # (grouped: a writer is its group's first rank, the others keep one slot)
        return buf[at * self.slot:(at + 1) * self.slot]

    def submit(self, plan=None, first_query=0, blob=None):
        """plan: a LiftoverPlan whose last run is exchanged; blob: a ready blob instead (CPU tests)"""
        if len(self._inflight) >= 3:
            # (the buffer about to be reused is still the target — or the unread result — of an exchange under way: completing it
            # here would throw its records away)
            raise RuntimeError("three exchanges in flight: wait() before the next submit")
        buf = self._bufs[self._turn % 3]
        self._turn += 1
        work = None
        if self.backend == "c_abi":
            if self.group is not None:
                self.last_bytes = plan.gather_writers(self.comm, self.group, first_query, buf, self.slot, bed_only=self.bed_only)
            elif self.root is None:
                self.last_bytes = plan.exchange(self.comm, first_query, buf, self.slot)
            else:
                self.last_bytes = plan.gather(self.comm, self.root, first_query, buf, self.slot, bed_only=self.bed_only)
            if buf.device.type == "cuda":  # (the library queues its all-gather on the current stream: an event marks its end)
                work = torch.cuda.Event()
                work.record()
        else:
            mine = self._mine(buf)
            if blob is not None:
                if blob.numel() > self.slot:
                    raise ValueError("blob of %d bytes does not fit the slot of %d" % (blob.numel(), self.slot))
                mine[:blob.numel()] = blob
                self.last_bytes = int(blob.numel())
            else:
                if plan.wire_capacity() > self.slot:
                    raise ValueError("this rank's records may need %d bytes, the slot has %d" % (plan.wire_capacity(), self.slot))
                b, self.last_format = plan.wire_blob(first_query, dst=mine, bed_only=self.bed_only)
                self.last_bytes = int(b.numel())
            if self.group is not None:  # to the group's writer: the sends and receives of one batch posted together
                if self.rank == self.writer:
                    ops = [dist.P2POp(dist.irecv, buf[(r - self.writer) * self.slot:(r - self.writer + 1) * self.slot], r)
                           for r in self.members[1:]]
                else:
                    ops = [dist.P2POp(dist.isend, mine.clone(), self.writer)]
                work = dist.batch_isend_irecv(ops) if ops else None
            elif self.root is not None:  # to the writer only
                parts = list(buf.view(self.world, self.slot).unbind(0)) if self.rank == self.root else None
                work = dist.gather(mine.clone(), gather_list=parts, dst=self.root, async_op=True)
            elif buf.device.type == "cuda":
                work = dist.all_gather_into_tensor(buf, mine, async_op=True)
            else:
                parts = list(buf.view(self.world, self.slot).unbind(0))
                work = dist.all_gather(parts, mine.clone(), async_op=True)
        self._inflight.append((work, buf))

    @property
    def in_flight(self):
        """exchanges submitted and not waited for (at most three: a fourth submit without a wait() is refused)"""
        return len(self._inflight)

    def wait(self):
        if not self._inflight:
            return None
        work, buf = self._inflight.pop(0)
        if work is not None:
            if isinstance(work, torch.cuda.Event):  # (c_abi): readers on other streams must see the gathered slots
                work.synchronize()
            elif isinstance(work, list):
                for w in work:
                    w.wait()
            else:
                work.wait()
        return buf

    def drain(self):
        out = []
        while self._inflight:
            out.append(self.wait())
        return out

    def slots(self, buf):
        """the ranks' blobs (each trimmed to what its header says it holds), in rank order (with a root: on the root only)"""
        if self.root is not None and self.rank != self.root:
            raise ValueError("the blobs were gathered on rank %d" % self.root)
        if self.group is not None and self.rank != self.writer:
            raise ValueError("the group's blobs were gathered on rank %d" % self.writer)
        out = []
        for r in range(self.world if self.group is None else len(self.members)):
            s = buf[r * self.slot:(r + 1) * self.slot]
            out.append(s[:blob_bytes(s)])
        return out


def writer_of(rank, group):
    """the first rank of `rank`'s group of `group` consecutive ranks: the group's writer"""
    return rank - rank % group


def text_placement(nbytes, device):
    """Level two of the collation by several writers, and it moves no records: every rank tells the bytes of text it holds (0 on a
    rank that is not a writer), and (offset, total) say where this rank's text belongs in the one output — the groups in rank
    order are the input's order (shard_bounds), so the writers' texts one behind the other are the single writer's text."""
    sizes = all_gather_counts(int(nbytes), device)
    rank = dist.get_rank()
    return sum(sizes[:rank]), sum(sizes)


def write_text_at(path, offset, text, total=None):
    """a writer's share of the output file, written where it belongs (the writers write side by side; the file ends at `total`)"""
    import os
    fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
    try:
        if total is not None and os.fstat(fd).st_size != total:
            os.ftruncate(fd, total)  # (never below what any writer writes: every share ends at or before `total`)
        done = 0
        view = memoryview(text)
        while done < len(view):
            done += os.pwrite(fd, view[done:], offset + done)
    finally:
        os.close(fd)


def line_shares(data, world):
    """`world` contiguous byte ranges of a text, cut behind line ends, in order and covering it: rank r's share of the lines (a line
    belongs to the share its first byte falls into; the shares' outputs one behind the other are the whole text's output, since a
    line's lifting knows nothing of the other lines: liftover/impl/halLiftover.cpp:46-92)."""
    n = len(data)
    cuts = [0]
    for r in range(1, world):
        at = max(cuts[-1], (n * r) // world)
        if at > 0 and at < n and data[at - 1:at] != b"\n":
            nl = data.find(b"\n", at)
            at = n if nl < 0 else nl + 1
        cuts.append(at)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def convert_sharded(convert, data, out_path, device=None):
    """One process per GPU, every rank a writer (groups of one: no records move): rank r converts its share of the input's lines
    (`convert`: bytes -> bytes; hal_amd.liftover_convert on the rank's own handle), an all-gather of the texts' sizes places them
    (text_placement), the ranks write side by side into `out_path`.  A rank whose share holds a malformed line behaves as the one
    process does (halLiftoverMain.cpp:143-149: what was lifted before the line is written, then the error): the ranks before it write
    everything, it writes what it had (the exception's partial_output), the ranks behind it nothing, and every rank raises.
    Returns the output's size."""
    device = device or torch.device("cpu")
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = line_shares(data, world)[rank]
    failure = None
    try:
        text = convert(data[lo:hi])
    except Exception as e:  # (the collectives below are still taken part in: the other ranks are waiting in theirs)
        failure = e
        text = getattr(e, "partial_output", b"") or b""
    if isinstance(text, str):
        text = text.encode()
    failed = all_gather_counts(1 if failure is not None else 0, device)
    first_bad = failed.index(1) if 1 in failed else world
    if rank > first_bad:
        text = b""
    offset, total = text_placement(len(text), device)
    write_text_at(out_path, offset, text, total)
    dist.barrier()
    if failure is not None and rank == first_bad:
        raise renumbered(failure, data[:lo])
    if first_bad < world:
        raise PeerFailed("rank %d met a malformed line: the output ends with what was lifted before it" % first_bad)
    return total


class PeerFailed(RuntimeError):
    """another rank's share held the malformed line: that rank reports it (the one process prints one message)"""


_BED_LINE_SUFFIX = re.compile(r"( in input bed line )(\d+)$")


def renumbered(failure, before):
    """The library numbers the lines of the text it was given; a rank's share begins inside the file.  The reference's scanner counts
    the lines it reads, and skips white space between them (liftover/impl/halBedScanner.cpp:47-59): the share's number plus the
    lines of `before` (the input in front of the share) that are not blank."""
    msg = str(failure)
    m = _BED_LINE_SUFFIX.search(msg)
    if not m or not before:
        return failure
    ahead = sum(1 for ln in before.split(b"\n") if ln.strip())
    if ahead == 0:
        return failure
    out = type(failure)(msg[:m.start()] + m.group(1) + str(int(m.group(2)) + ahead))
    for k, v in getattr(failure, "__dict__", {}).items():
        setattr(out, k, v)
    return out


def blob_bytes(slot):
    """length of the blob at the start of a slot, from its 32-byte header {"HGXW", u32 format, i64 first_query, u64 n_queries, u64 n_records}"""
    h = slot[:32].cpu().numpy().tobytes()
    if h[:4] != b"HGXW":
        raise ValueError("not a wire blob")
    fmt = int.from_bytes(h[4:8], "little")
    nq, nrec = int.from_bytes(h[16:24], "little"), int.from_bytes(h[24:32], "little")
    if fmt == 0:  # (hgx_liftover_exchange: the rank took part in the collective without a blob)
        if nq == 0:
            raise ValueError("a rank's records did not fit its slot (it needed %d bytes)" % nrec)
        raise ValueError("a rank had no blob for this batch (its own call reports the cause: a batch in flight, a HIP error)")
    if fmt in (8, 12):
        return 32 + (2 * nq + 7) // 8 * 8 + fmt * nrec
    return 32 + (20 if fmt == 20 else 40) * nrec
