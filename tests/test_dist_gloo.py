"""World-size-2 CPU test (gloo) of the multi-GPU host logic: contiguous query sharding and the all-gatherv of
fixed-width records (counts, then padded payloads) that bench.py and a multi-GPU driver use.  No GPU: the
per-rank "lift" is replaced by a deterministic stand-in producing a data-dependent number of records."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


from hal_amd.shard import (RecordCollator, all_gather_records, can_pack, decode_blob, encode_blob, offset_query_index, pack_records,
                           shard_bounds, split_blobs, unpack_records)


def fake_lift(q_lo, q_hi, base=0):
    """query i yields (i % 4) records whose bytes encode (i - base, k): base = q_lo imitates a rank that numbers
    its queries from 0, as the plan does."""
    out = []
    for i in range(q_lo, q_hi):
        for k in range(i % 4):
            r = np.zeros(40, dtype=np.uint8)
            r[:8] = np.frombuffer(np.int64(i - base).tobytes(), dtype=np.uint8)
            r[8] = k
            out.append(r)
    return torch.from_numpy(np.stack(out)) if out else torch.zeros((0, 40), dtype=torch.uint8)


def _real_records(n, seed):
    """n well-formed hgx_record rows (RECORD_DTYPE layout) with values near the limits of the packed form"""
    rng = np.random.default_rng(seed)
    dt = np.dtype([("query", "<i8"), ("tgt_start", "<i8"), ("tgt_end", "<i8"), ("src_start", "<i8"), ("tgt_seq", "<i4"), ("strand", "S1"),
                   ("tgt_reversed", "u1"), ("_pad", "S2")])
    r = np.zeros(n, dtype=dt)
    r["query"] = rng.integers(0, 2 ** 31 - 1, n)
    r["tgt_start"] = rng.integers(0, 2 ** 31 - 1, n)
    r["tgt_end"] = rng.integers(0, 2 ** 31 - 1, n)
    r["src_start"] = rng.integers(0, 2 ** 31 - 1, n)
    r["tgt_seq"] = rng.integers(0, 2 ** 16 - 1, n)
    r["strand"] = rng.choice([b"+", b"-", b"."], n)
    r["tgt_reversed"] = rng.integers(0, 2, n)
    return torch.from_numpy(r.view(np.uint8).reshape(n, 40).copy())


def _batch_records(nq, seed, wide=False):
    """records of one batch as a plan returns them: grouped by interval, shard-relative query index, lengths and sequence
    indices inside the 12-byte form unless wide"""
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 6, nq)
    counts[rng.integers(0, nq)] = 300  # one interval with many records
    n = int(counts.sum())
    dt = np.dtype([("query", "<i8"), ("tgt_start", "<i8"), ("tgt_end", "<i8"), ("src_start", "<i8"), ("tgt_seq", "<i4"), ("strand", "S1"),
                   ("tgt_reversed", "u1"), ("_pad", "S2")])
    r = np.zeros(n, dtype=dt)
    r["query"] = np.repeat(np.arange(nq), counts)
    top = 2 ** 31 - 2 ** 23 if not wide else 2 ** 40
    r["tgt_start"] = rng.integers(0, top, n)
    r["tgt_end"] = r["tgt_start"] + rng.integers(1, 2 ** 22 if not wide else 2 ** 24, n)
    r["tgt_end"][0] = r["tgt_start"][0] + (2 ** 22 - 1 if not wide else 2 ** 22)  # the longest length that fits / the first that does not
    r["src_start"] = rng.integers(0, top, n)
    r["tgt_seq"] = rng.integers(0, 128, n)
    r["strand"] = rng.choice([b"+", b"-", b"."], n)
    r["tgt_reversed"] = rng.integers(0, 2, n)
    return torch.from_numpy(r.view(np.uint8).reshape(n, 40).copy()), n


def test_wire_blob_roundtrip_and_format_choice():
    recs, n = _batch_records(500, 3)
    for fmt, per_record in ((None, 12), (12, 12), (20, 20), (40, 40)):
        blob = encode_blob(recs, 500, first_query=12345, fmt=fmt)
        assert blob.numel() == 32 + (1000 if per_record == 12 else 0) + per_record * n
        out, first, nq = decode_blob(blob)
        assert (first, nq) == (12345, 500) and torch.equal(out, offset_query_index(recs.clone(), 12345))
    wide, n = _batch_records(100, 4, wide=True)
    blob = encode_blob(wide, 100, first_query=7)  # a length of 2^22 does not fit: the encoder falls back
    assert blob[4] == 20
    raw = encode_blob(wide, 100, first_query=7, fmt=40)  # (coordinates beyond 31 bits: only the raw form is exact)
    assert torch.equal(decode_blob(raw)[0], offset_query_index(wide.clone(), 7))
    empty = encode_blob(torch.zeros((0, 40), dtype=torch.uint8), 0)
    assert empty.numel() == 32 and decode_blob(empty)[0].shape == (0, 40)
    none = encode_blob(torch.zeros((0, 40), dtype=torch.uint8), 9, first_query=3)  # intervals without records
    assert decode_blob(none)[0].shape == (0, 40) and decode_blob(none)[1:] == (3, 9)


def test_pack_roundtrip_and_limits():
    r = _real_records(1000, 7)
    assert torch.equal(unpack_records(pack_records(r)), r)
    assert pack_records(r).shape == (1000, 20)
    assert can_pack(2 ** 31 - 1, 10 ** 6, 100) and not can_pack(2 ** 31, 10, 1) and not can_pack(10, 2 ** 31, 1) and not can_pack(10, 10, 2 ** 16)


def _worker(rank, world, port, n, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(n, world, rank)
    gathered, counts = all_gather_records(offset_query_index(fake_lift(lo, hi, base=lo), lo))
    want = fake_lift(0, n)
    ok = gathered.shape == want.shape and bool(torch.equal(gathered, want)) and sum(counts) == want.shape[0]
    # the overlapped form: two batches in flight one after the other
    col = RecordCollator()
    assert col.wait() is None
    col.submit(offset_query_index(fake_lift(lo, hi, base=lo), lo))
    first = col.wait(flush=True)
    col.submit(fake_lift(0, 3 + rank))
    second = col.wait(flush=True)
    ok = ok and bool(torch.equal(first[0], want)) and first[1] == counts
    ok = ok and second[1] == [fake_lift(0, 3 + r).shape[0] for r in range(world)]
    ok = ok and bool(torch.equal(second[0], torch.cat([fake_lift(0, 3 + r) for r in range(world)], dim=0)))
    # the 20-byte wire form through the same exchange
    mine = _real_records(200 + 50 * rank, rank)
    col.submit(pack_records(mine))
    packed, pcounts = col.wait(flush=True)
    back = unpack_records(packed)
    expect = torch.cat([_real_records(200 + 50 * r, r) for r in range(world)], dim=0)
    ok = ok and packed.shape[1] == 20 and pcounts == [200 + 50 * r for r in range(world)] and bool(torch.equal(back, expect))
    # self-describing blobs of different formats and sizes through the same exchange
    nq = 300 + 20 * rank
    mine, _ = _batch_records(nq, 10 + rank)
    col.submit(encode_blob(mine, nq, first_query=1000 * rank, fmt=12 if rank == 0 else 20))
    gathered, sizes = col.wait(trim=False, flush=True)
    decoded = torch.cat([decode_blob(b)[0] for b in split_blobs(gathered, sizes)], dim=0)
    expect = torch.cat([offset_query_index(_batch_records(300 + 20 * r, 10 + r)[0], 1000 * r) for r in range(world)], dim=0)
    ok = ok and bool(torch.equal(decoded, expect))
    # a stream of batches the way bench.py drives it: wait(); submit() per step, drain() at the end — the payload of a
    # batch starts one submit later, nothing is lost or reordered, and a blob that is a prefix of a larger allocation
    # travels without a padding copy
    stream = RecordCollator()
    got = []
    for k in range(5):
        r = stream.wait(trim=False)
        if r is not None:
            got.append(r)
        nqk = 40 + 7 * k + 3 * rank
        blob = encode_blob(_batch_records(nqk, 100 + 10 * k + rank)[0], nqk, first_query=10000 * k + 100 * rank)
        room = torch.zeros(blob.numel() * 3, dtype=torch.uint8)
        room[:blob.numel()] = blob
        stream.submit(room[:blob.numel()])
    assert len(got) == 3  # (batch k-2 comes out in step k)
    got += stream.drain(trim=False)
    ok = ok and len(got) == 5 and stream.wait() is None
    for k, (gathered, sizes) in enumerate(got):
        decoded = torch.cat([decode_blob(b)[0] for b in split_blobs(gathered, sizes)], dim=0)
        expect = torch.cat([offset_query_index(_batch_records(40 + 7 * k + 3 * r, 100 + 10 * k + r)[0], 10000 * k + 100 * r)
                            for r in range(world)], dim=0)
        ok = ok and bool(torch.equal(decoded, expect))
    result[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allgatherv_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(2, port, 1001, result), nprocs=2, join=True)
    assert result[0] and result[1]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 1000003):
        for world in (1, 2, 4, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _slot_worker(rank, world, port, result):
    """The one-collective exchange (hal_amd.shard.SlotExchange) on records of a real lift: tests/golden/lifted_records.npz holds
    the hgx_record rows a GPU produced for 600 intervals (tests/golden/make_lifted_records.py), the rows of two uneven shards
    of them as two ranks would lift them, and the wire blobs the DEVICE wrote for those shards."""
    from hal_amd.shard import SlotExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLD, "lifted_records.npz"))
    bounds = [int(b) for b in z["bounds"]]
    mine = torch.from_numpy(z["shard%d" % rank].copy())
    nq = bounds[rank + 1] - bounds[rank]
    blob = encode_blob(mine, nq, first_query=bounds[rank])
    # the torch encoder writes the bytes the device wrote (so what travels here is what travels between GPUs)
    ok = bool(torch.equal(blob, torch.from_numpy(z["blob%d" % rank].copy()))) and int(z["formats"][rank]) == 12
    slot = 16384
    ex = SlotExchange(world, rank, slot, "cpu", backend="torch")
    assert ex.wait() is None
    whole = torch.from_numpy(z["whole"].copy())

    def check(buf):
        parts = ex.slots(buf)
        good = [p.numel() for p in parts] == [int(z["blob0"].shape[0]), int(z["blob1"].shape[0])]
        decoded = torch.cat([decode_blob(p)[0] for p in parts], dim=0)
        return good and bool(torch.equal(decoded, whole))  # rank-major concatenation of the shards = the unsharded lift

    seen = 0
    for _ in range(7):  # a stream of batches: three buffers rotate, a buffer is read before the exchange after next reuses it
        if ex.in_flight == 3:
            ok = ok and check(ex.wait())
            seen += 1
        ex.submit(blob=blob)
    for buf in ex.drain():
        ok = ok and check(buf)
        seen += 1
    ok = ok and seen == 7
    for _ in range(3):
        ex.submit(blob=blob)
    try:  # (a fourth submit without a wait would write over a buffer nobody has read: refused, nothing is lost)
        ex.submit(blob=blob)
        ok = False
    except RuntimeError:
        pass
    done = ex.drain()
    ok = ok and len(done) == 3 and all(check(b) for b in done)
    try:
        SlotExchange(world, rank, 64, "cpu", backend="torch").submit(blob=blob)
        ok = False
    except ValueError:
        pass
    # the writer's collation: every rank's blob on rank 0 only (torch.distributed.gather; hgx_liftover_gather from the library),
    # in the 8-byte form — what a writer of BED lines needs: everything but the source start
    blob8 = encode_blob(mine, nq, first_query=bounds[rank], fmt=8)
    gx = SlotExchange(world, rank, slot, "cpu", backend="torch", root=0, bed_only=True)
    want8 = whole.view(torch.int64).view(-1, 5).clone()
    want8[:, 3] = -1
    for _ in range(4):
        gx.submit(blob=blob8)
        buf = gx.wait()
        if rank == 0:
            parts = gx.slots(buf)
            ok = ok and [p.numel() for p in parts] == [32 + (2 * (bounds[r + 1] - bounds[r]) + 7) // 8 * 8 + 8 * int(z["shard%d" % r].shape[0])
                                                       for r in range(world)]
            got8 = torch.cat([decode_blob(p)[0] for p in parts], dim=0).view(torch.int64).view(-1, 5)
            ok = ok and bool(torch.equal(got8, want8))
        else:
            try:
                gx.slots(buf)
                ok = False
            except ValueError:
                pass
    result[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_one_collective_exchange_of_real_lifted_records_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_slot_worker, args=(2, port, result), nprocs=2, join=True)
    assert result[0] and result[1]


def _lines(recs):
    """BED-like lines of hgx_record rows (what a writer renders; names stand in as indices here)"""
    r = recs.contiguous().view(torch.int64).view(-1, 5).numpy()
    out = []
    for q, ts, te, ss, tail in r:
        out.append("q%d\tseq%d\t%d\t%d\t%s\n" % (q, tail & 0xFFFFFFFF, ts, te, chr((tail >> 32) & 0xFF)))
    return "".join(out).encode()


def _writers_worker(rank, world, port, group, path, result):
    """Several writers (hal_amd.shard.SlotExchange(group=), text_placement, write_text_at) on the really-lifted records of
    tests/golden/lifted_records.npz, as four ranks would lift them: four uneven shards of the 600 intervals, each rank's rows with
    shard-relative query indices as a plan returns them."""
    from hal_amd.shard import SlotExchange, text_placement, write_text_at, writer_of
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLD, "lifted_records.npz"))
    whole = torch.from_numpy(z["whole"].copy())
    nq_all = int(z["bounds"][-1])
    b1 = int(z["bounds"][1])
    cuts = [0, b1 // 3, b1, b1 + 3, nq_all] if world == 4 else [shard_bounds(nq_all, world, r)[0] for r in range(world)] + [nq_all]
    q = whole.view(torch.int64).view(-1, 5)[:, 0]

    def shard(r):
        rows = whole[(q >= cuts[r]) & (q < cuts[r + 1])].clone()
        return offset_query_index(rows, -cuts[r]), cuts[r + 1] - cuts[r]

    mine, nq = shard(rank)
    ok = True
    if world == 4 and group == 2:  # (the four shards are the golden's two, cut again: the device's own blobs decode to them)
        two = [torch.from_numpy(z["shard%d" % k].copy()) for k in (0, 1)]
        ok = int(z["bounds"][1]) == cuts[2] and bool(torch.equal(torch.cat([offset_query_index(shard(r)[0], cuts[r] - cuts[0 if r < 2 else 2])
                                                                             for r in range(4)]), torch.cat(two)))
    blob = encode_blob(mine, nq, first_query=cuts[rank])
    slot = 16384
    ex = SlotExchange(world, rank, slot, "cpu", backend="torch", group=group)
    writer = writer_of(rank, group)
    for step in range(4):  # a stream of batches, two in flight
        ex.submit(blob=blob)
        if step % 2 == 0:
            ex.submit(blob=blob)
        bufs = [ex.wait()] + ([ex.wait()] if step % 2 == 0 else [])
        for buf in bufs:
            if rank == writer:
                parts = ex.slots(buf)
                members = list(range(writer, min(writer + group, world)))
                ok = ok and len(parts) == len(members)
                decoded = torch.cat([decode_blob(p)[0] for p in parts], dim=0)
                want = whole[(q >= cuts[members[0]]) & (q < cuts[members[-1] + 1])]
                ok = ok and bool(torch.equal(decoded, want))  # the group's stretch of the unsharded lift, in its order
                text = _lines(decoded)
            else:
                try:
                    ex.slots(buf)
                    ok = False
                except ValueError:
                    pass
                text = b""
            # level two: where the writers' texts go
            offset, total = text_placement(len(text), torch.device("cpu"))
            ok = ok and total == len(_lines(whole))
            out = "%s.%d" % (path, step)
            write_text_at(out, offset, text, total)
            dist.barrier()
            if rank == world - 1:
                ok = ok and open(out, "rb").read() == _lines(whole)  # the single writer's file
            dist.barrier()
    result[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _spawn_writers(world, group, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_writers_worker, args=(world, port, group, str(tmp_path / "out.bed"), result), nprocs=world, join=True)
    assert all(result[r] for r in range(world)), dict(result)


def test_two_writers_of_four_ranks_place_their_texts_world4(tmp_path):
    _spawn_writers(4, 2, tmp_path)


def _golden_lift_inputs():
    """the alignment and the BED lines tests/golden/make_lifted_records.py lifted (same options, same seeds; no device needed)"""
    import hal_amd
    opts = hal_amd.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
                               min_segments=200, max_segments=600, seed=2, with_dna=False)
    al = hal_amd.Alignment.random(opts, device=-1)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    g = torch.Generator().manual_seed(3)
    starts = torch.randint(0, length - 400, (600,), generator=g)
    lens = torch.randint(1, 400, (600,), generator=g)
    lines = ["%s\t%d\t%d\tiv%d\t%d\t%s\n" % (name, int(starts[i]), int(starts[i] + lens[i]), i, i % 1000, "+-."[i % 3]) for i in range(600)]
    return al, src, tgt, lines


def _library_writers_worker(rank, world, port, group, path, img, want, result):
    """the writers' text made INSIDE the library (hgx_liftover_render_blobs: VERDICT r05 next 9): four ranks' blobs of the really
    lifted records, two writers, each renders its group's slots from the group's input lines; side by side they write halLiftover's
    file (the oracle's, made by the test's parent)"""
    import hal_amd
    from hal_amd.shard import SlotExchange, text_placement, write_text_at, writer_of
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    al, src, tgt, lines = _golden_lift_inputs()
    z = np.load(os.path.join(GOLD, "lifted_records.npz"))
    whole = torch.from_numpy(z["whole"].copy())
    b1 = int(z["bounds"][1])
    cuts = [0, b1 // 3, b1, b1 + 3, 600]
    q = whole.view(torch.int64).view(-1, 5)[:, 0]
    rows = offset_query_index(whole[(q >= cuts[rank]) & (q < cuts[rank + 1])].clone(), -cuts[rank])
    ok = True
    for fmt in (None, 8, 40):  # (the 12-byte form by default, the writers' 8-byte form, raw rows)
        blob = encode_blob(rows, cuts[rank + 1] - cuts[rank], first_query=cuts[rank], fmt=fmt)
        ex = SlotExchange(world, rank, 65536, "cpu", backend="torch", group=group)
        ex.submit(blob=blob)
        buf = ex.wait()
        writer = writer_of(rank, group)
        text = b""
        if rank == writer:
            members = list(range(writer, min(writer + group, world)))
            bed = "".join(lines[cuts[members[0]]:cuts[members[-1] + 1]])
            text = hal_amd.liftover_render_blobs(al, src, tgt, bed, ex.slots(buf))
        offset, total = text_placement(len(text), torch.device("cpu"))
        out = "%s.%s" % (path, fmt)
        write_text_at(out, offset, text, total)
        dist.barrier()
        if rank == 0:
            ok = ok and open(out, "rb").read() == want
        dist.barrier()
    if rank == 0:  # what does not fit is refused with a message, not rendered
        try:
            hal_amd.liftover_render_blobs(al, src, tgt, "".join(lines[:10]), [encode_blob(rows, cuts[1], first_query=0)])
            ok = False
        except hal_amd.HgxError as e:
            ok = ok and "intervals" in str(e)
    result[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_writers_render_their_groups_blobs_inside_the_library_world4(oracle_bin, tmp_path):
    import subprocess
    al, src, tgt, lines = _golden_lift_inputs()
    img = str(tmp_path / "al.hgx")
    al.save(img)
    bed = str(tmp_path / "in.bed")
    open(bed, "w").write("".join(lines))
    out = str(tmp_path / "want.bed")
    subprocess.check_call([oracle_bin, "liftover", img, "Genome_9", bed, "Genome_2", out])
    want = open(out, "rb").read()
    assert want.count(b"\n") == 2186  # (the golden's records, one line each)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_library_writers_worker, args=(4, port, 2, str(tmp_path / "out.bed"), img, want, result), nprocs=4, join=True)
    assert all(result[r] for r in range(4)), dict(result)


def test_writers_with_a_ragged_last_group_and_groups_of_one(tmp_path):
    _spawn_writers(3, 2, tmp_path)  # groups {0, 1} and {2}
    _spawn_writers(2, 1, tmp_path)  # every rank its own writer: no records move at all


class _Malformed(Exception):
    pass


def _stand_in_convert(share):
    """a line-by-line conversion with a data-dependent number of output lines; a line beginning with '!' is malformed: what came
    before it is the exception's partial_output (the contract of hal_amd.liftover_convert)"""
    out = []
    n_read = 0  # (the library numbers the lines of the text it is given, blank ones not counted: halBedScanner.cpp:47-59)
    for line in share.split(b"\n"):
        if not line.strip():
            continue
        n_read += 1
        if line.startswith(b"!"):
            e = _Malformed("malformed: " + line.decode() + " in input bed line %d" % n_read)
            e.partial_output = b"".join(out)
            raise e
        for k in range(len(line) % 4):
            out.append(line.upper() + b"\t%d\n" % k)
    return b"".join(out)


def _sharded_worker(rank, world, port, path, result):
    from hal_amd.shard import convert_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    lines = [b"chr%d\t%d\t%d" % (rng.integers(1, 20), rng.integers(0, 10 ** 6), rng.integers(0, 10 ** 6)) + b"x" * int(rng.integers(0, 5))
             for _ in range(997)]
    ok = True
    for tail in (b"\n", b""):  # (with and without a newline at the end of the input)
        data = b"\n".join(lines) + tail
        total = convert_sharded(_stand_in_convert, data, path)
        want = _stand_in_convert(data)
        ok = ok and total == len(want) and open(path, "rb").read() == want
        dist.barrier()
    # a malformed line in the middle share: the output ends with what was lifted before it, every rank raises
    bad = list(lines)
    bad[500] = b"!" + bad[500]
    bad.insert(10, b"")  # (a blank line in the first rank's share: read over, not counted)
    bad.insert(300, b"  ")
    data = b"\n".join(bad) + b"\n"
    try:
        convert_sharded(_stand_in_convert, data, path + ".bad")
        ok = False
    except _Malformed as e:
        # the message is the one process's: the line's number in the whole file, not in the rank's share
        ok = ok and rank == 1 and str(e).endswith(" in input bed line 501") and e.partial_output is not None
        try:
            _stand_in_convert(data)
        except _Malformed as whole:
            ok = ok and str(whole) == str(e)
    except RuntimeError:
        ok = ok and rank != 1
    ok = ok and open(path + ".bad", "rb").read() == _stand_in_convert(b"\n".join(bad[:502]) + b"\n")
    result[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_every_rank_a_writer_of_its_share_of_the_lines_world3(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_sharded_worker, args=(3, port, str(tmp_path / "out.bed"), result), nprocs=3, join=True)
    assert all(result[r] for r in range(3)), dict(result)
