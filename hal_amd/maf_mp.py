"""hal2mafMP.py (maf/hal2mafMP.py) over the GPUs of a node, one process per GPU and every process a writer:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m hal_amd.maf_mp \\
        <halFile> <mafFile> [--refGenome G] [--refSequence S] [--refStart A] [--length L] [--sliceSize K] [--targetGenomes a,b]
        [--noAncestors] [--noDupes] [--onlyOrthologs] [--onlySequenceNames] [--keepEmptyRefBlocks] [--maxBlockLen M] [--unique]

(the files first: the launcher takes options that follow the module's name for its own — and "--start" wherever it stands, for
its --start-method: under the launcher it is spelled --refStart)

The reference cuts every reference sequence into slices of sliceSize columns (computeSlices, maf/hal2mafMP.py:63-79), runs one hal2maf
per slice in a pool of processes and concatenates the outputs, dropping the header of every slice but the first
(concatenateSlices, :176-190).  Here rank r exports a contiguous run of every sequence's slices on GPU r (hgx_maf_export_multi over
its own handle: the slices of a handle share the per-base tracks), the ranks exchange the sizes of their texts — nothing else, over
the launcher's gloo group — and write side by side into the one MAF file.  The text is hgx_maf_export_multi's with the same slice size
(= hal2mafMP.py's)."""
import argparse
import os
import sys


def slices_of(seq_len, start, length, slice_size, world):
    """computeSlices for one sequence: [(start, length)], maf/hal2mafMP.py:63-79 (slice_size 0: the range divided evenly over the ranks)"""
    if start >= seq_len or start + length > seq_len:
        raise ValueError("Invalid range specified for convertGenome")
    n = seq_len - start if length < 1 else length
    size = slice_size if slice_size > 0 else (n + world - 1) // world
    if size < 1 or size >= n:
        return [(start, n)]
    out = [(start + i * size, size) for i in range(n // size)]
    if n % size:
        out.append((start + (n // size) * size, n % size))
    return out


def run(hal_path, maf_path, ref_genome=None, ref_sequence=None, start=0, length=0, slice_size=0, device=None, targets=None, **opts):
    """the calling process's part of the job (torch.distributed is initialised); returns the output's size.  opts: the keyword
    arguments of hal_amd.maf_export_multi (no_dupes, no_ancestors, only_sequence_names, only_orthologs, keep_empty_ref_blocks,
    max_block_len, unique)"""
    import torch
    import torch.distributed as dist
    import hal_amd
    from hal_amd import shard
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:  # (HGX_MP_DEVICE: every rank on that device — the tests' box has one GPU)
        device = int(os.environ.get("HGX_MP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    cpu = torch.device("cpu")
    al, failure, plan = None, None, []
    try:
        al = hal_amd.Alignment.open(hal_path, device=device)
        ref = al.genome_id(ref_genome) if ref_genome else 0
        if ref < 0:
            raise hal_amd.HgxError("Reference genome, %s, not found" % ref_genome)
        tg = None
        if targets:
            tg = [al.genome_id(t) for t in targets]
            if min(tg) < 0:
                raise hal_amd.HgxError("Target genome, %s, not found" % targets[tg.index(min(tg))])
        seqs = al.sequences(ref)
        for si, (name, _, seq_len) in enumerate(seqs):
            if ref_sequence is not None and name != ref_sequence:
                continue
            if seq_len == 0 and ref_sequence is None:
                continue
            plan.append((si, slices_of(seq_len, start, length, slice_size, world)))
        if ref_sequence is not None and not plan:
            raise hal_amd.HgxError("Reference sequence, %s, not found" % ref_sequence)
    except Exception as e:  # (the other ranks are told before anybody waits in a collective of the export)
        failure = e
    bad = shard.all_gather_counts(1 if failure is not None else 0, cpu)
    if failure is not None:
        raise failure
    if 1 in bad:
        al.close()
        raise RuntimeError("rank %d could not open its inputs" % bad.index(1))
    base, first_text = 0, True
    try:
        for si, sl in plan:
            # this rank's run of the sequence's slices: a range of whole slices, exported with the same slice size
            lo, hi = shard.shard_bounds(len(sl), world, rank)
            text, err = b"", None
            if hi > lo:
                size = sl[lo][1]  # (every slice but a sequence's last has the slice size)
                try:
                    text = hal_amd.maf_export_multi([al], ref, si, start=sl[lo][0], length=sum(n for _, n in sl[lo:hi]), slice_size=size,
                                                    targets=tg, **opts).encode()
                except Exception as e:
                    err = e
                if not (first_text and lo == 0):  # concatenateSlices: the header of the very first slice only
                    lines = text.split(b"\n")
                    k = 0
                    while k < len(lines) and lines[k].startswith(b"#"):
                        k += 1
                    text = b"\n".join(lines[k:])
            failed = shard.all_gather_counts(1 if err is not None else 0, cpu)
            if err is not None:
                raise err
            if 1 in failed:
                raise RuntimeError("rank %d failed in its export" % failed.index(1))
            offset, total = shard.text_placement(len(text), cpu)
            shard.write_text_at(maf_path, base + offset, text, base + total)
            base += total
            first_text = False
        return base
    finally:
        al.close()
        dist.barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="hal_amd.maf_mp", description=__doc__.split("\n\n")[0])
    ap.add_argument("halFile")
    ap.add_argument("mafFile")
    ap.add_argument("--refGenome")
    ap.add_argument("--refSequence")
    # (--refStart: the same under torch.distributed.run, whose own parser takes "--start" for its --start-method)
    ap.add_argument("--start", "--refStart", dest="start", type=int, default=0)
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--sliceSize", type=int, default=0)
    ap.add_argument("--targetGenomes")
    ap.add_argument("--maxBlockLen", type=int, default=1000)
    for flag in ("noAncestors", "noDupes", "onlyOrthologs", "onlySequenceNames", "keepEmptyRefBlocks", "unique"):
        ap.add_argument("--" + flag, action="store_true")
    a = ap.parse_args(argv)
    import torch.distributed as dist
    dist.init_process_group("gloo")  # (sizes only: RANK / WORLD_SIZE / MASTER_* from the launcher)
    try:
        run(a.halFile, a.mafFile, ref_genome=a.refGenome, ref_sequence=a.refSequence, start=a.start, length=a.length, slice_size=a.sliceSize,
            targets=a.targetGenomes.split(",") if a.targetGenomes else None, no_dupes=a.noDupes, no_ancestors=a.noAncestors,
            only_sequence_names=a.onlySequenceNames, only_orthologs=a.onlyOrthologs, keep_empty_ref_blocks=a.keepEmptyRefBlocks,
            max_block_len=a.maxBlockLen, unique=a.unique)
    except Exception as e:
        sys.stderr.write("hal exception caught: %s\n" % e)
        return 1
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
