"""halLiftover over the GPUs of a node, one process per GPU and every process a writer:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m hal_amd.liftover_mp \\
        <halFile> <srcGenome> <srcBed> <tgtGenome> <tgtBed> [--noDupes] [--outPSL] [--bedType T]

The reference's way to use more hardware is a pool of halLiftover processes over pieces of the BED file whose outputs are put
together afterwards (stats/halStats.py:16,38; maf/hal2mafMP.py:176-190 for hal2maf); here rank r lifts its share of the lines on
GPU r (hal_amd.shard.line_shares: a line's lifting knows nothing of the other lines, liftover/impl/halLiftover.cpp:46-92), the
ranks exchange nothing but the sizes of their texts (eight bytes each, over the launcher's own gloo group: no GPU collective) and
write side by side into the one output file (hal_amd.shard.convert_sharded).  The arguments are halLiftover's
(liftover/halLiftoverMain.cpp:18-60); the output is halLiftover's, byte for byte; a malformed line ends the output where the one
process ends it, with the one process's message and exit code 1."""
import argparse
import os
import sys


def run(hal_path, src_genome, src_bed, tgt_genome, tgt_bed, no_dupes=False, out_psl=False, bed_type=0, device=None):
    """the calling process's part of the job (torch.distributed is initialised); returns the output's size"""
    import torch.distributed as dist
    import hal_amd
    from hal_amd import shard
    import torch
    if device is None:  # (HGX_MP_DEVICE: every rank on that device — the tests' box has one GPU)
        device = int(os.environ.get("HGX_MP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    al, failure = None, None
    try:
        al = hal_amd.Alignment.open(hal_path, device=device)
        src, tgt = al.genome_id(src_genome), al.genome_id(tgt_genome)
        if src < 0 or tgt < 0:
            raise hal_amd.HgxError("Genome %s not found in alignment" % (src_genome if src < 0 else tgt_genome))
        with open(src_bed, "rb") as f:
            data = f.read()
    except Exception as e:  # (the other ranks are told before anybody waits in a collective of the conversion)
        failure = e
    bad = shard.all_gather_counts(1 if failure is not None else 0, torch.device("cpu"))
    if failure is not None:
        raise failure
    if 1 in bad:
        al.close()
        raise RuntimeError("rank %d could not open its inputs" % bad.index(1))

    def convert(share):
        try:
            return hal_amd.liftover_convert(al, src, share, tgt, bed_type=bed_type, traverse_dupes=not no_dupes, out_psl=out_psl).encode()
        except hal_amd.HgxError as e:
            e.partial_output = (getattr(e, "partial_output", "") or "").encode()
            raise
    try:
        return shard.convert_sharded(convert, data, tgt_bed)
    finally:
        al.close()
        dist.barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="hal_amd.liftover_mp", description=__doc__.split("\n\n")[0])
    ap.add_argument("halFile")
    ap.add_argument("srcGenome")
    ap.add_argument("srcBed")
    ap.add_argument("tgtGenome")
    ap.add_argument("tgtBed")
    ap.add_argument("--noDupes", action="store_true")
    ap.add_argument("--outPSL", action="store_true")
    ap.add_argument("--bedType", type=int, default=0)
    a = ap.parse_args(argv)
    import torch.distributed as dist
    dist.init_process_group("gloo")  # (sizes only: RANK / WORLD_SIZE / MASTER_* from the launcher)
    try:
        run(a.halFile, a.srcGenome, a.srcBed, a.tgtGenome, a.tgtBed, no_dupes=a.noDupes, out_psl=a.outPSL, bed_type=a.bedType)
    except Exception as e:
        from hal_amd import shard
        if not isinstance(e, shard.PeerFailed):  # (the rank that met the line prints the one message, numbered in the whole file)
            sys.stderr.write("hal exception caught: %s\n" % e)
        return 1
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
