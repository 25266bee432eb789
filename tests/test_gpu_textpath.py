"""The parallel text path of Liftover::convert (hgx_liftover_text.cpp) against the general one-line-at-a-time path
(HGX_TEXT_GENERAL=1) and the oracle: every column count, echoed fields, blank space, odd integers, skipped lines, malformed
lines in the middle of the input (same message, same partial output)."""
import os

import numpy as np
import pytest

from util import oracle_liftover
from test_gpu_liftover import _rand_alignment

pytestmark = pytest.mark.gpu


def _both(hal, al, src, bed, tgt, **kw):
    res = []
    for general in (False, True):
        if general:
            os.environ["HGX_TEXT_GENERAL"] = "1"
        try:
            try:
                res.append(("ok", hal.liftover_convert(al, src, bed, tgt, **kw)))
            except hal.HgxError as e:
                res.append(("error: " + str(e), e.partial_output))
        finally:
            os.environ.pop("HGX_TEXT_GENERAL", None)
    assert res[0] == res[1], (res[0][0], res[1][0])
    return res[0]


def _lines(name, length, n, cols, rng, extras=0):
    out = []
    for i in range(n):
        ln = int(rng.integers(1, 300))
        a = int(rng.integers(0, length - ln))
        f = [name, str(a), str(a + ln), "n%d" % i, str(int(rng.integers(0, 1000))), "+-."[int(rng.integers(0, 3))],
             str(a if rng.integers(0, 2) else 0), str(a + ln if rng.integers(0, 2) else 0),
             ["255,0,0", "7", "1,2", "3,4,5,"][int(rng.integers(0, 4))]]
        f = f[:cols] + ["x%d" % k if k % 2 == 0 else "" for k in range(extras)]
        out.append("\t".join(f))
    return out


@pytest.mark.parametrize("cols", [3, 4, 5, 6, 8, 9])
def test_every_column_count(hal, oracle_bin, tmp_path, cols):
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(cols)
    for extras in (0, 3):
        bed = "\n".join(_lines(name, length, 400, cols, rng, extras)) + "\n"
        kw = {"bed_type": cols} if extras else {}
        status, text = _both(hal, al, src, bed, tgt, **kw)
        assert status == "ok" and text.count("\n") > 400
        assert text == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path, bed_type=cols if extras else 0)


def test_blank_space_odd_integers_and_skipped_lines(hal, oracle_bin, tmp_path):
    al, img = _rand_alignment(hal, tmp_path, 5)
    src, tgt = al.genome_id("Genome_3"), al.genome_id("Genome_1")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(1)
    body = _lines(name, length, 300, 6, rng)
    body[3] = "  \t " + body[3]                                # leading blanks are skipped with the blank lines
    body[10] = body[10] + "\r"                                  # (the carriage return stays in the echoed strand field? no: it is extra text of field 5)
    body[20] = body[20].replace("\t", "\t ", 2)                 # blanks in front of integers
    f = body[30].split("\t")
    f[1], f[2], f[4] = "+" + f[1], f[2] + "abc", "007"
    body[30] = "\t".join(f)
    body[40] = "nosuchseq\t1\t20\tq\t0\t+"
    body[41] = "%s\t%d\t%d\tq\t0\t-" % (name, length - 5, length + 50)  # end past the sequence: skipped
    body[42] = body[42] + "\t"                                  # a trailing tab adds no field
    bed = "\n\n  \n".join(body[:50]) + "\n" + "\n".join(body[50:]) + "\n\n"
    status, text = _both(hal, al, src, bed, tgt)
    assert status == "ok"
    assert text == oracle_liftover(oracle_bin, img, "Genome_3", "Genome_1", bed, tmp_path)


@pytest.mark.parametrize("bad", ["chr\t5", "%s\t50\t50\tq\t0\t+", "%s\t10\t20\tq\t0\t*", "%s\tx1\t20\tq\t0\t+", "%s\t10\t20\tq\tscore\t+",
                                 "%s\t10\t99999999999999999999\tq\t0\t+"])
def test_malformed_line_in_the_middle(hal, tmp_path, bad):
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(3)
    body = _lines(name, length, 5000, 6, rng)
    body.insert(3777, bad % name if "%s" in bad else bad)
    bed = "\n".join(body) + "\n"
    status, partial = _both(hal, al, src, bed, tgt)
    assert status.startswith("error: ") and status.endswith("in input bed line 3778"), status
    assert partial == hal.liftover_convert(al, src, "\n".join(body[:3777]) + "\n", tgt)


def test_mixed_column_counts_and_bed12_take_the_general_path(hal, oracle_bin, tmp_path):
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(4)
    body = _lines(name, length, 200, 8, rng) + _lines(name, length, 200, 3, rng) + _lines(name, length, 200, 7, rng)
    bed = "\n".join(body) + "\n"
    status, text = _both(hal, al, src, bed, tgt)
    assert status == "ok" and text == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path)


def test_a_million_lines(hal, tmp_path):
    """10 Mb genomes, 1 M BED6 lines: the parallel path (chunks over all cores) gives the bytes of the general path."""
    import torch
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=70000, max_segments=140000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    n = 1000000
    g = torch.Generator().manual_seed(5)
    starts = torch.randint(0, length - 1100, (n,), generator=g).numpy()
    lens = torch.randint(50, 1000, (n,), generator=g).numpy()
    bed = "".join("%s\t%d\t%d\tq%d\t%d\t%s\n" % (name, a, a + b, i, i % 1000, "+-"[i & 1]) for i, (a, b) in enumerate(zip(starts, lens)))
    status, text = _both(hal, al, src, bed, tgt)
    assert status == "ok" and text.count("\n") > 2 * n


def test_lines_shared_out_over_device_clones(hal, oracle_bin, tmp_path):
    """hgx_liftover_convert_multi: three handles of one alignment (clones on the one GPU of the test box: the sharding, staging
    and collation are the same code as with three GPUs) give the bytes of the single handle and of the oracle, also with a
    malformed line in the last share, and so does the CLI twin with --devices."""
    import subprocess
    al, img = _rand_alignment(hal, tmp_path, 2)
    clones = [al, al.clone_to_device(0), al.clone_to_device(0)]
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(9)
    body = _lines(name, length, 30000, 6, rng)
    bed = "\n".join(body) + "\n"
    one = hal.liftover_convert(al, src, bed, tgt)
    assert hal.liftover_convert_multi(clones, src, bed, tgt) == one
    assert one == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path)
    assert hal.liftover_convert_multi(clones, src, bed, tgt, traverse_dupes=False) == hal.liftover_convert(al, src, bed, tgt, traverse_dupes=False)
    bad = body[:29000] + ["%s\t10\t5\tq\t0\t+" % name] + body[29000:]
    with pytest.raises(hal.HgxError, match="in input bed line 29001") as e:
        hal.liftover_convert_multi(clones, src, "\n".join(bad) + "\n", tgt)
    assert e.value.partial_output == hal.liftover_convert(al, src, "\n".join(body[:29000]) + "\n", tgt)
    # BED12 goes the general way on the first handle
    b12 = "".join("%s\t%d\t%d\tn\t0\t+\t%d\t%d\t0\t2\t10,10,\t0,%d,\n" % (name, a, a + 60, a, a + 60, 50) for a in range(100, 20000, 700))
    assert hal.liftover_convert_multi(clones, src, b12, tgt) == hal.liftover_convert(al, src, b12, tgt)
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hal_amd", "_build", "halLiftover")
    inp, outp = str(tmp_path / "in.bed"), str(tmp_path / "out.bed")
    open(inp, "w").write(bed)
    subprocess.check_call([tool, "--devices", "0,0", img, "Genome_9", inp, "Genome_2", outp])
    assert open(outp).read() == one


def test_device_batches_of_bounded_size(hal, oracle_bin, tmp_path, monkeypatch):
    """The text path lifts its chunks in groups of at most batchLines intervals per device (HGX_BATCH_LINES here): an input of many
    groups — on one handle and dealt over three — gives the bytes of one batch, a malformed line in a later group ends the output
    where it stands, and the plan that serves it was sized for a group, not for the input."""
    al, img = _rand_alignment(hal, tmp_path, 2)
    clones = [al, al.clone_to_device(0), al.clone_to_device(0)]
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(12)
    body = _lines(name, length, 60000, 6, rng)
    bed = "\n".join(body) + "\n"
    one = hal.liftover_convert(al, src, bed, tgt)
    assert one == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path)
    for lines in ("1", "700", "5000"):
        monkeypatch.setenv("HGX_BATCH_LINES", lines)
        assert hal.liftover_convert(al, src, bed, tgt) == one, lines
        assert hal.liftover_convert_multi(clones, src, bed, tgt) == one, lines
    monkeypatch.setenv("HGX_BATCH_LINES", "900")
    bad = body[:41000] + ["%s\t10\t5\tq\t0\t+" % name] + body[41000:]
    with pytest.raises(hal.HgxError, match="in input bed line 41001") as e:
        hal.liftover_convert_multi(clones, src, "\n".join(bad) + "\n", tgt)
    monkeypatch.delenv("HGX_BATCH_LINES")
    assert e.value.partial_output == hal.liftover_convert(al, src, "\n".join(body[:41000]) + "\n", tgt)


@pytest.mark.parametrize("table", [False, True])
def test_negative_start_lands_in_the_sequence_in_front(hal, oracle_bin, tmp_path, monkeypatch, table):
    """chromStart < 0 is not checked by the reference (halLiftover.cpp:52-66 looks at the end only; halBlockLiftover.cpp:48 adds the
    sequence's start): the interval begins in the sequence in front.  Text path, general path and oracle agree — with the walk
    and with the merged table (whose chains are not cut at the source's sequence boundaries for this reason)."""
    import halfix
    if table:
        monkeypatch.setenv("HGX_COMPOSED_UP", "1")
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(3, n_genomes=6, max_seqs=5))
    al = hal.Alignment.open(img, device=0)
    checked = 0
    for s in range(al.num_genomes):
        seqs = [q for q in al.sequences(s) if q[2] > 0]
        for k in range(1, len(seqs)):
            name, start, length = seqs[k]
            back = min(7, start)
            if back == 0:
                continue
            bed = "%s\t%d\t%d\tneg\t0\t+\n%s\t0\t%d\tpos\t0\t-\n" % (name, -back, min(length, 9), name, min(length, 9))
            for t in range(al.num_genomes):
                status, text = _both(hal, al, s, bed, t)
                assert status == "ok"
                assert text == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path), (s, k, t)
                checked += 1
    assert checked > 10
