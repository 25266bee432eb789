"""The command-line twins (halLiftover, hal2maf, halAlignmentDepth) against the reference's CLI goldens."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "hal_amd", "_build")


@pytest.fixture(scope="module")
def small_hal(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("cli") / "small.hgx")
    subprocess.check_call([os.path.join(BIN, "hgxRandGen"), "--preset", "small", "--seed", "0", "--testRand", p],
                          stderr=subprocess.DEVNULL)
    return p


def test_halLiftover_cli_goldens(hal, small_hal, tmp_path):
    out = str(tmp_path / "o.bed")
    subprocess.check_call([os.path.join(BIN, "halLiftover"), small_hal, "Genome_0",
                           os.path.join(GOLD, "ref_liftover", "test1.bed3"), "Genome_2", out])
    assert open(out).read() == open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed3Test.bed")).read()
    subprocess.check_call([os.path.join(BIN, "halLiftover"), "--bedType", "4", small_hal, "Genome_0",
                           os.path.join(GOLD, "ref_liftover", "test1.bed4+2"), "Genome_2", out])
    assert open(out).read() == open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed4ExtraTest.bed")).read()
    # stdin/stdout streaming and --append
    bed = open(os.path.join(GOLD, "ref_liftover", "test1.bed3"), "rb").read()
    r = subprocess.run([os.path.join(BIN, "halLiftover"), small_hal, "Genome_0", "stdin", "Genome_2", "stdout"], input=bed,
                       stdout=subprocess.PIPE, check=True)
    assert r.stdout.decode() == open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed3Test.bed")).read()
    r = subprocess.run([os.path.join(BIN, "halLiftover"), small_hal, "Nope", "stdin", "Genome_2", "stdout"], input=bed,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"srcGenome, Nope, not found in alignment" in r.stderr


def test_hal2maf_cli_goldens(hal, small_hal, tmp_path):
    out = str(tmp_path / "o.maf")
    subprocess.check_call([os.path.join(BIN, "hal2maf"), small_hal, out])
    assert open(out).read() == open(os.path.join(GOLD, "ref_maf", "hal2mafSmallTest.maf")).read()
    subprocess.check_call([os.path.join(BIN, "hal2maf"), "--refGenome", "Genome_2", "--refSequence", "Genome_2_seq", "--start", "1000",
                           "--length", "2000", small_hal, out])
    assert open(out).read() == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqPartTest.maf")).read()


def test_halAlignmentDepth_cli_vs_oracle(hal, oracle_bin, small_hal, tmp_path):
    out, want = str(tmp_path / "o.wig"), str(tmp_path / "w.wig")
    for args in ([], ["--countDupes"], ["--step", "13"], ["--start", "10", "--length", "500", "--refSequence", "Genome_3_seq"]):
        subprocess.check_call([os.path.join(BIN, "halAlignmentDepth"), small_hal, "Genome_3", "--outWiggle", out] + args)
        subprocess.check_call([oracle_bin, "depth", small_hal, "Genome_3", want] + args)
        assert open(out).read() == open(want).read(), args
