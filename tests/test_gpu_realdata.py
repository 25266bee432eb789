"""Parity on a real (non-synthetic) alignment: the reference's own HDF5 test file paf/tests/input/mr.hal (evolver
mouse / rat chr6 under their ancestor `mr`, 55 k segments per genome, three sequences in the root, real indels,
inversions and duplications, real DNA), imported by the product's HDF5 reader.  HIP path vs oracle for liftover in every
direction (BED6, PSL), alignment depth and MAF."""
import os

import numpy as np
import pytest

from util import oracle_liftover
from test_gpu_columns import _oracle
from test_gpu_multiseq import _bed

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mr(hal, tmp_path_factory):
    try:
        al = hal.Alignment.open(os.path.join(GOLD, "ref_hdf5", "mr.hal"), device=0)
    except hal.HgxError as e:
        if "HDF5 C library" in str(e):
            pytest.skip("libhdf5 not loadable here: %s" % e)
        raise
    img = str(tmp_path_factory.mktemp("mr") / "mr.hgx")
    al.save(img)
    return al, img


@pytest.mark.parametrize("src,tgt", [("simMouse_chr6", "simRat_chr6"), ("simRat_chr6", "simMouse_chr6"), ("simMouse_chr6", "mr"),
                                     ("mr", "simRat_chr6"), ("mr", "mr")])
def test_liftover_mouse_rat(hal, oracle_bin, tmp_path, mr, src, tgt):
    al, img = mr
    s, t = al.genome_id(src), al.genome_id(tgt)
    bed = _bed(al, s, 3000, 17 + s * 3 + t)
    rng = np.random.default_rng(5)
    # longer intervals as well: they cross many segments, indels and inversions
    for name, _, length in al.sequences(s):
        for _ in range(40):
            ln = int(rng.integers(1000, 20000))
            st = int(rng.integers(0, length - ln))
            bed += "%s\t%d\t%d\tlong\t0\t%s\n" % (name, st, st + ln, "+-"[int(rng.integers(0, 2))])
    for nd in (False, True):
        got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
        want = oracle_liftover(oracle_bin, img, src, tgt, bed, tmp_path, no_dupes=nd)
        assert got == want, (src, tgt, nd)
        assert got.count("\n") > 1000
    bedp = bed.replace("\t.\n", "\t+\n")
    assert hal.liftover_convert(al, s, bedp, t, out_psl=True) == oracle_liftover(oracle_bin, img, src, tgt, bedp, tmp_path, psl=True)


def test_columns_mouse_rat(hal, oracle_bin, tmp_path, mr):
    al, img = mr
    for name in ("simMouse_chr6", "mr"):
        g = al.genome_id(name)
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        sname, _, slen = al.sequences(g)[0]
        args = ("--refGenome", name, "--refSequence", sname, "--start", str(slen // 3), "--length", "60000")
        assert al.maf_export(g, 0, start=slen // 3, length=60000) == _oracle(oracle_bin, "maf", img, tmp_path, *args), name
        if name != "mr":  # an ancestral reference with --noAncestors is an error in the reference too (hal2maf.cpp)
            assert al.maf_export(g, 0, start=slen // 3, length=60000, no_ancestors=True, no_dupes=True) == \
                _oracle(oracle_bin, "maf", img, tmp_path, *args, "--noAncestors", "--noDupes"), name
        else:
            with pytest.raises(hal.HgxError, match="noAncestors option is invalid"):
                al.maf_export(g, 0, no_ancestors=True)
        assert al.maf_export(g, 0, start=slen // 3, length=20000, unique=True) == \
            _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--refSequence", sname, "--start", str(slen // 3), "--length", "20000",
                    "--unique"), name
