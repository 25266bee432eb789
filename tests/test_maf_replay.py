"""hal2maf's host side — MafBlock's state machine on flat arrays, the log of pairings, the rendering threads; the column-by-column
paths of --unique, --maxRefGap and --printTree (visit caches, the stack of inserted and deleted ranges, the block's tree)
(hal_amd/csrc/hgx_columns_host.cpp: MafExport::RunMachine; maf/impl/halMafBlock.cpp:36-82, 294-450, maf/impl/halMafExport.cpp:51-87) —
without a GPU: the device's batches of a list of exports were recorded on a GPU box (tests/golden/make_maf_batches.py); the
profiling build of the library (make hostprof-lib) plays them back to the host side, and the text must be the oracle's.  The
product library has no such switch: it fails without a device (test_capi_host.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

SCRIPT = r'''
import os, subprocess, sys, tempfile
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import hal_amd as hal
import make_maf_batches as mk
oracle = %r
bad = n = trees = 0
with tempfile.TemporaryDirectory() as tmp:
    for img, name, got, args in mk.cases(hal, -1, tmp):
        out = os.path.join(tmp, "o.maf")
        r = subprocess.run([oracle, "maf", img, out] + args, stderr=subprocess.PIPE)
        n += 1
        if got is None:  # (--printTree where the reference has no tree either)
            ok = r.returncode != 0 and any(m in r.stderr.decode() for m in mk.NO_TREE)
            trees -= 1
        else:
            ok = r.returncode == 0 and got == open(out).read()
        if "--printTree" in args:
            trees += 1
        if not ok:
            bad += 1
            print("DIFFERENT", img, name, args)
print("exports %%d different %%d trees %%d" %% (n, bad, trees))
''' % (ROOT, os.path.join(ROOT, "tests"), GOLD, os.path.join(ROOT, "oracle", "_build", "hal_oracle"))


def test_recorded_device_batches_through_the_host_state_machine(oracle_bin):
    lib = os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hal_amd", "csrc"), "hostprof-lib"])  # (nothing to do when it is current)
    env = dict(os.environ, HGX_LIB_PATH=lib, HGX_MAF_REPLAY=os.path.join(GOLD, "maf_batches.bin"))
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, check=True, stdout=subprocess.PIPE).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[0] == "exports" and int(last[1]) >= 120 and int(last[3]) == 0 and int(last[5]) >= 8, out  # (8+ exports with trees)


GLOBAL_SCRIPT = r'''
import os, subprocess, sys, tempfile
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import hal_amd as hal
import make_maf_batches as mk
with tempfile.TemporaryDirectory() as tmp:
    img, got = mk.global_case(hal, -1, tmp)
    out = os.path.join(tmp, "o.maf")
    subprocess.check_call([%r, "maf", img, out, "--global"])
    print("global", "same" if got == open(out).read() else "DIFFERENT", len(got))
''' % (ROOT, os.path.join(ROOT, "tests"), GOLD, os.path.join(ROOT, "oracle", "_build", "hal_oracle"))


def test_global_export_where_an_abandoned_walk_lies_under_a_paralogy_cycle(oracle_bin):
    """hal2maf --global over the alignment a randomised soak found: in the last column of a leaf's pass the walk is abandoned at a
    base an earlier leaf wrote, inside the subtree of a member of a paralogy cycle — whose remaining members the reference still
    inserts (updateNextTopDup's loop does not look at _break, api/impl/halColumnIterator.cpp:653-680)."""
    lib = os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hal_amd", "csrc"), "hostprof-lib"])  # (nothing to do when it is current)
    env = dict(os.environ, HGX_LIB_PATH=lib, HGX_MAF_REPLAY=os.path.join(GOLD, "maf_global_batches.bin"))
    out = subprocess.run([sys.executable, "-c", GLOBAL_SCRIPT], env=env, check=True, stdout=subprocess.PIPE).stdout.decode()
    assert out.strip().splitlines()[-1].startswith("global same"), out


def test_oracle_written_batches_through_the_host_state_machine(oracle_bin):
    """The plain export's host side over batches the ORACLE writes (hal_oracle columns --batches: which columns are heads, the
    heads' rows, cut into chunks of 1 .. 2^21 columns so that blocks, runs and sequences cross batch ends): random alignments with
    several sequences a genome, every genome as the reference, --noDupes / --noAncestors / --maxBlockLen / --keepEmptyRefBlocks /
    --onlySequenceNames; the text must be the oracle's (profiles/scripts/r04_cpu_maf_soak.py; profiles/r04y_cpu_maf_soak.txt has
    the long runs)."""
    lib = os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hal_amd", "csrc"), "all", "hostprof-lib"])
    assert os.path.exists(lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r04_cpu_maf_soak.py"), "7000", "12"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["alignments", "12"] and int(last[3]) >= 60 and last[4:] == ["different", "0"], out


def test_oracle_written_records_through_the_general_liftover_path(oracle_bin):
    """halLiftover's general path (BED12 blocks, PSL, mixed column counts, malformed lines: hal_amd/csrc/hgx_liftover_host.cpp,
    Liftover::convertGeneral — a batch's lines dealt to the host's threads) over records the ORACLE writes (hal_oracle liftover
    --records: every lifted interval's hgx_record rows), batches of 1 .. 4 M intervals; text and partial text before an error must be
    the oracle's (profiles/scripts/r04_cpu_liftover_soak.py; profiles/r04y_cpu_liftover_soak.txt has the long run)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hal_amd", "csrc"), "all", "hostprof-lib"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r04_cpu_liftover_soak.py"), "9000", "12"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["alignments", "12"] and int(last[3]) >= 60 and last[4:] == ["different", "0"], out


def test_walk_over_slices_of_the_export_is_the_one_thread_walk(oracle_bin):
    """hal2maf's block state machine over slices of the export side by side (MafExport::walkSliced: cold run-ups, the count of
    blocks told from round to round, slices accepted when they begin in the very state the slice before ended in, the first
    unsettled slice walked from that state) on batches the oracle writes: exports of thousands of blocks (block-length limits of
    1 .. 5: the column map's keys are reset at every thousandth block), genomes of several sequences that come and go, 5 .. 200
    slices, run-ups of 3 .. 4096 heads, 1 / 3 / 8 threads, --unique / --targetGenomes / --noDupes; every text the oracle's, every
    walk settled (profiles/scripts/r05_cpu_maf_sliced_soak.py; profiles/r05_cpu_maf_sliced_soak.txt has the long runs, and the
    full-size config-3 export — 27 slices, 1.79 M blocks, 1.81 GB — is byte-identical too: profiles/r05_notes.md)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hal_amd", "csrc"), "all", "hostprof-lib"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r05_cpu_maf_sliced_soak.py"), "500", "20"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["alignments", "20"] and int(last[3]) >= 40 and last[4:8] == ["different", "0", "not", "settled"] and last[8] == "0", out
    # batches that arrive while the rounds run (MafExport::Arrivals: the walk takes them in as they come)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r05_cpu_maf_sliced_soak.py"), "520", "6"],
                         env=dict(os.environ, HGX_MAF_FEED_DELAY_US="1500"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["alignments", "6"] and last[4:8] == ["different", "0", "not", "settled"] and last[8] == "0", out
    # and the earlier soak's shapes (chunks of 1 .. 2^21 columns, every option of the state machine) with the walk over slices forced
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r04_cpu_maf_soak.py"), "7100", "8"],
                         env=dict(os.environ, HGX_MAF_SLICED="1", HGX_MAF_RUNUP="50"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["alignments", "8"] and last[4:] == ["different", "0"], out
