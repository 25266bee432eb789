"""hal2maf's walk over slices of the export (hal_amd/csrc/hgx_columns_host.cpp: MafExport::walkSliced — cold run-ups, counts of blocks
told from round to round, the first unsettled slice walked from the state the slice before stopped in, slices accepted although
their count was off when nothing they decided hung on it) soaked on a machine WITHOUT a GPU against the oracle's text, where it is
hard: exports of thousands of blocks (short block-length limits: the column map's keys are reset at every thousandth block), genomes
of several sequences that come and go (keys whose entries decide), batches of a few dozen to a few thousand columns (5 .. 200 slices),
run-ups from 3 heads (nothing known at the seam) to 4096, with and without --unique / --targetGenomes / --noDupes.
usage: python profiles/scripts/r05_cpu_maf_sliced_soak.py [first seed] [alignments]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import halfix
ORACLE = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
TOOL = os.path.join(ROOT, "hal_amd", "_build", "hal2maf")
LIB = os.environ.get("HGX_SOAK_PRELOAD", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
exports = different = notsettled = rounds_total = 0
with tempfile.TemporaryDirectory() as tmp:
    img, rec, want, got = (os.path.join(tmp, n) for n in ("a.hgx", "rec.bin", "want.maf", "got.maf"))
    for seed in range(first, first + count):
        rng = random.Random(1000 + seed)
        al = halfix.random_multiseq_alignment(seed, n_genomes=rng.randint(3, 9), max_children=rng.randint(1, 3), root_len=rng.choice([3000, 6000, 12000]),
                                              max_seqs=rng.choice([1, 4, 12]))
        halfix.write_hgx(img, al)
        refs = [gd for gd in al if gd["tStart"][-1] > 1500 or gd["bStart"][-1] > 1500]
        for gd in rng.sample(refs, min(3, len(refs))):
            name, leaf = gd["name"], not gd["children"]
            col, host = [], []
            if rng.random() < 0.2:
                col.append("--noDupes")
            if leaf and rng.random() < 0.5:
                col.append("--noAncestors")
            if rng.random() < 0.3:
                col.append("--unique")
            if rng.random() < 0.25 and len(al) > 2:
                col += ["--targetGenomes", ",".join(g["name"] for g in rng.sample(al, rng.randint(1, min(3, len(al)))))]
            host += ["--maxBlockLen", str(rng.choice([1, 1, 2, 3, 5, 1000]))]
            if rng.random() < 0.3:
                host.append("--keepEmptyRefBlocks")
            chunk = rng.choice([37, 150, 400, 1100])
            runup = rng.choice([3, 40, 300, 4096])
            subprocess.run([ORACLE, "columns", img, name, "--batches", rec, "--chunk", str(chunk)] + col, check=True, stderr=subprocess.DEVNULL)
            subprocess.run([ORACLE, "maf", img, want, "--refGenome", name] + col + host, check=True)
            r = subprocess.run([TOOL, "--device", "-1", "--refGenome", name] + col + host + [img, got],
                               env=dict(os.environ, LD_PRELOAD=LIB, HGX_MAF_REPLAY=rec, HGX_MAF_SLICED="1", HGX_MAF_RUNUP=str(runup), HGX_MAF_TIMING="1",
                                        HGX_MAF_WALK_THREADS=str(rng.choice([1, 3, 8]))), stderr=subprocess.PIPE)
            exports += 1
            err = r.stderr.decode()
            if "NOT settled" in err:
                notsettled += 1
            for line in err.split("\n"):
                if "round(s)" in line:
                    rounds_total += int(line.split(" round(s)")[0].split()[-1])
            a, b = open(want).read(), (open(got).read() if r.returncode == 0 else None)
            if a != b:
                different += 1
                print("DIFFERENT seed %d genome %s chunk %d runup %d %s rc %d %s" % (seed, name, chunk, runup, col + host, r.returncode, err[-400:]), flush=True)
        if (seed - first) % 10 == 9:
            print("alignments %d exports %d different %d not settled %d rounds %d" % (seed - first + 1, exports, different, notsettled, rounds_total), flush=True)
print("alignments %d exports %d different %d not settled %d rounds %d" % (count, exports, different, notsettled, rounds_total))
sys.exit(1 if different else 0)
