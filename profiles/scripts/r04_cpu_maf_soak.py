"""hal2maf's host side (the block state machine on flat arrays, the log, the rendering threads: hal_amd/csrc/hgx_columns_host.cpp,
RunMachine) soaked on a machine WITHOUT a GPU: for random alignments (tests/halfix.py, several sequences a genome) the oracle writes
the plain export's columns in the layout of the library's recorded device batches (hal_oracle columns --batches: which columns are
heads, the heads' rows; with --unique: which columns the iterator passes over or walks without writing) with small and odd chunk sizes, the profiling build of the library (make -C hal_amd/csrc hostprof-lib) plays
them back through hal2maf (HGX_MAF_REPLAY, --device -1), and the text must be the oracle's own hal2maf text.
usage: python profiles/scripts/r04_cpu_maf_soak.py [first seed] [alignments]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import halfix
ORACLE = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
TOOL = os.path.join(ROOT, "hal_amd", "_build", "hal2maf")
LIB = os.environ.get("HGX_SOAK_PRELOAD", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))  # (e.g. a sanitizer build behind its runtime)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
exports = different = 0
with tempfile.TemporaryDirectory() as tmp:
    img, rec, want, got = (os.path.join(tmp, n) for n in ("a.hgx", "rec.bin", "want.maf", "got.maf"))
    for seed in range(first, first + count):
        rng = random.Random(seed)
        al = halfix.random_multiseq_alignment(seed, n_genomes=rng.randint(2, 9), max_children=rng.randint(1, 3), root_len=rng.choice([60, 200, 700, 2500]))
        halfix.write_hgx(img, al)
        for gd in al:
            name, leaf = gd["name"], not gd["children"]
            for _ in range(2):
                col, host = [], []  # options of the columns, options of the block state machine only
                if rng.random() < 0.3:
                    col.append("--noDupes")
                if leaf and rng.random() < 0.5:
                    col.append("--noAncestors")
                if rng.random() < 0.4:  # (which columns the iterator walks and writes: marks 2 and 3 of the batches)
                    col.append("--unique")
                if rng.random() < 0.25 and len(al) > 2:  # (the columns' scope: the spanning tree of the targets and the reference)
                    col += ["--targetGenomes", ",".join(g["name"] for g in rng.sample(al, rng.randint(1, min(3, len(al)))))]
                if rng.random() < 0.5:
                    host += ["--maxBlockLen", str(rng.choice([1, 2, 5, 17, 100]))]
                if rng.random() < 0.3:
                    host.append("--keepEmptyRefBlocks")
                if rng.random() < 0.3:
                    host.append("--onlySequenceNames")
                chunk = rng.choice([1, 3, 16, 101, 1 << 21])
                subprocess.run([ORACLE, "columns", img, name, "--batches", rec, "--chunk", str(chunk)] + col, check=True, stderr=subprocess.DEVNULL)
                subprocess.run([ORACLE, "maf", img, want, "--refGenome", name] + col + host, check=True)
                r = subprocess.run([TOOL, "--device", "-1", "--refGenome", name] + col + host + [img, got],
                                   env=dict(os.environ, LD_PRELOAD=LIB, HGX_MAF_REPLAY=rec, HGX_MAF_THREADS=os.environ.get("HGX_MAF_THREADS", "")), stderr=subprocess.PIPE)
                exports += 1
                a, b = open(want).read(), (open(got).read() if r.returncode == 0 else None)
                if a != b:
                    different += 1
                    print("DIFFERENT seed %d genome %s chunk %d %s rc %d %s" % (seed, name, chunk, col + host, r.returncode, r.stderr.decode()[-300:]), flush=True)
        if (seed - first) % 25 == 24:
            print("alignments %d exports %d different %d" % (seed - first + 1, exports, different), flush=True)
print("alignments %d exports %d different %d" % (count, exports, different))
sys.exit(1 if different else 0)
