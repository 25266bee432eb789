"""hal2maf's host state machine against the oracle without a GPU: `dump` (on a GPU box, HGX_LIB_PATH=hal_amd/libhgx_hostprof.so)
records the device's batches of a fixed list of exports, `replay` (anywhere, same library) plays them back through the state
machine with device -1 and compares every export with the oracle's text.
usage: HGX_LIB_PATH=hal_amd/libhgx_hostprof.so python profiles/scripts/maf_replay_cases.py dump|replay <file>"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mode, path = sys.argv[1], sys.argv[2]
os.environ["HGX_MAF_DUMP" if mode == "dump" else "HGX_MAF_REPLAY"] = path
os.environ.setdefault("HGX_LIB_PATH", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))
import halfix
import hal_amd as hal

ORACLE = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
device = 0 if mode == "dump" else -1
bad = n = 0
with tempfile.TemporaryDirectory() as T:
    def oracle(img, *args):
        out = os.path.join(T, "o.maf")
        subprocess.check_call([ORACLE, "maf", img, out] + list(args))
        return open(out).read()

    for seed in range(8):
        img = os.path.join(T, "ms%d.hgx" % seed)
        halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=6 + seed % 3))
        al = hal.Alignment.open(img, device=device)
        for g in range(al.num_genomes):
            nm = al.genome_name(g)
            leaf = not al.genome_children(g)
            cases = [({}, []), (dict(max_block_len=5, keep_empty_ref_blocks=True), ["--maxBlockLen", "5", "--keepEmptyRefBlocks"]),
                     (dict(no_dupes=True), ["--noDupes"]), (dict(only_orthologs=True, only_sequence_names=True), ["--onlyOrthologs", "--onlySequenceNames"])]
            if leaf:
                cases.append((dict(no_ancestors=True, max_block_len=0), ["--noAncestors", "--maxBlockLen", "0"]))
            for kw, args in cases:
                got = al.maf_export(g, **kw)
                want = oracle(img, "--refGenome", nm, *args)
                n += 1
                if got != want:
                    bad += 1
                    print("DIFFERENT: seed", seed, nm, kw)
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=8, max_segment_length=40,
                           min_segments=300, max_segments=700, seed=2, with_dna=True)
    img = os.path.join(T, "rand.hgx")
    if mode == "dump":
        hal.Alignment.random(opts, device=0).save(img)
        import shutil
        shutil.copy(img, path + ".rand.hgx")
    else:
        img = path + ".rand.hgx"
    al = hal.Alignment.open(img, device=device)
    for g in range(al.num_genomes):
        for kw, args in (({}, []), (dict(max_block_len=11, no_dupes=True), ["--maxBlockLen", "11", "--noDupes"])):
            got = al.maf_export(g, **kw)
            n += 1
            if got != oracle(img, "--refGenome", al.genome_name(g), *args):
                bad += 1
                print("DIFFERENT: rand", al.genome_name(g), kw)
print("%d exports, %d different from the oracle" % (n, bad))
sys.exit(1 if bad else 0)
