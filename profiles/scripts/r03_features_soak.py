"""Randomised parity soak of the entry points added in round 3, HIP path vs oracle: hal2maf --maxRefGap (the replayed iterator
stack), --global, --printTree, and halGetBlocksInTargetRange with adjacencies, over many seeds and shapes of both generators.
Not part of the test suite; run as  python profiles/scripts/r03_features_soak.py [seconds]."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import hal_amd as hal
import halfix

ORA = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "11")))
t0 = time.time()
tmp = tempfile.mkdtemp()
NO_TREE = ("has no parent in a genome with bottom segments", "no block entry continues at this base")
counts = dict(gap=0, glob=0, tree=0, tree_refused=0, viz=0)


def omaf(img, *a):
    out = os.path.join(tmp, "o.maf")
    r = subprocess.run([ORA, "maf", img, out] + list(a), stderr=subprocess.PIPE)
    return (open(out).read() if r.returncode == 0 else None), r.stderr.decode()


rounds = 0
while time.time() - t0 < budget:
    rounds += 1
    img = os.path.join(tmp, "a.hgx")
    seed = int(rng.integers(0, 10000))
    if rng.integers(0, 2) == 0:
        halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=int(rng.integers(3, 10)), max_children=int(rng.integers(1, 4)),
                                                              root_len=int(rng.integers(100, 900))))
        al = hal.Alignment.open(img, device=0)
        kind = "multiseq seed %d" % seed
    else:
        lo = int(rng.integers(3, 40)); hi = lo + int(rng.integers(1, 100))
        nlo = int(rng.integers(20, 300)); nhi = nlo + int(rng.integers(1, 500))
        opts = hal.RandOptions(mean_degree=float(rng.uniform(1.1, 2.5)), max_branch_length=float(rng.choice([0.5, 1.5, 3.0])), min_genomes=2,
                               max_genomes=int(rng.integers(2, 10)), min_segment_length=lo, max_segment_length=hi, min_segments=nlo, max_segments=nhi,
                               seed=seed, with_dna=True)
        try:
            al = hal.Alignment.random(opts, device=0)
        except hal.HgxError as e:
            if "runaway tree" in str(e):
                continue
            raise
        al.save(img)
        kind = "randgen seed %d seg %d-%d n %d-%d" % (seed, lo, hi, nlo, nhi)
    n = al.num_genomes
    # --maxRefGap
    for _ in range(2):
        g = int(rng.integers(0, n))
        if al.genome_length(g) == 0:
            continue
        name = al.genome_name(g)
        gap = int(rng.choice([2, 9, 60, 1000]))
        flags, kw = ["--refGenome", name, "--maxRefGap", str(gap)], dict(max_ref_gap=gap)
        if rng.integers(0, 3) == 0: flags.append("--noDupes"); kw["no_dupes"] = True
        if rng.integers(0, 3) == 0: flags.append("--unique"); kw["unique"] = True
        if rng.integers(0, 3) == 0: flags += ["--maxBlockLen", "13"]; kw["max_block_len"] = 13
        if rng.integers(0, 4) == 0: flags.append("--onlyOrthologs"); kw["only_orthologs"] = True
        want, _ = omaf(img, *flags)
        assert want is not None and al.maf_export(g, **kw) == want, ("maxRefGap", kind, name, kw)
        counts["gap"] += 1
    # --global
    if rng.integers(0, 3) == 0:
        flags, kw = ["--global"], {}
        if rng.integers(0, 2): flags.append("--noDupes"); kw["no_dupes"] = True
        if rng.integers(0, 2): flags.append("--noAncestors"); kw["no_ancestors"] = True
        want, _ = omaf(img, *flags)
        assert want is not None and al.maf_export_global(**kw) == want, ("global", kind, kw)
        counts["glob"] += 1
    # --printTree
    g = int(rng.integers(0, n))
    if al.genome_length(g) > 0:
        name = al.genome_name(g)
        flags, kw = ["--refGenome", name, "--printTree"], dict(print_tree=True)
        if rng.integers(0, 3) == 0: flags += ["--maxBlockLen", "11"]; kw["max_block_len"] = 11
        want, err = omaf(img, *flags)
        try:
            got = al.maf_export(g, **kw)
            assert want is not None and got == want, ("printTree", kind, name, kw)
            counts["tree"] += 1
        except hal.HgxError as e:
            assert want is None and any(m in str(e) and m in err for m in NO_TREE), ("printTree refusal", kind, name, kw, str(e), err)
            counts["tree_refused"] += 1
    # halGetBlocksInTargetRange
    for _ in range(3):
        q, t = int(rng.integers(0, n)), int(rng.integers(0, n))
        tseqs = [s for s in al.sequences(t) if s[2] > 0]
        if not tseqs or al.genome_length(q) == 0:
            continue
        chrom, _, length = tseqs[int(rng.integers(0, len(tseqs)))]
        size = int(min(length, rng.choice([1, 30, 400, 5000])))
        a = int(rng.integers(0, length - size + 1))
        dup_mode = int(rng.integers(0, 3)); adj = bool(rng.integers(0, 2)); seq = bool(rng.integers(0, 4) == 0)
        cmd = [ORA, "blockviz", img, al.genome_name(q), al.genome_name(t), chrom, str(a), str(a + size), "--dupMode", str(dup_mode)]
        if seq: cmd.append("--doSeq")
        if not adj: cmd.append("--noAdj")
        want = subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout.decode()
        blocks, dupes = al.blocks_in_target_range(al.genome_name(q), al.genome_name(t), chrom, a, a + size, seq=seq, dup_mode=dup_mode, adjacencies=adj)
        assert hal.format_block_results(blocks, dupes) == want, ("blockviz", kind, al.genome_name(q), al.genome_name(t), chrom, a, size, dup_mode, adj, seq)
        counts["viz"] += 1
    del al
print("features soak: %d alignments: %d --maxRefGap exports, %d --global, %d --printTree (+ %d refused by both), %d halGetBlocksInTargetRange "
      "calls, all identical to the oracle (%.0f s)" % (rounds, counts["gap"], counts["glob"], counts["tree"], counts["tree_refused"], counts["viz"],
                                                       time.time() - t0))
