"""Randomised parity soak, HIP path vs oracle: many seeds and shapes of halRandGen alignments and of the independent
multi-sequence generator, every genome pair, random option mixes.  Not part of the test suite (minutes of GPU time);
run as  python profiles/scripts/soak_parity.py [seconds]."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import hal_amd as hal
import halfix
from util import oracle_liftover, random_bed

ORA = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0 = time.time()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
checks = lines = 0
tmp = tempfile.mkdtemp()

def omaf(img, *a):
    out = os.path.join(tmp, "o.maf"); subprocess.check_call([ORA, "maf", img, out] + list(a)); return open(out).read()
def odepth(img, *a):
    out = os.path.join(tmp, "o.wig"); subprocess.check_call([ORA, "depth", img, a[0], out] + list(a[1:])); return open(out).read()

round_ = 0
while time.time() - t0 < budget:
    round_ += 1
    img = os.path.join(tmp, "a.hgx")
    if rng.integers(0, 3) == 0:
        seed = int(rng.integers(0, 10000))
        halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=int(rng.integers(3, 12)), max_children=int(rng.integers(1, 4)),
                                                              root_len=int(rng.integers(100, 1500))))
        al = hal.Alignment.open(img, device=0)
        has_dna = True
        kind = "multiseq seed %d" % seed
    else:
        seed = int(rng.integers(0, 10000))
        lo = int(rng.integers(3, 60)); hi = lo + int(rng.integers(1, 150))
        nlo = int(rng.integers(20, 800)); nhi = nlo + int(rng.integers(1, 1500))
        opts = hal.RandOptions(mean_degree=float(rng.uniform(1.1, 2.5)), max_branch_length=float(rng.choice([0.5, 1.5, 3.0, 5.0])),
                               min_genomes=2, max_genomes=int(rng.integers(2, 16)), min_segment_length=lo, max_segment_length=hi,
                               min_segments=nlo, max_segments=nhi, seed=seed, with_dna=bool(rng.integers(0, 2)))
        try:
            al = hal.Alignment.random(opts, device=0)
        except hal.HgxError as e:
            if "runaway tree" in str(e):
                continue  # the reference generator does not terminate for this seed either
            raise
        al.save(img)
        has_dna = opts.with_dna
        kind = "randgen seed %d seg %d-%d n %d-%d" % (seed, lo, hi, nlo, nhi)
    n = al.num_genomes
    for _ in range(6):
        s, t = int(rng.integers(0, n)), int(rng.integers(0, n))
        seqs = [q for q in al.sequences(s) if q[2] > 0]
        if not seqs:
            continue
        name, _, length = seqs[int(rng.integers(0, len(seqs)))]
        mx = int(rng.choice([30, 300, 3000, 30000]))
        bed = random_bed(name, length, int(rng.integers(1, 400)), 1, min(mx, length), int(rng.integers(0, 1 << 30)), strands="+-.")
        nd = bool(rng.integers(0, 3) == 0)
        kw, okw = {"traverse_dupes": not nd}, {"no_dupes": nd}
        m = al.mrca(s, t)
        if rng.integers(0, 4) == 0 and al.genome_parent(m) >= 0:
            kw["coalescence_limit"] = al.genome_parent(m); okw["coalescence_limit"] = al.genome_name(al.genome_parent(m))
        if has_dna and rng.integers(0, 4) == 0:
            bedp = bed.replace("\t.\n", "\t+\n")
            got = hal.liftover_convert(al, s, bedp, t, out_psl=True, **kw)
            want = oracle_liftover(ORA, img, al.genome_name(s), al.genome_name(t), bedp, tmp, psl=True, **okw)
        else:
            got = hal.liftover_convert(al, s, bed, t, **kw)
            want = oracle_liftover(ORA, img, al.genome_name(s), al.genome_name(t), bed, tmp, **okw)
        assert got == want, ("liftover", kind, al.genome_name(s), al.genome_name(t), kw)
        checks += 1; lines += got.count("\n")
    g = int(rng.integers(0, n))
    if al.genome_length(g) > 0:
        name = al.genome_name(g)
        assert al.alignment_depth(g) == odepth(img, name), ("depth", kind, name)
        flags, kw = ["--refGenome", name], {}
        if rng.integers(0, 2): flags.append("--noDupes"); kw["no_dupes"] = True
        if rng.integers(0, 2): flags.append("--unique"); kw["unique"] = True
        if rng.integers(0, 3) == 0: flags += ["--maxBlockLen", "17"]; kw["max_block_len"] = 17
        if rng.integers(0, 3) == 0: flags.append("--onlyOrthologs"); kw["only_orthologs"] = True
        assert al.maf_export(g, **kw) == omaf(img, *flags), ("maf", kind, name, kw)
        checks += 2
    del al
print("soak: %d rounds, %d checks, %d liftover lines, all identical to the oracle (%.0f s)" % (round_, checks, lines, time.time() - t0))
