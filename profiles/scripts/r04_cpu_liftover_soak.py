"""halLiftover's host side — the parallel text path for inputs of one column count (hal_amd/csrc/hgx_liftover_text.cpp) and the general path — BED12 blocks, PSL output, lines of mixed column counts, one BedLine at a time: hal_amd/csrc/hgx_liftover_host.cpp,
Liftover::convertGeneral, the lines of a batch dealt to the host's threads — soaked on a machine WITHOUT a GPU: for random alignments
(tests/halfix.py) and random BED inputs the oracle writes both the expected text and, with --records, every lifted interval's records as
the device hands them to the host side; the profiling build of the library (make -C hal_amd/csrc hostprof-lib) plays the records back
(HGX_LIFT_REPLAY, halLiftover --device -1, HGX_TEXT_GENERAL=1) with batches of 1 .. 4 M intervals and the text parsed in pieces of 40
bytes .. 16 KB a thread, and the text must be the oracle's.
usage: python profiles/scripts/r04_cpu_liftover_soak.py [first seed] [alignments]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import halfix
ORACLE = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
TOOL = os.path.join(ROOT, "hal_amd", "_build", "halLiftover")
LIB = os.environ.get("HGX_SOAK_PRELOAD", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))  # (e.g. a sanitizer build behind its runtime)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def bed_lines(rng, seqs, n, cols, extras):
    """n lines of `cols` columns (12: with blocks, in any order) over the genome's sequences, a few of them to be skipped"""
    out = []
    for i in range(n):
        name, _, length = rng.choice(seqs)[:3]
        if length < 2:
            continue
        ln = rng.randint(1, min(length, rng.choice([5, 40, 300])))
        a = rng.randrange(0, length - ln + 1)
        b = a + ln
        if rng.random() < 0.02:
            name = "nowhere%d" % rng.randint(0, 2)   # unknown sequence: skipped, reported once
        elif rng.random() < 0.02:
            b = length + rng.randint(1, 9)            # past the end: skipped
        strand = rng.choice("+-.") if cols != 12 else rng.choice("+-")
        odd = lambda v: rng.choice([" %d", "+%d", "%dx", "%d ", "0%d"]) % v if rng.random() < 0.03 else str(v)  # (read like operator>>)
        f = [name, odd(a), odd(b), "n%d" % i, odd(rng.randint(0, 1000)), strand, str(a if rng.random() < 0.5 else 0),
             str(b if rng.random() < 0.5 else 0), rng.choice(["255,0,0", "7", "1,2", "3,4,5,"])]
        if cols == 12:
            nb = rng.randint(1, min(5, max(1, ln // 2)))
            cuts = sorted(rng.sample(range(1, ln), 2 * nb - 1)) if ln > 2 * nb else list(range(1, 2 * nb))
            edges = [0] + cuts + [ln]
            sizes = [edges[2 * k + 1] - edges[2 * k] for k in range(nb)]
            starts = [edges[2 * k] for k in range(nb)]
            if ln <= 2 * nb:
                nb, sizes, starts = 1, [ln], [0]
            order = list(range(nb))
            rng.shuffle(order)
            f += [str(nb), ",".join(str(sizes[k]) for k in order) + ",", ",".join(str(starts[k]) for k in order) + ","]
        else:
            f = f[:cols] + ["x%d" % k if k % 2 == 0 else "" for k in range(extras)]
        out.append("\t".join(f))
    return out


exports = different = 0
with tempfile.TemporaryDirectory() as tmp:
    img, bed, rec, want, got = (os.path.join(tmp, n) for n in ("a.hgx", "in.bed", "rec.bin", "want.txt", "got.txt"))
    for seed in range(first, first + count):
        rng = random.Random(seed)
        al = halfix.random_multiseq_alignment(seed, n_genomes=rng.randint(2, 8), max_children=rng.randint(1, 3), root_len=rng.choice([80, 300, 1200]))
        halfix.write_hgx(img, al)
        for _ in range(6):
            src, tgt = rng.sample(al, 2)
            seqs = [s for s in src["seqs"] if s[2] > 0]
            if not seqs:
                continue
            shape = rng.choice(["bed12", "bed12", "mixed", "uniform", "extras", "thick"])
            opts, body = [], []
            if shape == "bed12":
                body = bed_lines(rng, seqs, rng.choice([3, 80, 400]), 12, 0)
            elif shape == "mixed":  # (the reference's line object keeps the fields of longer lines before)
                for _k in range(rng.randint(2, 4)):
                    body += bed_lines(rng, seqs, rng.choice([5, 120]), rng.choice([3, 4, 5, 6, 7, 8, 9, 12]), 0)
            elif shape == "thick":  # (lines of seven columns take their thick end from the last longer line: it decides what their thick start becomes)
                body = (bed_lines(rng, seqs, rng.choice([3, 150]), rng.choice([8, 9]), 0) + bed_lines(rng, seqs, rng.choice([0, 2, 150]), rng.choice([3, 4, 5, 6]), 0) +
                        bed_lines(rng, seqs, rng.choice([5, 200]), 7, 0) + bed_lines(rng, seqs, rng.choice([0, 40]), rng.choice([3, 6, 8]), 0))
            elif shape == "uniform":
                body = bed_lines(rng, seqs, rng.choice([10, 300]), rng.choice([3, 4, 5, 6, 7, 8, 9]), 0)
            else:
                cols = rng.choice([3, 4, 6, 9])
                body = bed_lines(rng, seqs, rng.choice([10, 200]), cols, rng.randint(1, 3))
                opts += ["--bedType", str(cols)]
            if rng.random() < 0.5:
                opts.append(rng.choice(["--outPSL", "--outPSLWithName"]))
            if rng.random() < 0.3:
                opts.append("--noDupes")
            if body and rng.random() < 0.12:  # a malformed line: what was lifted before it is written, the conversion ends there
                body.insert(rng.randrange(len(body) + 1), rng.choice(["chrBroken\t5", "%s\tx\t9" % seqs[0][0], "%s\t7\t3\tn\t0\t+" % seqs[0][0]]))
            text = "\n".join(body) + "\n"
            if rng.random() < 0.2:
                text = "\n  \n" + text.replace("\n", "\n\n", 3)
            open(bed, "w").write(text)
            r0 = subprocess.run([ORACLE, "liftover", img, src["name"], bed, tgt["name"], want, "--records", rec] + opts, stderr=subprocess.PIPE)
            batch = rng.choice(["1", "7", "100", "4000000"])
            env = dict(os.environ, LD_PRELOAD=LIB, HGX_LIFT_REPLAY=rec, HGX_BATCH_LINES=batch, HGX_PARSE_PIECE=rng.choice(["40", "300", "16384"]),
                       HGX_TEXT_THREADS=rng.choice(["1", "3", "8"]))
            if rng.random() < 0.5:  # (the records in the device's 8-byte form, when they fit it)
                env["HGX_REPLAY_PACKED"] = "1"
            if rng.random() < 0.5:  # (else: inputs of one column count, BED out, take the parallel text path of hgx_liftover_text.cpp)
                env["HGX_TEXT_GENERAL"] = "1"
            psl = any(o.startswith("--outPSL") for o in opts)
            if not psl and shape in ("uniform", "extras") and rng.random() < 0.5:
                # several handles (hgx_liftover_convert_multi: the lines shared out, one round or several, the shares' texts collated)
                env2 = dict(env, HGX_LIB_PATH=LIB.split()[-1], LD_PRELOAD=" ".join(LIB.split()[:-1]))
                env2.pop("HGX_REPLAY_PACKED", None)
                env2.pop("HGX_TEXT_GENERAL", None)
                r1 = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "r04_cpu_liftover_multi.py"), img, src["name"], bed, tgt["name"], got,
                                     str(rng.randint(2, 4))] + opts, env=env2, stderr=subprocess.PIPE)
            else:
                r1 = subprocess.run([TOOL, "--device", "-1"] + opts + [img, src["name"], bed, tgt["name"], got], env=env, stderr=subprocess.PIPE)
            exports += 1
            a = open(want).read() if os.path.exists(want) else None
            b = open(got).read() if os.path.exists(got) else None
            same = a == b and (r0.returncode == 0) == (r1.returncode == 0)
            if not same:
                different += 1
                print("DIFFERENT seed %d %s -> %s %s %s batch %s general %s rc %d / %d\n  oracle: %s\n  library: %s" % (
                    seed, src["name"], tgt["name"], shape, opts, batch, env.get("HGX_TEXT_GENERAL"), r0.returncode, r1.returncode, r0.stderr.decode()[-200:],
                    r1.stderr.decode()[-200:]), flush=True)
            for f in (want, got):
                if os.path.exists(f):
                    os.remove(f)
        if (seed - first) % 25 == 24:
            print("alignments %d conversions %d different %d" % (seed - first + 1, exports, different), flush=True)
print("alignments %d conversions %d different %d" % (count, exports, different))
sys.exit(1 if different else 0)
