"""Shared test helpers: BED generation, running the oracle."""
import os
import subprocess

import numpy as np


def random_bed(seq_name, seq_len, n, min_len, max_len, seed, strands="+-", bed6=True):
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=n)
    lens = np.minimum(lens, seq_len)
    starts = (rng.random(n) * (seq_len - lens + 1)).astype(np.int64)
    st = rng.integers(0, len(strands), size=n)
    lines = []
    for i in range(n):
        if bed6:
            lines.append("%s\t%d\t%d\tq%d\t0\t%s\n" % (seq_name, starts[i], starts[i] + lens[i], i, strands[st[i]]))
        else:
            lines.append("%s\t%d\t%d\n" % (seq_name, starts[i], starts[i] + lens[i]))
    return "".join(lines)


def oracle_liftover(oracle_bin, image_path, src, tgt, bed_text, tmpdir, no_dupes=False, bed_type=0, stats=False, psl=False,
                    psl_with_name=False, coalescence_limit=None):
    inp = os.path.join(str(tmpdir), "oracle_in.bed")
    out = os.path.join(str(tmpdir), "oracle_out.bed")
    with open(inp, "w") as f:
        f.write(bed_text)
    cmd = [oracle_bin, "liftover", image_path, src, inp, tgt, out]
    if no_dupes:
        cmd.append("--noDupes")
    if bed_type:
        cmd += ["--bedType", str(bed_type)]
    if coalescence_limit:
        cmd += ["--coalescenceLimit", coalescence_limit]
    if stats:
        cmd.append("--stats")
    if psl_with_name:
        cmd.append("--outPSLWithName")
    elif psl:
        cmd.append("--outPSL")
    res = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(out) as f:
        text = f.read()
    return (text, res.stdout.decode()) if stats else text
