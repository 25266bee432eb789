"""The alternative code paths behind environment switches give the same bytes as the defaults: one launch per up level
(HGX_LEVEL_SYNC_UP), the composed up table forced on small batches (HGX_COMPOSED_UP), the per-column depth kernel (HGX_COLUMNS_PER_BASE), the per-column MAF path (HGX_MAF_PER_COLUMN),
the run-compressed MAF path on MafBlock's own containers (HGX_MAF_MAP_STATE) and in small device batches (HGX_MAF_CHUNK), the
64-bit instantiations (HGX_FORCE_WIDE).  Each switch is read once per process, so every variant runs in its own process."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import hal_amd as hal
from util import random_bed
opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=8, max_segment_length=40,
                       min_segments=300, max_segments=700, seed=2, with_dna=True)
al = hal.Alignment.random(opts, device=0)
h = hashlib.sha256()
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
name, _, length = al.sequences(src)[0]
bed = random_bed(name, length, 3000, 1, 400, 5, strands="+-.")
h.update(hal.liftover_convert(al, src, bed, tgt).encode())
h.update(hal.liftover_convert(al, src, bed, al.genome_id("Genome_0"), traverse_dupes=False).encode())
for g in ("Genome_9", "Genome_0", "Genome_3"):
    gi = al.genome_id(g)
    h.update(al.alignment_depth(gi).encode())
    h.update(al.alignment_depth(gi, step=7, count_dupes=True).encode())
    h.update(al.maf_export(gi).encode())
    h.update(al.maf_export(gi, no_dupes=True, max_block_len=11).encode())
import os, tempfile
import halfix
with tempfile.TemporaryDirectory() as T:  # several sequences per genome, duplications, short blocks
    for seed in (0, 1, 3):
        p = os.path.join(T, "ms%%d.hgx" %% seed)
        halfix.write_hgx(p, halfix.random_multiseq_alignment(seed, n_genomes=6))
        al2 = hal.Alignment.open(p, device=0)
        for gi in range(al2.num_genomes):
            h.update(al2.maf_export(gi).encode())
            h.update(al2.maf_export(gi, max_block_len=5, keep_empty_ref_blocks=True).encode())
            h.update(al2.maf_export(gi, only_orthologs=True, only_sequence_names=True).encode())
st = al.columns_depth_stats(src, 0, al.genome_length(src))
assert st["top_derefs"] > 0 and st["bottom_derefs"] > 0
print(h.hexdigest())
''' % (ROOT, os.path.join(ROOT, "tests"))


def _digest(**env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=e, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    return out.strip().splitlines()[-1]


def test_switchable_paths_agree():
    base = _digest()
    assert len(base) == 64
    for env in ({"HGX_LEVEL_SYNC_UP": "1"}, {"HGX_COLUMNS_PER_BASE": "1"}, {"HGX_MAF_PER_COLUMN": "1"}, {"HGX_MAF_MAP_STATE": "1"},
                {"HGX_MAF_CHUNK": "97"}, {"HGX_FORCE_WIDE": "1"},
                {"HGX_COMPOSED_UP": "1"}, {"HGX_COMPOSED_UP": "1", "HGX_FORCE_WIDE": "1"}):
        assert _digest(**env) == base, env


def test_plan_timing_modes(hal, tmp_path, monkeypatch):
    import torch
    monkeypatch.setenv("HGX_COMPOSED_UP", "1")  # (a walking plan would switch to its table in the middle of the test)
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
                           min_segments=200, max_segments=600, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, ss, length = al.sequences(src)[0]
    n = 2000
    starts = torch.randint(0, length - 200, (n,))
    gs, ge, st = (starts + ss).cuda(), (starts + 150 + ss).cuda(), torch.full((n,), ord("+"), dtype=torch.uint8).cuda()
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    plan.run(gs, ge, st)
    last = plan.kernel_times()
    walk = "k_up_chain" if "k_up_chain" in last else "k_lift_merged" if "k_lift_merged" in last else "k_locate_through"
    ctr = "k_down_ring" if walk == "k_up_chain" else walk
    assert last[walk]["launches"] == 1 and last[ctr]["top_derefs"] > 0
    plan.set_timing(2)  # accumulate
    for _ in range(3):
        plan.run(gs, ge, st)
    acc = plan.kernel_times()
    assert acc[walk]["launches"] == 3 and acc[ctr]["top_derefs"] == 3 * last[ctr]["top_derefs"]
    assert plan.kernel_times() == {}  # the window restarts after a read
    plan.set_timing(0)
    plan.run(gs, ge, st)
    assert plan.kernel_times() == {}
    plan.set_timing(1)
    plan.run(gs, ge, st)
    assert plan.kernel_times()[walk]["launches"] == 1
