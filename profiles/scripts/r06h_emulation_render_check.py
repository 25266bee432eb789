import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import hal_amd as hal
import halfix
n_diff = 0
total = 0
for seed in (1, 2, 3, 5):
    o = dict(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
             min_segments=200, max_segments=600, seed=seed, with_dna=True)
    al = hal.Alignment.random(hal.RandOptions(**o), device=0)
    for g in range(al.num_genomes):
        for kw in ({}, {"no_ancestors": True}, {"keep_empty_ref_blocks": True} , {"max_block_len": 7}, {"only_sequence_names": True, "no_dupes": True}, {"max_block_len": 3, "keep_empty_ref_blocks": True, "no_ancestors": True}, {"unique": True}):
            try:
                os.environ["HGX_MAF_DEVICE_RENDER"] = "1"; os.environ["HGX_MAF_DEVICE_RENDER_MIN_BLOCKS"] = "1"
                a = al.maf_export(g, **kw)
                os.environ["HGX_MAF_DEVICE_RENDER"] = "0"
                b = al.maf_export(g, **kw)
            except (TypeError, hal.HgxError) as e:
                continue
            total += 1
            if a != b:
                n_diff += 1
                print("DIFFERENT", seed, g, kw, len(a), len(b))
                for i,(x,y) in enumerate(zip(a.split("\n"), b.split("\n"))):
                    if x != y:
                        print(i, repr(x[:200])); print(i, repr(y[:200])); break
print("exports", total, "different", n_diff)
