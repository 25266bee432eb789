import sys, time, json
sys.path.insert(0,'.')
import torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
g = al.genome_id("Genome_9"); n = al.genome_length(g)
out = torch.empty(n, dtype=torch.int32, device='cuda')
for mode in (False, True):
    for rep in range(3):
        ms = al.columns_depth_device(g, 0, n, out.data_ptr(), count_dupes=mode)
    print("cfg2 Genome_9 depth countDupes=%s: %d columns, %.3f ms, %.1f M columns/s, mean depth %.2f" % (mode, n, ms, n/ms/1e3, out.float().mean().item()))
g0 = al.genome_id("Genome_0"); n0 = al.genome_length(g0)
out0 = torch.empty(n0, dtype=torch.int32, device='cuda')
ms = al.columns_depth_device(g0, 0, n0, out0.data_ptr())
print("cfg2 root depth: %d columns %.3f ms %.1f M col/s" % (n0, ms, n0/ms/1e3))
t=time.time()
o = hal_amd.RandOptions(mean_degree=2.0, max_branch_length=3.0, min_genomes=2, max_genomes=50, min_segment_length=50, max_segment_length=200, min_segments=700000, max_segments=1400000, seed=0, with_dna=False)
al5 = hal_amd.Alignment.random(o, device=0)
print("cfg5 gen %.1fs" % (time.time()-t), al5.newick[:200])
depths = {i: 0 for i in range(al5.num_genomes)}
def depth(i):
    d=0
    while al5.genome_parent(i) >= 0: i = al5.genome_parent(i); d+=1
    return d
leaf = max(range(al5.num_genomes), key=depth)
n = al5.genome_length(leaf)
out = torch.empty(n, dtype=torch.int32, device='cuda')
for rep in range(2):
    ms = al5.columns_depth_device(leaf, 0, n, out.data_ptr())
print("cfg5 %s (depth %d) depth: %d columns %.3f ms %.1f M col/s mean %.2f max %d" % (al5.genome_name(leaf), depth(leaf), n, ms, n/ms/1e3, out.float().mean().item(), out.max().item()))
