"""halAlignmentDepth's wig text end to end (config 2's Genome_9, config 5's Genome_44): the library's own clock, three runs each"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
g = al.genome_id("Genome_9")
for rep in range(4):
    t = time.perf_counter()
    b = al.alignment_depth_bytes(g)
    dt = time.perf_counter() - t
    print("cfg2 wig: %d bytes in %.2f ms" % (b, dt * 1e3), flush=True)
al5 = hal_amd.Alignment.random(bench.workload_options(1.0, "cfg4"), device=0)
g5 = al5.genome_id("Genome_44")
for rep in range(3):
    t = time.perf_counter()
    b = al5.alignment_depth_bytes(g5)
    dt = time.perf_counter() - t
    print("cfg5 wig: %d bytes in %.2f ms" % (b, dt * 1e3), flush=True)
