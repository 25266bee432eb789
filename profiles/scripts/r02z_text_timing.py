"""Stages of hgx_liftover_convert on the cfg2 batch (HGX_TEXT_TIMING)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hal_amd
from bench import workload_options, make_queries

al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
seq_name, ss, length = al.sequences(src)[0]
starts, lens, strand = make_queries(length, 1000000, 1234)
bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(a), int(a + b), chr(int(c))) for a, b, c in zip(starts.numpy(), lens.numpy(), strand.numpy())).encode()
hal_amd.liftover_convert_bytes(al, src, bed, tgt)
os.environ["HGX_TEXT_TIMING"] = "1"
for _ in range(4):
    t0 = time.perf_counter()
    nb, nl = hal_amd.liftover_convert_bytes(al, src, bed, tgt)
    print("convert: %.1f ms, %d bytes" % ((time.perf_counter() - t0) * 1e3, nb), flush=True)
