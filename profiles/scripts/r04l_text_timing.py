"""Liftover::convert (BED text -> BED text) on the cfg2 batch, with the text path's own phase timing (HGX_TEXT_TIMING)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hal_amd
from bench import workload_options, make_queries
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
seq_name, ss, length = al.sequences(src)[0]
nq = 1000000
st, ln, sd = make_queries(length, nq, 1234)
bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(a), int(a + b), chr(int(c))) for a, b, c in zip(st.numpy(), ln.numpy(), sd.numpy())).encode()
hal_amd.liftover_convert_bytes(al, src, bed[:bed.index(b"\n", 4000000) + 1], tgt)
hal_amd.liftover_convert_bytes(al, src, bed, tgt)
os.environ["HGX_TEXT_TIMING"] = "1"
for _ in range(4):
    t0 = time.perf_counter()
    hal_amd.liftover_convert_bytes(al, src, bed, tgt, count_lines=False)
    print("convert %.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
# where the time outside the library's phases goes: the call itself against releasing its 119 MB of text
import ctypes as C
from hal_amd._lib import lib
for _ in range(3):
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    t0 = time.perf_counter()
    lib.hgx_liftover_convert(al._h, src, bed, len(bed), tgt, 0, 1, 0, 0, -1, C.byref(out), C.byref(n), C.byref(err))
    t1 = time.perf_counter()
    lib.hgx_free(out)
    t2 = time.perf_counter()
    print("call %.2f ms, hgx_free %.2f ms, %d bytes" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), n.value), flush=True)
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
