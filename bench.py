#!/usr/bin/env python3
"""bench.py — halLiftover block-mapping hot path on MI355X (BASELINE.json configs[1]).

Workload (synthetic, generated in-process by hal_amd's restatement of halRandGen):
  10-genome alignment, ~100 Mb and ~1 M segments per genome (halRandGen --seed 2 --minGenomes 2 --maxGenomes 10
  --meanDegree 1.5 --minSegmentLength 50 --maxSegmentLength 200 --minSegments 700000 --maxSegments 1400000
  --maxBranchLength 3, DNA draws skipped), 1 M BED6 intervals of 50..1000 bp, random strand, lifted from the
  deepest leaf Genome_9 to Genome_2 (5 hops up, 1 down), duplications traversed.
A "step" = one pass of the device-resident liftover over the whole interval batch (inputs already in HBM):
classify against the merged whole-path table, per-interval overlap breaking + merging + ordering, totals, record
compaction.  With --gpus N every rank lifts its own 1 M-interval shard against a replicated image (weak scaling,
no collective on the mapping path); collating every rank's records on every rank with an all-gather (RCCL) is
measured beside as `collated` (and the gather to one writer rank as `collated_to_writer`), or inside the timed step with
--exchange-in-step 1.  Those two legs run last, under a watchdog: their collectives have not met a multi-GPU node yet, and a hang or
an error there must not cost the line.

Prints one JSON line (rank 0).  `roofline` is for the kernel with the largest device time; `cpu_baseline` times the
oracle (bit-identical CPU restatement of the reference, oracle/) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def workload_options(scale, workload="cfg2", dna=False):
    import hal_amd
    if workload == "cfg4":  # BASELINE configs 4/5: the 50-genome alignment (--seed 0 --meanDegree 2 --maxGenomes 50)
        return hal_amd.RandOptions(mean_degree=2.0, max_branch_length=3.0, min_genomes=2, max_genomes=50, min_segment_length=50,
                                   max_segment_length=200, min_segments=int(700000 * scale), max_segments=int(1400000 * scale),
                                   seed=0, with_dna=dna)
    return hal_amd.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                               max_segment_length=200, min_segments=int(700000 * scale), max_segments=int(1400000 * scale),
                               seed=2, with_dna=dna)


def make_queries(length, n, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(50, 1001, (n,), generator=g, dtype=torch.int64)
    starts = (torch.rand(n, generator=g, dtype=torch.float64) * (length - 1001)).to(torch.int64)
    strand = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    return starts, lens, strand


def cpu_baseline(al, src_name, tgt_name, starts, lens, strand, seq_name, sample, all_cores_sample=0, img=None):
    """Oracle (oracle/_build/hal_oracle) on the first `sample` intervals of this rank's batch, single thread; and, as a
    fairness row, `all_cores_sample` intervals split over one oracle process per host core (the reference's own way of
    scaling: a pool of processes, stats/halStats.py:16,38)."""
    oracle = oracle_bin()
    with tempfile.TemporaryDirectory() as tmp:
        if img is None:
            img = os.path.join(tmp, "bench.hgx")
            al.save(img)

        def write_bed(path, lo, hi):
            with open(path, "w") as f:
                for i in range(lo, hi):
                    s, l = int(starts[i]), int(lens[i])
                    f.write("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, s, s + l, chr(int(strand[i]))))

        bed = os.path.join(tmp, "in.bed")
        write_bed(bed, 0, sample)
        out = subprocess.run([oracle, "liftover", img, src_name, bed, tgt_name, os.path.join(tmp, "out.bed"), "--stats"],
                             check=True, stdout=subprocess.PIPE).stdout.decode()
        st = json.loads(out)
        with open(os.path.join(tmp, "out.bed")) as f:
            text = f.read()
        multi = None
        cores = oracle_pool_size(img)
        if all_cores_sample > 0 and cores > 1:
            per = all_cores_sample // cores
            procs = []
            for c in range(cores):
                b = os.path.join(tmp, "in%d.bed" % c)
                write_bed(b, c * per, (c + 1) * per)
                procs.append(subprocess.Popen([oracle, "liftover", img, src_name, b, tgt_name, os.path.join(tmp, "out%d.bed" % c),
                                               "--stats"], stdout=subprocess.PIPE))
            secs = []
            for pr in procs:
                o, _ = pr.communicate()
                secs.append(json.loads(o.decode())["map_seconds"])
            # every process maps `per` intervals concurrently; the slowest one bounds the pool
            multi = {"value": per * cores / max(secs), "unit": "intervals/s", "cores": cores,
                     "sample": "%d intervals over %d oracle processes (mapping time of the slowest process)" % (per * cores, cores)}
    return st, text, multi


_T_BENCH = time.time()


def _mark(what):
    """a line of progress on stderr: seconds since the start, memory available (a leg that takes the box down is the last one named)"""
    avail = ""
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = ", %.0f GB of memory available" % (int(line.split()[1]) / 1048576.0)
    except OSError:
        pass
    sys.stderr.write("[bench +%.1f s%s] %s\n" % (time.time() - _T_BENCH, avail, what))
    sys.stderr.flush()


def oracle_pool_size(img_path):
    """How many oracle processes may run side by side.  Every one of them loads the whole image into memory of its own (1.1 GB for
    config 2's alignment with its DNA, several GB for the 50-genome one): one per host core of a 256-thread box is hundreds of GB,
    and a box driven out of memory dies without a word (profiles/r06_notes.md: every lost GPU box of round 5 and the first of
    round 6 were running these pools).  The pool is what fits a quarter of the memory that is free now — MemAvailable, and the
    cgroup's own limit where there is one — at 1.3 times the image's size a process, never more than the CPUs the process may keep busy (host_cpus) or 64."""
    try:
        per = int(1.3 * os.path.getsize(img_path)) + (256 << 20)
    except OSError:
        per = 4 << 30
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(lim).read().strip()
            if m != "max":
                left = int(m) - int(open(use).read().strip())
                avail = left if avail is None else min(avail, left)
        except (OSError, ValueError):
            pass
    if avail is None:
        avail = 16 << 30
    return max(1, min(host_cpus(), 64, int(0.25 * avail // per)))


def host_cpus():
    """the CPUs this process may keep busy: os.cpu_count() cut to its affinity mask and to the cgroup's CPU quota (the GPU boxes'
    containers: 256 hardware threads under cpu.max = "1600000 100000", i.e. 16 — more processes than that only take turns)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def oracle_bin():
    oracle = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return oracle


def host_cpu_model():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return ""


def cpu_columns_baseline(img, kind, ref_name, seq_name, length, tmp, tag, all_cores_total=0, extra=()):
    """The oracle's column loop — `maf`: MafExport::convertSequence (maf/impl/halMafExport.cpp:46-81), `depth`: printSequence
    (alignmentDepth/halAlignmentDepth.cpp:215-308) — single-threaded over the reference's first `length` columns (the time of
    the loop itself: image load and file writing are outside), its text for the parity gate, and, as the fairness row, one
    process per host core over consecutive slices (the reference's own way to use more cores: hal2mafMP.py's slices)."""
    oracle = oracle_bin()

    def cmd(out, start, n):
        c = [oracle, kind, img] + ([out] if kind == "maf" else [ref_name, out]) + (["--refGenome", ref_name] if kind == "maf" else [])
        return c + list(extra) + ["--refSequence", seq_name, "--start", str(start), "--length", str(n), "--stats"]
    out = os.path.join(tmp, "%s.%s" % (tag, kind))
    st = json.loads(subprocess.run(cmd(out, 0, length), check=True, stdout=subprocess.PIPE).stdout.decode())
    with open(out, "rb") as f:
        text = f.read()
    res = {"value": length / st["seconds"], "unit": "columns/s", "cores": 1, "kind": "port", "host_cpu": host_cpu_model(),
           "sample": "the oracle's %s over the first %d columns of %s, the column loop's time only (image load and the file excluded)"
                     % ("MafExport::convertSequence --noAncestors" if kind == "maf" else "halAlignmentDepth printSequence", length, ref_name),
           "seconds": st["seconds"]}
    cores = oracle_pool_size(img)
    if all_cores_total > 0 and cores > 1:
        per = max(1, all_cores_total // cores)
        procs = [subprocess.Popen(cmd(os.path.join(tmp, "%s.%d.%s" % (tag, c, kind)), c * per, per), stdout=subprocess.PIPE) for c in range(cores)]
        secs = [json.loads(pr.communicate()[0].decode())["seconds"] for pr in procs]
        for c in range(cores):
            try:
                os.unlink(os.path.join(tmp, "%s.%d.%s" % (tag, c, kind)))
            except OSError:
                pass
        res["all_cores"] = {"value": per * cores / max(secs), "unit": "columns/s", "cores": cores,
                            "sample": "%d columns in slices of %d over %d oracle processes (the loop time of the slowest)" % (per * cores, per, cores)}
    os.unlink(out)
    return res, text


def maf_prefix_matches(oracle_text, gpu_prefix):
    """The export of a whole genome begins with the export of its first L columns, but for the slice's last block (it ends where the
    slice does): the oracle's text up to its last block is held against the timed export's text."""
    cut = oracle_text.rfind(b"\na")
    return cut > 0 and len(gpu_prefix) >= cut and oracle_text[:cut] == gpu_prefix[:cut]


def leg_hal2maf_full(args, local=0):
    """BASELINE config 3 as stated — hal2maf --refGenome <leaf> --noAncestors over the WHOLE reference genome, end to end to MAF text in host
    memory — with its device-stage report, its CPU figure and its parity gate.  Run by a process of its own (main: --leg hal2maf_full): the
    export takes its heads from round 5's per-base tracks and walks its blocks over slices of the export, code whose first run on a GPU
    is the driver's; whatever happens to it must not cost the line."""
    import hal_amd
    al = hal_amd.Alignment.random(workload_options(args.scale, "cfg2", dna="fast"), device=local)
    src_name = "Genome_9"
    src = al.genome_id(src_name)
    seq_name = al.sequences(src)[0][0]
    ncols = al.genome_length(src)
    al.maf_export_bytes(src, start=0, length=min(ncols, 200000), no_ancestors=True)  # (DNA upload, code objects)
    runs = []
    for _ in range(2):  # (the second export finds the text's memory and the rendering buffers of the first: both are listed)
        nbytes, maf_head, m_s = al.maf_export_bytes(src, no_ancestors=True, prefix=56 * args.cpu_columns + 65536)
        runs.append(m_s)
    dt_m = min(runs)
    try:
        tracks = al.maf_tracks_info()
    except Exception as e:
        tracks = {"error": str(e)[:200]}
    # the same export by round 4's path (every column walked on the device, one thread's block state machine), for the comparison
    round4 = None
    if "HGX_MAF_SWEEP" not in os.environ and "HGX_MAF_SLICED" not in os.environ:
        try:
            os.environ["HGX_MAF_SWEEP"] = os.environ["HGX_MAF_SLICED"] = "0"
            n4, head4, s4 = al.maf_export_bytes(src, no_ancestors=True, prefix=65536)
            round4 = {"what": "HGX_MAF_SWEEP=0 HGX_MAF_SLICED=0: round 4's device stage and walk, one export after the two above", "seconds": s4,
                      "value": ncols / s4, "same_size": n4 == nbytes, "same_beginning": head4 == maf_head[:65536]}
        except Exception as e:
            round4 = {"error": str(e)[:200]}
        finally:
            os.environ.pop("HGX_MAF_SWEEP", None)
            os.environ.pop("HGX_MAF_SLICED", None)
    leg = {"metric": "MAF columns/sec (hal2maf --refGenome %s --noAncestors over the whole genome, end to end to MAF text in host memory; the better "
                     "of two exports)" % src_name,
           "value": ncols / dt_m, "unit": "columns/s", "columns": ncols, "seconds": dt_m, "runs_seconds": runs, "maf_bytes": nbytes,
           "process": "a child of bench.py (its own alignment, its own HIP context)", "by_round_4s_path": round4,
           "device_stage": dict(tracks, what="hgx_maf_tracks_info after the two exports: the per-base tracks the heads are taken from (hgx_maf_kernels.hpp) "
                                             "— built once (build_ms), every chunk's kernels timed with HIP events (device_ms_served over "
                                             "columns_served); state says whether the first chunk's heads were the column walk's (else the walk is "
                                             "used: round 4's stage); last_export: the host side — who walked the blocks, rounds, seconds")}
    ncc = min(args.cpu_columns, ncols)
    if args.cpu_sample > 0 and ncc > 0:
        try:
            with tempfile.TemporaryDirectory() as tmp:
                img = os.path.join(tmp, "bench.hgx")
                al.save(img)
                basem, textm = cpu_columns_baseline(img, "maf", src_name, seq_name, ncc, tmp, "cfg2m",
                                                    all_cores_total=4 * ncc if args.cpu_all_cores else 0, extra=("--noAncestors",))
            basem["parity_with_gpu"] = maf_prefix_matches(textm, maf_head)
            basem["parity"] = ("the oracle's MAF of the first %d columns, up to its last block (a slice ends its last block where it ends), is the "
                               "beginning of the timed export's text" % ncc)
            basem["maf_bytes"] = len(textm)
            leg["cpu_baseline"] = basem
        except Exception as e:
            leg["cpu_baseline"] = {"error": str(e)[:300]}
    # the line so far is out before round 5's second new device path runs (the parent reads the child's LAST line, whatever its end)
    print(json.dumps(leg), flush=True)
    try:
        # --unique over the whole genome — what hal2mafMP.py runs every slice with: the columns' classes (passed over / walked for
        # their keys / written) from the per-base tracks (hgx_maf_kernels.hpp: unique_stretches) against round 4's lane-a-column walk
        al.maf_tracks_info(drop=True)
        asked = os.environ.get("HGX_MAF_SWEEP")
        n_t, head_t, s_t = al.maf_export_bytes(src, no_ancestors=True, unique=True, prefix=1 << 20)
        n_t2, _, s_t2 = al.maf_export_bytes(src, no_ancestors=True, unique=True, prefix=1 << 20)
        info = al.maf_tracks_info()
        os.environ["HGX_MAF_SWEEP"] = "0"
        n_w, head_w, s_w = al.maf_export_bytes(src, no_ancestors=True, unique=True, prefix=1 << 20)
        if asked is None:
            os.environ.pop("HGX_MAF_SWEEP", None)
        else:
            os.environ["HGX_MAF_SWEEP"] = asked
        leg["unique"] = {"what": "the same export with --unique: heads and classes from the per-base tracks (the better of two) / by the column walk "
                                 "(HGX_MAF_SWEEP=0)",
                         "value": ncols / min(s_t, s_t2), "unit": "columns/s", "seconds": min(s_t, s_t2), "runs_seconds": [s_t, s_t2],
                         "by_the_column_walk": {"value": ncols / s_w, "seconds": s_w},
                         "same_text": n_t == n_w and n_t2 == n_w and head_t == head_w, "maf_bytes": n_t,
                         "device_stage": {k: info.get(k) for k in ("state_unique", "chunks_served_unique", "build_ms", "device_ms_served",
                                                                   "columns_served", "marked_columns")}}
        print(json.dumps(leg), flush=True)
        # hal2mafMP.py's recipe over the whole genome: slices of a million columns, --unique, an export each, two handles of this GPU —
        # the slices of a handle share its tracks
        clones = [al, al.clone_to_device(local)]
        al.maf_tracks_info(drop=True)
        t0 = time.perf_counter()
        nb_t = hal_amd.maf_export_multi(clones, src, 0, start=0, length=ncols, slice_size=1000000, no_ancestors=True, unique=True, size_only=True)
        s_mt = time.perf_counter() - t0
        infos = [c.maf_tracks_info() for c in clones]
        os.environ["HGX_MAF_SWEEP"] = "0"
        t0 = time.perf_counter()
        nb_w = hal_amd.maf_export_multi(clones, src, 0, start=0, length=ncols, slice_size=1000000, no_ancestors=True, unique=True, size_only=True)
        s_mw = time.perf_counter() - t0
        if asked is None:
            os.environ.pop("HGX_MAF_SWEEP", None)
        else:
            os.environ["HGX_MAF_SWEEP"] = asked
        leg["unique"]["export_multi"] = {"what": "hgx_maf_export_multi over the whole genome: %d slices of 1 M columns, --unique, two handles of this one GPU; "
                                                 "from each handle's tracks / by the column walk" % ((ncols + 999999) // 1000000),
                                         "value": ncols / s_mt, "unit": "columns/s", "seconds": s_mt, "by_the_column_walk": {"value": ncols / s_mw, "seconds": s_mw},
                                         "same_size": nb_t == nb_w, "maf_bytes": nb_t,
                                         "tracks": [{k: i.get(k) for k in ("tracks", "state_unique", "chunks_served_unique", "build_ms")} for i in infos]}
    except Exception as e:
        leg.setdefault("unique", {})["error"] = str(e)[:300]
    return leg


def leg_child_command():
    """this file, run by this interpreter (nothing from the environment enters the child's code; the dry run without a GPU replaces
    this function from its own script: tests/support/bench_fakes.py)"""
    return [sys.executable, os.path.join(ROOT, "bench.py")]


def run_leg_in_child(name, args, env_extra=None, timeout=300.0):
    """`python bench.py --leg <name>` in a process of its own; its JSON object, or {"error": ...}"""
    cmd = leg_child_command() + ["--leg", name, "--scale", str(args.scale), "--cpu-sample", str(args.cpu_sample), "--cpu-columns",
                                 str(args.cpu_columns), "--cpu-all-cores", str(args.cpu_all_cores)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env_extra or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "the child did not finish within %.0f s" % timeout}
    lines = [l for l in r.stdout.decode(errors="replace").strip().splitlines() if l.startswith("{")]
    if not lines:
        return {"error": "the child ended with code %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
    try:
        leg = json.loads(lines[-1])
    except Exception as e:
        return {"error": "the child's line does not parse: %s" % e}
    if r.returncode != 0:  # (a leg prints its line as far as it has got before it goes on: what it had is kept, its end is told)
        leg["child_ended_with"] = {"code": r.returncode, "stderr": r.stderr.decode(errors="replace")[-300:]}
    return leg


def lib_sha16():
    import hashlib
    with open(os.path.join(ROOT, "hal_amd", "libhgx.so"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def kernel_sources_sha16():
    """identity of the device code: the .hip files and the headers they include (host-only changes leave it alone)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "hal_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "hal_amd", "csrc", "*.hpp"))):
        with open(path, "rb") as f:
            text = f.read()
        if path.endswith(".hpp") and b"__global__" not in text and b"__device__" not in text:
            continue  # a host-side header
        h.update(os.path.basename(path).encode())
        h.update(text)
    return h.hexdigest()[:16]


_KERNEL_CODE = {}


def kernel_code_sha16s(lib_path=None):
    """{kernel name: hash of its machine code} for every hgx kernel of the library: the gfx950 code objects are taken out of
    libhgx.so (llvm-objdump --offloading), and of every FUNC symbol hgx::k_* the bytes of its body are hashed — all instantiations
    of a template under the template's name.  A PMC measurement of a kernel stays valid exactly as long as this stays the same:
    host-side edits and edits to other kernels leave it alone (what the file-level hashes could not tell apart).  {} when the
    tools are not there."""
    import hashlib
    import re
    import shutil
    import struct
    lib_path = lib_path or os.path.join(ROOT, "hal_amd", "libhgx.so")
    if lib_path in _KERNEL_CODE:
        return _KERNEL_CODE[lib_path]
    out = {}
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    try:
        with tempfile.TemporaryDirectory() as tmp:
            copy = os.path.join(tmp, "lib.so")
            shutil.copyfile(lib_path, copy)
            subprocess.run([objdump, "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            per = {}
            for name in sorted(os.listdir(tmp)):
                if "amdgcn" not in name:
                    continue
                data = open(os.path.join(tmp, name), "rb").read()
                if data[:4] != b"\x7fELF" or data[4] != 2:
                    continue
                shoff, = struct.unpack_from("<Q", data, 0x28)
                shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
                secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
                for sec in secs:
                    if sec[1] != 2:  # SHT_SYMTAB
                        continue
                    stroff = secs[sec[6]][4]
                    for k in range(sec[5] // 24):
                        st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", data, sec[4] + 24 * k)
                        if (st_info & 15) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= shnum:  # STT_FUNC with a body
                            continue
                        end = data.index(b"\0", stroff + st_name)
                        sym = data[stroff + st_name:end].decode("ascii", "replace")
                        m = re.match(r"_ZN3hgx(?:L|\d+_GLOBAL__N_1)?(\d+)(k_\w+)", sym)
                        if not m:
                            continue
                        base = m.group(2)[:int(m.group(1))]
                        host = secs[st_shndx]
                        off = host[4] + (st_value - host[3])
                        per.setdefault(base, {})[sym] = hashlib.sha256(data[off:off + st_size]).hexdigest()
            for base, syms in per.items():
                h = hashlib.sha256()
                for sym in sorted(syms):
                    h.update(sym.encode())
                    h.update(syms[sym].encode())
                out[base] = h.hexdigest()[:16]
    except Exception:
        out = {}
    _KERNEL_CODE[lib_path] = out
    return out


def pmc_traffic(kernel, form="kernels"):
    """HBM bytes per launch of `kernel` from the PMC passes of profiles/scripts/r04_pmc.py (profiles/pmc_traffic.json; form
    "kernels": one batch repeated, "rotating": four batches in turn), or
    None when the file was made with other device code than the one running (the file records the hash of the library and of
    the device sources it was built from; either one matching will do: a host-only change rebuilds the library without touching
    a kernel)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    # (per kernel first: the hash of the kernel's own machine code as the PMC run recorded it — host-side commits and edits to other
    # kernels do not void a measurement; then the whole library / all device sources, as files of earlier rounds have it)
    recorded = d.get("kernel_code_sha16", {}).get(kernel)
    if recorded is not None and recorded == kernel_code_sha16s().get(kernel):
        return d.get(form, {}).get(kernel), d.get("source", "") + " [kernel code %s]" % recorded
    if d.get("libhgx_sha16") != lib_sha16() and d.get("kernel_sources_sha16") != kernel_sources_sha16():
        return None, "profiles/pmc_traffic.json was measured on other code of %s (library %s); rerun profiles/scripts/r05_pmc.py" % (kernel, d.get("libhgx_sha16"))
    return d.get(form, {}).get(kernel), d.get("source", "")


def timed_steps(step, k, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(k):
        r = step()
    sync()
    return time.perf_counter() - t0, r


def steady_legs(hal_amd, torch, al, src, tgt, d_gs, d_ge, d_st, steps, sync, rec_bytes=16, streams=None):
    """The timed configuration's three measurements on another alignment / another table width: two plans with a batch each in
    flight (hgx_liftover_submit / _collect), one plan batch after batch (hgx_liftover_run_device), and the kernels of the latter
    with HIP events around every launch, each priced by its own bytes (plan_kernel_bytes)."""
    nq = d_gs.numel()
    plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(2)]
    for p in plans:
        for _ in range(3):  # (change-over to the table, workspace growth, the launches for intervals passed on)
            p.run(d_gs, d_ge, d_st)
        p.set_timing(0)
    # (the two streams of the main measurement where there are some: later streams of a process may share a hardware queue, and
    # two batches on one queue do not overlap — seen as a wide leg with no gain from the second batch, profiles/r03a_bench.log)
    streams = streams or [torch.cuda.Stream(), torch.cuda.Stream()]
    pending = [False, False]

    def two(k_steps):
        for i in range(k_steps):
            k = i & 1
            if pending[k]:
                plans[k].collect()
                pending[k] = False
            plans[k].submit(d_gs, d_ge, d_st, stream=streams[k])
            pending[k] = True
        for k in (0, 1):
            if pending[k]:
                plans[k].collect()
                pending[k] = False
    two(10)
    dt2, _ = timed_steps(lambda: two(steps), 1, sync)
    for _ in range(5):
        plans[0].run(d_gs, d_ge, d_st)
    dt1, _ = timed_steps(lambda: plans[0].run(d_gs, d_ge, d_st), steps, sync)
    plans[0].set_timing(2)
    for _ in range(steps):
        plans[0].run(d_gs, d_ge, d_st)
    kt = plans[0].kernel_times()
    st = plans[0].stats()
    plans[0].set_timing(1)
    per_alg = plan_kernel_bytes(kt, st, steps, rec_bytes)
    kernels = []
    for kname, kv in sorted(kt.items(), key=lambda kv: -kv[1]["ms"]):
        avg = kv["ms"] / max(1, kv["launches"])
        gbs = per_alg.get(kname, 0.0) / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        kernels.append({"kernel": kname, "kernel_avg_ms": avg, "launches_per_step": kv["launches"] / steps,
                        "algorithmic_bytes_per_launch": per_alg.get(kname, 0.0), "achieved": gbs, "frac": gbs / HBM_PEAK_GBS})
    return {"ms_per_step": 1e3 * dt2 / steps, "value": nq * steps / dt2, "unit": "intervals/s", "batches_in_flight": 2, "steps": steps,
            "one_plan": {"ms_per_step": 1e3 * dt1 / steps, "value": nq * steps / dt1},
            "kernels_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in sorted(kt.items())},
            "roofline_kernels": kernels, "records_per_step": st["records"], "general_queries": st["general_queries"],
            "deferred_queries": st["deferred_queries"], "composed_kind": st["composed_kind"], "table_records": st["composed_records"],
            "table_build_ms": st["composed_build_ms"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (a step is 0.12 ms at config 2: a hundred of them are a stable mean)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--queries", type=int, default=1000000, help="intervals per GPU")
    ap.add_argument("--scale", type=float, default=1.0, help="genome size multiplier (1.0 = ~100 Mb/genome)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4"],
                    help="cfg2: 10-genome alignment, Genome_9 -> Genome_2 (the metric's configuration); cfg4: 50-genome alignment, "
                         "Genome_44 -> Genome_2 (BASELINE configs 4 and 5, one GPU's shard)")
    ap.add_argument("--target", default="Genome_2")
    ap.add_argument("--cpu-sample", type=int, default=300000, help="intervals timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-all-cores", type=int, default=1, help="also time the oracle sharded over every host core")
    ap.add_argument("--cpu-columns", type=int, default=4000000,
                    help="columns of the reference genome the oracle's hal2maf and halAlignmentDepth loops are timed on, beside the column "
                         "legs (the parity gate of those legs is the same slice; 0 = skip)")
    ap.add_argument("--cpu-columns-cfg5", type=int, default=1000000, help="the same for config 5's depth scan on the 50-genome alignment")
    ap.add_argument("--columns", type=int, default=1, help="also time the column-depth kernel over the whole source genome (0 = skip)")
    ap.add_argument("--maf-columns", type=int, default=8000000,
                    help="hal2maf (BASELINE config 3) over the first N reference columns, end to end to MAF text (0 = skip; one GPU only)")
    ap.add_argument("--maf-full", type=int, default=1, help="hal2maf over the WHOLE reference genome (BASELINE config 3 as it is stated; one GPU only)")
    ap.add_argument("--wide", type=int, default=1, help="the same steps on int64-coordinate tables (HGX_FORCE_WIDE=1 copy of the alignment; one GPU only)")
    ap.add_argument("--cfg4", type=int, default=1,
                    help="BASELINE configs 4 and 5 on one GPU: a 1.25 M-interval shard Genome_44 -> Genome_2 and the whole-genome depth scan of "
                         "Genome_44 on the 50-genome alignment (generating it takes about a minute; 0 = skip; one GPU only)")
    ap.add_argument("--text-path", type=int, default=1, help="also time Liftover::convert (BED text in, BED text out) on the batch (one GPU only)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight on one GPU without an exchange: 2 (two plans, two streams) or 1")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="extra leg: the same step repeated for about this long")
    ap.add_argument("--rotating", type=int, default=4,
                    help="the timed loop takes this many DISTINCT batches (other seeds), each with a plan and output buffers of its own, in turn "
                         "with two in flight — one rotation's working set (0.65 GB at 4) is past the 256 MiB Infinity Cache (0 or 1: one batch "
                         "repeated, which stays in that cache; measured beside as `cached` otherwise)")
    ap.add_argument("--features", type=int, default=1,
                    help="extra legs for the other entry points of the path: halGetBlocksInTargetRange (ranges/s), hal2maf --unique, "
                         "--maxRefGap 100 and hgx_maf_export_multi over device clones (one GPU only)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "c_abi"],
                    help="who issues the batch's one all-gather: torch.distributed (the launcher's communicator, default) or the library "
                         "itself through hgx_liftover_exchange (RCCL loaded by libhgx.so)")
    ap.add_argument("--exchange-in-step", type=int, default=0,
                    help="N > 1: 1 = every timed step ends with the all-gather of all ranks' records (link-bound); 0 (default) = the ranks map "
                         "their shards as one GPU does and the collated form is measured beside")
    ap.add_argument("--leg", default="", help="run ONE leg and print its JSON object (bench.py starts itself with this: hal2maf_full)")
    ap.add_argument("--exchange-selftest", type=int, default=0,
                    help="1: with one GPU, run the multi-GPU code path (wire blob, overlapped all-gatherv, collective settle exit) on a "
                         "one-rank RCCL group; the JSON line then says so in config.exchange")
    args = ap.parse_args()

    if args.leg == "hal2maf_full":
        print(json.dumps(leg_hal2maf_full(args)), flush=True)
        return
    import torch
    import torch.distributed as dist
    import hal_amd
    from hal_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # The shards are independent (liftover/impl/halLiftover.cpp:46-92: a line's lifting knows nothing of the other lines): with N
    # ranks the timed step is what it is with one — every rank maps its batch, the records stay in its HBM — and `value` is the
    # ranks' intervals over the slowest rank's time.  Collating every rank's records on every rank (one all-gather per batch) is
    # an output step the path does not need; it is measured beside, as `collated` (--exchange-in-step 1 puts it into the timed
    # step instead, as rounds 1 and 2 did).
    synced = world > 1 or bool(args.exchange_selftest)                              # a process group, barriers around the timed region
    exchanging = bool(args.exchange_selftest) or (world > 1 and bool(args.exchange_in_step))  # the all-gather is part of the timed step
    if synced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sync = torch.cuda.synchronize
    want_maf = args.maf_columns > 0 and world == 1 and not args.exchange_selftest

    t0 = time.time()
    # (DNA only for the hal2maf leg: "fast" gives the alignment of with_dna=False plus bases from a separate generator)
    al = hal_amd.Alignment.random(workload_options(args.scale, args.workload, dna="fast" if want_maf else False), device=local)
    gen_s = time.time() - t0
    src_name, tgt_name = ("Genome_9" if args.workload == "cfg2" else "Genome_44"), args.target
    src, tgt = al.genome_id(src_name), al.genome_id(tgt_name)
    seq_name, seq_start, length = al.sequences(src)[0]
    nq = args.queries
    starts, lens, strand = make_queries(length, nq, 1234 + rank)  # disjoint shards: every rank has its own intervals
    d_gs = (starts + seq_start).to(dev)
    d_ge = (starts + lens - 1 + seq_start).to(dev)
    d_st = strand.to(dev)

    _mark("cold")
    # ---- cold: a fresh plan with the default policy, one pass over the batch (whatever the policy does on its first
    # batch — here it builds the table of the whole path and its merged form — is inside the time).  The process has run the
    # same code once before on a 1 %-scale alignment, so the one-time loading of HIP code objects (≈120 ms in a fresh
    # process, profiles/r02e_bench.log) is not in it; everything that belongs to the alignment, the plan and the batch is. ----
    small = hal_amd.Alignment.random(workload_options(0.01, args.workload), device=local)
    s_src, s_tgt = small.genome_id(src_name), small.genome_id(tgt_name)
    _, s_ss, s_len = small.sequences(s_src)[0]
    s_st, s_ln, s_sd = make_queries(s_len, 50000, 7)
    s_plan = hal_amd.LiftoverPlan(small, s_src, s_tgt, max_queries=50000)
    s_plan.run((s_st + s_ss).to(dev), (s_st + s_ln - 1 + s_ss).to(dev), s_sd.to(dev))
    del s_plan, small
    sync()
    t0 = time.perf_counter()
    plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
    t_plan = time.perf_counter() - t0
    ptr, nrec = plan.run(d_gs, d_ge, d_st)
    sync()
    cold_s = time.perf_counter() - t0
    cold_stats = plan.stats()
    cold_runs = [1e3 * cold_s]
    # (once more on a second copy of the alignment on this device — fresh device tables, nothing cached: the first figure is at
    # the mercy of the allocator, 9 ms in most runs and 70-80 ms in some; `cold.ms` is the better of the two, both are listed)
    al2 = al.clone_to_device(local)
    sync()
    t0b = time.perf_counter()
    plan_b = hal_amd.LiftoverPlan(al2, src, tgt, max_queries=nq)
    t_plan_b = time.perf_counter() - t0b
    plan_b.run(d_gs, d_ge, d_st)
    sync()
    cold_b = time.perf_counter() - t0b
    cold_runs.append(1e3 * cold_b)
    if cold_b < cold_s:
        cold_s, t_plan, cold_stats = cold_b, t_plan_b, plan_b.stats()
    del plan_b, al2
    # (a third time with a device synchronisation at every phase boundary — HGX_BUILD_TIMING=2 — for the breakdown: its phases
    # add up to its own wall time, which is a little above the un-instrumented runs')
    cold_phases = None
    if rank == 0:
        al3 = al.clone_to_device(local)
        sync()
        os.environ["HGX_BUILD_TIMING"] = "2"
        try:
            t0c = time.perf_counter()
            plan_c = hal_amd.LiftoverPlan(al3, src, tgt, max_queries=nq)
            sync()
            t_plan_c = time.perf_counter() - t0c
            plan_c.run(d_gs, d_ge, d_st)
            sync()
            cold_c = time.perf_counter() - t0c
        finally:
            del os.environ["HGX_BUILD_TIMING"]
        ph = [("plan creation (schedule, chain / down / locate tables, workspaces)", 1e3 * t_plan_c)] + hal_amd.build_phases()
        listed = sum(ms for _, ms in ph)
        ph.append(("host side of the run outside the phases above", max(0.0, 1e3 * cold_c - listed)))
        cold_phases = {"ms": 1e3 * cold_c, "phases_ms": [[k, round(v, 4)] for k, v in ph]}
        del plan_c, al3
    passes_before_timing = 1

    # The one exchange step of the path: the batch's records of every rank, as self-describing wire blobs (12 bytes per record + 2
    # per interval when every field fits: hgx_liftover_wire_blob) in equal slots of one buffer, ONE all-gather per batch
    # (hal_amd/shard.py: SlotExchange) — issued by torch.distributed or, --exchange c_abi, by the library itself
    # (hgx_liftover_exchange).  Three buffers rotate: wait() completes the oldest exchange under way, so every timed step pays for
    # one whole exchange while the following batches are mapped.
    exchange = None
    if synced:
        cap = torch.tensor([plan.wire_capacity()], dtype=torch.int64, device=dev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)  # (one slot size for all ranks, with room for batches that differ a little)
        slot = int(cap.item()) * 5 // 4
        comm = None
        if args.exchange == "c_abi":
            uid = torch.tensor(list(hal_amd.Comm.unique_id()) if rank == 0 else [0] * 128, dtype=torch.uint8, device=dev)
            dist.broadcast(uid, 0)
            comm = hal_amd.Comm(bytes(uid.cpu().tolist()), rank, world, local)
        exchange = shard.SlotExchange(world, rank, slot, dev, backend=args.exchange, comm=comm)
    wire = {"format": None, "bytes": 0}

    # One GPU without an exchange: TWO plans of the alignment with a batch each in flight on two streams
    # (hgx_liftover_submit / hgx_liftover_collect) — a step is still one pass of one plan over the batch, the next step of the
    # other plan is queued behind it before this one is waited for, so the end of one batch's launches (a few wavefronts
    # finishing general intervals) and the host's launch and wake-up times overlap the other batch.  `one_plan` reports the
    # same steps through one plan, batch after batch.  With an exchange every step carries a collective and one plan is used.
    in_flight = 1 if exchanging or args.in_flight <= 1 else 2
    # The timed loop takes K DISTINCT batches (other seeds), each with a plan — and so record, answer and count buffers — of its
    # own, in turn, two in flight: a batch's inputs and buffers are touched again only after the K - 1 others have gone through
    # (K = 4: 0.65 GB per rotation, past the 256 MiB Infinity Cache).  One batch passed through two plans again and again — what
    # `value` was up to round 4 — has its whole working set in that cache; it is measured beside, as `cached`.
    K = args.rotating if (in_flight == 2 and args.rotating > 1) else 1
    plans, batches, streams = [plan], [(d_gs, d_ge, d_st)], [torch.cuda.current_stream()]
    if in_flight == 2:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for k in range(1, max(K, 2)):
            if K > 1:
                s_k, l_k, d_k = make_queries(length, nq, 5000 + 17 * k + 101 * rank)
                batches.append(((s_k + seq_start).to(dev), (s_k + l_k - 1 + seq_start).to(dev), d_k.to(dev)))
            else:
                batches.append(batches[0])
            pk = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
            for _ in range(3):  # (the plan's own change-over to the table, which is cached in the alignment by now)
                pk.run(*batches[k])
            pk.set_timing(0)
            plans.append(pk)
    slots = len(plans)
    pending, records = [False] * slots, [nrec] * slots
    counter = {"i": 0}

    def step_one():
        ptr, nrec = plan.run(d_gs, d_ge, d_st)
        if exchanging:
            exchange.wait()
            exchange.submit(plan, first_query=rank * nq)
            wire["format"], wire["bytes"] = exchange.last_format, exchange.last_bytes
        return nrec

    def finish(k):
        if pending[k]:
            _, records[k] = plans[k].collect()
            pending[k] = False

    def step():
        if in_flight == 1:
            records[0] = step_one()
            return records[0]
        i = counter["i"]
        counter["i"] += 1
        k = i % slots
        finish((i - 2) % slots)  # (two in flight: the batch two steps back is waited for)
        finish(k)
        plans[k].submit(*batches[k], stream=streams[i & 1])
        pending[k] = True
        return records[k]

    def drain():
        for k in range(slots):
            finish(k)

    # settle (untimed): the first runs of a fresh process pay for lazy code-object loads and workspace growth, and a process
    # that starts while the previous GPU process is still being torn down sees extra host time per run for a second or two.
    # Run until ten consecutive runs are within 5 % of the fastest one seen, for at most 4 s.
    plan.set_timing(0)  # no kernel events in the timed loop (the kernel times come from a separate loop below)
    best, streak, t_settle = None, 0, time.perf_counter()
    while True:
        t_s = time.perf_counter()
        step()
        drain()
        sync()
        passes_before_timing += 1
        dt = time.perf_counter() - t_s
        if best is None or dt < best:
            best = dt
        streak = streak + 1 if dt <= 1.05 * best else 0
        done = streak >= 10 or time.perf_counter() - t_settle >= 4.0
        if synced:  # the ranks leave the loop together
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break
    for _ in range(args.warmup):
        step()
        passes_before_timing += 1
    drain()
    if synced:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # the batches still in flight belong to the timed region
    nrec = sum(records) // slots  # (records of a step: the batches' mean)
    if exchanging:
        exchange.drain()  # the exchanges still under way belong to the timed region
    sync()
    if synced:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    nrec_all = nrec
    if synced:
        tot = torch.tensor([nrec], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        nrec_all = int(tot.item())
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    _mark("with an exchange in the step")
    # ---- with an exchange in the step: the same steps without it (what the ranks map when nobody collates) ----
    mapping_only = None
    if exchanging:
        dist.barrier()
        sync()
        t0m = time.perf_counter()
        for _ in range(args.steps):
            plan.run(d_gs, d_ge, d_st)
        sync()
        dist.barrier()
        tm = torch.tensor([time.perf_counter() - t0m], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        mapping_only = {"value": world * nq * args.steps / float(tm.item()), "unit": "intervals/s", "ms_per_step": 1e3 * float(tm.item()) / args.steps,
                        "what": "the same steps without the all-gather of the records (one plan per rank, every batch waited for): the "
                                "shards are independent, so this is what the ranks map; `value` includes collating every rank's records "
                                "on every rank, which the links bound (%.1f MB per rank and step)" % (wire["bytes"] / 1e6)}

    _mark("N ranks without an exchange in the step")
    # ---- N ranks without an exchange in the step: the same steps with every rank's records collated on every rank ----
    # (these two legs carry collectives that no multi-GPU node has run yet: they are made at the very end, under a watchdog, so that
    # whatever happens to them the line with everything else is printed — see the end of main)
    def leg_collated():
        def step_collated():
            plan.run(d_gs, d_ge, d_st)
            exchange.wait()
            exchange.submit(plan, first_query=rank * nq)
            wire["format"], wire["bytes"] = exchange.last_format, exchange.last_bytes
        for _ in range(3):
            step_collated()
        exchange.drain()
        dist.barrier()
        sync()
        t0c = time.perf_counter()
        for _ in range(args.steps):
            step_collated()
        exchange.drain()
        sync()
        dist.barrier()
        tc = torch.tensor([time.perf_counter() - t0c], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        return {"value": world * nq * args.steps / float(tc.item()), "unit": "intervals/s", "ms_per_step": 1e3 * float(tc.item()) / args.steps,
                    "wire_format_bytes_per_record": wire["format"], "wire_MB_per_rank_and_step": wire["bytes"] / 1e6,
                    "what": "the same steps (one plan per rank) each followed by one all-gather of every rank's records of the batch to every "
                            "rank (%s), overlapped with the next batches: bound by the links, not by the kernels; not part of `value`"
                            % ("hgx_liftover_exchange: RCCL from the library" if args.exchange == "c_abi" else "torch.distributed")}

    _mark("the same steps with every rank's records gathered on rank 0")
    # ---- the same steps with every rank's records gathered on rank 0 only, in the 8-byte form: what a writer of the BED file needs
    # (a rank sends its blob once; nobody receives N of them) ----
    def leg_to_writer():
        wx = shard.SlotExchange(world, rank, exchange.slot, dev, backend=args.exchange, comm=exchange.comm, root=0, bed_only=True)

        def step_to_writer():
            plan.run(d_gs, d_ge, d_st)
            wx.wait()
            wx.submit(plan, first_query=rank * nq)
        for _ in range(3):
            step_to_writer()
        wx.drain()
        dist.barrier()
        sync()
        t0w = time.perf_counter()
        for _ in range(args.steps):
            step_to_writer()
        wx.drain()
        sync()
        dist.barrier()
        tw = torch.tensor([time.perf_counter() - t0w], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        res = {"value": world * nq * args.steps / float(tw.item()), "unit": "intervals/s", "ms_per_step": 1e3 * float(tw.item()) / args.steps,
                     "wire_format_bytes_per_record": wx.last_format, "wire_MB_per_rank_and_step": wx.last_bytes / 1e6,
                     "what": "the same steps (one plan per rank) each followed by a gather of the ranks' records of the batch to rank 0 only "
                             "(%s), 8 bytes a record (no source coordinates: what a writer of BED lines needs), overlapped with the next batches; "
                             "not part of `value`" % ("hgx_liftover_gather: RCCL send / recv from the library" if args.exchange == "c_abi"
                                                      else "torch.distributed.gather")}
        del wx
        return res

    _mark("sustained")
    # ---- sustained: the same step for a couple of seconds (a region long enough for outside observers: rocm-smi, the driver) ----
    sustained = None
    if args.sustained_seconds > 0 and not exchanging:
        k = max(args.steps, int(args.sustained_seconds / max(elapsed / args.steps, 1e-5)))

        def all_steps():
            for _ in range(k):
                step()
            drain()
        dt_s, _ = timed_steps(all_steps, 1, sync)
        sustained = {"steps": k, "seconds": dt_s, "value": nq * k / dt_s, "ms_per_step": 1e3 * dt_s / k, "in_flight": in_flight}
    _mark("cached")
    # ---- cached: ONE batch through two plans again and again (`value` up to round 4): inputs, tables and even the records written
    # (165 MB per plan) stay in the 256 MiB Infinity Cache, and FETCH_SIZE counts what is served from there ----
    cached = None
    if K > 1 and not exchanging:
        plan_c = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
        for _ in range(3):
            plan_c.run(d_gs, d_ge, d_st)
        plan_c.set_timing(0)
        c_plans, c_pending = [plan, plan_c], [False, False]

        def c_loop(n_steps):
            for i in range(n_steps):
                k = i & 1
                if c_pending[k]:
                    c_plans[k].collect()
                    c_pending[k] = False
                c_plans[k].submit(d_gs, d_ge, d_st, stream=streams[k])
                c_pending[k] = True
            for k in (0, 1):
                if c_pending[k]:
                    c_plans[k].collect()
                    c_pending[k] = False
        c_loop(10)
        n_c = max(args.steps, 16)
        c_runs = [timed_steps(lambda: c_loop(n_c), 1, sync)[0] for _ in range(3)]
        cached = {"value": nq * n_c / min(c_runs), "unit": "intervals/s", "ms_per_step": 1e3 * min(c_runs) / n_c,
                  "runs_ms_per_step": [1e3 * t / n_c for t in c_runs], "steps": n_c, "batches_in_flight": 2,
                  "what": "ONE batch passed through two plans again and again, two in flight (the headline of rounds 2-4): the working set of a "
                          "plan fits the 256 MiB Infinity Cache; `value` rotates %d distinct batches with buffers of their own" % K}
        del plan_c, c_plans

    _mark("the same number of steps through ONE plan, batch after batch")
    # ---- the same number of steps through ONE plan, batch after batch (what `value` was before batches were kept in flight) ----
    one_plan = None
    if in_flight == 2:
        dt_1, _ = timed_steps(step_one, args.steps, sync)
        one_plan = {"ms_per_step": 1e3 * dt_1 / args.steps, "value": nq * args.steps / dt_1, "unit": "intervals/s", "steps": args.steps,
                    "what": "hgx_liftover_run_device, one plan: every batch waited for before the next one is launched.  A batch that "
                            "runs by itself has its general intervals found and finished by workgroups at the head of k_lift_classify's "
                            "grid — scouts: each looks at a share of the batch and finishes what it finds (round 6; HGX_LIFT_SCOUT=0: "
                            "round 3's pass in front, k_lift_general_list); the batches kept in flight for `value` overlap their "
                            "launches' tails and are spared the look"}
        # the kernels of these steps (HIP events around every launch, a loop of its own)
        plan.set_timing(2)
        for _ in range(args.steps):
            plan.run(d_gs, d_ge, d_st)
        kt_one = plan.kernel_times()
        one_bytes = plan_kernel_bytes(kt_one, plan.stats(), args.steps)
        one_plan["kernels_ms_per_step"] = {k: round(v["ms"] / args.steps, 4) for k, v in sorted(kt_one.items())}
        one_plan["roofline_kernels"] = [
            {"kernel": k, "kernel_avg_ms": v["ms"] / max(1, v["launches"]), "algorithmic_bytes_per_launch": one_bytes.get(k, 0.0),
             "achieved": one_bytes.get(k, 0.0) / (v["ms"] / max(1, v["launches"]) * 1e-3) / 1e9 if v["ms"] > 0 else 0.0,
             "frac": one_bytes.get(k, 0.0) / (v["ms"] / max(1, v["launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBS if v["ms"] > 0 else 0.0}
            for k, v in sorted(kt_one.items(), key=lambda kv: -kv[1]["ms"])]
        plan.set_timing(1)

    _mark("kernel times")
    # ---- kernel times: the steps of the timed region again with HIP events around every launch (untimed) — through one plan, in
    # the form the batches in flight are launched in (no pass for the general intervals in front, see one_plan) ----
    n_kt = (max(args.steps, 4 * K) + K - 1) // K * K  # (every batch of the rotation as often as the others)
    kt_total = {}
    for pk in plans[:K]:
        if in_flight == 2:
            pk.set_workers(0)
        pk.set_timing(2)
    for i in range(n_kt):
        plans[i % K].run(*batches[i % K])
    for pk in plans[:K]:
        for kname, kv in pk.kernel_times().items():
            acc = kt_total.setdefault(kname, {"ms": 0.0, "launches": 0, "top_derefs": 0, "bot_derefs": 0})
            for f in acc:
                acc[f] += kv.get(f, 0)
        pk.set_timing(1 if pk is plan else 0)
        pk.set_workers(-1)
    kt_acc = {k: {"ms": v["ms"], "launches": v["launches"]} for k, v in kt_total.items()}

    col_result = None
    if args.columns:
        # secondary metric of BASELINE.json ("MAF columns/sec", configs 3 and 5): halAlignmentDepth's per-column closure over
        # the whole source genome (ColumnIterator path).  Columns are independent (api/impl/halColumnIterator.cpp:785-787):
        # rank r scans the contiguous range shard_bounds(ncol, world, r) and the int32 results are all-gathered.
        ncol = al.genome_length(src)
        lo, hi = shard.shard_bounds(ncol, world, rank)
        mine = torch.empty(max(hi - lo, 1), dtype=torch.int32, device=dev)
        al.columns_depth_device(src, lo, hi - lo, mine.data_ptr())
        col_ms = min(al.columns_depth_device(src, lo, hi - lo, mine.data_ptr()) for _ in range(3))
        gather_ms = 0.0
        depth_sum = float(mine[:hi - lo].double().sum().item())
        if synced:
            per = (ncol + world - 1) // world
            padded = torch.zeros(per, dtype=torch.int32, device=dev)
            padded[:hi - lo] = mine[:hi - lo]
            allv = torch.empty(per * world, dtype=torch.int32, device=dev)
            sync()
            dist.barrier()
            t_g = time.perf_counter()
            dist.all_gather_into_tensor(allv, padded)
            sync()
            gather_ms = (time.perf_counter() - t_g) * 1e3
            t = torch.tensor([col_ms, gather_ms, depth_sum], dtype=torch.float64, device=dev)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            col_ms, gather_ms, depth_sum = float(tmax[0].item()), float(tmax[1].item()), float(t[2].item())
        col_result = (ncol, col_ms, gather_ms, depth_sum / ncol)

    if rank == 0:
        kept = {}  # the beginnings of the timed column legs' texts, for their parity gates
        st = plan.stats()
        value = world * nq * args.steps / elapsed
        _mark("walk")
        # ---- walk: the level-by-level kernels (HGX_COMPOSED_UP=0), the "per-query-interval graph chase" itself.  Its
        # dereference counts are also the SURVEY 8(d) figure of the batch: algorithmic bytes are a property of the input,
        # counted by the reference's own walk, whatever the timed plan reads instead. ----
        os.environ["HGX_COMPOSED_UP"] = "0"
        try:
            walk_plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
        finally:
            del os.environ["HGX_COMPOSED_UP"]
        for _ in range(12):
            walk_plan.run(d_gs, d_ge, d_st)
        walk_plan.set_timing(0)
        walk_dt, _ = timed_steps(lambda: walk_plan.run(d_gs, d_ge, d_st), 5, sync)
        walk_plan.set_timing(2)
        for _ in range(5):
            walk_plan.run(d_gs, d_ge, d_st)
        walk_kt = walk_plan.kernel_times()
        wst = walk_plan.stats()
        assert wst["records"] == st["records"]
        walk_bytes = plan_kernel_bytes(walk_kt, wst, 5)
        walk_dom = max(walk_kt.items(), key=lambda kv: kv[1]["ms"])[0]
        walk_dom_ms = walk_kt[walk_dom]["ms"] / max(1, walk_kt[walk_dom]["launches"])
        walk_traffic, _ = pmc_traffic(walk_dom)
        walk = {"ms_per_step": 1e3 * walk_dt / 5, "value": nq * 5 / walk_dt, "unit": "intervals/s",
                "kernels_ms_per_step": {k: round(v["ms"] / 5, 4) for k, v in sorted(walk_kt.items())},
                "roofline": {"bound": "hbm", "kernel": walk_dom, "achieved": walk_bytes[walk_dom] / (walk_dom_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": walk_bytes[walk_dom] / (walk_dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": walk_traffic, "kernel_avg_ms": walk_dom_ms, "algorithmic_bytes_per_launch": walk_bytes[walk_dom]}}
        table_records = 0
        if st["composed_records"]:
            table_kernel = {3: "k_lift_merged", 2: "k_locate_through"}.get(st["composed_kind"], "k_locate_composed")
            table_records = kt_total[table_kernel]["top_derefs"] // n_kt  # table records dereferenced per step
        wcounts = dict(top_derefs=wst["top_derefs"], bottom_derefs=wst["bottom_derefs"], source_pieces=wst["source_pieces"],
                       mapped_pieces=wst["mapped_pieces"])
        del walk_plan
        # --- roofline of the dominant kernel (device time from HIP events on the launch stream) ---
        dom = max(kt_acc.items(), key=lambda kv: kv[1]["ms"])
        dom_name, dom_ms, dom_launches = dom[0], dom[1]["ms"], dom[1]["launches"]
        Q, T, B, R = st["queries"], wcounts["top_derefs"], wcounts["bottom_derefs"], st["records"]
        alg_walk = 24 * Q + 25 * T + 25 * B + 40 * R  # SURVEY 8(d), per step, by the reference's walk
        kern_ms_total = sum(v["ms"] for v in kt_acc.values()) / n_kt
        per_kernel_alg = plan_kernel_bytes(kt_total, dict(st, records=nrec), n_kt)  # (records: the batches' mean)
        own_bytes = sum(per_kernel_alg[k] * v["launches"] for k, v in kt_acc.items()) / n_kt  # what the timed kernels must move
        pmc_form = "rotating" if K > 1 else "kernels"
        dom_bytes_per_launch = per_kernel_alg.get(dom_name, 0.0)
        dom_avg_ms = dom_ms / max(1, dom_launches)
        achieved = dom_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic(dom_name, pmc_form)
        # every timed kernel of the step priced the same way (the object above is the one that takes the most time)
        per_kernel_roofline = []
        for kname, kv in sorted(kt_acc.items(), key=lambda kv: -kv[1]["ms"]):
            k_avg_ms = kv["ms"] / max(1, kv["launches"])
            k_bytes = per_kernel_alg.get(kname, 0.0)
            k_gbs = k_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
            per_kernel_roofline.append({"kernel": kname, "kernel_avg_ms": k_avg_ms, "algorithmic_bytes_per_launch": k_bytes,
                                        "achieved": k_gbs, "frac": k_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(kname, pmc_form)[0]})
        # measured device-copy rate on this GPU (read + write bytes of a 1 GiB device-to-device copy), SURVEY 8(d)
        cp_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        cp_dst = torch.empty_like(cp_src)
        cp_dst.copy_(cp_src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cp_dst.copy_(cp_src)
        e1.record()
        sync()
        copy_gbs = 5 * 2 * cp_src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del cp_src, cp_dst
        kind_text = {3: "merged table of the whole path src->MRCA->target (chains of mergeable pieces), single-pass kernels",
                     2: "table of the whole path src->MRCA->target", 1: "table of the up phase src->MRCA"}
        out = {
            "metric": "lifted BED intervals/sec", "value": value, "unit": "intervals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64" if al.genome_length(src) >= 2 ** 31 else "int32", "data": "synthetic",
            "config": {"workload": "halRandGen %s ~100 Mb/genome HAL (%s, scale %g), halLiftover of %d BED6 intervals "
                                   "per GPU, %s -> %s, dupes on" % ("10-genome" if args.workload == "cfg2" else "50-genome",
                                                                     "seed 2" if args.workload == "cfg2" else "seed 0", args.scale, nq,
                                                                     src_name, tgt_name),
                       "intervals_per_gpu": nq, "records_per_step": nrec_all, "parallelism": "query-shard x%d" % world,
                       "regime": ("steady state: the timed steps are passes %d.. over the batches, through %s; the first plan's first pass "
                                  "(see `cold`) built its %s: %d records (%.0f MB) in %.1f ms on the device; %d table records "
                                  "dereferenced per step, %d of the %d intervals took the general (overlap-breaking) route; "
                                  "`walk` is the same batch without any table"
                                  % (passes_before_timing + 1,
                                     "one plan" if in_flight == 1 else "%d plans of the alignment taken in turn, two batches in flight on two streams "
                                     "(hgx_liftover_submit / _collect; `one_plan`: the same steps batch after batch through one plan)" % slots,
                                     kind_text.get(st["composed_kind"], "?"), st["composed_records"],
                                     st["composed_records"] * 16 / 1e6, st["composed_build_ms"], table_records, st["general_queries"], nq))
                       if st["composed_records"] else "level-by-level walk (k_up_chain)",
                       "exchange": ("one all-gather per batch of self-describing wire blobs in equal slots (%s; format %s, %.1f MB per rank and "
                                    "step), overlapped with the next batches"
                                    % ("hgx_liftover_exchange: RCCL from the library" if args.exchange == "c_abi" else "torch.distributed",
                                       wire["format"], wire["bytes"] / 1e6))
                       if exchanging else ("none in the timed step: the ranks' shards are independent and their records stay in their HBM; "
                                           "`collated` has the same steps with the all-gather" if world > 1 else "none (one GPU)"),
                       "newick": al.newick, "generate_s": round(gen_s, 2)},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         "kernel_avg_ms": dom_avg_ms, "kernel_launches_per_step": dom_launches / n_kt,
                         "algorithmic_bytes_per_launch": dom_bytes_per_launch,
                         "kernels": per_kernel_roofline,
                         "whole_step": {"kernel_ms_per_step": kern_ms_total, "bytes_the_timed_kernels_must_move": own_bytes,
                                        "achieved_GBs": own_bytes / (kern_ms_total * 1e-3) / 1e9 if kern_ms_total > 0 else 0.0,
                                        "frac": own_bytes / (kern_ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS if kern_ms_total > 0 else 0.0,
                                        "reference_walk_bytes_per_step": alg_walk,
                                        "batches_in_flight": in_flight,
                                        "achieved_GBs_at_step_rate": own_bytes / (elapsed / args.steps) / 1e9,
                                        "frac_at_step_rate": own_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                        "note": "kernel times come from a loop of its own through one plan with HIP events around every launch, in the "
                                                "form the batches in flight are launched in (general intervals finished by the wavefronts that meet them: "
                                                "k_lift_classify's launch then ends one such interval's latency after its last tile, a tail the other "
                                                "batch's launches fill; one_plan.roofline_kernels has the form a batch that runs by itself takes, with "
                                                "k_lift_classify at " + ("%.3f" % (one_plan["kernels_ms_per_step"].get("k_lift_classify", float("nan"))) if one_plan else "?") + " ms); with "
                                                "two batches in flight the launches of the two overlap, so ms_per_step is below kernel_ms_per_step; "
                                                "achieved_GBs_at_step_rate = the same bytes over ms_per_step, what the GPU moves per second in the timed loop.  "
                                                "bytes_the_timed_kernels_must_move prices every timed kernel by its own inputs and outputs "
                                                "(DESIGN.md 5); reference_walk_bytes_per_step is SURVEY 8(d)'s 24Q+25T+25B+40R counted by the "
                                                "level walk of the same batch — a table reads far fewer records, so that figure divided by "
                                                "the table path's time is not a bandwidth"}},
            "cold": {"ms": 1e3 * cold_s, "runs_ms": cold_runs, "value": nq / cold_s, "unit": "intervals/s", "plan_create_ms": 1e3 * t_plan,
                     "table_build_ms": cold_stats["composed_build_ms"], "composed_kind": cold_stats["composed_kind"],
                     "instrumented": cold_phases,
                     "what": "fresh plan (default policy: the table is built when the first batch reaches a quarter of the source's segments) + "
                             "one pass over the batch, wall clock with the device synchronised, in a process that has loaded its HIP code "
                             "objects on a 1 %-scale alignment before (a fresh process adds ~120 ms of module loading once)"},
            "walk": walk,
            "kernels_ms_per_step": {k: round(v["ms"] / n_kt, 4) for k, v in sorted(kt_acc.items())},
            "counts_per_step": {"queries": Q, "source_pieces": wcounts["source_pieces"], "top_derefs": T, "bottom_derefs": B,
                                "mapped_pieces": wcounts["mapped_pieces"], "records": R, "deferred_queries": st["deferred_queries"],
                                "general_queries": st["general_queries"], "table_records_dereferenced": table_records},
        }
        if sustained:
            out["sustained"] = sustained
        if one_plan:
            out["one_plan"] = one_plan
        if cached:
            out["cached"] = cached
        out["roofline"]["form"] = (("`value`'s: %d distinct batches in turn, each with a plan and buffers of its own, two in flight (0.65 GB per "
                                    "rotation at 4: past the Infinity Cache); kernel_avg_ms from the same rotation one batch at a time with HIP "
                                    "events; traffic from the PMC passes over this form" % K) if K > 1 else "one batch repeated")
        out["config"]["batches_rotating"] = K
        out["config"]["working_set_bytes_per_rotation"] = K * (17 * nq + 12 * nq + 4 * nq) + 40 * sum(records[:K]) + 16 * st["composed_records"]
        if mapping_only:
            out["mapping_only"] = mapping_only
        out["timed_step"] = "map+allgather" if exchanging else "map_only"
        out["config"]["batches_in_flight"] = in_flight
        # The line as it stands is complete (metric, value, roofline); the legs below stand beside it.  Should one of them hang — a
        # kernel that never ends, a child that cannot be reaped — the line is printed as far as it has got after a quarter of an hour
        legs_done = None
        if world == 1:
            import threading
            legs_done = threading.Event()

            def line_anyway(line_so_far=out, done=legs_done):
                if done.wait(900.0):
                    return
                for _ in range(5):
                    try:
                        snap = json.dumps(dict(line_so_far, secondary_legs="timed out after 900 s: the legs that had finished are here"))
                        break
                    except RuntimeError:  # (the main thread added a key meanwhile)
                        time.sleep(0.05)
                else:
                    return
                sys.stdout.flush()
                print(snap, flush=True)
                os._exit(0)
            threading.Thread(target=line_anyway, daemon=True).start()
        if col_result:
            ncol, col_ms, gather_ms, mean_depth = col_result
            cst = al.columns_depth_stats(src, 0, ncol)
            # algorithmic bytes of the column path: 25 B per segment record of the closure, computed once per
            # reference piece (= per run of columns with one walk, which is how the kernel works), + the 4-byte result
            col_bytes = 25.0 * (cst["top_derefs"] + cst["bottom_derefs"]) + 4.0 * ncol
            col_gbs = col_bytes / world / (col_ms * 1e-3) / 1e9
            col_kernel = "k_sweep_up"  # (requests of a million columns or more take the tree sweeps; k_depth_runs otherwise)
            col_traffic, col_src = pmc_traffic(col_kernel)
            sweep_bytes = sweep_design_bytes(al, src)
            out["columns"] = {"metric": "alignment-depth columns/sec (ColumnIterator closure per reference base)",
                              "value": ncol / (col_ms * 1e-3), "unit": "columns/s", "columns": ncol, "kernel_ms": col_ms,
                              "n_gpus": world, "all_gather_ms": gather_ms,
                              "reference_genome": src_name, "mean_depth": mean_depth,
                              "roofline": {"bound": "hbm", "kernel": col_kernel, "achieved": col_gbs, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": col_gbs / HBM_PEAK_GBS, "traffic": col_traffic, "traffic_source": col_src,
                                           "algorithmic_bytes_per_launch": col_bytes / world,
                                           "top_derefs": cst["top_derefs"], "bottom_derefs": cst["bottom_derefs"],
                                           "sweeps_own_bytes": sweep_bytes,
                                           "sweeps_own_GBs": sweep_bytes / (col_ms * 1e-3) / 1e9 if world == 1 else None,
                                           "sweeps_own_frac": sweep_bytes / (col_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
                                           "note": "achieved = SURVEY 8(d)'s 25 B per segment record of the closure (counted by the column walk's "
                                                   "counting instantiation) + 4 B per column, over the time of all sweep kernels of one call "
                                                   "(k_sweep_fill/up/top/down/out) — the reference's per-column walk priced at the sweeps' time: an effective rate, "
                                                   "above the peak where a sweep visits a segment record once and the walk visits it from every reference "
                                                   "piece that reaches it (50-genome alignments).  sweeps_own_*: what the two sweeps themselves must move "
                                                   "(bench.py: sweep_design_bytes: the per-base genome-set tracks written and read once, the segment "
                                                   "records, the 4-byte depths).  traffic: k_sweep_up launches only; per-GPU figures when n_gpus > 1"}}
        _mark("hal2maf over the first columns")
        if want_maf:
            # halAlignmentDepth as the tool delivers it: the wig text of the whole genome in host memory (values over PCIe, the lines made
            # by the host's threads)
            try:
                al.alignment_depth_bytes(src, length=min(al.genome_length(src), 2000000))
                wruns = []
                for _ in range(2):  # (the text's beginning is kept for the parity gate below: cpu_baseline; the call itself is timed)
                    wbytes, wig_head, w_s = al.alignment_depth_bytes(src, prefix=4 * args.cpu_columns + 4096)
                    wruns.append(w_s)
                kept["wig_head"] = wig_head
                out.setdefault("columns", {})["depth_wig"] = {"what": "hgx_alignment_depth end to end: wig text of %s's whole genome in host memory (PCIe inclusive); "
                                                                      "the better of two" % src_name,
                                                              "value": al.genome_length(src) / min(wruns), "unit": "columns/s", "seconds": min(wruns),
                                                              "runs_seconds": wruns, "wig_bytes": wbytes}
            except Exception as e:  # (a leg beside the line, not the line)
                out.setdefault("columns", {})["depth_wig"] = {"error": str(e)[:300]}
            # BASELINE config 3: hal2maf --refGenome <leaf> --noAncestors, end to end (column kernels, row fetch, block state
            # machine, text rendering) over the first N reference columns
            ncols = min(args.maf_columns, al.genome_length(src))
            al.maf_export_bytes(src, start=0, length=min(ncols, 200000), no_ancestors=True)  # (DNA upload, code objects)
            nbytes, maf_head, dt_m = al.maf_export_bytes(src, start=0, length=ncols, no_ancestors=True, prefix=56 * args.cpu_columns + 65536)
            if ncols > args.cpu_columns:
                kept["maf_head"] = maf_head
            out.setdefault("columns", {})["hal2maf"] = {"metric": "MAF columns/sec (hal2maf --refGenome %s --noAncestors, end to end to MAF text in host memory)" % src_name,
                                                        "value": ncols / dt_m, "unit": "columns/s", "columns": ncols, "seconds": dt_m,
                                                        "maf_bytes": nbytes}
        _mark("wide: int64 tables")
        if args.wide and world == 1 and not args.exchange_selftest:
            # the reference's own coordinate width (hal_index_t = int64, api/inc/halDefs.h:34; what an alignment with a genome of
            # 2^31 bases or more — every mammalian one — runs on): the same alignment, batch and steps on int64 tables
            os.environ["HGX_FORCE_WIDE"] = "1"
            try:
                alw = al.clone_to_device(local)
            finally:
                del os.environ["HGX_FORCE_WIDE"]
            w = steady_legs(hal_amd, torch, alw, src, tgt, d_gs, d_ge, d_st, args.steps, sync, rec_bytes=32,
                            streams=streams if in_flight == 2 else None)
            w["what"] = ("the timed configuration on int64-coordinate tables (HGX_FORCE_WIDE=1 copy of the same alignment: 32-byte table "
                         "records): same batch, same records")
            w["records_match"] = w["records_per_step"] == st["records"]  # (the same batch: the rotation's first)
            w["ratio_to_int32_step"] = w["ms_per_step"] / (1e3 * elapsed / args.steps)
            if one_plan:
                w["one_plan"]["ratio_to_int32"] = w["one_plan"]["ms_per_step"] / one_plan["ms_per_step"]
            out["wide"] = w
            del alw
        _mark("configs 4 and 5 on the 50-genome alignment")
        if args.cfg4 and world == 1 and not args.exchange_selftest and args.workload == "cfg2":
            # BASELINE configs 4 and 5 on this GPU: one GPU's 1.25 M-interval shard of the 10 M intervals Genome_44 -> Genome_2 on the
            # 50-genome alignment, and the whole-genome depth scan of Genome_44
            t0 = time.time()
            al4 = hal_amd.Alignment.random(workload_options(args.scale, "cfg4"), device=local)
            gen4 = time.time() - t0
            s4, t4 = al4.genome_id("Genome_44"), al4.genome_id("Genome_2")
            _, ss4, len4 = al4.sequences(s4)[0]
            nq4 = 1250000
            st4, ln4, sd4 = make_queries(len4, nq4, 1234)
            c4 = steady_legs(hal_amd, torch, al4, s4, t4, (st4 + ss4).to(dev), (st4 + ln4 - 1 + ss4).to(dev), sd4.to(dev), max(5, args.steps // 4), sync,
                             streams=streams if in_flight == 2 else None)
            c4["what"] = ("BASELINE config 4, one GPU's shard: halRandGen 50-genome alignment (seed 0, meanDegree 2), %d BED6 intervals "
                          "Genome_44 -> Genome_2" % nq4)
            c4["generate_s"] = round(gen4, 2)
            out["cfg4"] = c4
            ncol4 = al4.genome_length(s4)
            d4 = torch.empty(ncol4, dtype=torch.int32, device=dev)
            al4.columns_depth_device(s4, 0, ncol4, d4.data_ptr())
            ms4 = min(al4.columns_depth_device(s4, 0, ncol4, d4.data_ptr()) for _ in range(3))
            sw4 = sweep_design_bytes(al4, s4)
            out["cfg5"] = {"what": "BASELINE config 5 on one GPU: halAlignmentDepth of Genome_44 over its whole genome on the 50-genome alignment",
                           "value": ncol4 / (ms4 * 1e-3), "unit": "columns/s", "columns": ncol4, "kernel_ms": ms4,
                           "mean_depth": float(d4.double().mean().item()),
                           "sweeps_own_bytes": sw4, "sweeps_own_GBs": sw4 / (ms4 * 1e-3) / 1e9, "sweeps_own_frac": sw4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the same as the tool delivers it: wig text in host memory (values over PCIe, the lines made by the host's threads)
            try:
                al4.alignment_depth_bytes(s4, length=min(ncol4, 2000000))
                wruns = []
                for _ in range(2):
                    wbytes, wig4_head, w_s = al4.alignment_depth_bytes(s4, prefix=4 * args.cpu_columns_cfg5 + 4096)
                    wruns.append(w_s)
                out["cfg5"]["wig"] = {"what": "hgx_alignment_depth end to end: the whole genome's wig text in host memory (PCIe inclusive); the better of two",
                                      "value": ncol4 / min(wruns), "unit": "columns/s", "seconds": min(wruns), "runs_seconds": wruns, "wig_bytes": wbytes}
            except Exception as e:  # (a leg beside the line, not the line)
                out["cfg5"]["wig"] = {"error": str(e)[:300]}
                wig4_head = None
            _mark("config 5: the oracle's depth loop (cpu_baseline)")
            if args.cpu_columns_cfg5 > 0 and args.cpu_sample > 0:
                # the CPU figure beside config 5, and its parity gate: the oracle's halAlignmentDepth loop over the genome's first columns
                try:
                    with tempfile.TemporaryDirectory() as tmp4:
                        img4 = os.path.join(tmp4, "cfg4.hgx")
                        al4.save(img4)
                        n5 = min(args.cpu_columns_cfg5, ncol4)
                        base5, text5 = cpu_columns_baseline(img4, "depth", "Genome_44", al4.sequences(s4)[0][0], n5, tmp4, "cfg5",
                                                            all_cores_total=4 * n5 if args.cpu_all_cores else 0)
                    base5["parity_with_gpu"] = wig4_head is not None and wig4_head[:len(text5)] == text5
                    base5["parity"] = "the oracle's wig of the first %d columns is the beginning of the timed call's text" % n5
                    out["cfg5"]["cpu_baseline"] = base5
                except Exception as e:
                    out["cfg5"]["cpu_baseline"] = {"error": str(e)[:300]}
            del al4, d4
        _mark("the text path")
        if args.text_path and world == 1 and not args.exchange_selftest:
            # Liftover::convert as halLiftover runs it: BED text in, BED text out (parse, H2D, kernels, D2H, format), PCIe inclusive
            sn, ln, tn = starts.numpy(), lens.numpy(), strand.numpy()
            bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(a), int(a + b), chr(int(c))) for a, b, c in zip(sn, ln, tn)).encode()
            hal_amd.liftover_convert_bytes(al, src, bed[:bed.index(b"\n", min(4000000, len(bed) // 2)) + 1], tgt)  # (code objects, plan, pinned buffers)
            out_bytes, out_lines = hal_amd.liftover_convert_bytes(al, src, bed, tgt)
            best_t = None
            for _ in range(3):
                t0 = time.perf_counter()
                hal_amd.liftover_convert_bytes(al, src, bed, tgt, count_lines=False)
                dt_t = time.perf_counter() - t0
                best_t = dt_t if best_t is None else min(best_t, dt_t)
            out["end_to_end"] = {"what": "hgx_liftover_convert = Liftover::convert: BED6 text of the batch in host memory -> lifted BED text "
                                         "in host memory (tokenise, H2D, kernels, D2H, render; PCIe inclusive, never `value`); best of 3",
                                 "value": nq / best_t, "unit": "intervals/s", "seconds": best_t, "lines_in": nq, "lines_out": out_lines,
                                 "bytes_in": len(bed), "bytes_out": out_bytes, "host_threads": min(64, host_cpus())}
        _mark("features")
        if args.features and want_maf:
            # the other entry points of the path, each as a number: halGetBlocksInTargetRange (a browser's call: latency per call and
            # ranges per second when a call carries many), hal2maf --unique (what hal2mafMP.py runs every slice with), --maxRefGap,
            # and hgx_maf_export_multi's slices over two handles of this GPU
            feats = {}
            sn, ln = starts.numpy(), lens.numpy()
            t_chrom = seq_name
            r1k = [(int(a), int(a + b)) for a, b in zip(sn[:1000], ln[:1000])]
            big = [(int(a), int(a) + 100000) for a in sn[:64] if int(a) + 100000 < length]
            kwv = dict(dup_mode=2, adjacencies=True, decode=False)
            al.blocks_in_target_ranges(tgt_name, src_name, t_chrom, r1k[:10], **kwv)  # (plans, tables, code objects)
            al.blocks_in_target_ranges(tgt_name, src_name, t_chrom, r1k, **kwv)
            lat = []
            for r in r1k[:100]:
                t0 = time.perf_counter()
                al.blocks_in_target_ranges(tgt_name, src_name, t_chrom, [r], **kwv)
                lat.append(time.perf_counter() - t0)
            lat.sort()
            lat_big = []
            for r in big[:20]:
                t0 = time.perf_counter()
                al.blocks_in_target_ranges(tgt_name, src_name, t_chrom, [r], **kwv)
                lat_big.append(time.perf_counter() - t0)
            lat_big.sort()
            t0 = time.perf_counter()
            for _ in range(3):
                al.blocks_in_target_ranges(tgt_name, src_name, t_chrom, r1k, **kwv)
            dt_b = (time.perf_counter() - t0) / 3
            feats["blocks_in_target_range"] = {
                "what": "hgx_get_blocks_in_target_range[s] = halGetBlocksInTargetRange (target %s, query %s, dupMode HAL_QUERY_AND_TARGET_DUPS, "
                        "adjacencies mapped back): the batch's first intervals (50..1000 bases) as ranges, results left in library memory and "
                        "released" % (src_name, tgt_name),
                "one_range_per_call": {"median_us": 1e6 * lat[len(lat) // 2], "p90_us": 1e6 * lat[int(0.9 * len(lat))], "ranges_per_s": len(lat) / sum(lat),
                                       "calls": len(lat)},
                "one_100kb_range_per_call": {"median_us": 1e6 * lat_big[len(lat_big) // 2] if lat_big else None, "calls": len(lat_big)},
                "thousand_ranges_per_call": {"ranges_per_s": len(r1k) / dt_b, "ms_per_call": 1e3 * dt_b}}
            ncu = min(2000000, al.genome_length(src))
            al.maf_export_bytes(src, 0, start=0, length=min(ncu, 200000), no_ancestors=True, unique=True)
            t0 = time.perf_counter()
            nb_u = al.maf_export_bytes(src, 0, start=0, length=ncu, no_ancestors=True, unique=True)
            dt_u = time.perf_counter() - t0
            t0 = time.perf_counter()
            nb_p = al.maf_export_bytes(src, 0, start=0, length=ncu, no_ancestors=True)
            dt_p = time.perf_counter() - t0
            feats["hal2maf_unique"] = {"what": "hal2maf --refGenome %s --noAncestors --unique over the first %d columns, to MAF text in host memory; "
                                               "which columns the visit cache lets through is decided on the device (k_column_unique_count), "
                                               "the written ones take the run-compressed path" % (src_name, ncu),
                                       "value": ncu / dt_u, "unit": "columns/s", "seconds": dt_u, "maf_bytes": nb_u,
                                       "same_range_without_unique": {"value": ncu / dt_p, "seconds": dt_p, "maf_bytes": nb_p}}
            ncg = min(200000, al.genome_length(src))
            al.maf_export_bytes(src, 0, start=0, length=20000, no_ancestors=True, max_ref_gap=100)
            t0 = time.perf_counter()
            nb_g = al.maf_export_bytes(src, 0, start=0, length=ncg, no_ancestors=True, max_ref_gap=100)
            dt_g = time.perf_counter() - t0
            feats["hal2maf_max_ref_gap"] = {"what": "hal2maf --noAncestors --maxRefGap 100 over the first %d columns (the iterator's stack of inserted and "
                                                    "deleted ranges and its visit caches are replayed on the host over device columns: host-bound)" % ncg,
                                            "value": ncg / dt_g, "unit": "columns/s", "seconds": dt_g, "maf_bytes": nb_g}
            clones = [al, al.clone_to_device(local)]
            ncm = min(8000000, al.genome_length(src))
            hal_amd.maf_export_multi(clones, src, 0, start=0, length=min(ncm, 400000), slice_size=100000, no_ancestors=True, unique=True, size_only=True)
            t0 = time.perf_counter()
            nb_m = hal_amd.maf_export_multi(clones, src, 0, start=0, length=ncm, slice_size=1000000, no_ancestors=True, unique=True, size_only=True)
            dt_m2 = time.perf_counter() - t0
            feats["maf_export_multi"] = {"what": "hgx_maf_export_multi: hal2mafMP.py's recipe (slices of 1 M reference columns, --unique, an export each) over "
                                                 "two handles of this one GPU, first %d columns" % ncm,
                                         "value": ncm / dt_m2, "unit": "columns/s", "seconds": dt_m2, "maf_bytes": nb_m, "handles": 2, "gpus": 1}
            del clones
            try:
                import numpy as np
                if not want_maf:
                    raise RuntimeError("skipped: the alignment was generated without DNA (--maf-columns 0)")
                # halLiftover --outPSL of BED12 lines: the general path of Liftover::convert (a BedLine a line, blocks lifted one by one,
                # assignBlocksToIntervals, PSL columns counted from both genomes' DNA on the host), parsed and rendered by the host's threads
                npsl = 200000
                rs = np.random.default_rng(11)
                b12 = []
                for i, (a0, l0) in enumerate(zip(sn[:npsl], ln[:npsl])):
                    a0, l0 = int(a0), int(l0)
                    cut = sorted(int(x) for x in rs.choice(np.arange(1, l0), size=3, replace=False))
                    b12.append("%s\t%d\t%d\tn%d\t0\t%s\t%d\t%d\t0\t2\t%d,%d,\t0,%d," % (seq_name, a0, a0 + l0, i, "+-"[i & 1], a0, a0 + l0,
                                                                                       cut[0], l0 - cut[1], cut[1]))
                psl_in = ("\n".join(b12) + "\n").encode()
                hal_amd.liftover_convert_bytes(al, src, ("\n".join(b12[:2000]) + "\n").encode(), tgt, out_psl=True)
                t0 = time.perf_counter()
                nb_psl, ln_psl = hal_amd.liftover_convert_bytes(al, src, psl_in, tgt, out_psl=True)
                dt_psl = time.perf_counter() - t0
                feats["liftover_psl"] = {"what": "hgx_liftover_convert --outPSL: %d BED12 lines of two blocks (the batch's first intervals) to PSL text in host "
                                                 "memory: the general path of Liftover::convert (hgx_liftover_host.cpp), needs the alignment's DNA" % npsl,
                                         "value": npsl / dt_psl, "unit": "lines/s", "seconds": dt_psl, "bytes_in": len(psl_in), "bytes_out": nb_psl,
                                         "lines_out": ln_psl}
            except Exception as e:  # (a leg beside the line, not the line)
                feats["liftover_psl"] = {"error": str(e)[:300]}
            out["features"] = feats
        if args.cpu_sample > 0 and world == 1:  # (the CPU baseline: rank 0 at N = 1 only — with more ranks the others would stand waiting)
            sample = min(args.cpu_sample, nq)
            cpu_tmp = tempfile.TemporaryDirectory()
            img = os.path.join(cpu_tmp.name, "bench.hgx")
            al.save(img)
            cst, text, multi = cpu_baseline(al, src_name, tgt_name, starts, lens, strand, seq_name, sample,
                                            all_cores_sample=min(nq, 4 * sample) if args.cpu_all_cores else 0, img=img)
            _mark("the CPU figures beside the column metric")
            # ---- the CPU figures beside the column metric ("MAF columns/sec ... vs CPU ref"), each with the parity gate of the timed leg
            # it stands beside: the oracle's loops over the reference genome's first --cpu-columns columns ----
            ncc = min(args.cpu_columns, al.genome_length(src))
            if ncc > 0 and "columns" in out:
                try:
                    based, textd = cpu_columns_baseline(img, "depth", src_name, seq_name, ncc, cpu_tmp.name, "cfg2d",
                                                        all_cores_total=4 * ncc if args.cpu_all_cores else 0)
                    if "wig_head" in kept:
                        based["parity_with_gpu"] = kept["wig_head"][:len(textd)] == textd
                        based["parity"] = "the oracle's wig of the first %d columns is the beginning of columns.depth_wig's timed text" % ncc
                    else:  # (no wig leg in this run: the values of the depth kernel's leg)
                        import numpy as np
                        vals = al.columns_depth(src, 0, ncc)
                        based["parity_with_gpu"] = bool(np.array_equal(np.array(textd.split(b"\n")[1:-1], dtype=np.int64), vals))
                        based["parity"] = "the oracle's values of the first %d columns against hgx_columns_depth's" % ncc
                    out["columns"]["cpu_baseline"] = based
                except Exception as e:  # (a leg beside the line, not the line)
                    out["columns"]["cpu_baseline"] = {"error": str(e)[:300]}
                if want_maf and "hal2maf" in out["columns"] and "maf_head" in kept:
                    try:  # (the CPU figure beside the 8 M-column leg when the whole-genome leg is off; else that leg's child brings its own)
                        if not args.maf_full:
                            basem, textm = cpu_columns_baseline(img, "maf", src_name, seq_name, ncc, cpu_tmp.name, "cfg2m",
                                                                all_cores_total=4 * ncc if args.cpu_all_cores else 0, extra=("--noAncestors",))
                            basem["parity_with_gpu"] = maf_prefix_matches(textm, kept["maf_head"])
                            basem["parity"] = ("the oracle's MAF of the first %d columns, up to its last block, is the beginning of the timed export's "
                                               "text" % ncc)
                            out["columns"]["hal2maf"]["cpu_baseline"] = basem
                    except Exception as e:
                        out["columns"]["hal2maf_cpu_baseline"] = {"error": str(e)[:300]}
            _mark("halGetBlocksInTargetRange on the CPU")
            # ---- halGetBlocksInTargetRange on the CPU: the oracle's getBlocksInTargetRange, a call per range, the same ranges ----
            if "features" in out and "blocks_in_target_range" in out["features"]:
                try:
                    rfile, ofile = os.path.join(cpu_tmp.name, "ranges.txt"), os.path.join(cpu_tmp.name, "viz.out")
                    rr = [(int(a), int(a + b)) for a, b in zip(starts.numpy()[:100], lens.numpy()[:100])]
                    with open(rfile, "w") as f:
                        f.write("".join("%d %d\n" % r for r in rr))
                    vst = json.loads(subprocess.run([oracle_bin(), "blockviz", img, tgt_name, src_name, seq_name, "--ranges", rfile, "--out", ofile,
                                                     "--dupMode", "2", "--stats"], check=True, stdout=subprocess.PIPE).stdout.decode())
                    want = open(ofile).read()
                    got = al.blocks_in_target_ranges(tgt_name, src_name, seq_name, rr, dup_mode=2, adjacencies=True)
                    got_text = "".join("# %d %d\n" % r + hal_amd.format_block_results(b, d) for r, (b, d) in zip(rr, got))
                    out["features"]["blocks_in_target_range"]["cpu_baseline"] = {
                        "median_us": None, "mean_us": 1e6 * vst["seconds"] / max(1, vst["ranges"]), "ranges_per_s": vst["ranges"] / vst["seconds"],
                        "cores": 1, "kind": "port", "host_cpu": host_cpu_model(),
                        "sample": "the oracle's getBlocksInTargetRange (blockViz/impl/halBlockViz.cpp:759-827), one call per range, the first %d "
                                  "ranges of one_range_per_call" % len(rr),
                        "parity_with_gpu": got_text == want}
                except Exception as e:
                    out["features"]["blocks_in_target_range"]["cpu_baseline"] = {"error": str(e)[:300]}
            # parity spot check of the timed configuration: GPU records of the sampled intervals vs the oracle's text
            ptr, n = plan.run(d_gs[:sample].contiguous(), d_ge[:sample].contiguous(), d_st[:sample].contiguous())
            import numpy as np
            recs = plan.records_to_tensor(ptr, n).cpu().numpy().view(hal_amd.RECORD_DTYPE).reshape(-1)
            tname = al.sequences(tgt)[0][0]
            gpu_text = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (tname, r["tgt_start"], r["tgt_end"], r["strand"].decode()) for r in recs)
            cpu_model = host_cpu_model()
            cpu_tmp.cleanup()
            out["cpu_baseline"] = {"value": cst["intervals"] / cst["map_seconds"], "unit": "intervals/s", "cores": 1,
                                   "kind": "port", "host_cpu": cpu_model,
                                   "sample": "first %d intervals of rank 0's batch, oracle liftInterval+sort time only "
                                             "(BED parse/print and image load excluded)" % sample,
                                   "parity_with_gpu": gpu_text == text}
            if multi:
                out["cpu_baseline"]["all_cores"] = multi
        _mark("config 3 in a child process")
        if want_maf and args.maf_full:
            # BASELINE config 3 as stated: hal2maf over the full reference genome — in a process of its own (leg_hal2maf_full), the last
            # thing this rank does: the line above is complete whatever becomes of it.  Should it fail with round 5's device stage and
            # walk, it is run once more with round 4's (HGX_MAF_SWEEP=0, HGX_MAF_SLICED=0) and both outcomes are in the line.
            leg = run_leg_in_child("hal2maf_full", args)
            if "error" in leg:
                again = run_leg_in_child("hal2maf_full", args, env_extra={"HGX_MAF_SWEEP": "0", "HGX_MAF_SLICED": "0"})
                again["first_attempt"] = leg["error"]
                again["path"] = "the column walk and one thread's block state machine (HGX_MAF_SWEEP=0 HGX_MAF_SLICED=0) after the default path failed"
                leg = again
            out.setdefault("columns", {})["hal2maf_full"] = leg
        if legs_done is not None:
            legs_done.set()
    else:
        out = None
    _mark("the collated legs, last and under a watchdog")
    # ---- the collated legs, last and under a watchdog: a collective that hangs or fails here costs these legs, not the line ----
    want_collated = synced and not exchanging and world > 1
    want_writer = synced and (world > 1 or bool(args.exchange_selftest))
    if want_collated or want_writer:
        import threading
        finished = threading.Event()

        def give_up():
            if finished.wait(180.0):
                return
            if out is not None:
                out["collated_legs"] = "timed out after 180 s: not measured"
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
            os._exit(0)
        threading.Thread(target=give_up, daemon=True).start()
        legs = {}
        try:
            if want_collated:
                legs["collated"] = leg_collated()
            if want_writer:
                legs["collated_to_writer"] = leg_to_writer()
        except Exception as e:  # (every rank fails alike on a call that is not there; a hang is the watchdog's)
            legs["collated_legs_error"] = "%s: %s" % (type(e).__name__, e)
        finished.set()
        if out is not None:
            for k, v in legs.items():
                out[k] = v
            if "collated" in legs:
                # BASELINE config 4 names the all-gatherv as part of its workload: its number is the collated one, quoted beside `value`
                out["config4_as_stated"] = {"value": legs["collated"]["value"], "unit": "intervals/s", "ms_per_step": legs["collated"]["ms_per_step"],
                                            "what": "`collated`: every step's records gathered on every rank (RCCL all-gather of wire blobs); "
                                                    "`value` is the same steps without it"}
    result_line = json.dumps(out) if out is not None else None
    if synced:
        dist.destroy_process_group()
    # the JSON line is the last thing on stdout: whatever C libraries have buffered (RCCL prints its version banner through
    # stdio) goes out first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if result_line is not None:
        print(result_line, flush=True)


def sweep_design_bytes(al, ref):
    """What halAlignmentDepth's two tree sweeps (hgx_columns.hip: sweepTracks) must move for a scan of genome `ref` with every genome
    in scope: bottom-up, every genome with children writes its per-base genome-set track once and reads its children's (a child
    without children of its own carries a constant: no track), with one bottom-segment record and child link per segment and child
    and one top-segment record per child segment; a genome's set holds the genomes of its own subtree, in a word just wide enough
    for them.  Top-down along the path root -> ref, a genome reads its parent's depth (the root's: its track) and writes its
    depth per base (a byte: at most 64 genomes are counted per group); the result is read and written once (int32)."""
    n = al.num_genomes
    size = [0] * n

    def subtree(g):
        size[g] = 1 + sum(subtree(c) for c in al.genome_children(g))
        return size[g]
    roots = [g for g in range(n) if al.genome_parent(g) < 0]
    for r in roots:
        subtree(r)
    if n > 64:  # (groups of 64 genomes: one numbering, 8-byte words, a pass per group)
        word = [8.0] * n
        passes = (n + 63) // 64
    else:
        word = [1.0 if size[g] <= 8 else 2.0 if size[g] <= 16 else 4.0 if size[g] <= 32 else 8.0 for g in range(n)]
        passes = 1
    has_track = [len(al.genome_children(g)) > 0 for g in range(n)]
    total = 0.0
    for g in range(n):
        kids = al.genome_children(g)
        if not kids:
            continue
        total += word[g] * al.genome_length(g)  # its own track
        total += al.num_bottom_segments(g) * (8.0 + 4.0 * len(kids))  # BotRec + child links
        for c in kids:
            total += 16.0 * al.num_top_segments(c)  # the child's TopRec
            if has_track[c]:
                total += word[c] * al.genome_length(c)
    path = [ref]
    while al.genome_parent(path[-1]) >= 0:
        path.append(al.genome_parent(path[-1]))
    for c in path[:-1]:  # (every genome on the path below the root)
        p = al.genome_parent(c)
        total += 16.0 * al.num_top_segments(c) + 8.0 * al.num_top_segments(c)  # TopRec + the parent's BotRec behind it
        total += 1.0 * al.genome_length(c)  # depth written (a byte)
        total += (1.0 if al.genome_parent(p) >= 0 else word[p]) * al.genome_length(c)  # the parent's depth (the root's: its track)
    total *= passes
    total += 5.0 * al.genome_length(ref)  # depth read (a byte), result written (int32)
    return total


def plan_kernel_bytes(kt, st, steps, rec_bytes=16):
    """Algorithmic bytes per launch of each kernel: 25 B per segment record logically
    dereferenced by that kernel, 24 B per query for the locate kernel, 40 B per record written by the finishing kernel.
    kt: kernel times and dereference counts accumulated over `steps` runs; st: the counts of one run; rec_bytes: size of a
    table record (16 with int32 coordinates, 32 with int64)."""
    out = {}
    for name, v in kt.items():
        launches = max(1, v["launches"])
        t, b = v.get("top_derefs", 0), v.get("bot_derefs", 0)
        bytes_ = 25.0 * (t + b)
        if name in ("k_locate_composed", "k_locate_through"):
            # the table kernels' own figure: 16 B per composed record that overlaps its interval (counted in the top slot)
            # and the piece it leaves (29 B in a frontier, a 32-byte MappedRec from the whole-path kernel, which also writes
            # 8 B of offset and count per interval), instead of the segment records of the walk they replace
            bytes_ = (rec_bytes + (29.0 if name == "k_locate_composed" else 32.0)) * t
            if name == "k_locate_through":
                bytes_ += 8.0 * st["queries"] * steps
        if name == "k_lift_merged":
            # the storing kernel: 37 B per interval (its two ends, the strand, the 8-byte answer and the line count
            # k_lift_classify left, the 4-byte offset it writes), 16 B per merged record that overlaps its interval (top slot),
            # 40 B per record written
            bytes_ = rec_bytes * t + (16.0 + 1.0 + 8.0 + 4.0 + 4.0 + 4.0) * st["queries"] * steps + 40.0 * st["records"] * steps
        if name == "k_lift_classify":
            # the counting kernel: interval ends, one 4-byte bucket entry, the 8-byte answer and the 4-byte line count, and 16 B
            # per merged record that overlaps its interval (k_lift_merged's top slot: the same records) or unmerged record a
            # general interval clips (its own top slot)
            merged_records = kt.get("k_lift_merged", {}).get("top_derefs", 0)
            bytes_ = (16.0 + 4.0 + 8.0 + 4.0) * st["queries"] * steps + rec_bytes * (merged_records + t)
        if name == "k_lift_general_list":
            # the pass in front of k_lift_classify: interval ends, a bit per interval written
            bytes_ = (16.0 + 0.125) * st["queries"] * steps
        if name in ("k_locate_expand", "k_locate_composed", "k_locate_through"):
            bytes_ += 24.0 * st["queries"] * steps
        if name in ("k_finish_fast", "k_finish_lds", "k_finish_big"):
            # 40 B per output record.  The fast kernels write nearly all of them; the split between the fast and
            # the general kernel is not counted, so each is credited with all records (an upper bound for either)
            bytes_ += 40.0 * st["records"] * steps
        if name == "k_compact_records":
            bytes_ += 80.0 * st["records"] * steps
        out[name] = bytes_ / launches
    return out


if __name__ == "__main__":
    main()
