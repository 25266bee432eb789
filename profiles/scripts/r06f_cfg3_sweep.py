"""Config 3 (hal2maf --noAncestors over Genome_9's whole genome, text in host memory) under several settings of the host side's
threads, in one process: the GPU boxes' containers have a CPU quota of 16 (cpu.max "1600000 100000") beside their 256 hardware
threads — which settings finish first when the quota, not the cores, is the budget?  Each line: the setting, the seconds of two
exports, the library's own account of the last one (rounds, waits, CPU seconds per stage with HGX_MAF_TIMING on stderr)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HGX_MAF_TIMING", "1")
import hal_amd  # noqa: E402
from bench import workload_options  # noqa: E402


def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e6
    except Exception:
        return 0, 0.0


def main():
    al = hal_amd.Alignment.random(workload_options(1.0, "cfg2", dna="fast"), device=0)
    src = al.genome_id("Genome_9")
    ncols = al.genome_length(src)
    al.maf_export_bytes(src, start=0, length=200000, no_ancestors=True)
    settings = [
        {},
        {"HGX_MAF_SLICED": "0"},
        {"HGX_MAF_SLICED": "1", "HGX_MAF_WALK_THREADS": "8", "HGX_MAF_RENDER_THREADS": "4", "HGX_MAF_RENDERS_IN_FLIGHT": "2"},
        {"HGX_MAF_SLICED": "1", "HGX_MAF_WALK_THREADS": "12", "HGX_MAF_RENDER_THREADS": "6", "HGX_MAF_RENDERS_IN_FLIGHT": "2"},
        {"HGX_MAF_SLICED": "1", "HGX_MAF_WALK_THREADS": "16", "HGX_MAF_RENDER_THREADS": "8", "HGX_MAF_RENDERS_IN_FLIGHT": "2"},
        {"HGX_MAF_SLICED": "1", "HGX_MAF_WALK_THREADS": "16", "HGX_MAF_RENDER_THREADS": "16", "HGX_MAF_RENDERS_IN_FLIGHT": "1"},
        {"HGX_MAF_SLICED": "1", "HGX_MAF_WALK_THREADS": "32", "HGX_MAF_RENDER_THREADS": "16", "HGX_MAF_RENDERS_IN_FLIGHT": "2"},
        {"HGX_MAF_SLICED": "0", "HGX_MAF_RENDER_THREADS": "15", "HGX_MAF_RENDERS_IN_FLIGHT": "1"},
        {"HGX_MAF_SLICED": "0", "HGX_MAF_RENDER_THREADS": "8", "HGX_MAF_RENDERS_IN_FLIGHT": "2"},
        {"HGX_MAF_SLICED": "0", "HGX_MAF_SWEEP": "0"},
    ]
    extra = os.environ.get("R06F_EXTRA")
    if extra:
        settings = json.loads(extra)
    for st in settings:
        for k, v in st.items():
            os.environ[k] = v
        runs = []
        th0 = throttled()
        for _ in range(2):
            sys.stderr.write("==== %s\n" % json.dumps(st))
            sys.stderr.flush()
            nbytes, _, s = al.maf_export_bytes(src, no_ancestors=True, prefix=4096)
            runs.append(round(s, 4))
        th1 = throttled()
        info = al.maf_tracks_info().get("last_export", {})
        print(json.dumps({"setting": st, "seconds": runs, "columns_per_s": round(ncols / min(runs)), "maf_bytes": nbytes,
                          "throttled_periods": th1[0] - th0[0], "throttled_s": round(th1[1] - th0[1], 3),
                          "last_export": {k: info.get(k) for k in ("walk", "slices", "rounds", "walk_threads", "seconds_until_the_batches_were_there",
                                                                     "seconds_of_the_rounds", "seconds")}}), flush=True)
        for k in st:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
