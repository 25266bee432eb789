"""halAlignmentDepth by tree sweeps on cfg2 / cfg5: the tracks in the subtree's own numbering and width (default) against one numbering and width for all (HGX_SWEEP_LOCAL=0)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hal_amd
from bench import workload_options
for wl, refname in (("cfg2", "Genome_9"), ("cfg4", "Genome_44")):
    al = hal_amd.Alignment.random(workload_options(1.0, wl), device=0)
    g = al.genome_id(refname)
    n = al.genome_length(g)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["HGX_SWEEP_LOCAL"] = mode
        al.columns_depth_device(g, 0, n, out.data_ptr())
        ms = min(al.columns_depth_device(g, 0, n, out.data_ptr()) for _ in range(3))
        res.setdefault(mode, []).append(round(ms, 3))
        chk = int(out.long().sum().item())
        print(wl, "local" if mode == "1" else "global", ms, "ms  sum", chk, flush=True)
