"""Depth-kernel time against the number of resident blocks (HGX_COL_GRID): the frame stacks live in scratch, and the
hot part of 8192 wavefronts' stacks does not fit the L2s."""
import os, subprocess, sys, json
for grid in (256, 512, 768, 1024, 1536, 2048, 4096):
    env = dict(os.environ, HGX_COL_GRID=str(grid))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--queries", "1000"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()[-1]
    c = json.loads(out)["columns"]
    print("grid %5d: %.2f ms, %.3f G columns/s" % (grid, c["kernel_ms"], c["value"] / 1e9), flush=True)
