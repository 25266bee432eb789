"""bench.py's control flow run end to end without a GPU (tests/support/bench_fakes.py): torch.cuda replaced by no-ops, the liftover
plans by fakes, everything else real at toy size — the alignment generator, the column engine on the host-side emulation, the
oracle's CPU baselines with their parity gates.  One JSON line with every leg must come out; the numbers in it mean nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.fixture(scope="module")
def emulation_lib(tmp_path_factory):
    lib = str(tmp_path_factory.mktemp("emu") / "libhgx_cpu.so")
    subprocess.check_call([os.path.join(ROOT, "tests", "cpp", "build_cpu_emulation.sh"), lib])
    return lib


def test_bench_with_three_ranks(emulation_lib, tmp_path):
    """the N-rank control flow (python -m torch.distributed.run ... bench.py --gpus N): gloo in the place of RCCL, every ordinal of the
    emulated device the host — barriers, the slowest rank's time, the rotating batches per rank, the collated legs under their
    watchdog; ONE line from rank 0, without the CPU baselines (N = 1 only)"""
    script = tmp_path / "bench_n.py"
    script.write_text("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                      "import bench_fakes; bench_fakes.install()\n"
                      "import bench\n"
                      "sys.argv = ['bench.py', '--gpus', '3', '--steps', '5', '--warmup', '2', '--scale', '0.002', '--queries', '2000',\n"
                      "            '--sustained-seconds', '0.05', '--cpu-sample', '500']\n"
                      "bench.main()\n" % (ROOT, os.path.join(ROOT, "tests", "support")))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HGX_LIB_PATH=emulation_lib, HGX_COL_GRID="4", HGX_EMULATED_DEVICES="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["scaling"] == "weak" and out["steps"] == 5 and out["config"]["batches_rotating"] == 4
    assert out["config"]["parallelism"] == "query-shard x3" and "roofline" in out and "cpu_baseline" not in out
    for leg in ("collated", "collated_to_writer", "config4_as_stated"):
        assert out[leg]["value"] > 0, leg
    assert "collated_legs_error" not in out and "collated_legs" not in out


def test_bench_prints_its_line_with_every_leg(emulation_lib, tmp_path):
    lib = emulation_lib
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import bench_fakes; bench_fakes.install()\n"
            "import bench\n"
            "bench_fakes.install_leg_child(bench)\n"
            "sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--scale', '0.002', '--queries', '3000', '--maf-columns', '20000',\n"
            "            '--cpu-sample', '500', '--cpu-columns', '20000', '--cpu-columns-cfg5', '5000', '--sustained-seconds', '0.05']\n"
            "bench.main()\n") % (ROOT, os.path.join(ROOT, "tests", "support"))
    env = dict(os.environ, HGX_LIB_PATH=lib, HGX_COL_GRID="4", HGX_MAF_SWEEP="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    line = r.stdout.decode().strip().splitlines()[-1]
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["steps"] == 6 and out["warmup"] == 2 and out["n_gpus"] == 1 and out["config"]["batches_rotating"] == 4
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernels", "form"}
    assert "cached" in out and "one_plan" in out and "walk" in out and "cold" in out
    cols = out["columns"]
    assert cols["cpu_baseline"]["parity_with_gpu"] is True, cols["cpu_baseline"]      # the real column engine (emulated) vs the oracle
    assert cols["hal2maf_full"]["cpu_baseline"]["parity_with_gpu"] is True, cols["hal2maf_full"]["cpu_baseline"]
    assert cols["hal2maf_full"]["device_stage"]["state"].startswith("checked"), cols["hal2maf_full"]["device_stage"]
    assert cols["hal2maf_full"]["device_stage"]["last_export"]["walk"] in ("one thread", "slices of the export side by side")
    assert cols["hal2maf_full"]["process"].startswith("a child")
    uniq = cols["hal2maf_full"]["unique"]
    assert uniq["same_text"] is True and uniq["device_stage"]["state_unique"].startswith("checked"), uniq
    assert out["cfg5"]["cpu_baseline"]["parity_with_gpu"] is True, out["cfg5"]
    assert "all_cores" in cols["cpu_baseline"] and "all_cores" in out["cpu_baseline"]
    assert "blocks_in_target_range" in out["features"] and "cpu_baseline" in out["features"]["blocks_in_target_range"]
