"""The column engine on a machine without a GPU: libhgx's own sources for that path — the launch sequences, buffers and kernels of
hal_amd/csrc/hgx_columns.hip, the upload code of hgx_device_image.hip, the host side above them — compiled by g++ against
tests/cpp/hipshim (device memory is host memory, a launch is a loop over the grid's threads, barriers by fibers) and run through
the same ctypes binding and the same `-m gpu` tests, in a process of their own, against the oracle.  What it is for: this
container has no GPU, and the GPU box is minutes away and rationed; the emulation runs the engine's host-side launch code and the
kernels' arithmetic — not their timing, not the wavefront — under the tests that the GPU run repeats.  Test infrastructure:
libhgx.so is never built this way, the liftover engine (DPP, readlane, ballots) is not part of it, and no test marked `gpu`
counts as run by it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the round-5 device stage of hal2maf and its neighbours: a subset sized for the CPU suite (the whole column suite passes under the
# emulation too — profiles/r05_notes.md has that run — but takes a quarter of an hour)
CASES = [
    "tests/test_gpu_zz_round5.py::test_maf_tracks_reference_goldens",
    "tests/test_gpu_zz_round5.py::test_maf_tracks_unique_small",
    "tests/test_gpu_zz_round5.py::test_hal2maf_over_the_ranks_of_a_node_every_rank_a_writer",  # (1, 2 and 3 processes, for real)
    "tests/test_gpu_zz_round5.py::test_count_dupes_sweep_over_a_polytomy_with_segment_tails",
    "tests/test_gpu_zz_round5.py::test_depth_wig_through_several_chunks[97]",
    "tests/test_gpu_columns.py::test_reference_cli_goldens_hal2maf",
    "tests/test_gpu_columns.py::test_depth_handbuilt_and_columns_api",
    "tests/test_gpu_columns.py::test_maf_handbuilt_inversions",
    "tests/test_gpu_columns.py::test_reference_goldens_of_the_multiprocess_driver",
    "tests/test_gpu_columns.py::test_reference_unit_tests_of_the_column_iterator",
]


@pytest.fixture(scope="module")
def emulation_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libhgx_cpu.so")
    subprocess.check_call([os.path.join(ROOT, "tests", "cpp", "build_cpu_emulation.sh"), out])
    return out


def test_column_engine_on_the_host_side_emulation(emulation_lib):
    env = dict(os.environ, HGX_LIB_PATH=emulation_lib, HGX_COL_GRID="4", HGX_SWEEP_GRID="16")
    env.pop("HGX_MAF_SWEEP", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + CASES, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = r.stdout.decode()[-3000:]
    assert r.returncode == 0, tail
    assert "%d passed" % len(CASES) in tail, tail


def test_emulation_refuses_the_liftover_engine(emulation_lib):
    """the emulation is the column engine only: a liftover through it fails loudly instead of answering"""
    code = ("import hal_amd, sys\n"
            "al = hal_amd.Alignment.random(hal_amd.RandOptions.preset('small', seed=0), device=0)\n"
            "try:\n"
            "    hal_amd.liftover_convert(al, 0, 'Genome_0_seq\\t1\\t50\\n', 2)\n"
            "except hal_amd.HgxError as e:\n"
            "    assert 'not part of the host-side emulation' in str(e), e\n"
            "    sys.exit(0)\n"
            "sys.exit(1)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, HGX_LIB_PATH=emulation_lib), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
