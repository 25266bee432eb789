#!/bin/bash
# The host side of hal2maf and of both halLiftover text paths under AddressSanitizer + UndefinedBehaviorSanitizer and under
# ThreadSanitizer, on a machine without a GPU: the CPU replays (oracle as the device) of profiles/scripts/r04_cpu_maf_soak.py and
# r04_cpu_liftover_soak.py with the sanitizer builds of the library (make -C hal_amd/csrc asan-lib tsan-lib) — the pools of batches and
# host blocks, the device-stage thread, the rendering threads, the parsing pieces.  usage: r04_cpu_sanitizers.sh [alignments]
cd "$(dirname "$0")/../.." || exit 1
N=${1:-25}
make -s -C hal_amd/csrc all hostprof-lib asan-lib tsan-lib || exit 1
make -s -C oracle || exit 1
G=$(dirname "$(gcc -print-file-name=libasan.so)")
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0"
echo "== address + undefined"
HGX_SOAK_PRELOAD="$G/libasan.so $G/libubsan.so $PWD/hal_amd/libhgx_asan.so" python profiles/scripts/r04_cpu_liftover_soak.py 7000 $N | tail -1
HGX_SOAK_PRELOAD="$G/libasan.so $G/libubsan.so $PWD/hal_amd/libhgx_asan.so" python profiles/scripts/r04_cpu_maf_soak.py 7000 $N | tail -1
echo "== thread"
HGX_SOAK_PRELOAD="$G/libtsan.so $PWD/hal_amd/libhgx_tsan.so" python profiles/scripts/r04_cpu_liftover_soak.py 7000 $N | tail -1
HGX_SOAK_PRELOAD="$G/libtsan.so $PWD/hal_amd/libhgx_tsan.so" python profiles/scripts/r04_cpu_maf_soak.py 7000 $N | tail -1
