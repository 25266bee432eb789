#!/bin/bash
# round 3, second GPU call: suite on the build with the workspace cache and the batched text path; bench; int64 two-in-flight diagnosis
O=gpurun_out/r03b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
for w in 0 1; do timeout 300 python profiles/scripts/r03b_wide_inflight.py $w 200 > $O/inflight_$w.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$w -- python $GRAFT_REPO_ROOT/profiles/scripts/r03b_wide_inflight.py $w 40 > $GRAFT_REPO_ROOT/$O/trace_$w.log 2>&1
  f=$(find /tmp/kt$w -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && grep "k_lift" "$f" | tail -400 > $GRAFT_REPO_ROOT/$O/kernel_trace_$w.csv
  [ -n "$f" ] && head -1 "$f" > $GRAFT_REPO_ROOT/$O/kernel_trace_header.csv
done
cd $GRAFT_REPO_ROOT
tail -3 $O/tests.log; cat $O/inflight_*.log | grep wide=
